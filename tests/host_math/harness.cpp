/* TEST INFRASTRUCTURE ONLY - tests/test_device_math_on_host.py.
 *
 * jrender_amd/csrc/softras_device.h - the per-(pixel, face) arithmetic of the HIP raster kernels - compiled for the host
 * (shim/hip/hip_runtime.h) and compared with the oracle's functions PAIR BY PAIR: every face against every pixel centre of its
 * widened box.  What is compared, bit for bit: faces_info, the border decision from the record's box, barycentrics, sign and
 * squared distance (plain IEEE and reciprocal-refinement instantiations, the decision the index buffer depends on), barycentric
 * distance, clipped weights, depth, the texel choice, and the reference-form coverage.  Paths that use a device approximation
 * (v_rcp / v_exp: the default colour path, the inside pairs' 2nd / 3rd projection) are NOT comparable on the host and not compared.
 * Build: g++ -O2 -ffp-contract=off -fno-fast-math (the flags of the device build), optionally -fsanitize=address,undefined. */
#include "hip/hip_runtime.h"
#include "../../jrender_amd/csrc/softras_device.h"

extern "C" {
void hm_orc_pair(const float* f, const float* fi, float xp, float yp, float rad, float sigma, float* out);
void hm_orc_face_setup(const float* f, float* info);
}

static inline bool same(float a, float b) { uint32_t x, y; memcpy(&x, &a, 4); memcpy(&y, &b, 4); return x == y || (a != a && b != b); }

template <bool FAST>
static void device_pair(const jr::FaceGeo& g, float xp, float yp, jr::RasterParams& p, float* out) {
    using namespace jr;
    out[0] = (float)(xp > g.xhi || xp < g.xlo || yp > g.yhi || yp < g.ylo);
    const Bary b = barycentric(g, xp, yp);
    out[1] = b.w0; out[2] = b.w1; out[3] = b.w2;
    float sign, dis;
    euclidean_sign_dis<FAST, true>(g, g.meta, b, xp, yp, sign, dis);
    out[4] = sign; out[5] = dis;
    out[6] = barycentric_dist(b);
    const Bary c = barycentric_clip<FAST>(b);
    out[7] = c.w0; out[8] = c.w1; out[9] = c.w2;
    out[10] = depth_of<FAST>(g, c);
    out[11] = coverage_fast<1>(-sign * dis, p);
    for (int R = 1; R <= 5; R++) out[11 + R] = (float)surface_texel(c, R);
    // the split forms the heavy-tile pipeline uses must agree with the joint one
    out[17] = sign < 0 ? euclidean_outside_dis<FAST>(g, g.meta, b, xp, yp) : euclidean_inside_dis<FAST, true>(g, b);
    const Dist d = euclidean_p2f<FAST>(g, g.meta, b, xp, yp);                         // the backward's form
    out[18] = d.sign; out[19] = d.dx * d.dx + d.dy * d.dy;
}

extern "C" {

/* faces [nf, 9]; every pixel centre of an IS x IS image inside the face's box widened by rad (capped at max_px per axis).
 * counts[0] = pairs compared, [1] = pairs skipped in the reference's undefined corner, [2] = faces whose record is not FLAG_SAFE,
 * [3..] = bit mismatches per quantity: 3 faces_info words, 4 border, 5 w, 6 sign, 7 dis (IEEE), 8 dis (refinement, safe faces),
 * 9 barycentric distance, 10 clipped w, 11 zp (IEEE), 12 zp (refinement, safe faces), 13 reference-form coverage, 14 texel,
 * 15 split forms vs joint, 16 backward form vs forward form, 17 cull decision dis >= thr */
void hm_compare(const float* faces, long nf, int IS, float sigma, float dist_eps_log, int max_px, long* counts) {
    using namespace jr;
    RasterParams p;
    memset(&p, 0, sizeof(p));
    p.IS = IS; p.sigma = sigma; p.dist_eps = dist_eps_log; p.thr = dist_eps_log * sigma; p.rad = sqrtf(p.thr);
    p.consts_safe = 0;
    for (long fn = 0; fn < nf; fn++) {
        const float* f = faces + fn * 9;
        float info[27], oinfo[27];
        face_setup(f, info);
        hm_orc_face_setup(f, oinfo);
        for (int k = 0; k < 27; k++) counts[3] += !same(info[k], oinfo[k]);
        FaceGeo g;
        build_face_geo(g, f, info, p.rad, (int)(fn & 0xffff));
        const bool safe = face_safe(g.meta);
        counts[2] += !safe;
        // pixel columns / rows around the widened box
        auto lo = [&](float v) { double a = floor(((double)v * IS + IS - 1.0) * 0.5) - 1; return (int)fmax(0.0, fmin(a, IS - 1.0)); };
        auto hi = [&](float v) { double a = ceil(((double)v * IS + IS - 1.0) * 0.5) + 1; return (int)fmax(0.0, fmin(a, IS - 1.0)); };
        int x0 = lo(g.xlo), x1 = hi(g.xhi), y0 = lo(g.ylo), y1 = hi(g.yhi);
        if (!(g.xlo == g.xlo) || !(g.ylo == g.ylo)) { x0 = y0 = 0; x1 = y1 = max_px < IS ? max_px : IS - 1; }
        if (x1 - x0 > max_px) x1 = x0 + max_px;
        if (y1 - y0 > max_px) y1 = y0 + max_px;
        for (int yi = y0; yi <= y1; yi++)
            for (int xi = x0; xi <= x1; xi++) {
                const float xp = pixel_centre(xi, IS), yp = pixel_centre(yi, IS);
                float o[18], a[20], b[20];
                hm_orc_pair(f, oinfo, xp, yp, p.rad, sigma, o);
                if (o[17] != 0.f) { counts[1]++; continue; }
                device_pair<false>(g, xp, yp, p, a);
                device_pair<true>(g, xp, yp, p, b);
                counts[0]++;
                counts[4] += a[0] != o[0];
                counts[5] += !same(a[1], o[1]) || !same(a[2], o[2]) || !same(a[3], o[3]);
                counts[6] += a[4] != o[4];
                counts[7] += !same(a[5], o[5]);
                if (safe) counts[8] += !same(b[5], o[5]) || b[4] != o[4];
                counts[9] += !same(a[6], o[6]);
                counts[10] += !same(a[7], o[7]) || !same(a[8], o[8]) || !same(a[9], o[9]);
                counts[11] += !same(a[10], o[10]);
                if (safe) counts[12] += !same(b[10], o[10]) || !same(b[7], o[7]) || !same(b[8], o[8]) || !same(b[9], o[9]);
                counts[13] += !same(a[11], o[11]);
                for (int R = 1; R <= 5; R++) counts[14] += a[11 + R] != o[11 + R];
                counts[15] += !same(a[17], a[5]) || (safe && !same(b[17], b[5]));
                counts[16] += a[18] != a[4] || !same(a[19], a[5]);
                counts[17] += ((a[4] < 0 && a[5] >= p.thr) != (o[4] < 0 && o[5] >= p.thr)) || (safe && ((b[4] < 0 && b[5] >= p.thr) != (o[4] < 0 && o[5] >= p.thr)));
            }
    }
}

}  /* extern "C" */
