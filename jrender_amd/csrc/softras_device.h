// SoftRas per-(pixel, face) arithmetic for gfx950 — shared by the forward and backward kernels.
//
// The values computed here are the ones the reference's kernels compute
// (jrender/renderer/dr/softras/cuda/soft_rasterize.py, "SRK"), in the same
// precision and association order, because the per-pixel face-index buffer has
// to match bit for bit.  This translation unit MUST be
// built with -ffp-contract=off (no FMA contraction) and without fast-math; the
// `double` islands below are where the reference's bare literals promote.
//
// What is NOT the reference's: the data organisation.  Every face gets one packed
// 144-byte geometry record (FaceGeo) written once per forward by the setup kernel:
// border box incl. cull radius, face_inv, vertices, edge-difference vectors of the
// Gram matrix and the edge denominators, obtuse vertex, facing.  The raster kernels
// copy the records of the faces a wavefront needs into LDS and every lane reads the
// record of ITS current face — no dynamic register indexing, no scattered global
// gathers in the inner loop.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jr {

constexpr int TILE = 8;          // pixels per side of a wavefront's tile (8x8 = 64 lanes)
constexpr int BIN = 32;          // pixels per side of a binning cell (4x4 tiles)
constexpr int SUBS = BIN / TILE; // tiles per bin side
constexpr int CHUNK = 64;        // faces staged per pass = one per lane
constexpr int MAX_IMAGE = 4096;

struct RasterParams {
    int B, NF, T, R, IS, K;
    float near_, far_, eps, sigma, dist_eps, gamma;
    float thr;   // dist_eps * sigma           (SRK:289)
    float rad;   // sqrtf(thr)                 (SRK:316)
    int dist, rgb, alpha, tex, double_side;
    float bg[3];
    int bins_x, bins_y;   // ceil(IS / BIN)
};

// Packed per-face geometry, 36 floats = 9 x 16 B (global, one per face per forward).
struct FaceGeo {
    float xlo, xhi, ylo, yhi;      // border box incl. cull radius        (SRK:28-34, :316)
    float inv[9];                  // face_inv                            (SRK:205-217)
    float z[3];
    float x0, y0, x1, y1, x2, y2;
    int obt;                       // index of the (first) obtuse vertex or -1   (SRK:227-235)
    int front;                     // check_face_frontside                (SRK:37-40)
    float A[9];                    // A[e] = sym[e] - sym[e+1]            (SRK:77-79)
    float Dn[3];                   // A[e][e] - A[e][e+1]                 (SRK:81 denominator)
};
static_assert(sizeof(FaceGeo) == 144, "FaceGeo layout");

// LDS record = geometry + what the colour path needs.  52 dwords = 13 x 16 B: an ODD number of
// 16-byte slots, so lanes that read records of different faces spread over all LDS bank groups.
struct FaceRec {
    FaceGeo g;
    int id;
    float col[9];                  // T==1 surface colour (3) or the three vertex colours (9)
    int pad[6];
};
static_assert(sizeof(FaceRec) == 208, "FaceRec layout");

// Per-face preprocessing = faces_info of the reference (SRK:176-236).
__device__ inline void face_setup(const float* __restrict__ f, float* __restrict__ info) {
    const float x0 = f[0], y0 = f[1], x1 = f[3], y1 = f[4], x2 = f[6], y2 = f[7];
    float star[9];
    star[0] = y1 - y2; star[1] = x2 - x1; star[2] = x1 * y2 - x2 * y1;
    star[3] = y2 - y0; star[4] = x0 - x2; star[5] = x2 * y0 - x0 * y2;
    star[6] = y0 - y1; star[7] = x1 - x0; star[8] = x0 * y1 - x1 * y0;
    float det = (x2 * (y0 - y1) + x0 * (y1 - y2)) + x1 * (y2 - y0);
    det = det > 0 ? (float)fmax((double)det, 1e-10) : (float)fmin((double)det, -1e-10);
#pragma unroll
    for (int k = 0; k < 9; k++) info[k] = star[k] / det;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int k = 0; k < 3; k++)
            info[9 + 3 * j + k] = (f[3 * j] * f[3 * k] + f[3 * j + 1] * f[3 * k + 1]) + 1.f;
    const float d0 = (x1 - x0) * (x2 - x0) + (y1 - y0) * (y2 - y0);
    const float d1 = (x2 - x1) * (x0 - x1) + (y2 - y1) * (y0 - y1);
    const float d2 = (x0 - x2) * (x1 - x2) + (y0 - y2) * (y1 - y2);
    const int obt = d0 < 0 ? 0 : (d1 < 0 ? 1 : (d2 < 0 ? 2 : -1));
    info[18] = obt == 0 ? 1.f : 0.f;
    info[19] = obt == 1 ? 1.f : 0.f;
    info[20] = obt == 2 ? 1.f : 0.f;
#pragma unroll
    for (int k = 21; k < 27; k++) info[k] = 0.f;
}

// Geometry record from the face and its faces_info.
__device__ inline void build_face_geo(FaceGeo& r, const float* __restrict__ f,
                                      const float* __restrict__ fi, float rad) {
    const float x0 = f[0], y0 = f[1], x1 = f[3], y1 = f[4], x2 = f[6], y2 = f[7];
    r.xhi = fmaxf(fmaxf(x0, x1), x2) + rad;
    r.xlo = fminf(fminf(x0, x1), x2) - rad;
    r.yhi = fmaxf(fmaxf(y0, y1), y2) + rad;
    r.ylo = fminf(fminf(y0, y1), y2) - rad;
#pragma unroll
    for (int k = 0; k < 9; k++) r.inv[k] = fi[k];
    r.z[0] = f[2]; r.z[1] = f[5]; r.z[2] = f[8];
    r.x0 = x0; r.y0 = y0; r.x1 = x1; r.y1 = y1; r.x2 = x2; r.y2 = y2;
    const float* sym = fi + 9;
#pragma unroll
    for (int e = 0; e < 3; e++) {
        const int e1 = (e + 1) % 3;
#pragma unroll
        for (int c = 0; c < 3; c++) r.A[3 * e + c] = sym[3 * e + c] - sym[3 * e1 + c];
        r.Dn[e] = r.A[3 * e + e] - r.A[3 * e + e1];
    }
    r.obt = fi[18] == 1.f ? 0 : (fi[19] == 1.f ? 1 : (fi[20] == 1.f ? 2 : -1));
    r.front = ((y2 - y0) * (x1 - x0) < (y1 - y0) * (x2 - x0)) ? 1 : 0;
}

// pixel centre in NDC: (2*i + 1 - IS) / IS evaluated in double, rounded once (SRK:280-283)
__device__ inline float pixel_centre(int i, int is) {
    return (float)((2. * i + 1. - is) / is);
}

struct Bary { float w0, w1, w2; };

__device__ inline Bary barycentric(const FaceGeo& r, float x, float y) {             // SRK:20-25
    Bary b;
    b.w0 = (r.inv[0] * x + r.inv[1] * y) + r.inv[2];
    b.w1 = (r.inv[3] * x + r.inv[4] * y) + r.inv[5];
    b.w2 = (r.inv[6] * x + r.inv[7] * y) + r.inv[8];
    return b;
}

__device__ inline bool pixel_inside(const Bary& b) {                                 // SRK:43-46
    return b.w0 <= 1 && b.w0 >= 0 && b.w1 <= 1 && b.w1 >= 0 && b.w2 <= 1 && b.w2 >= 0;
}

// max(min(v, 1.), 0.) / min(max(v, 0.), 1.): the reference evaluates these in double because of
// the bare literals; both are exact selections, so the float form returns the same value
// (NaN -> 1 resp. 0 in both, like CUDA's fmin/fmax; only the sign of a zero may differ, which
// nothing downstream observes).
__device__ inline float clamp01(float v) { return fmaxf(fminf(v, 1.f), 0.f); }           // SRK:51
__device__ inline float clamp01_maxfirst(float v) { return fminf(fmaxf(v, 0.f), 1.f); }  // SRK:138

__device__ inline Bary barycentric_clip(Bary b) {                                    // SRK:49-54
    b.w0 = clamp01(b.w0); b.w1 = clamp01(b.w1); b.w2 = clamp01(b.w2);
    // max(w_sum, 1e-5) compares in double and stores (float)1e-5 when clamped == fmaxf(s, 1e-5f)
    const float s = fmaxf((b.w0 + b.w1) + b.w2, 1e-5f);
    b.w0 = b.w0 / s; b.w1 = b.w1 / s; b.w2 = b.w2 / s;
    return b;
}

// depth of the clipped barycentric point, 1./(sum w/z) (SRK:364, :1296).  The reference divides in
// double and rounds to float; for a float divisor that equals the correctly rounded float quotient
// (53 >= 2*24+2 bits: double rounding is innocuous for division), i.e. IEEE 1.0f/s.
__device__ inline float depth_of(const FaceGeo& r, const Bary& c) {
    const float s = (c.w0 / r.z[0] + c.w1 / r.z[1]) + c.w2 / r.z[2];
    return 1.0f / s;
}

struct Dist {
    float sign, dx, dy;   // sign: +1 inside / -1 outside; (dx,dy) = nearest point - pixel
    float t0, t1, t2;     // nearest-point barycentric minus w   (SRK:98-100, :139)
};

__device__ inline float edge_param(const Bary& b, const float* A3, float a_v1, float dn) {
    return (((b.w0 * A3[0] + b.w1 * A3[1]) + b.w2 * A3[2]) - a_v1) / dn;       // SRK:81 / :132
}

// squared-distance machinery, euclidean mode (SRK:57-147).  No dynamic register indexing:
// the edge is selected with v_cndmask chains.
__device__ inline Dist euclidean_p2f(const FaceGeo& r, const Bary& b, float xp, float yp) {
    Dist d;
    if (b.w0 > 0 && b.w1 > 0 && b.w2 > 0 && b.w0 < 1 && b.w1 < 1 && b.w2 < 1) {
        float best = 100000000.f, bx = 0.f, by = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 3; e++) {
            const int e1 = (e + 1) % 3;
            const float tv = edge_param(b, &r.A[3 * e], r.A[3 * e + e1], r.Dn[e]);
            const float tn = 1 - tv;
            float u0 = (e == 0) ? tv : ((e1 == 0) ? tn : 0.f);
            float u1 = (e == 1) ? tv : ((e1 == 1) ? tn : 0.f);
            float u2 = (e == 2) ? tv : ((e1 == 2) ? tn : 0.f);
            u0 -= b.w0; u1 -= b.w1; u2 -= b.w2;
            const float ex = (u0 * r.x0 + u1 * r.x1) + u2 * r.x2;
            const float ey = (u0 * r.y0 + u1 * r.y1) + u2 * r.y2;
            const float dd = ex * ex + ey * ey;
            if (dd < best) { best = dd; bx = ex; by = ey; s0 = u0; s1 = u1; s2 = u2; }
        }
        d.sign = 1.f; d.dx = bx; d.dy = by; d.t0 = s0; d.t1 = s1; d.t2 = s2;
        return d;
    }
    int v0 = -1;                                                                      // SRK:107-121
    if (b.w1 <= 0 && b.w2 <= 0) {
        v0 = 0;
        if (r.obt == 0 && (xp - r.x0) * (r.x2 - r.x0) + (yp - r.y0) * (r.y2 - r.y0) > 0) v0 = 2;
    } else if (b.w2 <= 0 && b.w0 <= 0) {
        v0 = 1;
        if (r.obt == 1 && (xp - r.x1) * (r.x0 - r.x1) + (yp - r.y1) * (r.y0 - r.y1) > 0) v0 = 0;
    } else if (b.w0 <= 0 && b.w1 <= 0) {
        v0 = 2;
        if (r.obt == 2 && (xp - r.x2) * (r.x1 - r.x2) + (yp - r.y2) * (r.y1 - r.y2) > 0) v0 = 1;
    } else if (b.w0 <= 0) v0 = 1;
    else if (b.w1 <= 0) v0 = 2;
    else if (b.w2 <= 0) v0 = 0;
    d.sign = -1.f;
    if (v0 < 0) {
        // Reference indexes t[-1]/a0[-1] here (undefined behaviour; only reachable when some
        // w >= 1 by rounding while none is <= 0).  Defined like the oracle: distance 0.
        d.dx = 0.f; d.dy = 0.f; d.t0 = 0.f - b.w0; d.t1 = 0.f - b.w1; d.t2 = 0.f - b.w2;
        return d;
    }
    const float a0 = v0 == 0 ? r.A[0] : (v0 == 1 ? r.A[3] : r.A[6]);
    const float a1 = v0 == 0 ? r.A[1] : (v0 == 1 ? r.A[4] : r.A[7]);
    const float a2 = v0 == 0 ? r.A[2] : (v0 == 1 ? r.A[5] : r.A[8]);
    const float av1 = v0 == 0 ? a1 : (v0 == 1 ? a2 : a0);
    const float dn = v0 == 0 ? r.Dn[0] : (v0 == 1 ? r.Dn[1] : r.Dn[2]);
    const float tv = (((b.w0 * a0 + b.w1 * a1) + b.w2 * a2) - av1) / dn;                 // SRK:132
    const float tn = 1 - tv;
    // t[v0] = tv, t[v1] = 1 - tv, t[v2] = 0 with v1 = v0+1, v2 = v0+2 (mod 3)
    float u0 = v0 == 0 ? tv : (v0 == 2 ? tn : 0.f);
    float u1 = v0 == 1 ? tv : (v0 == 0 ? tn : 0.f);
    float u2 = v0 == 2 ? tv : (v0 == 1 ? tn : 0.f);
    u0 = clamp01_maxfirst(u0) - b.w0;                                                     // SRK:137-140
    u1 = clamp01_maxfirst(u1) - b.w1;
    u2 = clamp01_maxfirst(u2) - b.w2;
    d.dx = (u0 * r.x0 + u1 * r.x1) + u2 * r.x2;
    d.dy = (u0 * r.y0 + u1 * r.y1) + u2 * r.y2;
    d.t0 = u0; d.t1 = u1; d.t2 = u2;
    return d;
}

__device__ inline float barycentric_dist(const Bary& b) {                             // SRK:150-154
    const float m = b.w0 > b.w1 ? (b.w1 > b.w2 ? b.w2 : b.w1) : (b.w0 > b.w2 ? b.w2 : b.w0);
    return m > 0 ? m * m : -m * m;
}

// sigmoid coverage: 1./(1.+exp(x)) with float exp and a double add/divide (SRK:338, :344)
__device__ inline float coverage(float neg_arg) {
    return (float)(1. / (1. + (double)expf(neg_arg)));
}

// 'surface' sampler texel choice (SRK:159-166, identical in SRK:1138-1145)
__device__ inline int surface_texel(const Bary& c, int R) {
    const int wx = (int)fminf(c.w0 * R, (float)(R - 1));
    const int wy = (int)fminf(c.w1 * R, (float)(R - 1));
    if (((c.w0 + c.w1) * R - wx) - wy <= 1) return wy * R + wx;
    return (R - 1 - wy) * R + (R - 1 - wx);
}

// ---- wavefront helpers (64 lanes) ----------------------------------------------------------
__device__ inline unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// Value of a 64-bit wave-uniform table entry selected by a small per-lane index (0..7): the
// compiler lowers this to v_cndmask chains on SGPR operands.
__device__ inline unsigned long long select8(const unsigned long long (&tab)[8], int idx) {
    unsigned long long m = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) m = idx == c ? tab[c] : m;
    return m;
}

}  // namespace jr
