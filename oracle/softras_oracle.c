/*
 * softras_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.  The shipped path (jrender_amd/) never links or calls it.
 *
 * A plain-C restatement of the arithmetic of jrender's SoftRas op
 * (reference file jrender/renderer/dr/softras/cuda/soft_rasterize.py = "SRK",
 * wrapper jrender/renderer/dr/softras/soft_rasterize.py = "SRW"), written as
 * ordinary loops over images/pixels/faces rather than CUDA thread blocks.
 * Every function cites the reference lines it follows.  It is pinned against
 * the reference's own kernels compiled for the host (oracle/_ref, built by
 * oracle/build_ref.py): tests/test_oracle_vs_ref.py demands bit-identical
 * outputs, and tests/golden/ holds vectors generated from oracle/_ref.
 *
 * Precision contract: scalar_t = float; the bare
 * literals `1.` `0.` `1e-5` `1e-6` `1e-10` `2.` in the reference are double,
 * so those islands are evaluated in double and rounded back to float here,
 * exactly where the reference does.  Build with -ffp-contract=off (no FMA).
 *
 * Documented deviation from undefined behaviour in the reference:
 *   (1) SRK:1154-1174 returns an uninitialised local for non-sampled texels;
 *       the oracle (and the patched oracle/_ref) use 0.
 *   (2) SRK:107-121: when a point is not strictly inside (some w >= 1 by
 *       rounding) yet no w <= 0, v0 stays -1 and the reference indexes
 *       t[-1]/a0[-1] (stack garbage).  The oracle defines that corner as
 *       "distance 0, outside sign" (t = -w, dis = 0, sign = -1) and counts the
 *       event in orc_ub_events() so tests can exclude such pixels.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int B, NF, T, R, IS, K;
    float near_, far_, eps, sigma, dist_eps, gamma;
    int dist, rgb, alpha, tex_type, double_side;
    float bg[3];
} orc_params;

static long g_ub_events = 0;
long orc_ub_events(void) { return g_ub_events; }
void orc_ub_reset(void) { g_ub_events = 0; }

/* CUDA max/min on floats ignore a NaN operand, like fmaxf/fminf. */
static inline float maxf(float a, float b) { return fmaxf(a, b); }
static inline float minf(float a, float b) { return fminf(a, b); }

/* ---- per-face preprocessing: SRK:176-236 -------------------------------- */
void orc_face_setup(const float* f, float* info /* 27, pre-zeroed */) {
    const float x0 = f[0], y0 = f[1], x1 = f[3], y1 = f[4], x2 = f[6], y2 = f[7];
    float star[9];
    star[0] = y1 - y2; star[1] = x2 - x1; star[2] = x1 * y2 - x2 * y1;
    star[3] = y2 - y0; star[4] = x0 - x2; star[5] = x2 * y0 - x0 * y2;
    star[6] = y0 - y1; star[7] = x1 - x0; star[8] = x0 * y1 - x1 * y0;
    float det = (x2 * (y0 - y1) + x0 * (y1 - y2)) + x1 * (y2 - y0);            /* SRK:209-212 */
    det = det > 0 ? (float)fmax((double)det, 1e-10) : (float)fmin((double)det, -1e-10); /* SRK:213 */
    for (int k = 0; k < 9; k++) info[k] = star[k] / det;
    for (int j = 0; j < 3; j++)                                                  /* SRK:219-225 */
        for (int k = 0; k < 3; k++)
            info[9 + j * 3 + k] = (f[j * 3] * f[k * 3] + f[j * 3 + 1] * f[k * 3 + 1]) + 1;
    const float px[3] = {x0, x1, x2}, py[3] = {y0, y1, y2};
    for (int k = 0; k < 3; k++) {                                                /* SRK:227-235 */
        const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
        if ((px[k1] - px[k]) * (px[k2] - px[k]) + (py[k1] - py[k]) * (py[k2] - py[k]) < 0) {
            info[18 + k] = 1;
            break;
        }
    }
}

/* ---- small helpers: SRK:20-54 ------------------------------------------- */
static inline void bary(float* w, float x, float y, const float* fi) {          /* SRK:20-25 */
    for (int k = 0; k < 3; k++) w[k] = (fi[3 * k] * x + fi[3 * k + 1] * y) + fi[3 * k + 2];
}
static inline int outside_border(float x, float y, const float* f, float thr) { /* SRK:28-34 */
    return x > maxf(maxf(f[0], f[3]), f[6]) + thr || x < minf(minf(f[0], f[3]), f[6]) - thr ||
           y > maxf(maxf(f[1], f[4]), f[7]) + thr || y < minf(minf(f[1], f[4]), f[7]) - thr;
}
static inline int frontside(const float* f) {                                   /* SRK:37-40 */
    return (f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0]);
}
static inline int pixel_inside(const float* w) {                                /* SRK:43-46 */
    return w[0] <= 1 && w[0] >= 0 && w[1] <= 1 && w[1] >= 0 && w[2] <= 1 && w[2] >= 0;
}
static inline void bary_clip(float* w) {                                        /* SRK:49-54 */
    for (int k = 0; k < 3; k++) w[k] = (float)fmax(fmin((double)w[k], 1.), 0.);
    const float s = (float)fmax((double)((w[0] + w[1]) + w[2]), 1e-5);
    for (int k = 0; k < 3; k++) w[k] = w[k] / s;
}

/* ---- squared euclidean pixel-to-triangle distance: SRK:57-147 ------------ */
static void euclid(float* sign, float* dx, float* dy, const float* w, float* t,
                   const float* f, const float* fi, float xp, float yp) {
    const float* sym = fi + 9;
    const float* obt = fi + 18;
    if (w[0] > 0 && w[1] > 0 && w[2] > 0 && w[0] < 1 && w[1] < 1 && w[2] < 1) {
        float best = 100000000.f, bx = 0, by = 0;
        for (int k = 0; k < 3; k++) {                                            /* SRK:73-102 */
            const int v0 = k, v1 = (k + 1) % 3, v2 = (k + 2) % 3;
            float a[3], t0[3];
            for (int c = 0; c < 3; c++) a[c] = sym[3 * v0 + c] - sym[3 * v1 + c];
            t0[v0] = ((((w[0] * a[0] + w[1] * a[1]) + w[2] * a[2]) - a[v1])) / (a[v0] - a[v1]);
            t0[v1] = 1 - t0[v0];
            t0[v2] = 0;
            for (int c = 0; c < 3; c++) t0[c] -= w[c];
            const float ex = (t0[0] * f[0] + t0[1] * f[3]) + t0[2] * f[6];
            const float ey = (t0[0] * f[1] + t0[1] * f[4]) + t0[2] * f[7];
            const float d = ex * ex + ey * ey;
            if (d < best) { best = d; bx = ex; by = ey; t[0] = t0[0]; t[1] = t0[1]; t[2] = t0[2]; }
        }
        *dx = bx; *dy = by; *sign = 1;
        return;
    }
    int v0 = -1;                                                                  /* SRK:107-121 */
    if (w[1] <= 0 && w[2] <= 0) {
        v0 = 0;
        if (obt[0] == 1 && (xp - f[0]) * (f[6] - f[0]) + (yp - f[1]) * (f[7] - f[1]) > 0) v0 = 2;
    } else if (w[2] <= 0 && w[0] <= 0) {
        v0 = 1;
        if (obt[1] == 1 && (xp - f[3]) * (f[0] - f[3]) + (yp - f[4]) * (f[1] - f[4]) > 0) v0 = 0;
    } else if (w[0] <= 0 && w[1] <= 0) {
        v0 = 2;
        if (obt[2] == 1 && (xp - f[6]) * (f[3] - f[6]) + (yp - f[7]) * (f[4] - f[7]) > 0) v0 = 1;
    } else if (w[0] <= 0) v0 = 1;
    else if (w[1] <= 0) v0 = 2;
    else if (w[2] <= 0) v0 = 0;
    *sign = -1;
    if (v0 < 0) {                       /* documented deviation (2): reference UB */
#pragma omp atomic
        g_ub_events++;
        for (int c = 0; c < 3; c++) t[c] = 0 - w[c];
        *dx = 0; *dy = 0;
        return;
    }
    const int v1 = (v0 + 1) % 3, v2 = (v0 + 2) % 3;
    float a[3];
    for (int c = 0; c < 3; c++) a[c] = sym[3 * v0 + c] - sym[3 * v1 + c];
    t[v0] = (((w[0] * a[0] + w[1] * a[1]) + w[2] * a[2]) - a[v1]) / (a[v0] - a[v1]);  /* SRK:132 */
    t[v1] = 1 - t[v0];
    t[v2] = 0;
    for (int c = 0; c < 3; c++) {                                                /* SRK:137-140 */
        t[c] = (float)fmin(fmax((double)t[c], 0.), 1.);
        t[c] -= w[c];
    }
    *dx = (t[0] * f[0] + t[1] * f[3]) + t[2] * f[6];
    *dy = (t[0] * f[1] + t[1] * f[4]) + t[2] * f[7];
}

#ifdef ORC_EXPERIMENT_FAST_D
/* EXPERIMENT ONLY (tests/probe_fast_distance.py builds a separate library with -DORC_EXPERIMENT_FAST_D -mfma; the oracle proper never
 * defines it).  VERDICT r5 "next" #2 asks what a FAST distance for OUTSIDE pairs would cost in parity: the same expression tree as
 * euclid()'s outside branch with floating-point contraction on (FMA) and the quotient taken as a product with a float reciprocal -
 * what a `v_rcp_f32` + `v_fma_f32` path on the GPU would compute.  The cull DECISION stays on the exact distance; only the
 * coverage D of the surviving outside pairs is taken from this one.  Counters say how many pairs sit in a guard band around the
 * threshold (where a GPU kernel would have to fall back to the exact tree) and how often the fast distance alone would have
 * decided differently. */
static long g_exp_pairs = 0, g_exp_outside = 0, g_exp_guard = 0, g_exp_flips = 0;
static double g_exp_max_rel = 0;
void orc_exp_counters(double* out /*5*/) {
    out[0] = (double)g_exp_pairs; out[1] = (double)g_exp_outside; out[2] = (double)g_exp_guard; out[3] = (double)g_exp_flips; out[4] = g_exp_max_rel;
}
void orc_exp_reset(void) { g_exp_pairs = g_exp_outside = g_exp_guard = g_exp_flips = 0; g_exp_max_rel = 0; }
#ifndef ORC_EXPERIMENT_NO_FMA
__attribute__((optimize("fp-contract=fast"), noinline))
#endif
static float euclid_outside_fast(const float* w, const float* f, const float* fi, float xp, float yp) {
    const float* sym = fi + 9;
    const float* obt = fi + 18;
    int v0 = -1;
    if (w[1] <= 0 && w[2] <= 0) {
        v0 = 0;
        if (obt[0] == 1 && (xp - f[0]) * (f[6] - f[0]) + (yp - f[1]) * (f[7] - f[1]) > 0) v0 = 2;
    } else if (w[2] <= 0 && w[0] <= 0) {
        v0 = 1;
        if (obt[1] == 1 && (xp - f[3]) * (f[0] - f[3]) + (yp - f[4]) * (f[1] - f[4]) > 0) v0 = 0;
    } else if (w[0] <= 0 && w[1] <= 0) {
        v0 = 2;
        if (obt[2] == 1 && (xp - f[6]) * (f[3] - f[6]) + (yp - f[7]) * (f[4] - f[7]) > 0) v0 = 1;
    } else if (w[0] <= 0) v0 = 1;
    else if (w[1] <= 0) v0 = 2;
    else if (w[2] <= 0) v0 = 0;
    if (v0 < 0) return 0.f;
    const int v1 = (v0 + 1) % 3, v2 = (v0 + 2) % 3;
    float a[3], t[3];
    for (int c = 0; c < 3; c++) a[c] = sym[3 * v0 + c] - sym[3 * v1 + c];
#ifndef ORC_EXPERIMENT_NO_RCP
    const float rcp = 1.0f / (a[v0] - a[v1]);
    t[v0] = (((w[0] * a[0] + w[1] * a[1]) + w[2] * a[2]) - a[v1]) * rcp;
#else
    t[v0] = (((w[0] * a[0] + w[1] * a[1]) + w[2] * a[2]) - a[v1]) / (a[v0] - a[v1]);
#endif
    t[v1] = 1 - t[v0];
    t[v2] = 0;
    for (int c = 0; c < 3; c++) t[c] = fminf(fmaxf(t[c], 0.f), 1.f) - w[c];
    const float dx = (t[0] * f[0] + t[1] * f[3]) + t[2] * f[6];
    const float dy = (t[0] * f[1] + t[1] * f[4]) + t[2] * f[7];
    return dx * dx + dy * dy;
}
#endif

static inline float bary_dist(const float* w) {                                 /* SRK:150-154 */
    float d = w[0] > w[1] ? (w[1] > w[2] ? w[2] : w[1]) : (w[0] > w[2] ? w[2] : w[0]);
    return d > 0 ? d * d : -d * d;
}

/* texel selection of the 'surface' sampler: SRK:159-166 (same in SRK:1138-1145) */
static inline int surface_texel(const float* w, int R) {
    const int wx = (int)minf(w[0] * R, (float)(R - 1));
    const int wy = (int)minf(w[1] * R, (float)(R - 1));
    if (((w[0] + w[1]) * R - wx) - wy <= 1) return wy * R + wx;
    return (R - 1 - wy) * R + (R - 1 - wx);
}
/* forward sampler: SRK:156-173 (vertex colours perspective-correct) */
static inline float sample_fwd(const float* tex, const float* w, int R, int k, int type,
                               const float* f, float z) {
    if (type == 0) return tex[surface_texel(w, R) * 3 + k];
    float c = (w[0] * tex[k] / f[2] + w[1] * tex[3 + k] / f[5]) + w[2] * tex[6 + k] / f[8];
    c *= z;
    return c;
}
/* backward's copy of the sampler: SRK:1135-1151 (vertex colours affine) */
static inline float sample_bwd(const float* tex, const float* w, int R, int k, int type) {
    if (type == 0) return tex[surface_texel(w, R) * 3 + k];
    return (w[0] * tex[k] + w[1] * tex[3 + k]) + w[2] * tex[6 + k];
}

/* ---- forward, one pixel: SRK:243-456 ------------------------------------- */
static void forward_pixel(const orc_params* p, const float* faces, const float* textures,
                          const float* infos, int bn, int pn,
                          float* out_rgba /*4*/, float* out_aggr /*2*/, int32_t* out_ids /*K*/) {
    const int is = p->IS, nf = p->NF, K = p->K;
    const int yi = is - 1 - (pn / is), xi = pn % is;
    const float yp = (float)((2. * yi + 1. - is) / is);                          /* SRK:282-283 */
    const float xp = (float)((2. * xi + 1. - is) / is);
    const float thr = p->dist_eps * p->sigma;                                    /* SRK:289 */
    const float rad = sqrtf(thr);                                                /* SRK:316 */

    float col[4] = {1.f, 1.f, 1.f, 0.f};
    if (p->alpha == 2) col[3] = 1.f;
    float ssum = expf(p->eps / p->gamma), smax = p->eps;                         /* SRK:294-295 */
    for (int k = 0; k < 3; k++) {                                                /* SRK:296-303 */
        if (p->rgb == 0) col[k] = p->bg[k];
        else if (p->rgb == 1) col[k] = p->bg[k] * ssum;
    }
    float depth_min = 10000000.f;
    int face_min = -1;
    int32_t qid[64];
    float qz[64];
    int qn = 0, qmax_slot = -1;
    float qmax = -1;

    for (int fn = 0; fn < nf; fn++) {
        const float* f = faces + ((size_t)bn * nf + fn) * 9;
        const float* fi = infos + ((size_t)bn * nf + fn) * 27;
        const float* tex = textures + ((size_t)bn * nf + fn) * p->T * 3;
        if (outside_border(xp, yp, f, rad)) continue;                            /* SRK:316 */
        float w[3], wc[3], t[3], sign = 0, dx = 0, dy = 0, dis, D;
        bary(w, xp, yp, fi);
        if (p->dist == 0) {                                                      /* SRK:331-345 */
            D = pixel_inside(w) ? 1.f : 0.f;
            if (D == 0.f) continue;
        } else if (p->dist == 1) {
            dis = bary_dist(w);
            if (-dis >= thr) continue;
            D = (float)(1. / (1. + (double)expf(-dis / p->sigma)));
        } else {
            euclid(&sign, &dx, &dy, w, t, f, fi, xp, yp);
            dis = dx * dx + dy * dy;
#ifdef ORC_EXPERIMENT_FAST_D
            float dis_fast = dis;
            if (sign < 0) {
                dis_fast = euclid_outside_fast(w, f, fi, xp, yp);
                const int guard = fabsf(dis_fast - thr) <= ORC_EXPERIMENT_GUARD * thr;
                const int flip = (dis_fast >= thr) != (dis >= thr);
#pragma omp critical(orc_exp)
                {
                    g_exp_outside++; g_exp_guard += guard; g_exp_flips += flip;
                    if (dis < thr && dis > 0) { const double r = fabs((double)dis_fast - dis) / dis; if (r > g_exp_max_rel) g_exp_max_rel = r; }
                }
            }
#pragma omp atomic
            g_exp_pairs++;
#endif
            if (sign < 0 && dis >= thr) continue;
#ifdef ORC_EXPERIMENT_FAST_D
            D = (float)(1. / (1. + (double)expf(-sign * dis_fast / p->sigma)));
#else
            D = (float)(1. / (1. + (double)expf(-sign * dis / p->sigma)));
#endif
        }
        if (p->alpha == 0) { if (D > 0.5) col[3] = 1.f; }                        /* SRK:350-358 */
        else if (p->alpha == 1) col[3] += D;
        else if (p->alpha == 2) col[3] = (float)((double)col[3] * (1. - (double)D));

        for (int k = 0; k < 3; k++) wc[k] = w[k];                                /* SRK:362-365 */
        bary_clip(wc);
        const float zp = (float)(1. / (double)((wc[0] / f[2] + wc[1] / f[5]) + wc[2] / f[8]));
        if (zp < p->near_ || zp > p->far_) continue;

        if (qn < K) {                                                            /* SRK:369-385 */
            qid[qn] = fn; qz[qn] = zp;
            if (zp > qmax) { qmax = zp; qmax_slot = qn; }
            qn++;
        } else if (zp < qmax) {
            qid[qmax_slot] = fn; qz[qmax_slot] = zp;
            qmax = -1;
            for (int k = 0; k < qn; k++)
                if (qz[k] > qmax) { qmax = qz[k]; qmax_slot = k; }
        }

        if (p->rgb == 0) {                                                       /* SRK:390-397 */
            if (zp < depth_min && pixel_inside(w) && (p->double_side || frontside(f))) {
                depth_min = zp; face_min = fn;
                for (int k = 0; k < 3; k++) col[k] = sample_fwd(tex, wc, p->R, k, p->tex_type, f, zp);
            }
        } else if (p->rgb == 1) {                                                /* SRK:399-419 */
            if (frontside(f) || p->double_side) {
                const float zn = (p->far_ - zp) / (p->far_ - p->near_);
                float ed = 1.f;
                if (zn > smax) { ed = expf((smax - zn) / p->gamma); smax = zn; }
                const float ez = expf((zn - smax) / p->gamma);
                ssum = ed * ssum + ez * D;
                for (int k = 0; k < 3; k++) {
                    const float c = sample_fwd(tex, wc, p->R, k, p->tex_type, f, zp);
                    col[k] = ed * col[k] + ez * D * c;
                }
            }
        }
    }
    /* finalise: SRK:426-455 */
    if (p->alpha == 0) out_rgba[3] = col[3];
    else if (p->alpha == 1) out_rgba[3] = col[3] / nf;
    else if (p->alpha == 2) out_rgba[3] = (float)(1. - (double)col[3]);
    else out_rgba[3] = 0;
    for (int k = 0; k < 3; k++) out_rgba[k] = p->bg[k];  /* the buffer the reference pre-fills (=0) */
    out_aggr[0] = 0; out_aggr[1] = 0;
    if (p->rgb == 0) {
        if (face_min != -1) for (int k = 0; k < 3; k++) out_rgba[k] = col[k];
        out_aggr[0] = depth_min; out_aggr[1] = (float)face_min;
    } else if (p->rgb == 1) {
        for (int k = 0; k < 3; k++) out_rgba[k] = col[k] / ssum;
        out_aggr[0] = ssum; out_aggr[1] = smax;
    }
    for (int k = 0; k < K; k++) out_ids[k] = k < qn ? qid[k] : -1;
}

static void fill_params(orc_params* p, int B, int NF, int T, int IS, int K, float near_, float far_,
                        float eps, float sigma, int dist, float dist_eps, float gamma, int rgb,
                        int alpha, int tex_type, int double_side, const float* bg) {
    p->B = B; p->NF = NF; p->T = T; p->R = (int)sqrt((double)T); p->IS = IS; p->K = K;   /* SRK:475 */
    p->near_ = near_; p->far_ = far_; p->eps = eps; p->sigma = sigma; p->dist_eps = dist_eps;
    p->gamma = gamma; p->dist = dist; p->rgb = rgb; p->alpha = alpha; p->tex_type = tex_type;
    p->double_side = double_side;
    for (int k = 0; k < 3; k++) p->bg[k] = bg ? bg[k] : 0.f;
}

/*
 * Forward op (SRW:34-103 + SRK:460-516).  Layouts are the reference's:
 *   faces [B,NF,9]  textures [B,NF,T,3]  faces_info [B,NF,27]
 *   aggrs_info [B,2,IS,IS]  soft_colors [B,4,IS,IS]  faces_id_buffer [B,K,IS,IS] (int32, -1 = empty)
 * bg = NULL reproduces the reference (background ignored => 0).
 */
int orc_softras_forward(const float* faces, const float* textures, float* faces_info,
                        float* aggrs_info, float* soft_colors, int32_t* faces_id_buffer,
                        int B, int NF, int T, int IS, int K, float near_, float far_, float eps,
                        float sigma, int dist, float dist_eps, float gamma, int rgb, int alpha,
                        int tex_type, int double_side, const float* bg, int nthreads) {
    if (K < 1 || K > 64) return 1;                        /* reference: q[64], unchecked (SRK:16) */
    orc_params p;
    fill_params(&p, B, NF, T, IS, K, near_, far_, eps, sigma, dist, dist_eps, gamma, rgb, alpha,
                tex_type, double_side, bg);
#ifdef _OPENMP
    /* 0 = all host cores (a serial backward before this call leaves the OpenMP default at 1) */
    omp_set_num_threads(nthreads > 0 ? nthreads : omp_get_num_procs());
#endif
    memset(faces_info, 0, sizeof(float) * (size_t)B * NF * 27);
    for (long i = 0; i < (long)B * NF; i++) orc_face_setup(faces + i * 9, faces_info + i * 27);
    const long pp = (long)IS * IS;
#pragma omp parallel for schedule(dynamic, 256)
    for (long i = 0; i < (long)B * pp; i++) {
        const int bn = (int)(i / pp), pn = (int)(i % pp);
        float rgba[4], aggr[2];
        int32_t ids[64];
        forward_pixel(&p, faces, textures, faces_info, bn, pn, rgba, aggr, ids);
        for (int k = 0; k < 4; k++) soft_colors[((size_t)bn * 4 + k) * pp + pn] = rgba[k];
        for (int k = 0; k < 2; k++) aggrs_info[((size_t)bn * 2 + k) * pp + pn] = aggr[k];
        for (int k = 0; k < K; k++) faces_id_buffer[((size_t)bn * K + k) * pp + pn] = ids[k];
    }
    return 0;
}

static inline void acc(float* dst, float v, int parallel) {
    if (parallel) {
#pragma omp atomic
        *dst += v;
    } else {
        *dst += v;
    }
}

/* ---- backward, one pixel: SRK:1177-1360 ---------------------------------- */
static void backward_pixel(const orc_params* p, const float* faces, const float* textures,
                           const float* soft_colors, const float* infos, const float* aggrs,
                           const int32_t* ids, const float* grad_rgba, float* grad_faces,
                           float* grad_textures, int bn, int pn, int parallel) {
    const int is = p->IS, nf = p->NF, K = p->K, T = p->T;
    const long pp = (long)is * is;
    const int yi = is - 1 - (pn / is), xi = pn % is;
    const float yp = (float)((2. * yi + 1 - is) / is);                           /* SRK:1220-1221 */
    const float xp = (float)((2. * xi + 1 - is) / is);
    const float thr = p->dist_eps * p->sigma;
    const float rad = sqrtf(thr);
    const float ssum = aggrs[((size_t)bn * 2 + 0) * pp + pn];
    const float smax = aggrs[((size_t)bn * 2 + 1) * pp + pn];
    float g[4], out[4];
    for (int k = 0; k < 4; k++) {
        g[k] = grad_rgba[((size_t)bn * 4 + k) * pp + pn];
        out[k] = soft_colors[((size_t)bn * 4 + k) * pp + pn];
    }
    for (int m = 0; m < K; m++) {
        /* the reference reads ids[b][y][x][m] from the transposed buffer (SRW:108,
         * SRK:1226,1234); this is the same element of the [B,K,IS,IS] layout. */
        const int fn = ids[((size_t)bn * K + m) * pp + pn];
        if (fn == -1) break;
        const float* f = faces + ((size_t)bn * nf + fn) * 9;
        const float* fi = infos + ((size_t)bn * nf + fn) * 27;
        const float* tex = textures + ((size_t)bn * nf + fn) * T * 3;
        if (outside_border(xp, yp, f, rad)) continue;                            /* SRK:1244 */
        float w[3], w0[3], t[3] = {0, 0, 0}, sign = 0, dx = 0, dy = 0, dis = 0, D;
        bary(w, xp, yp, fi);
        if (p->dist == 0) D = 1;                                                 /* SRK:1258-1270 */
        else if (p->dist == 1) {
            dis = bary_dist(w);
            for (int k = 0; k < 3; k++) t[k] = w[k];
            D = (float)(1. / (1. + (double)expf(-dis / p->sigma)));
        } else {
            euclid(&sign, &dx, &dy, w, t, f, fi, xp, yp);
            dis = dx * dx + dy * dy;
            D = (float)(1. / (1. + (double)expf(-sign * dis / p->sigma)));
        }
        float* gf = grad_faces + ((size_t)bn * nf + fn) * 9;
        float* gt = grad_textures + ((size_t)bn * nf + fn) * T * 3;
        float gv[3][3] = {{0}};
        float cxy = 0;
        float ca = g[3];                                                         /* SRK:1281-1291 */
        if (p->alpha == 1) ca /= nf;
        else if (p->alpha == 2)
            ca = (float)((double)ca * ((double)(1 - out[3]) / fmax((double)(1 - D), 1e-6)));
        cxy += ca;

        for (int k = 0; k < 3; k++) w0[k] = w[k];                                /* SRK:1294-1296 */
        bary_clip(w);
        const float zp = (float)(1. / (double)((w[0] / f[2] + w[1] / f[5]) + w[2] / f[8]));

        if (p->rgb == 0) {                                                       /* SRK:1299-1306 */
            if ((float)fn == smax) {
                const int texel = p->tex_type == 0 ? surface_texel(w, p->R) : -1;
                for (int k = 0; k < 3; k++)
                    for (int j = 0; j < T; j++) {
                        const float v = p->tex_type == 0 ? (j == texel ? g[k] : 0.f) : w[j] * g[k];
                        acc(&gt[3 * j + k], v, parallel);
                    }
            }
        } else if (p->rgb == 1) {                                                /* SRK:1308-1332 */
            float crgb = 0.f;
            const float zn = (p->far_ - zp) / (p->far_ - p->near_);
            const float zs = D * expf((zn - smax) / p->gamma) / ssum;
            const int texel = p->tex_type == 0 ? surface_texel(w, p->R) : -1;
            for (int k = 0; k < 3; k++) {
                for (int j = 0; j < T; j++) {
                    const float v = p->tex_type == 0 ? (j == texel ? g[k] : 0.f) : w[j] * g[k];
                    acc(&gt[3 * j + k], zs * v, parallel);
                }
                const float c = sample_bwd(tex, w, p->R, k, p->tex_type);
                crgb += g[k] * (c - out[k]);
            }
            crgb *= zs;
            cxy += crgb / D;
            const float cz = crgb / p->gamma / (p->near_ - p->far_) * zp * zp;
            gv[0][2] = cz * w[0] / f[2] / f[2];
            gv[1][2] = cz * w[1] / f[5] / f[5];
            gv[2][2] = cz * w[2] / f[8] / f[8];
        }
        cxy *= D * (1 - D) / p->sigma;                                           /* SRK:1336 */
        if (p->dist == 1) {                                                      /* SRK:1118-1132 */
            const int q = t[0] > t[1] ? (t[1] > t[2] ? 2 : 1) : (t[0] > t[2] ? 2 : 0);
            for (int l = 0; l < 2; l++)
                for (int k = 0; k < 3; k++) {
                    float s = 0;
                    s += -fi[3 * q + l] * fi[3 * k + 0] * xp;
                    s += -fi[3 * q + l] * fi[3 * k + 1] * yp;
                    s += -fi[3 * q + l] * fi[3 * k + 2] * 1;
                    float v = s * cxy;
                    v = (float)((double)v * (dis > 0 ? (2. * (double)sqrtf(dis)) : (2. * (double)sqrtf(-dis))));
                    gv[k][l] = v;
                }
        } else if (p->dist == 2) {                                               /* SRK:1341-1347 */
            for (int k = 0; k < 3; k++) {
                gv[k][0] = 2 * sign * cxy * (t[k] + w0[k]) * dx;
                gv[k][1] = 2 * sign * cxy * (t[k] + w0[k]) * dy;
            }
        }
        acc(&gf[0], gv[0][0], parallel); acc(&gf[1], gv[0][1], parallel);       /* SRK:1349-1358 */
        acc(&gf[3], gv[1][0], parallel); acc(&gf[4], gv[1][1], parallel);
        acc(&gf[6], gv[2][0], parallel); acc(&gf[7], gv[2][1], parallel);
        acc(&gf[2], gv[0][2], parallel); acc(&gf[5], gv[1][2], parallel);
        acc(&gf[8], gv[2][2], parallel);
    }
}

/*
 * Backward op (SRW:105-133 + SRK:1364-1411).  faces_id_buffer is taken in the
 * forward's own [B,K,IS,IS] layout (the reference transposes it to
 * [B,IS,IS,K] first, SRW:108 — a pure re-indexing).  Serial (nthreads<=1)
 * runs accumulate in pixel order = the order a serial run of the reference
 * kernel adds them, so results are deterministic.
 */
int orc_softras_backward(const float* faces, const float* textures, const float* soft_colors,
                         const float* faces_info, const float* aggrs_info,
                         const int32_t* faces_id_buffer, const float* grad_soft_colors,
                         float* grad_faces, float* grad_textures,
                         int B, int NF, int T, int IS, int K, float near_, float far_, float eps,
                         float sigma, int dist, float dist_eps, float gamma, int rgb, int alpha,
                         int tex_type, int double_side, int nthreads) {
    if (K < 1 || K > 64) return 1;
    orc_params p;
    fill_params(&p, B, NF, T, IS, K, near_, far_, eps, sigma, dist, dist_eps, gamma, rgb, alpha,
                tex_type, double_side, NULL);
    memset(grad_faces, 0, sizeof(float) * (size_t)B * NF * 9);
    memset(grad_textures, 0, sizeof(float) * (size_t)B * NF * T * 3);
    const long pp = (long)IS * IS;
    const int parallel = nthreads > 1;
#ifdef _OPENMP
    omp_set_num_threads(parallel ? nthreads : 1);
#endif
#pragma omp parallel for schedule(dynamic, 256) if (parallel)
    for (long i = 0; i < (long)B * pp; i++)
        backward_pixel(&p, faces, textures, soft_colors, faces_info, aggrs_info, faces_id_buffer,
                       grad_soft_colors, grad_faces, grad_textures, (int)(i / pp), (int)(i % pp),
                       parallel);
    return 0;
}

/*
 * Pixel-subset variants (test infrastructure for full-size parity checks where the whole
 * O(pixels x faces) image would take minutes on a CPU): the SAME per-pixel functions, run only
 * for the listed pixels.  pix[i] = global pixel index b*IS*IS + row*IS + col.  faces_info must be
 * precomputed (orc_faces_info).  Outputs are packed per listed pixel: rgba[n][4], aggr[n][2],
 * ids[n][K].
 */
int orc_faces_info(const float* faces, float* faces_info, long nfaces) {
    memset(faces_info, 0, sizeof(float) * (size_t)nfaces * 27);
    for (long i = 0; i < nfaces; i++) orc_face_setup(faces + i * 9, faces_info + i * 27);
    return 0;
}

int orc_softras_forward_subset(const float* faces, const float* textures, const float* faces_info,
                               const int64_t* pix, long npix, float* rgba, float* aggr, int32_t* ids,
                               int B, int NF, int T, int IS, int K, float near_, float far_, float eps,
                               float sigma, int dist, float dist_eps, float gamma, int rgb, int alpha,
                               int tex_type, int double_side, const float* bg, int nthreads) {
    if (K < 1 || K > 64) return 1;
    orc_params p;
    fill_params(&p, B, NF, T, IS, K, near_, far_, eps, sigma, dist, dist_eps, gamma, rgb, alpha,
                tex_type, double_side, bg);
#ifdef _OPENMP
    /* 0 = all host cores (a serial backward before this call leaves the OpenMP default at 1) */
    omp_set_num_threads(nthreads > 0 ? nthreads : omp_get_num_procs());
#endif
    const long pp = (long)IS * IS;
#pragma omp parallel for schedule(dynamic, 16)
    for (long i = 0; i < npix; i++) {
        int32_t tmp[64];
        forward_pixel(&p, faces, textures, faces_info, (int)(pix[i] / pp), (int)(pix[i] % pp),
                      rgba + i * 4, aggr + i * 2, tmp);
        for (int k = 0; k < K; k++) ids[i * K + k] = tmp[k];
    }
    return 0;
}

/* Backward restricted to the listed pixels: equals the full backward when grad_soft_colors is
 * zero everywhere else.  Full-layout inputs, as orc_softras_backward. */
int orc_softras_backward_subset(const float* faces, const float* textures, const float* soft_colors,
                                const float* faces_info, const float* aggrs_info,
                                const int32_t* faces_id_buffer, const float* grad_soft_colors,
                                const int64_t* pix, long npix, float* grad_faces, float* grad_textures,
                                int B, int NF, int T, int IS, int K, float near_, float far_, float eps,
                                float sigma, int dist, float dist_eps, float gamma, int rgb, int alpha,
                                int tex_type, int double_side) {
    if (K < 1 || K > 64) return 1;
    orc_params p;
    fill_params(&p, B, NF, T, IS, K, near_, far_, eps, sigma, dist, dist_eps, gamma, rgb, alpha,
                tex_type, double_side, NULL);
    memset(grad_faces, 0, sizeof(float) * (size_t)B * NF * 9);
    memset(grad_textures, 0, sizeof(float) * (size_t)B * NF * T * 3);
    const long pp = (long)IS * IS;
    for (long i = 0; i < npix; i++)
        backward_pixel(&p, faces, textures, soft_colors, faces_info, aggrs_info, faces_id_buffer,
                       grad_soft_colors, grad_faces, grad_textures, (int)(pix[i] / pp),
                       (int)(pix[i] % pp), 0);
    return 0;
}

int orc_num_procs(void) {
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}
