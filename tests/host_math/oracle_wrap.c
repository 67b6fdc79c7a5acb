/* TEST INFRASTRUCTURE ONLY - the oracle's per-pair functions (static in oracle/softras_oracle.c) behind plain entry points,
 * for tests/host_math/harness.cpp. */
#include "../../oracle/softras_oracle.c"

/* out: [0] border-culled (SRK:316), [1..3] w, [4] sign, [5] dis (euclidean, SRK:341), [6] barycentric distance (SRK:150-154),
 *      [7..9] clipped w, [10] zp (SRK:364), [11] D euclidean (SRK:344), [12..16] surface texel index for R = 1..5 (as float),
 *      [17] the reference's undefined corner (SRK:107-121: no w <= 0 although not strictly inside) */
void hm_orc_pair(const float* f, const float* fi, float xp, float yp, float rad, float sigma, float* out) {
    float w[3], wc[3], t[3], sign = 0, dx = 0, dy = 0;
    out[0] = (float)outside_border(xp, yp, f, rad);
    bary(w, xp, yp, fi);
    const long ub0 = g_ub_events;
    euclid(&sign, &dx, &dy, w, t, f, fi, xp, yp);
    out[17] = (float)(g_ub_events != ub0);
    out[1] = w[0]; out[2] = w[1]; out[3] = w[2];
    out[4] = sign; out[5] = dx * dx + dy * dy;
    out[6] = bary_dist(w);
    for (int k = 0; k < 3; k++) wc[k] = w[k];
    bary_clip(wc);
    out[7] = wc[0]; out[8] = wc[1]; out[9] = wc[2];
    out[10] = (float)(1. / (double)((wc[0] / f[2] + wc[1] / f[5]) + wc[2] / f[8]));
    out[11] = (float)(1. / (1. + (double)expf(-sign * out[5] / sigma)));
    for (int R = 1; R <= 5; R++) out[11 + R] = (float)surface_texel(wc, R);
}
void hm_orc_face_setup(const float* f, float* info) { memset(info, 0, 27 * sizeof(float)); orc_face_setup(f, info); }
