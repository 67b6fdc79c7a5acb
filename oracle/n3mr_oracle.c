/*
 * TEST INFRASTRUCTURE ONLY — never linked into, imported or called by the product (jrender_amd/).
 *
 * Plain-C restatement of the reference's NMR ("n3mr") renderer path, serial, single precision exactly
 * where the reference is single precision and double exactly where its C++ promotes to double:
 *
 *   N3K = /root/reference/jrender/renderer/dr/n3mr/cuda/rasterize.py   (five CUDA kernel strings)
 *   N3F = /root/reference/jrender/renderer/dr/n3mr/n3mr.py             (Function: fills, host ops)
 *
 *   n3o face_index pass     N3K:35-164   forward_face_index_map_cuda_kernel
 *   n3o texture sampling    N3K:228-298  forward_texture_sampling_cuda_kernel
 *   n3o pixel-map gradient  N3K:352-610  backward_pixel_map_cuda_kernel
 *   n3o texture gradient    N3K:660-694  backward_textures_cuda_kernel
 *   n3o depth gradient      N3K:739-788  backward_depth_map_cuda_kernel
 *
 * Pinned against the reference's own kernels compiled for the host (oracle/_ref/libn3mr_ref.so, built by
 * oracle/build_ref.py from where they lie): tests/test_oracle.py demands BIT-IDENTICAL maps and gradients
 * for every configuration it draws.  The entry points have the signatures of oracle/ref_n3mr_driver.cpp so
 * that oracle.N3mrOracle can load either library; this one needs nothing but gcc, i.e. it also builds on the
 * GPU box where /root/reference does not exist.  Build: -O2 -std=c11 -ffp-contract=off -fno-fast-math.
 *
 * Order of evaluation follows a SERIAL run of the reference (thread 0, 1, 2, ...): the per-pixel spin lock of
 * the z-buffer pass (N3K:140-161) then lets the LOWEST face index win a depth tie, and the float atomics of
 * the texture / depth gradients (N3K:691, :771) add in pixel order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- forward ------------------------------------------------------------------------------------- */

/* N3K:66-86: pixel-space vertices p = 0.5*(ndc*is + is - 1) and the inverse of [[x0 x1 x2][y0 y1 y2][1 1 1]]. */
static void face_inverse(const float* face, int is, float px[3], float py[3], float inv[9]) {
    for (int k = 0; k < 3; k++) {
        px[k] = 0.5f * (face[3 * k] * is + is - 1);
        py[k] = 0.5f * (face[3 * k + 1] * is + is - 1);
    }
    inv[0] = py[1] - py[2]; inv[1] = px[2] - px[1]; inv[2] = px[1] * py[2] - px[2] * py[1];
    inv[3] = py[2] - py[0]; inv[4] = px[0] - px[2]; inv[5] = px[2] * py[0] - px[0] * py[2];
    inv[6] = py[0] - py[1]; inv[7] = px[1] - px[0]; inv[8] = px[0] * py[1] - px[1] * py[0];
    const float den = (px[2] * (py[0] - py[1]) + px[0] * (py[1] - py[2])) + px[1] * (py[2] - py[0]);
    for (int k = 0; k < 9; k++) inv[k] = inv[k] / den;
}

/* N3K:35-164.  Serial over faces: a pixel is overwritten only by a strictly nearer face. */
static void face_index_pass(const float* faces, float* faces_inv, int32_t* face_index_map, float* weight_map,
                            float* depth_map, float* face_inv_map, int B, int NF, int is, float near_, float far_,
                            int return_depth) {
    for (int bn = 0; bn < B; bn++)
        for (int fn = 0; fn < NF; fn++) {
            const float* f = faces + ((size_t)bn * NF + fn) * 9;
            float* finv_out = faces_inv + ((size_t)bn * NF + fn) * 9;
            /* back side: skipped (N3K:63) */
            if ((f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0])) continue;
            float px[3], py[3], inv[9];
            face_inverse(f, is, px, py, inv);
            for (int k = 0; k < 9; k++) finv_out[k] = inv[k];
            /* bounding box in pixels (N3K:89-99) */
            float x_min = is, y_min = is, x_max = 0, y_max = 0;
            for (int k = 0; k < 3; k++) {
                if (px[k] < x_min) x_min = px[k];
                if (px[k] > x_max) x_max = px[k];
                if (py[k] < y_min) y_min = py[k];
                if (py[k] > y_max) y_max = py[k];
            }
            int ix0 = (int)x_min, ix1 = (int)x_max, iy0 = (int)y_min, iy1 = (int)y_max;
            if (ix0 < 0) ix0 = 0;
            if (iy0 < 0) iy0 = 0;
            if (ix1 > is - 1) ix1 = is - 1;
            if (iy1 > is - 1) iy1 = is - 1;
            for (int xi = ix0; xi <= ix1; xi++)
                for (int yi = iy0; yi <= iy1; yi++) {
                    /* pixel centre in NDC and the three edge tests (N3K:108-116) */
                    const float yp = (float)((2. * yi + 1 - is) / is);
                    const float xp = (float)((2. * xi + 1 - is) / is);
                    if (((yp - f[1]) * (f[3] - f[0]) < (xp - f[0]) * (f[4] - f[1])) ||
                        ((yp - f[4]) * (f[6] - f[3]) < (xp - f[3]) * (f[7] - f[4])) ||
                        ((yp - f[7]) * (f[0] - f[6]) < (xp - f[6]) * (f[1] - f[7])))
                        continue;
                    /* barycentric weights, clamped and renormalised; depth (N3K:120-134) */
                    float w[3], ws = 0.f;
                    for (int k = 0; k < 3; k++) {
                        const float v = (inv[3 * k] * xi + inv[3 * k + 1] * yi) + inv[3 * k + 2];
                        w[k] = (float)fmin(fmax((double)v, 0.), 1.);
                        ws += w[k];
                    }
                    for (int k = 0; k < 3; k++) w[k] = w[k] / ws;
                    const float zp = (float)(1. / (double)((w[0] / f[2] + w[1] / f[5]) + w[2] / f[8]));
                    if (zp <= near_ || far_ <= zp) continue;                             /* N3K:136 */
                    const size_t idx = (size_t)bn * is * is + (size_t)yi * is + xi;
                    if (zp < depth_map[idx]) {                                           /* N3K:143-156 */
                        depth_map[idx] = zp;
                        face_index_map[idx] = fn;
                        for (int k = 0; k < 3; k++) weight_map[3 * idx + k] = w[k];
                        if (return_depth)
                            for (int k = 0; k < 9; k++) face_inv_map[9 * idx + k] = inv[k];
                    }
                }
        }
}

/* N3K:228-298: trilinear sample of the face's texture cube at the perspective-corrected weights. */
static void texture_pass(const float* faces, const float* textures, const int32_t* face_index_map,
                         const float* weight_map, const float* depth_map, float* rgb_map,
                         int32_t* sampling_index_map, float* sampling_weight_map, int B, int NF, int is, int ts,
                         float eps) {
    const size_t P = (size_t)B * is * is;
    for (size_t i = 0; i < P; i++) {
        const int fn = face_index_map[i];
        if (fn < 0) continue;
        const int bn = (int)(i / ((size_t)is * is));
        const float* f = faces + ((size_t)bn * NF + fn) * 9;
        const float* tex = textures + ((size_t)bn * NF + fn) * ts * ts * ts * 3;
        const float* w = weight_map + 3 * i;
        const float depth = depth_map[i];
        float tif[3];
        for (int k = 0; k < 3; k++) {                                                   /* N3K:264-272 */
            float t = w[k] * (ts - 1) * (depth / f[3 * k + 2]);
            t = (float)fmax((double)t, 0.);
            t = (float)fmin((double)t, (double)(ts - 1 - eps));
            tif[k] = t;
        }
        float acc[3] = {0.f, 0.f, 0.f};
        for (int pn = 0; pn < 8; pn++) {                                                /* N3K:275-295 */
            float ww = 1;
            int ti[3];
            for (int k = 0; k < 3; k++) {
                if (((pn >> k) % 2) == 0) { ww *= 1 - (tif[k] - (int)tif[k]); ti[k] = (int)tif[k]; }
                else { ww *= tif[k] - (int)tif[k]; ti[k] = (int)tif[k] + 1; }
            }
            const int isc = ti[0] * ts * ts + ti[1] * ts + ti[2];
            for (int k = 0; k < 3; k++) acc[k] += ww * tex[isc * 3 + k];
            sampling_index_map[8 * i + pn] = isc;
            sampling_weight_map[8 * i + pn] = ww;
        }
        for (int k = 0; k < 3; k++) rgb_map[3 * i + k] = acc[k];
    }
}

int ref_n3mr_forward(const float* faces, const float* textures, float* faces_inv, int32_t* face_index_map,
                     float* weight_map, float* depth_map, float* face_inv_map, float* rgb_map,
                     int32_t* sampling_index_map, float* sampling_weight_map, int B, int NF, int TS, int IS,
                     float near_, float far_, float eps, int return_rgb, int return_depth) {
    const size_t P = (size_t)B * IS * IS;
    /* fills of the Function (N3F:57-93): index -1, depth far, everything else 0 */
    for (size_t i = 0; i < P; i++) { face_index_map[i] = -1; depth_map[i] = far_; }
    memset(weight_map, 0, sizeof(float) * P * 3);
    if (return_depth) memset(face_inv_map, 0, sizeof(float) * P * 9);
    face_index_pass(faces, faces_inv, face_index_map, weight_map, depth_map, face_inv_map, B, NF, IS, near_, far_,
                    return_depth);
    if (return_rgb) {
        memset(rgb_map, 0, sizeof(float) * P * 3);
        memset(sampling_index_map, 0, sizeof(int32_t) * P * 8);
        memset(sampling_weight_map, 0, sizeof(float) * P * 8);
        texture_pass(faces, textures, face_index_map, weight_map, depth_map, rgb_map, sampling_index_map,
                     sampling_weight_map, B, NF, IS, TS, eps);
    }
    return 0;
}

/* ---- backward ------------------------------------------------------------------------------------ */

/* -= diff / (dist +- eps) for the two vertices of the edge (N3K:496-505, :583-592); dist is formed in double */
static void push(float diff, int d0, int d1, float d1_cross, const float q[3][2], int is, float eps, float* ga,
                 float* gb) {
    const float e10 = q[1][0] - q[0][0];
    if (q[1][0] != d0) {
        float dist = (float)((double)(e10 / (q[1][0] - d0) * (d1 - d1_cross)) * 2. / is);
        dist = (0 < dist) ? dist + eps : dist - eps;
        *ga -= diff / dist;
    }
    if (q[0][0] != d0) {
        float dist = (float)((double)(e10 / (d0 - q[0][0]) * (d1 - d1_cross)) * 2. / is);
        dist = (0 < dist) ? dist + eps : dist - eps;
        *gb -= diff / dist;
    }
}

/* N3K:352-610: for each front face, each edge, each image axis: the scan positions the edge crosses. */
static void pixel_map_pass(const float* faces, const int32_t* face_index_map, const float* rgb_map,
                           const float* alpha_map, const float* grad_rgb_map, const float* grad_alpha_map,
                           float* grad_faces, int B, int NF, int is, float eps, int use_rgb, int use_a) {
    for (int bn = 0; bn < B; bn++)
        for (int fn = 0; fn < NF; fn++) {
            const float* face = faces + ((size_t)bn * NF + fn) * 9;
            float* g = grad_faces + ((size_t)bn * NF + fn) * 9;
            if ((face[7] - face[1]) * (face[3] - face[0]) < (face[4] - face[1]) * (face[6] - face[0])) continue;
            const size_t mbase = (size_t)bn * is * is;
            float acc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int edge = 0; edge < 3; edge++) {
                const int pis[3] = {edge % 3, (edge + 1) % 3, (edge + 2) % 3};
                float pp[3][2];
                for (int n = 0; n < 3; n++)
                    for (int d = 0; d < 2; d++) pp[n][d] = 0.5f * (face[3 * pis[n] + d] * is + is - 1);
                for (int axis = 0; axis < 2; axis++) {
                    float q[3][2];
                    for (int n = 0; n < 3; n++)
                        for (int d = 0; d < 2; d++) q[n][d] = pp[n][(d + axis) % 2];
                    const int direction = axis == 0 ? (q[0][0] < q[1][0] ? -1 : 1) : (q[0][0] < q[1][0] ? 1 : -1);
                    const int d0_from = (int)fmax((double)ceilf(fminf(q[0][0], q[1][0])), 0.);
                    const int d0_to = (int)fmin((double)fmaxf(q[0][0], q[1][0]), is - 1.);
                    float* ga = &acc[pis[0] * 3 + (1 - axis)];
                    float* gb = &acc[pis[1] * 3 + (1 - axis)];
                    for (int d0 = d0_from; d0 <= d0_to; d0++) {
                        const float d1_cross = (q[1][1] - q[0][1]) / (q[1][0] - q[0][0]) * (d0 - q[0][0]) + q[0][1];
                        const int d1_in = 0 < direction ? (int)floorf(d1_cross) : (int)ceilf(d1_cross);
                        const int d1_out = d1_in + direction;
                        if (d1_in < 0 || is <= d1_in) continue;
                        if (d1_out < 0 || is <= d1_out) continue;
                        const size_t idx_in = mbase + (axis == 0 ? (size_t)d1_in * is + d0 : (size_t)d0 * is + d1_in);
                        const size_t idx_out = mbase + (axis == 0 ? (size_t)d1_out * is + d0 : (size_t)d0 * is + d1_out);
                        const size_t moff = axis == 0 ? (size_t)is : 1;
                        float a_in = 0.f, a_out = 0.f;
                        const float *c_in = 0, *c_out = 0;
                        if (use_a) { a_in = alpha_map[idx_in]; a_out = alpha_map[idx_out]; }
                        if (use_rgb) { c_in = rgb_map + idx_in * 3; c_out = rgb_map + idx_out * 3; }
                        /* out: from the outside neighbour to the image border (N3K:450-507) */
                        if (face_index_map[idx_in] == fn) {
                            const int lim = 0 < direction ? is - 1 : 0;
                            int from = d1_out < lim ? d1_out : lim, to = d1_out > lim ? d1_out : lim;
                            if (from < 0) from = 0;
                            if (to > is - 1) to = is - 1;
                            size_t m = mbase + (axis == 0 ? (size_t)from * is + d0 : (size_t)d0 * is + from);
                            for (int d1 = from; d1 <= to; d1++, m += moff) {
                                float diff = 0;
                                if (use_a) diff += (alpha_map[m] - a_in) * grad_alpha_map[m];
                                if (use_rgb)
                                    for (int k = 0; k < 3; k++) diff += (rgb_map[m * 3 + k] - c_in[k]) * grad_rgb_map[m * 3 + k];
                                if (diff <= 0) continue;
                                push(diff, d0, d1, d1_cross, q, is, eps, ga, gb);
                            }
                        }
                        /* in: from the inside pixel to the opposite edge (N3K:510-594) */
                        {
                            float cross2;
                            if ((d0 - q[0][0]) * (d0 - q[2][0]) < 0)
                                cross2 = (q[2][1] - q[0][1]) / (q[2][0] - q[0][0]) * (d0 - q[0][0]) + q[0][1];
                            else
                                cross2 = (q[1][1] - q[2][1]) / (q[1][0] - q[2][0]) * (d0 - q[2][0]) + q[2][1];
                            const int lim = 0 < direction ? (int)ceilf(cross2) : (int)floorf(cross2);
                            int from = d1_in < lim ? d1_in : lim, to = d1_in > lim ? d1_in : lim;
                            if (from < 0) from = 0;
                            if (to > is - 1) to = is - 1;
                            size_t m = mbase + (axis == 0 ? (size_t)from * is + d0 : (size_t)d0 * is + from);
                            for (int d1 = from; d1 <= to; d1++, m += moff) {
                                if (face_index_map[m] != fn) continue;
                                float diff = 0;
                                if (use_a) diff += (alpha_map[m] - a_out) * grad_alpha_map[m];
                                if (use_rgb)
                                    for (int k = 0; k < 3; k++) diff += (rgb_map[m * 3 + k] - c_out[k]) * grad_rgb_map[m * 3 + k];
                                if (diff <= 0) continue;
                                push(diff, d0, d1, d1_cross, q, is, eps, ga, gb);
                            }
                        }
                    }
                }
            }
            for (int k = 0; k < 9; k++) g[k] = acc[k];                                  /* N3K:608-609: plain store */
        }
}

/* N3K:660-694 */
static void texture_grad_pass(const int32_t* face_index_map, const float* sampling_weight_map,
                              const int32_t* sampling_index_map, const float* grad_rgb_map, float* grad_textures,
                              int B, int NF, int is, int ts) {
    const size_t P = (size_t)B * is * is;
    for (size_t i = 0; i < P; i++) {
        const int fn = face_index_map[i];
        if (fn < 0) continue;
        const int bn = (int)(i / ((size_t)is * is));
        float* gt = grad_textures + ((size_t)bn * NF + fn) * ts * ts * ts * 3;
        for (int pn = 0; pn < 8; pn++) {
            const float w = sampling_weight_map[8 * i + pn];
            const int isc = sampling_index_map[8 * i + pn];
            for (int k = 0; k < 3; k++) gt[isc * 3 + k] += w * grad_rgb_map[3 * i + k];
        }
    }
}

/* N3K:739-788 */
static void depth_grad_pass(const float* faces, const float* depth_map, const int32_t* face_index_map,
                            const float* face_inv_map, const float* weight_map, const float* grad_depth_map,
                            float* grad_faces, int B, int NF, int is) {
    const size_t P = (size_t)B * is * is;
    for (size_t i = 0; i < P; i++) {
        const int fn = face_index_map[i];
        if (fn < 0) continue;
        const int bn = (int)(i / ((size_t)is * is));
        const float* face = faces + ((size_t)bn * NF + fn) * 9;
        float* gf = grad_faces + ((size_t)bn * NF + fn) * 9;
        const float depth = depth_map[i], depth2 = depth * depth, gd = grad_depth_map[i];
        const float* finv = face_inv_map + 9 * i;
        const float* w = weight_map + 3 * i;
        for (int k = 0; k < 3; k++) {                                                   /* N3K:768-771 */
            const float zk = face[3 * k + 2];
            gf[3 * k + 2] += gd * w[k] * depth2 / (zk * zk);
        }
        float tmp[3] = {0.f, 0.f, 0.f};                                                 /* N3K:773-779 */
        for (int k = 0; k < 3; k++)
            for (int l = 0; l < 3; l++) tmp[k] += -finv[3 * l + k] / face[3 * l + 2];
        for (int k = 0; k < 3; k++)
            for (int l = 0; l < 2; l++) gf[3 * k + l] += -gd * tmp[l] * w[k] * depth2 * is / 2;
    }
}

int ref_n3mr_backward(const float* faces, const int32_t* face_index_map, const float* weight_map,
                      const float* depth_map, const float* face_inv_map, const float* rgb_map,
                      const float* alpha_map, const float* sampling_weight_map, const int32_t* sampling_index_map,
                      const float* grad_rgb_map, const float* grad_alpha_map, const float* grad_depth_map,
                      float* grad_faces, float* grad_textures, int B, int NF, int TS, int IS, float eps,
                      int return_rgb, int return_alpha, int return_depth) {
    memset(grad_faces, 0, sizeof(float) * (size_t)B * NF * 9);
    if (return_rgb || return_alpha)
        pixel_map_pass(faces, face_index_map, rgb_map, alpha_map, grad_rgb_map, grad_alpha_map, grad_faces, B, NF, IS,
                       eps, return_rgb, return_alpha);
    if (return_rgb) {
        memset(grad_textures, 0, sizeof(float) * (size_t)B * NF * TS * TS * TS * 3);
        texture_grad_pass(face_index_map, sampling_weight_map, sampling_index_map, grad_rgb_map, grad_textures, B, NF,
                          IS, TS);
    }
    if (return_depth)
        depth_grad_pass(faces, depth_map, face_index_map, face_inv_map, weight_map, grad_depth_map, grad_faces, B, NF,
                        IS);
    return 0;
}
