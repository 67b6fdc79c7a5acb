// TEST INFRASTRUCTURE ONLY - tests/test_device_math_on_host.py.
//
// jrender_amd/csrc/softras_backward.hip compiled with every __device__ function ALSO built for the host (the macro below; nothing
// of the file is restated): jr::backward_pair - the gradient of ONE (pixel, face) pair, everything the backward kernel computes
// between fetching a pair and reducing its components - runs here on the CPU, pair by pair over a whole image, in the order of the
// saved face-index buffer.  The sums are taken in double, so the result can be held to the exact sum of the REFERENCE's float
// per-pair terms (oracle/_ref: ref_softras_backward_exactsum): what remains is the per-pair arithmetic, no atomic order.
// The loop around the call is this file's own (the kernel's is wavefront code): border test of the saved face (SRK:1244), the
// choice of the instantiation (reciprocal-refinement quotients for well-conditioned records, plain IEEE otherwise), and where the
// colour gradient goes (single texel / sampled texel / the three vertex colours) as softras_backward.hip:506-565 has it.
// The device approximations (v_rcp_f32, v_exp_f32) are the host's exact 1/x and exp2f here: the comparison is about the formulae.
#include <hip/hip_runtime.h>
#include <math.h>
#undef __device__
#define __device__ __attribute__((host)) __attribute__((device))
#if !defined(__HIP_DEVICE_COMPILE__)
// host pass only: the gfx950 builtins the pair arithmetic uses, as plain host functions (exact where the device approximates)
static inline float hm_rcpf(float x) { return 1.0f / x; }
static inline float hm_exp2f(float x) { return exp2f(x); }
static inline float hm_fmed3f(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }
#define __builtin_amdgcn_rcpf(x) hm_rcpf(x)
#define __builtin_amdgcn_exp2f(x) hm_exp2f(x)
#define __builtin_amdgcn_fmed3f(a, b, c) hm_fmed3f(a, b, c)
#endif
#include "../../jrender_amd/csrc/softras_backward.hip"

#include <math.h>
#include <stdint.h>
#include <string.h>

namespace {

// the per-call constants as jr_api.cpp:224-251 derives them (bin geometry left out: the pair arithmetic never reads it)
jr::RasterParams host_params(int NF, int T, int IS, int K, float near_, float far_, float eps, float sigma, int dist, float dist_eps,
                             float gamma, int rgb, int alpha, int tex, int double_side) {
    jr::RasterParams p;
    memset(&p, 0, sizeof(p));
    p.B = 1; p.NF = NF; p.T = T; p.R = (int)sqrt((double)T); p.IS = IS; p.K = K;
    p.near_ = near_; p.far_ = far_; p.eps = eps; p.sigma = sigma; p.dist_eps = dist_eps; p.gamma = gamma;
    p.thr = dist_eps * sigma; p.rad = sqrtf(p.thr);
    p.dist = dist; p.rgb = rgb; p.alpha = alpha; p.tex = tex; p.double_side = double_side ? 1 : 0;
    p.far_minus_near = far_ - near_; p.near_minus_far = near_ - far_;
    p.r_sigma = 1.0f / sigma; p.r_gamma = 1.0f / gamma;
    p.r_far_minus_near = 1.0f / p.far_minus_near; p.r_near_minus_far = 1.0f / p.near_minus_far;
    p.rs_log2e = (float)(1.4426950408889634 / (double)sigma); p.rg_log2e = (float)(1.4426950408889634 / (double)gamma);
    auto in_range = [](float v) { const float a = fabsf(v); return a >= 9.094947017729282e-13f && a <= 1.099511627776e12f; };
    p.consts_safe = in_range(sigma) && in_range(gamma) && in_range(p.far_minus_near) && in_range(near_) && in_range(far_) && (eps == 0.f || in_range(eps));
    return p;
}

template <int DIST, int RGB>
int pair(const jr::RasterParams& p, const jr::FaceRec& r, const float* vc, const jr::PixelGrad& q, float xp, float yp, const float* tbase,
         float (&gv)[9], float (&wcw)[3], float& tgs, bool& tex_on) {
    return (jr::face_safe(r.meta) && p.consts_safe)
        ? jr::backward_pair<DIST, RGB, true>(p, r, vc, nullptr, q, xp, yp, tbase, gv, wcw, tgs, tex_on, nullptr)
        : jr::backward_pair<DIST, RGB, false>(p, r, vc, nullptr, q, xp, yp, tbase, gv, wcw, tgs, tex_on, nullptr);
}

}  // namespace

extern "C" {

// One image (B = 1), the reference's layouts: faces [NF,9], textures [NF,T,3], soft_colors [4,IS,IS], aggrs [2,IS,IS], ids [K,IS,IS],
// grad_rgba [4,IS,IS] -> grad_faces [NF,9], grad_textures [NF,T,3] as DOUBLE sums of the float per-pair terms.  stats[0] = pairs.
int hm_backward_image(const float* faces, const float* textures, const float* soft_colors, const float* aggrs, const int32_t* ids,
                      const float* grad_rgba, int NF, int T, int IS, int K, float near_, float far_, float eps, float sigma, int dist,
                      float dist_eps, float gamma, int rgb, int alpha, int tex, int double_side,
                      double* grad_faces, double* grad_textures, long* stats) {
    using namespace jr;
    const RasterParams p = host_params(NF, T, IS, K, near_, far_, eps, sigma, dist, dist_eps, gamma, rgb, alpha, tex, double_side);
    if (dist < 0 || dist > 2 || rgb < 0 || rgb > 1) return 1;
    const long pp = (long)IS * IS;
    memset(grad_faces, 0, sizeof(double) * (size_t)NF * 9);
    memset(grad_textures, 0, sizeof(double) * (size_t)NF * T * 3);
    stats[0] = 0;
    for (long pn = 0; pn < pp; pn++) {
        const int row = (int)(pn / IS), xi = (int)(pn % IS);
        const float xp = pixel_centre(xi, IS), yp = pixel_centre(IS - 1 - row, IS);
        PixelGrad q;
        q.g0 = grad_rgba[0 * pp + pn]; q.g1 = grad_rgba[1 * pp + pn]; q.g2 = grad_rgba[2 * pp + pn]; q.g3 = grad_rgba[3 * pp + pn];
        q.o0 = soft_colors[0 * pp + pn]; q.o1 = soft_colors[1 * pp + pn]; q.o2 = soft_colors[2 * pp + pn]; q.o3 = soft_colors[3 * pp + pn];
        q.ssum = aggrs[0 * pp + pn]; q.smax = aggrs[1 * pp + pn];
        q.r_ssum = 1.0f / q.ssum;
        for (int m = 0; m < K; m++) {
            const int fn = ids[(long)m * pp + pn];
            if (fn < 0) break;                                                   // SRK:1234-1235
            const float* f = faces + (size_t)fn * 9;
            float info[27];
            face_setup(f, info);
            FaceRec r;
            build_face_geo(r, f, info, p.rad, fn);
            if (T == 1) { r.col[0] = textures[(size_t)fn * 3]; r.col[1] = textures[(size_t)fn * 3 + 1]; r.col[2] = textures[(size_t)fn * 3 + 2]; }
            if (xp > r.xhi || xp < r.xlo || yp > r.yhi || yp < r.ylo) continue;   // SRK:1244
            const float* vc = textures + (size_t)fn * T * 3;                     // vertex colours (texture_type 'vertex': T = 3)
            float gv[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, wcw[3] = {0.f, 0.f, 0.f}, tgs = 0.f;
            bool tex_on = false;
            int texel;
            if (dist == 0) texel = rgb == 0 ? pair<0, 0>(p, r, vc, q, xp, yp, textures, gv, wcw, tgs, tex_on) : pair<0, 1>(p, r, vc, q, xp, yp, textures, gv, wcw, tgs, tex_on);
            else if (dist == 1) texel = rgb == 0 ? pair<1, 0>(p, r, vc, q, xp, yp, textures, gv, wcw, tgs, tex_on) : pair<1, 1>(p, r, vc, q, xp, yp, textures, gv, wcw, tgs, tex_on);
            else texel = rgb == 0 ? pair<2, 0>(p, r, vc, q, xp, yp, textures, gv, wcw, tgs, tex_on) : pair<2, 1>(p, r, vc, q, xp, yp, textures, gv, wcw, tgs, tex_on);
            stats[0]++;
            for (int k = 0; k < 9; k++) grad_faces[(size_t)fn * 9 + k] += (double)gv[k];
            const float tw = tex_on ? tgs : 0.f;
            double* gt = grad_textures + (size_t)fn * T * 3;
            if (tex == 1) {                                                       // the three vertex colours (softras_backward.hip:554-565)
                for (int jv = 0; jv < 3; jv++) {
                    gt[3 * jv + 0] += (double)(tw * (wcw[jv] * q.g0));
                    gt[3 * jv + 1] += (double)(tw * (wcw[jv] * q.g1));
                    gt[3 * jv + 2] += (double)(tw * (wcw[jv] * q.g2));
                }
            } else if (T == 1) {                                                  // single texel (:531)
                gt[0] += (double)(tw * q.g0); gt[1] += (double)(tw * q.g1); gt[2] += (double)(tw * q.g2);
            } else if (tex_on) {                                                  // the sampled texel (:509-514)
                const float c[3] = {tgs * q.g0, tgs * q.g1, tgs * q.g2};
                for (int k = 0; k < 3; k++) {
                    gt[texel * 3 + k] += (double)c[k];
                    if (!isfinite(c[k]))                                          // (:515-526: the reference's 0 * inf poisons the face's other texels)
                        for (int jt = 0; jt < T; jt++) if (jt != texel) gt[jt * 3 + k] += (double)NAN;
                }
            }
        }
    }
    return 0;
}

}  // extern "C"
