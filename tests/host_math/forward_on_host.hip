// TEST INFRASTRUCTURE ONLY - tests/test_device_math_on_host.py.
//
// jrender_amd/csrc/softras_forward.hip compiled with every __device__ function ALSO built for the host (the macro below; nothing of
// the file is restated): jr::forward_pair - one (pixel, face) step of the raster loop: distance, cull, coverage, alpha, depth cull,
// K-nearest insert with its id store, online softmax / hard colour - together with init_colour_state, KBuffer and final_colour is the
// whole per-pixel state machine of the forward kernels.  Here it runs on the CPU: every pixel of an image walks the faces in ascending
// order (what the ordered lists guarantee on the device), through the same instantiation choice as the kernels (reciprocal-refinement
// quotients for well-conditioned records, DIST = 3 where the inside distance feeds a decision).  The loop around the call is this
// file's own (the kernels' is wavefront code: lists, LDS staging, ballots) and tests the border box of the record per pixel instead of
// per tile.  The device approximations (v_rcp_f32, v_exp_f32) are the host's exact 1/x and exp2f: the face-index buffer must come out
// bit for bit, colours within the colour tolerance.
#include <hip/hip_runtime.h>
#include <math.h>
#undef __device__
#define __device__ __attribute__((host)) __attribute__((device))
#if !defined(__HIP_DEVICE_COMPILE__)
static inline float hm_rcpf(float x) { return 1.0f / x; }
static inline float hm_exp2f(float x) { return exp2f(x); }
static inline float hm_fmed3f(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }
#define __builtin_amdgcn_rcpf(x) hm_rcpf(x)
#define __builtin_amdgcn_exp2f(x) hm_exp2f(x)
#define __builtin_amdgcn_fmed3f(a, b, c) hm_fmed3f(a, b, c)
#define __builtin_amdgcn_ballot_w64(p) ((p) ? 1ull : 0ull)
#define __builtin_amdgcn_mbcnt_lo(a, b) (0u)
#define __builtin_amdgcn_mbcnt_hi(a, b) (0u)
#endif
#include "../../jrender_amd/csrc/softras_forward.hip"

#include <stdint.h>
#include <string.h>

namespace {

jr::RasterParams host_params(int NF, int T, int IS, int K, float near_, float far_, float eps, float sigma, int dist, float dist_eps,
                             float gamma, int rgb, int alpha, int tex, int double_side) {      // jr_api.cpp:224-251
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

template <int DIST, int RGB, int KCAP>
void image(const jr::RasterParams& p, const jr::FaceGeo* rec, const float* textures, float* aggrs, float* rgba, int32_t* ids) {
    using namespace jr;
    const long pp = (long)p.IS * p.IS;
    for (long pn = 0; pn < pp; pn++) {
        const int row = (int)(pn / p.IS), xi = (int)(pn % p.IS);
        const float xp = pixel_centre(xi, p.IS), yp = pixel_centre(p.IS - 1 - row, p.IS);
        PixelState<KCAP> s;
        init_colour_state<RGB>(p, s);
        s.q.init(p.K, ids, (unsigned)pn, (unsigned)pp);
        PathCount pc;
        for (int fn = 0; fn < p.NF; fn++) {
            const FaceGeo& r = rec[fn];
            if (xp > r.xhi || xp < r.xlo || yp > r.yhi || yp < r.ylo) continue;        // SRK:316 (the kernels: exact pixel rectangle + per-tile ballots)
            const float* vc = textures + (size_t)fn * p.T * 3;
            if (face_safe(r.meta) && p.consts_safe) forward_pair<DIST, RGB, true, KCAP>(p, r, vc, textures, xp, yp, s, pc);
            else forward_pair<DIST, RGB, false, KCAP>(p, r, vc, textures, xp, yp, s, pc);
        }
        float o[6];
        final_colour<RGB>(p, s, o);
        for (int k = 0; k < 4; k++) rgba[k * pp + pn] = o[k];
        aggrs[pn] = o[4]; aggrs[pp + pn] = o[5];
    }
}

}  // namespace

extern "C" {

// One image (B = 1), the reference's layouts -> faces_info [NF,27], aggrs [2,IS,IS], rgba [4,IS,IS], ids [K,IS,IS] (K <= 64).
int hm_forward_image(const float* faces, const float* textures, int NF, int T, int IS, int K, float near_, float far_, float eps, float sigma,
                     int dist, float dist_eps, float gamma, int rgb, int alpha, int tex, int double_side,
                     float* faces_info, float* aggrs, float* rgba, int32_t* ids) {
    using namespace jr;
    if (K < 1 || K > 64 || dist < 0 || dist > 2 || rgb < 0 || rgb > 1 || !KBuffer<16>::IDS_GLOBAL) return 1;
    const RasterParams p = host_params(NF, T, IS, K, near_, far_, eps, sigma, dist, dist_eps, gamma, rgb, alpha, tex, double_side);
    FaceGeo* rec = new FaceGeo[NF > 0 ? NF : 1];
    for (int fn = 0; fn < NF; fn++) {
        face_setup(faces + (size_t)fn * 9, faces_info + (size_t)fn * 27);
        build_face_geo(rec[fn], faces + (size_t)fn * 9, faces_info + (size_t)fn * 27, p.rad, fn);
        if (T == 1) for (int c = 0; c < 3; c++) rec[fn].col[c] = textures[(size_t)fn * 3 + c];
    }
    for (long i = 0; i < (long)K * IS * IS; i++) ids[i] = -1;            // slots that are never filled (the kernels: store_ids at the end)
    // the kernels' choice of the distance instantiation (softras_forward.hip: launch_softras_forward)
    const int d = (p.dist == 2 && ((p.alpha == 0 && tune::fwd_hard_exact) || p.sigma < tune::fwd_exact_inside_sigma)) ? 3 : p.dist;
    // ... and of the K-buffer capacity: 16, 32 or 64 depth registers (softras_forward.hip:1481-1483)
#define HM_RUN_K(D, KC) (rgb == 0 ? image<D, 0, KC>(p, rec, textures, aggrs, rgba, ids) : image<D, 1, KC>(p, rec, textures, aggrs, rgba, ids))
#define HM_RUN(D) (K <= 16 ? HM_RUN_K(D, 16) : (K <= 32 ? HM_RUN_K(D, 32) : HM_RUN_K(D, 64)))
    if (d == 0) HM_RUN(0); else if (d == 1) HM_RUN(1); else if (d == 2) HM_RUN(2); else HM_RUN(3);
#undef HM_RUN
#undef HM_RUN_K
    delete[] rec;
    return 0;
}

}  // extern "C"
