/*
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build).  Not part of the product.
 *
 * Host driver for the reference's own NMR ("n3mr") kernels.  The kernel source is NOT in this
 * repository: oracle/build_ref.py extracts the `cuda_header` strings of
 * /root/reference/jrender/renderer/dr/n3mr/cuda/rasterize.py (N3K:20-164, :221-298, :344-610,
 * :652-694, :731-788) into oracle/_ref/n3k_*.inc (git-ignored) and this file compiles them under
 * oracle/ref_shim/.  Restated here is only the launch glue of the five ops (fills/memsets and
 * launch geometry, N3K:166-216, :300-339, :612-648, :696-727, :790-821) run SERIALLY, so the
 * racy per-pixel spin lock of the z-buffer kernel (N3K:140-161) resolves depth ties as
 * "lowest face index first" (SURVEY.md Appendix C).
 * image_size is a template argument of the reference's forward kernel: a fixed list is instantiated.
 */
#include "ref_shim/cuda_runtime.h"
#include <cassert>
#include <iostream>
#include <stdint.h>

thread_local uint3_shim blockIdx, threadIdx;
thread_local dim3 blockDim, gridDim;
atomic_shadow g_atomic_shadow[2] = {{nullptr, 0, nullptr}, {nullptr, 0, nullptr}};   /* unused here (ref_shim) */

static inline int atomicCAS(int32_t* addr, int cmp, int val) {
    int old = *addr;
    if (old == cmp) *addr = val;
    return old;
}
static inline void __threadfence() {}

namespace n3_fwd_index {
#include "_ref/n3k_fwd_index.inc"
}
namespace n3_fwd_tex {
#include "_ref/n3k_fwd_tex.inc"
}
namespace n3_bwd_pix {
#include "_ref/n3k_bwd_pix.inc"
}
namespace n3_bwd_tex {
#include "_ref/n3k_bwd_tex.inc"
}
namespace n3_bwd_depth {
#include "_ref/n3k_bwd_depth.inc"
}

#define RUN_BLOCKS(nthreads_total, threads, CALL)                       \
    do {                                                                \
        const long nb_ = ((long)(nthreads_total) - 1) / (threads) + 1;  \
        blockDim = dim3(threads); gridDim = dim3((unsigned)nb_);        \
        for (long b_ = 0; b_ < nb_; b_++) {                             \
            blockIdx.x = (unsigned)b_;                                  \
            for (int t_ = 0; t_ < (threads); t_++) { threadIdx.x = t_; CALL; } \
        }                                                               \
    } while (0)

template <int IS, int RD>
static void run_index(const float* faces, float* faces_inv, int32_t* fim, float* wm, float* dm, float* fivm,
                      int B, int NF, float near_, float far_, int nbits, int32_t* lock) {
    RUN_BLOCKS((long)B * (1L << nbits), 256,
               (n3_fwd_index::forward_face_index_map_cuda_kernel<float, IS, 1, 1, RD>(
                   faces, faces_inv, fim, wm, dm, fivm, B, NF, near_, far_, nbits, lock)));
}

extern "C" {

/* forward: face index / weight / depth / face_inv maps (N3K:5-216) then texture sampling (N3K:219-339).
 * Layouts are the reference's: maps [B,IS,IS,(c)] bottom-up rows, textures [B,NF,ts,ts,ts,3]. */
int ref_n3mr_forward(const float* faces, const float* textures, float* faces_inv, int32_t* face_index_map,
                     float* weight_map, float* depth_map, float* face_inv_map, float* rgb_map,
                     int32_t* sampling_index_map, float* sampling_weight_map, int B, int NF, int TS, int IS,
                     float near_, float far_, float eps, int return_rgb, int return_depth) {
    const long P = (long)B * IS * IS;
    for (long i = 0; i < P; i++) { face_index_map[i] = -1; depth_map[i] = far_; }
    memset(weight_map, 0, sizeof(float) * P * 3);
    if (return_depth) memset(face_inv_map, 0, sizeof(float) * P * 9);
    int32_t* lock = (int32_t*)calloc(P, sizeof(int32_t));
    int nbits = 0;
    while ((1L << nbits) < NF) nbits++;
#define CASE(S)                                                                                        \
    case S:                                                                                            \
        if (return_depth) run_index<S, 1>(faces, faces_inv, face_index_map, weight_map, depth_map,    \
                                          face_inv_map, B, NF, near_, far_, nbits, lock);              \
        else run_index<S, 0>(faces, faces_inv, face_index_map, weight_map, depth_map, face_inv_map, B, \
                             NF, near_, far_, nbits, lock);                                            \
        break;
    switch (IS) {
        CASE(4) CASE(8) CASE(16) CASE(24) CASE(32) CASE(48) CASE(64) CASE(96) CASE(128) CASE(256) CASE(512)
        CASE(1024) CASE(2048)
        default: free(lock); return 2;
    }
#undef CASE
    free(lock);
    if (return_rgb) {
        memset(rgb_map, 0, sizeof(float) * P * 3);
        memset(sampling_index_map, 0, sizeof(int32_t) * P * 8);
        memset(sampling_weight_map, 0, sizeof(float) * P * 8);
        RUN_BLOCKS(P, 512, (n3_fwd_tex::forward_texture_sampling_cuda_kernel<float>(
                               faces, textures, face_index_map, weight_map, depth_map, rgb_map,
                               sampling_index_map, sampling_weight_map, (size_t)B, NF, IS, TS, eps)));
    }
    return 0;
}

int ref_n3mr_backward(const float* faces, int32_t* face_index_map, float* weight_map, float* depth_map,
                      float* face_inv_map, float* rgb_map, float* alpha_map, float* sampling_weight_map,
                      int32_t* sampling_index_map, float* grad_rgb_map, float* grad_alpha_map,
                      float* grad_depth_map, float* grad_faces, float* grad_textures, int B, int NF, int TS,
                      int IS, float eps, int return_rgb, int return_alpha, int return_depth) {
    const long P = (long)B * IS * IS;
    memset(grad_faces, 0, sizeof(float) * (size_t)B * NF * 9);
    if (return_rgb || return_alpha)
        RUN_BLOCKS((long)B * NF, 512, (n3_bwd_pix::backward_pixel_map_cuda_kernel<float>(
                                          faces, face_index_map, rgb_map, alpha_map, grad_rgb_map, grad_alpha_map,
                                          grad_faces, (size_t)B, (size_t)NF, IS, eps, return_rgb, return_alpha)));
    if (return_rgb) {
        memset(grad_textures, 0, sizeof(float) * (size_t)B * NF * TS * TS * TS * 3);
        RUN_BLOCKS(P, 512, (n3_bwd_tex::backward_textures_cuda_kernel<float>(
                               face_index_map, sampling_weight_map, sampling_index_map, grad_rgb_map, grad_textures,
                               (size_t)B, (size_t)NF, IS, (size_t)TS)));
    }
    if (return_depth)
        RUN_BLOCKS(P, 512, (n3_bwd_depth::backward_depth_map_cuda_kernel<float>(
                               faces, depth_map, face_index_map, face_inv_map, weight_map, grad_depth_map,
                               grad_faces, (size_t)B, (size_t)NF, IS)));
    return 0;
}

}  /* extern "C" */
