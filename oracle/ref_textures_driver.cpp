/*
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build).  Not part of the product.
 *
 * Host driver for the reference's own texel-sampler kernels (load time).  The kernel
 * source is NOT in this repository: oracle/build_ref.py extracts the `cuda_header` strings of
 * /root/reference/jrender/io/utils/load_textures.py — _load_textures_for_softras (:11-69) and
 * _load_textures_for_n3mr (:103-219, one string per (texture_wrapping, use_bilinear) pair because the
 * reference substitutes both into the source text, :220-221) — into oracle/_ref/ltx_*.inc (git-ignored)
 * and this file compiles them under oracle/ref_shim/.  Restated here is only the launch glue (:71-99,
 * :223-247): 1024 threads per block over texture_size/3 texels, run SERIALLY.
 *
 * The outputs of the reference ops are FRESH buffers (jt.code allocates `textures.shape`), so texels of
 * faces with is_update == 0 are uninitialised there; the driver takes the caller's buffer as is.
 * The n3mr kernel rewrites `faces` in place from every thread of a face (:151-172): a serial run applies the
 * wrapping ts^3 times, which is idempotent except for a coordinate that is exactly 0 under REPEAT
 * (0 -> 1 -> 0 -> ...).  The driver therefore works on a private copy of `faces` per call and the
 * comparison fixtures avoid exact zeros.
 */
#include "ref_shim/cuda_runtime.h"
#include <stdint.h>
#include <vector>

thread_local uint3_shim blockIdx, threadIdx;
thread_local dim3 blockDim, gridDim;
atomic_shadow g_atomic_shadow[2] = {{nullptr, 0, nullptr}, {nullptr, 0, nullptr}};   /* unused here (ref_shim) */

/* fmod / round on float arguments resolve to the <cmath> float overloads, like CUDA's */

namespace ltx_softras {
#include "_ref/ltx_softras.inc"
}
#define LTX_N3(W, B) namespace ltx_n3_##W##_##B {
LTX_N3(0, 0)
#include "_ref/ltx_n3mr_0_0.inc"
}
LTX_N3(0, 1)
#include "_ref/ltx_n3mr_0_1.inc"
}
LTX_N3(1, 0)
#include "_ref/ltx_n3mr_1_0.inc"
}
LTX_N3(1, 1)
#include "_ref/ltx_n3mr_1_1.inc"
}
LTX_N3(2, 0)
#include "_ref/ltx_n3mr_2_0.inc"
}
LTX_N3(2, 1)
#include "_ref/ltx_n3mr_2_1.inc"
}
LTX_N3(3, 0)
#include "_ref/ltx_n3mr_3_0.inc"
}
LTX_N3(3, 1)
#include "_ref/ltx_n3mr_3_1.inc"
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

extern "C" {

/* image [H,W,3], faces [NF,3,2], is_update [NF], textures [NF,R*R,3] (in/out) */
int ref_load_textures_softras(const float* image, const float* faces, const int32_t* is_update, float* textures,
                              int NF, int R, int H, int W) {
    const size_t texture_size = (size_t)NF * R * R * 3;                   /* textures->num */
    RUN_BLOCKS(texture_size / 3, 1024,
               (ltx_softras::load_textures_cuda_kernel<float>(image, faces, is_update, textures, texture_size,
                                                               (size_t)R, (size_t)H, (size_t)W)));
    return 0;
}

/* image [H,W,3], faces [NF,3,2] (not modified: private copy), textures [NF,ts,ts,ts,3] (in/out) */
int ref_load_textures_n3mr(const float* image, const float* faces, const int32_t* is_update, float* textures,
                           int NF, int ts, int H, int W, int wrapping, int bilinear) {
    std::vector<float> f(faces, faces + (size_t)NF * 6);
    const int textures_size = NF * ts * ts * ts * 3;
#define LTX_CALL(Wr, Bi)                                                                                      \
    if (wrapping == Wr && bilinear == Bi) {                                                                   \
        RUN_BLOCKS(textures_size / 3, 1024,                                                                   \
                   (ltx_n3_##Wr##_##Bi::load_textures_cuda_kernel<float>(image, is_update, f.data(), textures, \
                                                                          textures_size, ts, H, W)));        \
        return 0;                                                                                             \
    }
    LTX_CALL(0, 0) LTX_CALL(0, 1) LTX_CALL(1, 0) LTX_CALL(1, 1)
    LTX_CALL(2, 0) LTX_CALL(2, 1) LTX_CALL(3, 0) LTX_CALL(3, 1)
    return 1;
}

}  /* extern "C" */
