/*
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build).  Not part of the product.
 *
 * Host driver for the reference's own SoftRas kernels.  The kernel source is
 * NOT in this repository: oracle/build_ref.py extracts the `cuda_header`
 * strings from /root/reference/jrender/renderer/dr/softras/cuda/soft_rasterize.py
 * at build time into oracle/_ref/srk_fwd.inc / srk_bwd.inc (git-ignored) and
 * this file compiles them under oracle/ref_shim/.  What this driver restates
 * is only the launch glue of the reference op:
 *   forward : SRK:467-516  (4 memsets, inv kernel 512 thr/block over B*NF,
 *                           pixel kernel 512 thr/block over B*IS*IS)
 *   backward: SRK:1374-1411 (2 memsets, pixel kernel 512 thr/block)
 * Scalar parameters arrive as float, exactly like the literals the reference
 * interpolates into its launch (SRK:485-516).
 *
 * Build flags (oracle/build_ref.py): -O2 -ffp-contract=off -fno-fast-math.
 */
#include "ref_shim/cuda_runtime.h"
#include <omp.h>
#include <stdint.h>

thread_local uint3_shim blockIdx, threadIdx;
thread_local dim3 blockDim, gridDim;
atomic_shadow g_atomic_shadow[2] = {{nullptr, 0, nullptr}, {nullptr, 0, nullptr}};

namespace ref_fwd {
#include "_ref/srk_fwd.inc"
}
namespace ref_bwd {
#include "_ref/srk_bwd.inc"
}

static void set_threads(int nthreads) {
    omp_set_num_threads(nthreads > 0 ? nthreads : omp_get_num_procs());   // 0 = all host cores
}

extern "C" {

int ref_softras_forward(const float* faces, const float* textures,
                        float* faces_info, float* aggrs_info, float* soft_colors,
                        int32_t* faces_id_buffer,
                        int B, int NF, int T, int IS, int K,
                        float near_, float far_, float eps, float sigma_val,
                        int func_id_dist, float dist_eps, float gamma_val,
                        int func_id_rgb, int func_id_alpha,
                        int texture_sample_type, int double_side, int nthreads) {
    set_threads(nthreads);
    const size_t P = (size_t)B * IS * IS;
    memset(faces_info, 0, sizeof(float) * (size_t)B * NF * 27);
    memset(aggrs_info, 0, sizeof(float) * P * 2);
    memset(soft_colors, 0, sizeof(float) * P * 4);
    memset(faces_id_buffer, 0xff, sizeof(int32_t) * P * K);
    const int texture_res = (int)sqrt((double)T);
    const int threads = 512;
    const long nb1 = ((long)B * NF - 1) / threads + 1;
#pragma omp parallel for schedule(static)
    for (long b = 0; b < nb1; b++) {
        blockDim = dim3(threads); gridDim = dim3((unsigned)nb1);
        blockIdx.x = (unsigned)b;
        for (int t = 0; t < threads; t++) {
            threadIdx.x = t;
            ref_fwd::forward_soft_rasterize_inv_cuda_kernel<float>(faces, faces_info, B, NF, IS);
        }
    }
    const long nb2 = ((long)P - 1) / threads + 1;
#pragma omp parallel for schedule(dynamic, 4)
    for (long b = 0; b < nb2; b++) {
        blockDim = dim3(threads); gridDim = dim3((unsigned)nb2);
        blockIdx.x = (unsigned)b;
        for (int t = 0; t < threads; t++) {
            threadIdx.x = t;
            ref_fwd::forward_soft_rasterize_cuda_kernel<float>(
                faces, textures, faces_info, aggrs_info, soft_colors, faces_id_buffer,
                B, NF, IS, K, T, texture_res, near_, far_, eps, sigma_val,
                func_id_dist, dist_eps, gamma_val, func_id_rgb, func_id_alpha,
                texture_sample_type, double_side);
        }
    }
    return 0;
}

/* faces_id_buffer_t is the TRANSPOSED buffer [B, IS, IS, K] exactly as the
 * reference passes it to the backward op (SRW:108, SRK:1226). */
int ref_softras_backward(const float* faces, const float* textures, const float* soft_colors,
                         const float* faces_info, const float* aggrs_info,
                         const int32_t* faces_id_buffer_t, const float* grad_soft_colors,
                         float* grad_faces, float* grad_textures,
                         int B, int NF, int T, int IS, int K,
                         float near_, float far_, float eps, float sigma_val,
                         int func_id_dist, float dist_eps, float gamma_val,
                         int func_id_rgb, int func_id_alpha,
                         int texture_sample_type, int double_side, int nthreads) {
    set_threads(nthreads);
    const size_t P = (size_t)B * IS * IS;
    memset(grad_faces, 0, sizeof(float) * (size_t)B * NF * 9);
    memset(grad_textures, 0, sizeof(float) * (size_t)B * NF * T * 3);
    const int texture_res = (int)sqrt((double)T);
    const int threads = 512;
    const long nb = ((long)P - 1) / threads + 1;
#pragma omp parallel for schedule(dynamic, 4)
    for (long b = 0; b < nb; b++) {
        blockDim = dim3(threads); gridDim = dim3((unsigned)nb);
        blockIdx.x = (unsigned)b;
        for (int t = 0; t < threads; t++) {
            threadIdx.x = t;
            ref_bwd::backward_soft_rasterize_cuda_kernel<float>(
                faces, faces_id_buffer_t, textures, soft_colors, faces_info, aggrs_info,
                grad_faces, grad_textures, const_cast<float*>(grad_soft_colors),
                B, NF, IS, K, T, texture_res, near_, far_, eps, sigma_val,
                func_id_dist, dist_eps, gamma_val, func_id_rgb, func_id_alpha,
                texture_sample_type, (bool)double_side);
        }
    }
    return 0;
}

/* The float backward once more, with every float atomic ALSO summed in double (ref_shim: atomic_shadow): grad_*_sum64 is
 * the exact sum of the reference's float per-pair terms, i.e. the reference's gradient without the order noise of its
 * float atomics.  Distance of any float run of the reference from it = pure order noise; distance of the HIP kernels
 * from it = order noise + their arithmetic differences. */
int ref_softras_backward_exactsum(const float* faces, const float* textures, const float* soft_colors,
                                  const float* faces_info, const float* aggrs_info,
                                  const int32_t* faces_id_buffer_t, const float* grad_soft_colors,
                                  float* grad_faces, float* grad_textures, double* grad_faces_sum64, double* grad_textures_sum64,
                                  int B, int NF, int T, int IS, int K,
                                  float near_, float far_, float eps, float sigma_val,
                                  int func_id_dist, float dist_eps, float gamma_val,
                                  int func_id_rgb, int func_id_alpha,
                                  int texture_sample_type, int double_side, int nthreads) {
    memset(grad_faces_sum64, 0, sizeof(double) * (size_t)B * NF * 9);
    memset(grad_textures_sum64, 0, sizeof(double) * (size_t)B * NF * T * 3);
    g_atomic_shadow[0] = {grad_faces, (size_t)B * NF * 9, grad_faces_sum64};
    g_atomic_shadow[1] = {grad_textures, (size_t)B * NF * T * 3, grad_textures_sum64};
    const int rc = ref_softras_backward(faces, textures, soft_colors, faces_info, aggrs_info, faces_id_buffer_t, grad_soft_colors,
                                        grad_faces, grad_textures, B, NF, T, IS, K, near_, far_, eps, sigma_val, func_id_dist,
                                        dist_eps, gamma_val, func_id_rgb, func_id_alpha, texture_sample_type, double_side, nthreads);
    g_atomic_shadow[0] = {nullptr, 0, nullptr};
    g_atomic_shadow[1] = {nullptr, 0, nullptr};
    return rc;
}

/* The SAME reference kernel instantiated for double (backward_soft_rasterize_cuda_kernel is a template over scalar_t,
 * SRK:1177): every float input is widened exactly, all per-pair arithmetic and the atomic sums run in double.  Its
 * result is the gradient as a function of the saved tensors that both float implementations - the reference's own
 * float instantiation and the HIP kernels - approximate; bench.py's `parity.vs_f64` measures both against it. */
int ref_softras_backward_f64(const float* faces, const float* textures, const float* soft_colors,
                             const float* faces_info, const float* aggrs_info,
                             const int32_t* faces_id_buffer_t, const float* grad_soft_colors,
                             double* grad_faces, double* grad_textures,
                             int B, int NF, int T, int IS, int K,
                             float near_, float far_, float eps, float sigma_val,
                             int func_id_dist, float dist_eps, float gamma_val,
                             int func_id_rgb, int func_id_alpha,
                             int texture_sample_type, int double_side, int nthreads) {
    set_threads(nthreads);
    const size_t P = (size_t)B * IS * IS;
    auto widen = [](const float* a, size_t n) {
        double* d = (double*)malloc(sizeof(double) * (n ? n : 1));
        if (d) for (size_t i = 0; i < n; i++) d[i] = (double)a[i];
        return d;
    };
    double* f = widen(faces, (size_t)B * NF * 9);
    double* tx = widen(textures, (size_t)B * NF * T * 3);
    double* sc = widen(soft_colors, P * 4);
    double* fi = widen(faces_info, (size_t)B * NF * 27);
    double* ag = widen(aggrs_info, P * 2);
    double* gs = widen(grad_soft_colors, P * 4);
    int rc = (f && tx && sc && fi && ag && gs) ? 0 : 1;
    if (!rc) {
        memset(grad_faces, 0, sizeof(double) * (size_t)B * NF * 9);
        memset(grad_textures, 0, sizeof(double) * (size_t)B * NF * T * 3);
        const int texture_res = (int)sqrt((double)T);
        const int threads = 512;
        const long nb = ((long)P - 1) / threads + 1;
#pragma omp parallel for schedule(dynamic, 4)
        for (long b = 0; b < nb; b++) {
            blockDim = dim3(threads); gridDim = dim3((unsigned)nb);
            blockIdx.x = (unsigned)b;
            for (int t = 0; t < threads; t++) {
                threadIdx.x = t;
                ref_bwd::backward_soft_rasterize_cuda_kernel<double>(
                    f, faces_id_buffer_t, tx, sc, fi, ag, grad_faces, grad_textures, gs,
                    B, NF, IS, K, T, texture_res, near_, far_, eps, sigma_val,
                    func_id_dist, dist_eps, gamma_val, func_id_rgb, func_id_alpha,
                    texture_sample_type, (bool)double_side);
            }
        }
    }
    free(f); free(tx); free(sc); free(fi); free(ag); free(gs);
    return rc;
}

int ref_num_procs(void) { return omp_get_num_procs(); }

}  /* extern "C" */
