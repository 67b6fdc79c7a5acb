/*
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build).  Not part of the product.
 *
 * Host driver for the reference's own COARSE-TO-FINE SoftRas forward (jrender/renderer/dr/softras/cuda/
 * soft_rasterize_coarse_to_fine.py, C2F).  The kernel source is NOT in this repository: oracle/build_ref.py extracts the
 * `cuda_header` string at build time into oracle/_ref/c2f_fwd.inc (git-ignored) and the literal the reference interpolates
 * for the bin margin (`blur_radius`, C2F:15) into oracle/_ref/c2f_params.h.  What this file restates is the launch glue
 * of C2F:765-873 only:
 *   memsets (outputs 0, ids -1)                                   C2F:774-777
 *   forward_soft_rasterize_inv_cuda_kernel, 512 thr/block         C2F:793-798
 *   TriangleBoundingBoxKernel <<<128, 256>>> (grid-stride)        C2F:800-807
 *   elems_per_bin = 0, bin_elems = -1                             C2F:809-818
 *   RasterizeCoarseCudaKernel, chunk 512, shared = bins^2*512/8   C2F:820-835   <- launched here as <<<1, 1>>>
 *   forward_soft_rasterize_cuda_kernel, 128 thr/block over B * (bins * bin_size)^2   C2F:841-871
 *
 * Why <<<1, 1>>> for the coarse kernel: the reference launches it as <<<64, 512>>>; its blocks race for segments of the
 * bin lists (`atomicAdd(elems_per_bin + ..., count)`, C2F:236), so the order of a bin's faces - and with it the K-nearest
 * buffer, the alpha product and the online softmax of every pixel - depends on block scheduling.  All its loops are
 * block- / grid-stride loops (`chunk += gridDim.x`, `e += blockDim.x`, `byx += blockDim.x`), so one block of one thread is
 * a legal launch of the same kernel: __syncthreads() has nothing to wait for, chunks are taken in ascending order, faces
 * inside a chunk are written in ascending order (C2F:263-272) - the lists come out ASCENDING BY FACE ID, the one order in
 * which the binned forward can be compared with the bin_size = 0 forward at all.
 */
#include "ref_shim/cuda_runtime.h"
#include <omp.h>
#include <vector>

thread_local uint3_shim blockIdx, threadIdx;
thread_local dim3 blockDim, gridDim;
atomic_shadow g_atomic_shadow[2] = {{nullptr, 0, nullptr}, {nullptr, 0, nullptr}};
namespace ref_c2f {
char sbuf[27 * 27 * 512 / 8];                 /* the coarse kernel's dynamic shared memory (`extern __shared__ char sbuf[]` inside the kernel, C2F:141): bins^2 x chunk_size bits, bins <= 27 (C2F:17) */
}

#include "_ref/c2f_params.h"                   /* C2F_BLUR_RADIUS: the literal of C2F:15 as the reference interpolates it */
namespace ref_c2f {
#include "_ref/c2f_fwd.inc"
}

extern "C" {

/* elems_per_bin_out [B, bins, bins] and bin_elems_out [B, bins, bins, M] are optional (nullptr): the lists the per-pixel kernel read. */
int ref_softras_forward_c2f(const float* faces, const float* textures,
                            float* faces_info, float* aggrs_info, float* soft_colors, int32_t* faces_id_buffer,
                            int B, int NF, int T, int IS, int K,
                            float near_, float far_, float eps, float sigma_val,
                            int func_id_dist, float dist_eps, float gamma_val,
                            int func_id_rgb, int func_id_alpha, int texture_sample_type, int double_side,
                            int bin_size, int max_elems_per_bin, int32_t* elems_per_bin_out, int32_t* bin_elems_out, int nthreads) {
    const int bins = 1 + (IS - 1) / bin_size;
    if (bins > 27 || bin_size <= 0 || max_elems_per_bin <= 0) return 1;           /* C2F:16-18 raises ValueError */
    omp_set_num_threads(nthreads > 0 ? nthreads : omp_get_num_procs());
    const size_t P = (size_t)B * IS * IS, F = (size_t)B * NF;
    memset(faces_info, 0, sizeof(float) * F * 27);
    memset(aggrs_info, 0, sizeof(float) * P * 2);
    memset(soft_colors, 0, sizeof(float) * P * 4);
    memset(faces_id_buffer, 0xff, sizeof(int32_t) * P * K);
    const int texture_res = (int)sqrt((double)T);

    const long nb1 = ((long)F - 1) / 512 + 1;
    for (long b = 0; b < nb1; b++) {
        blockDim = dim3(512); gridDim = dim3((unsigned)nb1); blockIdx.x = (unsigned)b;
        for (int t = 0; t < 512; t++) {
            threadIdx.x = t;
            ref_c2f::forward_soft_rasterize_inv_cuda_kernel<float>(faces, faces_info, B, NF, IS);
        }
    }
    std::vector<float> bboxes(F * 4);
    bool* should_skip = (bool*)calloc(F ? F : 1, sizeof(bool));
    blockDim = dim3(1); gridDim = dim3(1); blockIdx.x = 0; threadIdx.x = 0;      /* grid-stride loop: any launch computes the same boxes */
    ref_c2f::TriangleBoundingBoxKernel(faces, (int)F, C2F_BLUR_RADIUS, bboxes.data(), should_skip);

    const size_t nbin = (size_t)B * bins * bins;
    std::vector<int> elems_per_bin(nbin, 0), bin_elems(nbin * max_elems_per_bin, -1);
    blockDim = dim3(1); gridDim = dim3(1); blockIdx.x = 0; threadIdx.x = 0;      /* see the header: ascending lists */
    ref_c2f::RasterizeCoarseCudaKernel(bboxes.data(), should_skip, B, NF, IS, bin_size, 512, max_elems_per_bin,
                                       elems_per_bin.data(), bin_elems.data());
    free(should_skip);
    if (elems_per_bin_out) memcpy(elems_per_bin_out, elems_per_bin.data(), sizeof(int) * nbin);
    if (bin_elems_out) memcpy(bin_elems_out, bin_elems.data(), sizeof(int) * nbin * max_elems_per_bin);

    const long total = (long)B * bin_size * bin_size * bins * bins;
    const long nb4 = (total - 1) / 128 + 1;
    const int* be = bin_elems.data();
    const int* epb = elems_per_bin.data();
#pragma omp parallel for schedule(dynamic, 16)
    for (long b = 0; b < nb4; b++) {
        blockDim = dim3(128); gridDim = dim3((unsigned)nb4); blockIdx.x = (unsigned)b;
        for (int t = 0; t < 128; t++) {
            threadIdx.x = t;
            ref_c2f::forward_soft_rasterize_cuda_kernel<float>(
                faces, textures, faces_info, be, epb, aggrs_info, soft_colors, faces_id_buffer,
                bins, max_elems_per_bin, bin_size, B, NF, IS, K, T, texture_res, near_, far_, eps, sigma_val,
                func_id_dist, dist_eps, gamma_val, func_id_rgb, func_id_alpha, texture_sample_type, double_side);
        }
    }
    return 0;
}

}  /* extern "C" */
