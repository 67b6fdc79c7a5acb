// Small kernels adjacent to the SoftRas op which the reference ran as Jittor tensor ops.
//   face_vertices gather  : jrender/structures/utils/faces_vertices.py:4-19
//   its backward          : scatter-add (was Jittor autograd)
//   2x2 mean pool (+bwd)  : nn.pool(images, 2, "mean", stride=2), softras/rasterizer.py:54-55
// All are HBM-bound elementwise/gather passes: one coalesced read, one coalesced write.
#include "jr_kernels.h"

namespace jr {

__global__ __launch_bounds__(256) void k_face_vertices_fwd(const float* __restrict__ v,
                                                           const int32_t* __restrict__ faces,
                                                           float* __restrict__ fv, int B, int NV, int NF) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (b, face, corner)
    const long total = (long)B * NF * 3;
    if (i >= total) return;
    const long b = i / ((long)NF * 3);
    const long fc = i - b * NF * 3;
    const int vi = faces[fc];
    const float* src = v + (b * NV + vi) * 3;
    float* dst = fv + i * 3;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

__global__ __launch_bounds__(256) void k_face_vertices_bwd(const float* __restrict__ gfv,
                                                           const int32_t* __restrict__ faces,
                                                           float* __restrict__ gv, int B, int NV, int NF) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * NF * 3;
    if (i >= total) return;
    const long b = i / ((long)NF * 3);
    const long fc = i - b * NF * 3;
    const int vi = faces[fc];
    float* dst = gv + (b * NV + vi) * 3;
    const float* src = gfv + i * 3;
    atomicAdd(dst + 0, src[0]); atomicAdd(dst + 1, src[1]); atomicAdd(dst + 2, src[2]);
}

// All views share ONE vertex set (demo2-deform.py:45 repeats the vertices over the batch): the per-view
// gradients collapse into [NV,3].  One thread per (face, corner) sums its B views in registers (coalesced
// reads, 36 B apart per view), then one atomic per component.
__global__ __launch_bounds__(256) void k_face_vertices_bwd_shared(const float* __restrict__ gfv,
                                                                  const int32_t* __restrict__ faces,
                                                                  float* __restrict__ gv, int B, int NF) {
    const long fc = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (fc >= (long)NF * 3) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int b = 0; b < B; b++) {
        const float* src = gfv + ((long)b * NF * 3 + fc) * 3;
        s0 += src[0]; s1 += src[1]; s2 += src[2];
    }
    float* dst = gv + (long)faces[fc] * 3;
    atomicAdd(dst + 0, s0); atomicAdd(dst + 1, s1); atomicAdd(dst + 2, s2);
}

__global__ __launch_bounds__(256) void k_avgpool_fwd(const float* __restrict__ in, float* __restrict__ out,
                                                     long total, int Ho, int Wo) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % Wo);
    const long r = i / Wo;
    const int y = (int)(r % Ho);
    const long pl = r / Ho;
    const float* s = in + (pl * (2 * Ho) + 2 * y) * (long)(2 * Wo) + 2 * x;
    const float2 a = *reinterpret_cast<const float2*>(s);
    const float2 b = *reinterpret_cast<const float2*>(s + 2 * Wo);
    out[i] = (((a.x + a.y) + b.x) + b.y) / 4;
}

__global__ __launch_bounds__(256) void k_avgpool_bwd(const float* __restrict__ gout, float* __restrict__ gin,
                                                     long total, int Ho, int Wo) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % Wo);
    const long r = i / Wo;
    const int y = (int)(r % Ho);
    const long pl = r / Ho;
    const float g = gout[i] / 4;
    float* d = gin + (pl * (2 * Ho) + 2 * y) * (long)(2 * Wo) + 2 * x;
    *reinterpret_cast<float2*>(d) = make_float2(g, g);
    *reinterpret_cast<float2*>(d + 2 * Wo) = make_float2(g, g);
}

// NMR output transform (jrender/renderer/dr/n3mr/n3mr.py:240-256, Jittor tensor ops in the reference): the
// rasteriser's maps are NHWC with BOTTOM-UP rows; the functional API returns NCHW, top-down, 2x2-mean-pooled when
// anti_aliasing.  One thread per output pixel moves all C channels (C contiguous floats per input pixel).
template <int S>
__global__ __launch_bounds__(256) void k_n3mr_image_fwd(const float* __restrict__ in, float* __restrict__ out,
                                                        int B, int H, int W, int C) {
    const int Ho = H / S, Wo = W / S;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * Ho * Wo) return;
    const int x = (int)(i % Wo);
    const long r = i / Wo;
    const int y = (int)(r % Ho);
    const int b = (int)(r / Ho);
    for (int c = 0; c < C; c++) {
        float acc = 0.f;
#pragma unroll
        for (int dy = 0; dy < S; dy++)
#pragma unroll
            for (int dx = 0; dx < S; dx++)
                acc += in[(((long)b * H + (H - 1 - (S * y + dy))) * W + (S * x + dx)) * C + c];
        out[(((long)b * C + c) * Ho + y) * Wo + x] = S == 1 ? acc : acc / (S * S);
    }
}
template <int S>
__global__ __launch_bounds__(256) void k_n3mr_image_bwd(const float* __restrict__ gout, float* __restrict__ gin,
                                                        int B, int H, int W, int C) {
    const int Ho = H / S, Wo = W / S;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per INPUT pixel (b, row, col)
    if (i >= (long)B * H * W) return;
    const int col = (int)(i % W);
    const long r = i / W;
    const int row = (int)(r % H);
    const int b = (int)(r / H);
    const int y = (H - 1 - row) / S, x = col / S;
    for (int c = 0; c < C; c++) {
        const float g = gout[(((long)b * C + c) * Ho + y) * Wo + x];
        gin[i * C + c] = S == 1 ? g : g / (S * S);
    }
}

void launch_n3mr_image_forward(hipStream_t st, const float* in, float* out, int B, int H, int W, int C, int pool) {
    const long total = (long)B * (H / pool) * (W / pool);
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (pool == 2) k_n3mr_image_fwd<2><<<grid, 256, 0, st>>>(in, out, B, H, W, C);
    else k_n3mr_image_fwd<1><<<grid, 256, 0, st>>>(in, out, B, H, W, C);
}
void launch_n3mr_image_backward(hipStream_t st, const float* gout, float* gin, int B, int H, int W, int C, int pool) {
    const long total = (long)B * H * W;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (pool == 2) k_n3mr_image_bwd<2><<<grid, 256, 0, st>>>(gout, gin, B, H, W, C);
    else k_n3mr_image_bwd<1><<<grid, 256, 0, st>>>(gout, gin, B, H, W, C);
}

void launch_face_vertices_forward(hipStream_t st, const float* v, const int32_t* faces, float* fv, int B,
                                  int NV, int NF) {
    const long total = (long)B * NF * 3;
    k_face_vertices_fwd<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(v, faces, fv, B, NV, NF);
}
void launch_face_vertices_backward(hipStream_t st, const float* gfv, const int32_t* faces, float* gv, int B,
                                   int NV, int NF) {
    (void)hipMemsetAsync(gv, 0, sizeof(float) * (size_t)B * NV * 3, st);
    const long total = (long)B * NF * 3;
    k_face_vertices_bwd<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(gfv, faces, gv, B, NV, NF);
}
void launch_face_vertices_backward_shared(hipStream_t st, const float* gfv, const int32_t* faces, float* gv,
                                          int B, int NV, int NF) {
    (void)hipMemsetAsync(gv, 0, sizeof(float) * (size_t)NV * 3, st);
    if (B < 1) return;
    const long total = (long)NF * 3;
    k_face_vertices_bwd_shared<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(gfv, faces, gv, B, NF);
}
void launch_avgpool2x2_forward(hipStream_t st, const float* in, float* out, int planes, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)planes * Ho * Wo;
    k_avgpool_fwd<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(in, out, total, Ho, Wo);
}
void launch_avgpool2x2_backward(hipStream_t st, const float* gout, float* gin, int planes, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)planes * Ho * Wo;
    k_avgpool_bwd<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(gout, gin, total, Ho, Wo);
}

}  // namespace jr

// ---- self-test: div_known<true>(a, b, RN(1/b)) == a / b on its guarantee domain ----------------
// a == 0 or 2^-80 <= |a| <= 2^60, 2^-40 <= |b| <= 2^40 (softras_device.h).  Counts mismatching bits.
namespace jr {
__device__ inline uint32_t xs32(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
__device__ inline float rnd_float(uint32_t& s, int emin, int emax) {
    const uint32_t m = xs32(s) & 0x7fffffu;
    const uint32_t e = 127u + (uint32_t)(emin + (int)(xs32(s) % (uint32_t)(emax - emin + 1)));
    const uint32_t sign = xs32(s) & 0x80000000u;
    return __builtin_bit_cast(float, sign | (e << 23) | m);
}
__global__ __launch_bounds__(256) void k_selftest_div(unsigned long long per_thread, uint32_t seed,
                                                      unsigned long long* mismatches) {
    uint32_t s = seed ^ (0x9e3779b9u * (uint32_t)(blockIdx.x * blockDim.x + threadIdx.x + 1));
    unsigned long long bad = 0;
    for (unsigned long long i = 0; i < per_thread; i++) {
        const float b = rnd_float(s, -40, 40);
        float a = rnd_float(s, -80, 60);
        const uint32_t kind = xs32(s) & 15u;
        if (kind == 0) a = 0.f;
        else if (kind < 6) a = fabsf(rnd_float(s, -24, 0)) * 0.999f;      // like clipped barycentrics
        const float q_ref = a / b;
        const float q = div_known<true>(a, b, 1.0f / b);
        bad += __builtin_bit_cast(uint32_t, q) != __builtin_bit_cast(uint32_t, q_ref);
    }
    if (bad) atomicAdd(mismatches, bad);
}
void launch_selftest_div(hipStream_t st, unsigned long long n, uint32_t seed, unsigned long long* mismatches) {
    const int blocks = 2048, threads = 256;
    const unsigned long long per = (n + (unsigned long long)blocks * threads - 1) / ((unsigned long long)blocks * threads);
    k_selftest_div<<<blocks, threads, 0, st>>>(per, seed, mismatches);
}
}  // namespace jr

// ---- self-test: recip_exact(x) == 1.0f / x for EVERY float with exponent in [-40, 40] -----------
namespace jr {
__global__ __launch_bounds__(256) void k_selftest_rcp(unsigned long long* mismatches) {
    const unsigned m = blockIdx.x * blockDim.x + threadIdx.x;       // all 2^23 mantissas
    unsigned long long bad = 0;
    for (int e = -40; e <= 40; e++)
        for (unsigned s = 0; s < 2; s++) {
            const float x = __builtin_bit_cast(float, (s << 31) | ((unsigned)(127 + e) << 23) | m);
            bad += __builtin_bit_cast(unsigned, recip_exact(x)) != __builtin_bit_cast(unsigned, 1.0f / x);
        }
    if (bad) atomicAdd(mismatches, bad);
}
void launch_selftest_rcp(hipStream_t st, unsigned long long* mismatches) {
    k_selftest_rcp<<<(1u << 23) / 256, 256, 0, st>>>(mismatches);
}
}  // namespace jr
