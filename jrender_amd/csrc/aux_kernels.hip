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

// Camera step of the 'look_at' / 'look' modes on DEVICE vertices (jrender/renderer/transform/look_at.py:3-39, look.py:3-54,
// then perspective.py:4-17 or orthogonal.py:3-16; Jittor tensor ops in the reference).  The rotation (rows = camera axes)
// and the eye of every view are O(B) host work and arrive as two small arrays; vertices are [VB,NV,3] with VB = B, or
// VB = 1 when all views share one vertex set (demo2-deform.py:45 repeats it over the batch: here it is broadcast).
// KIND 0: rotation only, 1: perspective (param = tan(angle)), 2: orthogonal (param = scale).
__device__ inline void camera_point(const float* __restrict__ v, const float* __restrict__ e,
                                    const float* __restrict__ r, float c[3]) {
    const float d0 = v[0] - e[0], d1 = v[1] - e[1], d2 = v[2] - e[2];
    c[0] = (d0 * r[0] + d1 * r[1]) + d2 * r[2];
    c[1] = (d0 * r[3] + d1 * r[4]) + d2 * r[5];
    c[2] = (d0 * r[6] + d1 * r[7]) + d2 * r[8];
}
template <int KIND>
__global__ __launch_bounds__(256) void k_camera_fwd(const float* __restrict__ v, const float* __restrict__ eye,
                                                    const float* __restrict__ rot, float* __restrict__ out,
                                                    int B, int VB, int NV, float param) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (view, vertex)
    if (i >= (long)B * NV) return;
    const int b = (int)(i / NV), n = (int)(i - (long)b * NV);
    float c[3];
    camera_point(v + ((long)(VB == 1 ? 0 : b) * NV + n) * 3, eye + b * 3, rot + b * 9, c);
    float* o = out + i * 3;
    if (KIND == 1) { o[0] = c[0] / c[2] / param; o[1] = c[1] / c[2] / param; o[2] = c[2]; }
    else if (KIND == 2) { o[0] = c[0] * param; o[1] = c[1] * param; o[2] = c[2]; }
    else { o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; }
}
// VJP of the above w.r.t. the world-space vertices.  SHARED: one thread per vertex walks the B views in order and
// writes their sum (deterministic: no atomics), else one thread per (view, vertex).
template <int KIND>
__device__ inline void camera_point_vjp(const float* __restrict__ g, const float* __restrict__ v,
                                        const float* __restrict__ e, const float* __restrict__ r, float param,
                                        float gw[3]) {
    float gc[3] = {g[0], g[1], g[2]};
    if (KIND == 1) {
        float c[3];
        camera_point(v, e, r, c);
        gc[0] = g[0] / c[2] / param;
        gc[1] = g[1] / c[2] / param;
        gc[2] = g[2] - (g[0] * c[0] + g[1] * c[1]) / (c[2] * c[2]) / param;
    } else if (KIND == 2) {
        gc[0] = g[0] * param; gc[1] = g[1] * param;
    }
    gw[0] = (gc[0] * r[0] + gc[1] * r[3]) + gc[2] * r[6];
    gw[1] = (gc[0] * r[1] + gc[1] * r[4]) + gc[2] * r[7];
    gw[2] = (gc[0] * r[2] + gc[1] * r[5]) + gc[2] * r[8];
}
template <int KIND, bool SHARED>
__global__ __launch_bounds__(256) void k_camera_bwd(const float* __restrict__ gout, const float* __restrict__ v,
                                                    const float* __restrict__ eye, const float* __restrict__ rot,
                                                    float* __restrict__ gv, int B, int NV, float param) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    float gw[3];
    if (SHARED) {
        if (i >= NV) return;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int b = 0; b < B; b++) {
            camera_point_vjp<KIND>(gout + ((long)b * NV + i) * 3, v + i * 3, eye + b * 3, rot + b * 9, param, gw);
            s0 += gw[0]; s1 += gw[1]; s2 += gw[2];
        }
        gv[i * 3 + 0] = s0; gv[i * 3 + 1] = s1; gv[i * 3 + 2] = s2;
    } else {
        if (i >= (long)B * NV) return;
        const int b = (int)(i / NV);
        camera_point_vjp<KIND>(gout + i * 3, v + i * 3, eye + b * 3, rot + b * 9, param, gw);
        gv[i * 3 + 0] = gw[0]; gv[i * 3 + 1] = gw[1]; gv[i * 3 + 2] = gw[2];
    }
}
// Scatter-add of the face-vertex gradients AND the camera VJP in one pass for views that share ONE vertex set
// (demo2-deform.py:45): the VJP is linear, so it can be applied per (view, face corner) before the sum.  One thread
// per face corner walks the B views (reads 12 B apart across the threads of a wavefront), applies view b's VJP at the
// corner's vertex and keeps the sum of eight views in registers: 3 * B / 8 atomics per corner instead of 3 * B, and no
// [B,NV,3] intermediate.
constexpr int FCB_VIEWS = 8;
template <int KIND>
__global__ __launch_bounds__(256) void k_face_camera_bwd_shared(const float* __restrict__ gfv,
                                                                const int32_t* __restrict__ faces,
                                                                const float* __restrict__ v,
                                                                const float* __restrict__ eye,
                                                                const float* __restrict__ rot, float* __restrict__ gv,
                                                                int B, int NF, float param) {
    const long fc = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (fc >= (long)NF * 3) return;
    const int vi = faces[fc];
    const float* pv = v + (long)vi * 3;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, gw[3];
    // blockIdx.y = a chunk of FCB_VIEWS views: the walk over the views is a dependent chain of divisions, so it is kept short
    const int b0 = blockIdx.y * FCB_VIEWS, b1 = min(B, b0 + FCB_VIEWS);
    for (int b = b0; b < b1; b++) {
        camera_point_vjp<KIND>(gfv + ((long)b * NF * 3 + fc) * 3, pv, eye + b * 3, rot + b * 9, param, gw);
        s0 += gw[0]; s1 += gw[1]; s2 += gw[2];
    }
    float* dst = gv + (long)vi * 3;
    atomicAdd(dst + 0, s0); atomicAdd(dst + 1, s1); atomicAdd(dst + 2, s2);
}
void launch_face_camera_backward_shared(hipStream_t st, const float* gfv, const int32_t* faces, const float* v,
                                        const float* eye, const float* rot, float* gv, int B, int NV, int NF,
                                        int kind, float param) {
    (void)hipMemsetAsync(gv, 0, sizeof(float) * (size_t)NV * 3, st);
    if (B < 1) return;
    const dim3 grid((unsigned)(((long)NF * 3 + 255) / 256), (unsigned)((B + FCB_VIEWS - 1) / FCB_VIEWS));
    if (kind == 1) k_face_camera_bwd_shared<1><<<grid, 256, 0, st>>>(gfv, faces, v, eye, rot, gv, B, NF, param);
    else if (kind == 2) k_face_camera_bwd_shared<2><<<grid, 256, 0, st>>>(gfv, faces, v, eye, rot, gv, B, NF, param);
    else k_face_camera_bwd_shared<0><<<grid, 256, 0, st>>>(gfv, faces, v, eye, rot, gv, B, NF, param);
}
void launch_camera_forward(hipStream_t st, const float* v, const float* eye, const float* rot, float* out, int B,
                           int VB, int NV, int kind, float param) {
    const unsigned grid = (unsigned)(((long)B * NV + 255) / 256);
    if (kind == 1) k_camera_fwd<1><<<grid, 256, 0, st>>>(v, eye, rot, out, B, VB, NV, param);
    else if (kind == 2) k_camera_fwd<2><<<grid, 256, 0, st>>>(v, eye, rot, out, B, VB, NV, param);
    else k_camera_fwd<0><<<grid, 256, 0, st>>>(v, eye, rot, out, B, VB, NV, param);
}
template <int KIND>
static void launch_camera_backward_kind(hipStream_t st, const float* gout, const float* v, const float* eye,
                                        const float* rot, float* gv, int B, int VB, int NV, float param) {
    if (VB == 1 && B != 1)
        k_camera_bwd<KIND, true><<<(unsigned)((NV + 255) / 256), 256, 0, st>>>(gout, v, eye, rot, gv, B, NV, param);
    else
        k_camera_bwd<KIND, false><<<(unsigned)(((long)B * NV + 255) / 256), 256, 0, st>>>(gout, v, eye, rot, gv, B, NV, param);
}
void launch_camera_backward(hipStream_t st, const float* gout, const float* v, const float* eye, const float* rot,
                            float* gv, int B, int VB, int NV, int kind, float param) {
    if (kind == 1) launch_camera_backward_kind<1>(st, gout, v, eye, rot, gv, B, VB, NV, param);
    else if (kind == 2) launch_camera_backward_kind<2>(st, gout, v, eye, rot, gv, B, VB, NV, param);
    else launch_camera_backward_kind<0>(st, gout, v, eye, rot, gv, B, VB, NV, param);
}

// neg_iou_loss (jrender/loss/iou_loss.py:1-9) and its gradient, one workgroup per view: I = sum(p*t),
// U = sum(p + t - p*t) + 1e-6 (float terms like the reference's, summed in double), iou[b] = I / U;
// grad = -(t*U - I*(1 - t)) / U^2 / divisor (divisor = the number of views the mean runs over).
__global__ __launch_bounds__(256) void k_neg_iou(const float* __restrict__ predict, const float* __restrict__ target,
                                                 float* __restrict__ iou, float* __restrict__ grad, int n,
                                                 float divisor) {
    __shared__ double s_i[256], s_u[256];
    const float* p = predict + (long)blockIdx.x * n;
    const float* t = target + (long)blockIdx.x * n;
    double si = 0.0, su = 0.0;
    for (int k = threadIdx.x; k < n; k += 256) {
        const float pt = p[k] * t[k];
        si += (double)pt;
        su += (double)(p[k] + t[k] - pt);
    }
    s_i[threadIdx.x] = si; s_u[threadIdx.x] = su;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) { s_i[threadIdx.x] += s_i[threadIdx.x + w]; s_u[threadIdx.x] += s_u[threadIdx.x + w]; }
        __syncthreads();
    }
    const float I = (float)s_i[0];
    const float U = (float)s_u[0] + 1e-6f;
    if (threadIdx.x == 0) iou[blockIdx.x] = I / U;
    if (!grad) return;
    float* g = grad + (long)blockIdx.x * n;
    for (int k = threadIdx.x; k < n; k += 256) g[k] = -(t[k] * U - I * (1.f - t[k])) / (U * U) / divisor;
}
void launch_neg_iou_loss(hipStream_t st, const float* predict, const float* target, float* iou, float* grad, int B,
                         int n, float divisor) {
    k_neg_iou<<<(unsigned)B, 256, 0, st>>>(predict, target, iou, grad, n, divisor);
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
