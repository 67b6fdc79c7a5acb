// The optimiser side of the deformation loop (BASELINE configs[3]) on the device - round 5.
//
// The reference keeps all of it on the GPU as Jittor tensor ops + autograd:
//   Model.execute   demo2-deform.py:35-41   vertices = parametrisation(template, displace, center)
//   nn.Adam         demo2-deform.py:72      (restated in jrender_amd/optim.py; these kernels mirror that restatement
//                                           operation by operation, in float, without contraction)
//   loss terms      demo2-deform.py:85-88   neg-IoU + 0.03 Laplacian + 0.0003 flatten, one scalar per iteration
// Round 4 ran them in NumPy between two PCIe hops per iteration (VERDICT r4 missing 5).  All three are O(nv) element-wise
// work on a few thousand values: one small launch each, nothing here is bandwidth- or compute-bound.
#include "jr_kernels.h"

namespace jr {

__device__ inline float sign_of(float t) { return t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f); }

// demo2-deform.py:36-41:  base = log(|t| / (1 - |t|));  c = tanh(center);  u = sigmoid(base + displace) * sign(t);
//                         v = relu(u) * (1 - c) - relu(-u) * (c + 1) + c          (t = 0.5 * template vertex, |t| < 1)
__device__ inline void deform_terms(float t, float d, float cen, float& s, float& sg, float& c, float& u) {
    const float a = fabsf(t);
    const float base = logf(a / (1.f - a));
    c = tanhf(cen);
    s = 1.0f / (1.0f + expf(-(base + d)));
    sg = sign_of(t);
    u = s * sg;
}

__global__ __launch_bounds__(256) void k_deform_forward(const float* __restrict__ tmpl, const float* __restrict__ displace,
                                                        const float* __restrict__ center, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + (int)threadIdx.x;
    if (i >= n) return;
    float s, sg, c, u;
    deform_terms(tmpl[i], displace[i], center[i % 3], s, sg, c, u);
    out[i] = (fmaxf(u, 0.f) * (1.f - c) - fmaxf(-u, 0.f) * (c + 1.f)) + c;
}

// The VJP of the above for the upstream gradient g = w0 g0 + w1 g1 + w2 g2 (the silhouette term and the two
// regularisers, demo2-deform.py:85-88: combined here instead of in three more launches):
//   d/d displace = g * ((u > 0)(1 - c) + (u < 0)(c + 1)) * sign * s (1 - s)
//   d/d center_k = (1 - c_k^2) * sum over the vertices of g * (1 - relu(u) - relu(-u))       (two-stage double sum)
__global__ __launch_bounds__(256) void k_deform_backward(const float* __restrict__ tmpl, const float* __restrict__ displace,
                                                         const float* __restrict__ center, const float* __restrict__ g0,
                                                         float w0, const float* __restrict__ g1, float w1,
                                                         const float* __restrict__ g2, float w2,
                                                         float* __restrict__ grad_displace, float* __restrict__ grad_center,
                                                         double* acc, unsigned* ticket, int nv) {
    __shared__ double s_red[3][4];
    const int v = blockIdx.x * 256 + (int)threadIdx.x;
    double part[3] = {0.0, 0.0, 0.0};
    if (v < nv) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int i = v * 3 + k;
            float g = w0 * g0[i];
            if (g1) g = g + w1 * g1[i];
            if (g2) g = g + w2 * g2[i];
            float s, sg, c, u;
            deform_terms(tmpl[i], displace[i], center[k], s, sg, c, u);
            const float gu = g * ((u > 0.f ? (1.f - c) : 0.f) + (u < 0.f ? (c + 1.f) : 0.f));
            grad_displace[i] = gu * sg * s * (1.f - s);
            part[k] = (double)(g * ((1.f - fmaxf(u, 0.f)) - fmaxf(-u, 0.f)));
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double x = part[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
        if (lane == 0) s_red[k][w] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) atomicAdd(&acc[k], (s_red[k][0] + s_red[k][1]) + (s_red[k][2] + s_red[k][3]));
        __threadfence();
        if (atomicAdd(ticket, 1u) == gridDim.x - 1u) {             // the last workgroup publishes and clears (see loss_kernels.hip)
            __threadfence();
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const double total = atomicAdd(&acc[k], 0.0);
                const float c = tanhf(center[k]);
                grad_center[k] = (float)total * (1.f - c * c);
                atomicExch(reinterpret_cast<unsigned long long*>(&acc[k]), 0ull);
            }
            atomicExch(ticket, 0u);
        }
    }
}

// One Adam step of jrender_amd/optim.py (the restatement of demo2-deform.py:72), in place:
//   m = b0 m + (1 - b0) g;  v = b1 v + ((1 - b1) g) g;  p -= ((lr / c0) m) / (sqrt(v / c1) + eps)
// c0 = 1 - b0^t, c1 = 1 - b1^t come from the host (it owns the step count); float operations in the mirror's order.
// `iteration` (nullable): the step number lives on the DEVICE (step = *iteration + 1) - a launch captured into a HIP graph must not
// freeze a host scalar; the two bias corrections are then formed here, in double like the host forms them, by one thread per workgroup.
__global__ __launch_bounds__(256) void k_adam_step(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, float lr_over_c0, float b0,
                                                   float one_minus_b0, float b1, float one_minus_b1, float c1, float eps,
                                                   float weight_decay, const int* __restrict__ iteration, double lr, double beta0,
                                                   double beta1) {
    __shared__ float s_sched[2];
    if (iteration) {
        if (threadIdx.x == 0) {
            const int step = *iteration + 1;
            s_sched[0] = (float)(lr / (1.0 - pow(beta0, (double)step)));
            s_sched[1] = (float)(1.0 - pow(beta1, (double)step));
        }
        __syncthreads();
        lr_over_c0 = s_sched[0]; c1 = s_sched[1];
    }
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float gi = g[i];
    if (weight_decay != 0.f) gi = gi + weight_decay * p[i];
    float mi = m[i] * b0;
    mi = mi + one_minus_b0 * gi;
    float vi = v[i] * b1;
    vi = vi + (one_minus_b1 * gi) * gi;
    m[i] = mi; v[i] = vi;
    p[i] = p[i] - (lr_over_c0 * mi) / (sqrtf(vi / c1) + eps);
}

// dst[0] = (accumulate ? dst[0] : 0) + bias + scale * sum(src[0..n)): the per-iteration loss terms stay on the device
// (a history array the host reads once per N iterations); one workgroup, double sum.
// `iteration` (nullable) + stride: dst advances by stride * *iteration floats (the history row of a captured iteration).
__global__ __launch_bounds__(256) void k_scalar_accumulate(float* __restrict__ dst, const float* __restrict__ src, int n,
                                                           float scale, float bias, int accumulate,
                                                           const int* __restrict__ iteration, int stride) {
    __shared__ double s_red[4];
    if (iteration) dst += (size_t)stride * (size_t)*iteration;
    double part = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) part += (double)src[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double total = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        dst[0] = (accumulate ? dst[0] : 0.f) + bias + (float)((double)scale * total);
    }
}

void launch_deform_forward(hipStream_t st, const float* tmpl, const float* displace, const float* center, float* out, int nv) {
    const int n = nv * 3;
    k_deform_forward<<<(n + 255) / 256, 256, 0, st>>>(tmpl, displace, center, out, n);
}
void launch_deform_backward(hipStream_t st, const float* tmpl, const float* displace, const float* center, const float* g0,
                            float w0, const float* g1, float w1, const float* g2, float w2, float* grad_displace,
                            float* grad_center, double* acc, unsigned* ticket, int nv) {
    k_deform_backward<<<(nv + 255) / 256, 256, 0, st>>>(tmpl, displace, center, g0, w0, g1, w1, g2, w2, grad_displace,
                                                       grad_center, acc, ticket, nv);
}
void launch_adam_step(hipStream_t st, float* p, const float* g, float* m, float* v, size_t n, float lr_over_c0, float b0,
                      float one_minus_b0, float b1, float one_minus_b1, float c1, float eps, float weight_decay,
                      const int* iteration, double lr, double beta0, double beta1) {
    k_adam_step<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(p, g, m, v, n, lr_over_c0, b0, one_minus_b0, b1, one_minus_b1, c1,
                                                            eps, weight_decay, iteration, lr, beta0, beta1);
}
void launch_scalar_accumulate(hipStream_t st, float* dst, const float* src, int n, float scale, float bias, int accumulate,
                              const int* iteration, int stride) {
    k_scalar_accumulate<<<1, 256, 0, st>>>(dst, src, n, scale, bias, accumulate, iteration, stride);
}
__global__ void k_counter_add(int* __restrict__ counter, int delta) { *counter += delta; }
void launch_counter_add(hipStream_t st, int* counter, int delta) { k_counter_add<<<1, 1, 0, st>>>(counter, delta); }

}  // namespace jr
