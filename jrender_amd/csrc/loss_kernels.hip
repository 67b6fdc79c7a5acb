// The mesh regularisers of the deformation loop (BASELINE configs[3], demo2-deform.py:48-56) on the device:
//   LaplacianLoss  jrender/loss/laplacian_loss.py:5-37   sum((L x)^2) per mesh, L = the normalised graph Laplacian
//   FlattenLoss    jrender/loss/flatten_loss.py:5-80     sum((cos + 1)^2) over the edges shared by two faces
// Jittor tensor ops + autograd in the reference; value AND gradient in one launch each here.  Both are O(nv) work on
// meshes of a few thousand vertices — latency, not bandwidth: ONE workgroup per mesh of the batch walks its rows /
// edges, so the loss is a plain in-block sum (deterministic) and nothing needs a second launch.
#include "jr_kernels.h"

namespace jr {

constexpr int LOSS_WG = 1024;      // one workgroup per mesh: as many threads as a workgroup can have
__device__ inline double block_sum(double v, double* s) {          // LOSS_WG threads
    s[threadIdx.x] = v;
    __syncthreads();
    for (int w = LOSS_WG / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
        __syncthreads();
    }
    const double r = s[0];
    __syncthreads();
    return r;
}

// y = L x (CSR, float like the reference's matmul), loss = sum y^2, grad = 2 L^T y (CSR of the transpose: a gather,
// no atomics).  y goes through a global scratch row block of the same mesh, written and read by this workgroup only.
__global__ __launch_bounds__(LOSS_WG) void k_laplacian_loss(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                        const float* __restrict__ val, const int* __restrict__ rowptr_t,
                                                        const int* __restrict__ col_t, const float* __restrict__ val_t,
                                                        const float* __restrict__ x, float* __restrict__ y,
                                                        float* __restrict__ loss, float* __restrict__ grad, int nv,
                                                        float scale) {
    __shared__ double s_red[LOSS_WG];
    const long off = (long)blockIdx.x * nv * 3;
    x += off; y += off;
    double part = 0.0;
    for (int i = threadIdx.x; i < nv; i += LOSS_WG) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int k = rowptr[i]; k < rowptr[i + 1]; k++) {
            const float w = val[k];
            const float* p = x + (long)col[k] * 3;
            a0 += w * p[0]; a1 += w * p[1]; a2 += w * p[2];
        }
        y[i * 3 + 0] = a0; y[i * 3 + 1] = a1; y[i * 3 + 2] = a2;
        part += (double)(a0 * a0) + (double)(a1 * a1) + (double)(a2 * a2);
    }
    const double total = block_sum(part, s_red);       // its barriers also publish y to the whole workgroup
    if (threadIdx.x == 0) loss[blockIdx.x] = (float)total;
    if (!grad) return;
    __threadfence_block();
    grad += off;
    for (int j = threadIdx.x; j < nv; j += LOSS_WG) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int k = rowptr_t[j]; k < rowptr_t[j + 1]; k++) {
            const float w = val_t[k];
            const float* p = y + (long)col_t[k] * 3;
            a0 += w * p[0]; a1 += w * p[1]; a2 += w * p[2];
        }
        grad[j * 3 + 0] = 2.f * a0 * scale; grad[j * 3 + 1] = 2.f * a1 * scale; grad[j * 3 + 2] = 2.f * a2 * scale;
    }
}

struct D3 { double x, y, z; };
__device__ inline D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ inline D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline D3 operator*(D3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ inline double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline D3 load3(const float* p) { return {(double)p[0], (double)p[1], (double)p[2]}; }

// flatten_loss.py:44-66 for one (edge a, opposite vertex direction b): the component of b orthogonal to a and its length
// bl1 * sin(angle), with the reference's eps placements; the VJP mirrors it expression by expression.
struct Half { D3 cb; double l, al2, bl1, al1, ab, den, cosv, sinv, s; };
__device__ inline Half flatten_half(D3 a, D3 b, double eps) {
    Half h;
    h.al2 = dot(a, a);
    const double bl2 = dot(b, b);
    h.al1 = sqrt(h.al2 + eps);
    h.bl1 = sqrt(bl2 + eps);
    h.ab = dot(a, b);
    h.den = h.al1 * h.bl1 + eps;
    h.cosv = h.ab / h.den;
    h.sinv = sqrt(1.0 - h.cosv * h.cosv + eps);
    h.s = h.ab / (h.al2 + eps);
    h.cb = b - a * h.s;
    h.l = h.bl1 * h.sinv;
    return h;
}
__device__ inline void flatten_half_vjp(D3 a, D3 b, const Half& h, double eps, D3 dcb, double dl, D3& da, D3& db) {
    double dbl1 = h.sinv * dl;
    const double dsin = h.bl1 * dl;
    const double dcos = dsin * (-h.cosv / h.sinv);
    double dab = dcos / h.den;
    const double dal1 = -h.ab * h.bl1 / (h.den * h.den) * dcos;
    dbl1 += -h.ab * h.al1 / (h.den * h.den) * dcos;
    db = dcb;
    da = dcb * (-h.s);
    const double ds = -dot(a, dcb);
    dab += ds / (h.al2 + eps);
    double dal2 = -h.ab / ((h.al2 + eps) * (h.al2 + eps)) * ds;
    dal2 += dal1 / (2.0 * h.al1);
    const double dbl2 = dbl1 / (2.0 * h.bl1);
    da = da + a * (2.0 * dal2) + b * dab;
    db = db + b * (2.0 * dbl2) + a * dab;
}

// One workgroup per mesh: value (in-block double sum) and gradient (float atomics onto the four vertices of every edge
// pair; grad must arrive zeroed).  Arithmetic in double like the host mirror.
__global__ __launch_bounds__(LOSS_WG) void k_flatten_loss(const int* __restrict__ v0s, const int* __restrict__ v1s,
                                                      const int* __restrict__ v2s, const int* __restrict__ v3s,
                                                      const float* __restrict__ x, float* __restrict__ loss,
                                                      float* __restrict__ grad, int nv, int ne, float eps_f,
                                                      float scale) {
    __shared__ double s_red[LOSS_WG];
    const long off = (long)blockIdx.x * nv * 3;
    x += off;
    const double eps = (double)eps_f;
    double part = 0.0;
    for (int e = threadIdx.x; e < ne; e += LOSS_WG) {
        const int i0 = v0s[e], i1 = v1s[e], i2 = v2s[e], i3 = v3s[e];
        const D3 p0 = load3(x + (long)i0 * 3);
        const D3 a = load3(x + (long)i1 * 3) - p0, b1 = load3(x + (long)i2 * 3) - p0, b2 = load3(x + (long)i3 * 3) - p0;
        const Half h1 = flatten_half(a, b1, eps), h2 = flatten_half(a, b2, eps);
        const double num = dot(h1.cb, h2.cb);
        const double den = h1.l * h2.l + eps;
        const double cosv = num / den;
        part += (cosv + 1.0) * (cosv + 1.0);
        if (!grad) continue;
        const double g = 2.0 * (cosv + 1.0);
        D3 da1, db1, da2, db2;
        flatten_half_vjp(a, b1, h1, eps, h2.cb * (g / den), -num * h2.l / (den * den) * g, da1, db1);
        flatten_half_vjp(a, b2, h2, eps, h1.cb * (g / den), -num * h1.l / (den * den) * g, da2, db2);
        const D3 da = da1 + da2;
        const D3 d0 = (da + db1 + db2) * -1.0;
        float* gp = grad + off;
        const double sc = (double)scale;
        atomicAdd(gp + (long)i1 * 3 + 0, (float)(da.x * sc)); atomicAdd(gp + (long)i1 * 3 + 1, (float)(da.y * sc)); atomicAdd(gp + (long)i1 * 3 + 2, (float)(da.z * sc));
        atomicAdd(gp + (long)i2 * 3 + 0, (float)(db1.x * sc)); atomicAdd(gp + (long)i2 * 3 + 1, (float)(db1.y * sc)); atomicAdd(gp + (long)i2 * 3 + 2, (float)(db1.z * sc));
        atomicAdd(gp + (long)i3 * 3 + 0, (float)(db2.x * sc)); atomicAdd(gp + (long)i3 * 3 + 1, (float)(db2.y * sc)); atomicAdd(gp + (long)i3 * 3 + 2, (float)(db2.z * sc));
        atomicAdd(gp + (long)i0 * 3 + 0, (float)(d0.x * sc)); atomicAdd(gp + (long)i0 * 3 + 1, (float)(d0.y * sc)); atomicAdd(gp + (long)i0 * 3 + 2, (float)(d0.z * sc));
    }
    const double total = block_sum(part, s_red);
    if (threadIdx.x == 0) loss[blockIdx.x] = (float)total;
}

void launch_laplacian_loss(hipStream_t st, const int* rowptr, const int* col, const float* val, const int* rowptr_t,
                           const int* col_t, const float* val_t, const float* x, float* y, float* loss, float* grad,
                           int B, int nv, float scale) {
    k_laplacian_loss<<<(unsigned)B, LOSS_WG, 0, st>>>(rowptr, col, val, rowptr_t, col_t, val_t, x, y, loss, grad, nv, scale);
}
void launch_flatten_loss(hipStream_t st, const int* v0s, const int* v1s, const int* v2s, const int* v3s, const float* x,
                         float* loss, float* grad, int B, int nv, int ne, float eps, float scale) {
    if (grad) (void)hipMemsetAsync(grad, 0, sizeof(float) * (size_t)B * nv * 3, st);
    k_flatten_loss<<<(unsigned)B, LOSS_WG, 0, st>>>(v0s, v1s, v2s, v3s, x, loss, grad, nv, ne, eps, scale);
}

}  // namespace jr
