// Exhaustive check: which refinement of v_rcp_f32 equals the IEEE quotient 1.0f/x for EVERY float
// with exponent in [-40, 40] (all 2^23 mantissas, both signs)?
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ inline float rcp_nr1(float x) { float y = __builtin_amdgcn_rcpf(x); float e = __builtin_fmaf(-x, y, 1.f); return __builtin_fmaf(e, y, y); }
__device__ inline float rcp_nr2(float x) { float y = rcp_nr1(x); float e = __builtin_fmaf(-x, y, 1.f); return __builtin_fmaf(e, y, y); }
__global__ void k(unsigned long long* bad) {
    const unsigned m = blockIdx.x * blockDim.x + threadIdx.x;     // mantissa 0..2^23-1
    if (m >= (1u << 23)) return;
    unsigned long long b1 = 0, b2 = 0;
    for (int e = -40; e <= 40; e++)
        for (int s = 0; s < 2; s++) {
            const float x = __builtin_bit_cast(float, ((unsigned)s << 31) | ((unsigned)(127 + e) << 23) | m);
            const float r = 1.0f / x;
            b1 += __builtin_bit_cast(unsigned, rcp_nr1(x)) != __builtin_bit_cast(unsigned, r);
            b2 += __builtin_bit_cast(unsigned, rcp_nr2(x)) != __builtin_bit_cast(unsigned, r);
        }
    if (b1) atomicAdd(&bad[0], b1);
    if (b2) atomicAdd(&bad[1], b2);
}
int main() {
    unsigned long long* d; hipMalloc(&d, 16); hipMemset(d, 0, 16);
    k<<<(1 << 23) / 256, 256>>>(d); hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("mismatches vs 1.0f/x over %llu values: one Newton step %llu, two Newton steps %llu\n", 81ull * 2 * (1ull << 23), h[0], h[1]);
    return 0;
}
