// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the VALU instruction classes
// the SoftRas kernels are made of.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));
#define REPS 4096
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    const float c = 1.0000001f, e = 1e-9f;
    for (int i = 0; i < REPS; i++) {
        if (KIND == 0) { a0 *= c; a1 *= c; a2 *= c; a3 *= c; a4 *= c; a5 *= c; a6 *= c; a7 *= c; }                    // 8 v_mul_f32
        if (KIND == 1) { p0 *= c; p1 *= c; p2 *= c; p3 *= c; }                                                        // 4 v_pk_mul_f32 (8 flops)
        if (KIND == 2) { a0 = __builtin_fmaf(a0, c, e); a1 = __builtin_fmaf(a1, c, e); a2 = __builtin_fmaf(a2, c, e); a3 = __builtin_fmaf(a3, c, e);
                         a4 = __builtin_fmaf(a4, c, e); a5 = __builtin_fmaf(a5, c, e); a6 = __builtin_fmaf(a6, c, e); a7 = __builtin_fmaf(a7, c, e); }  // 8 v_fma_f32
        if (KIND == 3) { a0 = a0 > a1 ? a2 : a0; a1 = a1 > a2 ? a3 : a1; a2 = a2 > a3 ? a4 : a2; a3 = a3 > a4 ? a5 : a3;
                         a4 = a4 > a5 ? a6 : a4; a5 = a5 > a6 ? a7 : a5; a6 = a6 > a7 ? a0 : a6; a7 = a7 > a0 ? a1 : a7; }                           // 8 x (v_cmp + v_cndmask)
        if (KIND == 4) { a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_amdgcn_rcpf(a1); a2 = __builtin_amdgcn_rcpf(a2); a3 = __builtin_amdgcn_rcpf(a3);
                         a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_amdgcn_rcpf(a5); a6 = __builtin_amdgcn_rcpf(a6); a7 = __builtin_amdgcn_rcpf(a7); } // 8 v_rcp_f32
        if (KIND == 5) { d0 = d0 * 1.0000001 + 1e-9; d1 = d1 * 1.0000001 + 1e-9; d2 = d2 * 1.0000001 + 1e-9; d3 = d3 * 1.0000001 + 1e-9; }       // 4 mul + 4 add f64
        if (KIND == 6) { a0 = a0 / a1; a2 = a2 / a3; a4 = a4 / a5; a6 = a6 / a7; a1 += e; a3 += e; a5 += e; a7 += e; }                                // 4 IEEE divides + 4 adds
        if (KIND == 7) { a0 = fmaxf(a0, a1); a1 = fminf(a1, a2); a2 = fmaxf(a2, a3); a3 = fminf(a3, a4); a4 = fmaxf(a4, a5); a5 = fminf(a5, a6); a6 = fmaxf(a6, a7); a7 = fminf(a7, a0); } // 8 min/max
        if (KIND == 8) { a0 = __expf(a0 * 1e-6f); a1 = __expf(a1 * 1e-6f); a2 = __expf(a2 * 1e-6f); a3 = __expf(a3 * 1e-6f); }                        // 4 x (mul + v_exp)
        if (KIND == 9) { p0 = p0 * c + p1; p1 = p1 * c + p2; p2 = p2 * c + p3; p3 = p3 * c + p0; }                                                   // 4 pk_mul + 4 pk_add
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(d0 + d1 + d2 + d3);
}
template <int KIND> double run(float* out, const char* name, int instrs_per_iter) {
    const int blocks = 256 * 8, threads = 256;   // 8 blocks/CU -> 8 waves/SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, threads>>>(out, 1.0f); hipDeviceSynchronize();
    hipEventRecord(e0); k<KIND><<<blocks, threads>>>(out, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD = waves_per_SIMD * REPS * instrs_per_iter ; waves_per_SIMD = blocks*4 / (256 CUs * 4 SIMDs) = 8
    const double winst = 8.0 * REPS * instrs_per_iter;
    const double cyc = ms * 1e-3 * 2.4e9;          // at nominal 2.4 GHz (actual clock may be lower)
    printf("%-34s %8.3f ms  %6.2f cycles/wave-instr/SIMD (nominal 2.4 GHz)\n", name, ms, cyc / winst);
    return ms;
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>(out, "v_mul_f32 x8", 8);
    run<1>(out, "v_pk_mul_f32 x4 (8 flops)", 4);
    run<2>(out, "v_fma_f32 x8", 8);
    run<3>(out, "v_cmp+v_cndmask x8 (16 instr)", 16);
    run<4>(out, "v_rcp_f32 x8", 8);
    run<5>(out, "f64 mul+add x4 (8 instr or 4 fma)", 8);
    run<6>(out, "IEEE div x4 (+4 add) per-div cost", 4);
    run<7>(out, "v_min/v_max x8", 8);
    run<8>(out, "mul + v_exp x4 (8 instr)", 8);
    run<9>(out, "pk_mul + pk_add x4 (8 instr, 16 flops)", 8);
    return 0;
}
