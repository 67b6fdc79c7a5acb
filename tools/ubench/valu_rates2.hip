// Micro-benchmark 2: issue cost of individual VALU opcodes (inline asm, 8 independent chains).
// hipcc --offload-arch=gfx950 -O3 valu_rates2.hip -o /tmp/vr2 && /tmp/vr2
#include <hip/hip_runtime.h>
#include <cstdio>
#define REPS 4096
#define CHAIN8(OPSTR)                                                                                   \
    asm volatile(OPSTR(%0) OPSTR(%1) OPSTR(%2) OPSTR(%3) OPSTR(%4) OPSTR(%5) OPSTR(%6) OPSTR(%7)        \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)       \
                 : "v"(b), "v"(c), "s"(m))
#define OP_MOV(r) "v_mov_b32 " #r ", %8\n"
#define OP_ADDF(r) "v_add_f32 " #r ", " #r ", %8\n"
#define OP_ADDU(r) "v_add_u32 " #r ", " #r ", %8\n"
#define OP_AND(r) "v_and_b32 " #r ", " #r ", %8\n"
#define OP_LSHL(r) "v_lshlrev_b32 " #r ", 1, " #r "\n"
#define OP_BFE(r) "v_bfe_i32 " #r ", " #r ", 3, 1\n"
#define OP_BFI(r) "v_bfi_b32 " #r ", %8, %9, " #r "\n"
#define OP_CND(r) "v_cndmask_b32 " #r ", " #r ", %8, %10\n"
#define OP_CMP(r) "v_cmp_lt_f32 vcc, " #r ", %8\n"
#define OP_MAX3(r) "v_max3_f32 " #r ", " #r ", %8, %9\n"
#define OP_MED3(r) "v_med3_f32 " #r ", " #r ", %8, %9\n"
#define OP_MAX(r) "v_max_f32 " #r ", " #r ", %8\n"
#define OP_MAXI(r) "v_max_i32 " #r ", " #r ", %8\n"
#define OP_FMA(r) "v_fma_f32 " #r ", " #r ", %8, %9\n"
#define OP_BCNT(r) "v_bcnt_u32_b32 " #r ", " #r ", %8\n"
#define OP_DPP(r) "v_add_f32_dpp " #r ", " #r ", " #r " row_mirror row_mask:0xf bank_mask:0xf\n"
#define OP_PERM(r) "v_perm_b32 " #r ", " #r ", %8, %9\n"
#define OP_CVT(r) "v_cvt_f32_i32 " #r ", " #r "\n"
#define OP_RSQ(r) "v_rsq_f32 " #r ", " #r "\n"
#define OP_SQRT(r) "v_sqrt_f32 " #r ", " #r "\n"
#define OP_MULLO(r) "v_mul_lo_u32 " #r ", " #r ", %8\n"
#define OP_MAD24(r) "v_mad_u32_u24 " #r ", " #r ", %8, %9\n"
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = seed * 1.0000001f, c = seed + 3.f;
    unsigned long long m = 0x5555555555555555ull;
    for (int i = 0; i < REPS; i++) {
        if (KIND == 0) CHAIN8(OP_MOV);
        if (KIND == 1) CHAIN8(OP_ADDF);
        if (KIND == 2) CHAIN8(OP_ADDU);
        if (KIND == 3) CHAIN8(OP_AND);
        if (KIND == 4) CHAIN8(OP_LSHL);
        if (KIND == 5) CHAIN8(OP_BFE);
        if (KIND == 6) CHAIN8(OP_BFI);
        if (KIND == 7) CHAIN8(OP_CND);
        if (KIND == 8) CHAIN8(OP_CMP);
        if (KIND == 9) CHAIN8(OP_MAX3);
        if (KIND == 10) CHAIN8(OP_MED3);
        if (KIND == 11) CHAIN8(OP_MAX);
        if (KIND == 12) CHAIN8(OP_MAXI);
        if (KIND == 13) CHAIN8(OP_FMA);
        if (KIND == 14) CHAIN8(OP_BCNT);
        if (KIND == 15) CHAIN8(OP_DPP);
        if (KIND == 16) CHAIN8(OP_PERM);
        if (KIND == 17) CHAIN8(OP_CVT);
        if (KIND == 18) CHAIN8(OP_RSQ);
        if (KIND == 19) CHAIN8(OP_SQRT);
        if (KIND == 20) CHAIN8(OP_MULLO);
        if (KIND == 21) CHAIN8(OP_MAD24);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int KIND> void run(float* out, const char* name) {
    const int blocks = 256 * 8, threads = 256;   // 8 waves/SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, threads>>>(out, 1.0f); hipDeviceSynchronize();
    hipEventRecord(e0); k<KIND><<<blocks, threads>>>(out, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s %8.3f ms  %6.2f cycles/wave-instr/SIMD (nominal 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / (8.0 * REPS * 8));
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>(out, "v_mov_b32"); run<1>(out, "v_add_f32"); run<13>(out, "v_fma_f32"); run<2>(out, "v_add_u32"); run<3>(out, "v_and_b32");
    run<4>(out, "v_lshlrev_b32"); run<5>(out, "v_bfe_i32"); run<6>(out, "v_bfi_b32"); run<7>(out, "v_cndmask_b32 (sgpr)");
    run<8>(out, "v_cmp_lt_f32 (vcc)"); run<9>(out, "v_max3_f32"); run<10>(out, "v_med3_f32"); run<11>(out, "v_max_f32");
    run<12>(out, "v_max_i32"); run<14>(out, "v_bcnt_u32_b32"); run<15>(out, "v_add_f32_dpp"); run<16>(out, "v_perm_b32");
    run<17>(out, "v_cvt_f32_i32"); run<18>(out, "v_rsq_f32"); run<19>(out, "v_sqrt_f32"); run<20>(out, "v_mul_lo_u32"); run<21>(out, "v_mad_u32_u24");
    return 0;
}
