/* TEST INFRASTRUCTURE ONLY - tests/host_math/: lets jrender_amd/csrc/softras_device.h (the per-pair arithmetic the raster
 * kernels are built from) compile with g++ for the HOST, so that its exact paths can be compared with the oracle pair by pair
 * on the CPU and run under the sanitizers (there is no GPU AddressSanitizer on this pool).  Supplies only the names hipcc
 * would supply.  The gfx950 builtins that are APPROXIMATIONS on the device (v_rcp_f32, v_exp_f32) are exact here: the harness
 * compares the header's bit-critical paths, which must not depend on them, and says so where it skips a path that does. */
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>
#define __device__
#define __host__
#define __global__
#define __restrict__
#define __launch_bounds__(...)
/* v_rcp_f32 is good to 1 ulp: HM_RCP_ULP_OFF = 1 returns a neighbour of the correctly rounded reciprocal (up or down by a bit of the
 * argument) - the Newton step of recip_exact and the refinement quotients built on it must absorb that (on the device the identity
 * is checked exhaustively: jr_selftest_reciprocal) */
#ifndef HM_RCP_ULP_OFF
#define HM_RCP_ULP_OFF 0
#endif
static inline float __builtin_amdgcn_rcpf(float x) {
    float y = 1.0f / x;
    if (HM_RCP_ULP_OFF && y == y && fabsf(y) < 1e37f && fabsf(y) > 1e-37f) {
        uint32_t u; memcpy(&u, &x, 4);
        y = nextafterf(y, (u & 1u) ? INFINITY : -INFINITY);
    }
    return y;
}
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }
static inline unsigned long long __builtin_amdgcn_s_memtime() { return 0ull; }
static inline unsigned long long __builtin_amdgcn_ballot_w64(bool p) { return p ? 1ull : 0ull; }
struct hm_dim3 { unsigned x, y, z; };
static hm_dim3 threadIdx = {0, 0, 0};
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }
