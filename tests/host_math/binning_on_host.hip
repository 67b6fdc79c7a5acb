// TEST INFRASTRUCTURE ONLY - tests/test_device_math_on_host.py.
//
// jrender_amd/csrc/binning.hip compiled with every __device__ function ALSO built for the host (the macro below; nothing of the file is
// restated), so that jr::pixel_range - the EXACT pixel rectangle the list building, the forward's skipped box test and the tile masks rest
// on - can be checked on the CPU against its definition: the set of pixel centres c(i) with !(c(i) < vlo) && !(c(i) > vhi), the reference's
// border test on one axis (SRK:28-34), evaluated with the kernels' own pixel_centre and float compares.
#include <hip/hip_runtime.h>
#undef __device__
#define __device__ __attribute__((host)) __attribute__((device))
#include "../../jrender_amd/csrc/binning.hip"

#include <math.h>
#include <stdint.h>
#include <string.h>

static uint64_t hm_state;
static inline uint32_t hm_next() { hm_state = hm_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(hm_state >> 33); }
static inline float hm_uniform(float a, float b) { return a + (b - a) * (float)(hm_next() & 0xffffff) / 16777216.0f; }
static inline float hm_nudge(float v, int ulps) { for (int k = 0; k < abs(ulps); k++) v = nextafterf(v, ulps > 0 ? INFINITY : -INFINITY); return v; }

// one bound: a pixel centre moved by -3..3 ulps (the cases the inward walk exists for), a uniform value, far outside, +-inf, NaN
static float hm_bound(int is) {
    const uint32_t k = hm_next() % 16;
    if (k < 8) return hm_nudge(jr::pixel_centre((int)(hm_next() % (uint32_t)is), is), (int)(hm_next() % 7) - 3);
    if (k < 12) return hm_uniform(-1.2f, 1.2f);
    if (k == 12) return hm_uniform(-50.f, 50.f);
    if (k == 13) return (hm_next() & 1) ? INFINITY : -INFINITY;
    if (k == 14) return NAN;
    return hm_uniform(-1.f, 1.f) * 1e-6f;
}

extern "C" {

// -> number of cases whose range differs from the definition; first failing case in bad[4] = {vlo, vhi, lo, hi as floats}
long hm_pixel_range_check(int is, long cases, unsigned long long seed, float* bad) {
    hm_state = seed * 2654435761ull + (unsigned long long)is;
    long wrong = 0;
    for (long c = 0; c < cases; c++) {
        const float vlo = hm_bound(is), vhi = hm_bound(is);
        int lo, hi;
        jr::pixel_range(vlo, vhi, is, lo, hi);
        int first = -1, last = -1;
        for (int i = 0; i < is; i++) {
            const float x = jr::pixel_centre(i, is);
            if (!(x < vlo) && !(x > vhi)) { if (first < 0) first = i; last = i; }
        }
        // the passing centres are contiguous (pixel_centre is monotone): [first, last], or none
        const bool ok = first < 0 ? lo > hi : (lo == first && hi == last);
        if (!ok && !wrong++) { bad[0] = vlo; bad[1] = vhi; bad[2] = (float)lo; bad[3] = (float)hi; }
    }
    return wrong;
}

}  // extern "C"
