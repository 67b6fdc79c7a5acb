/*
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build).  Not part of the product.
 *
 * Minimal CUDA -> host shim: lets the reference's own SoftRas kernel source
 * (jrender/renderer/dr/softras/cuda/soft_rasterize.py, the `cuda_header`
 * strings at SRK:12-458 and SRK:975-1362) compile with g++ so that the
 * reference arithmetic can run on the CPU.  Nothing from the reference is
 * copied here: this file only supplies the names CUDA would supply.
 *
 * Semantics reproduced on purpose:
 *   - CUDA's min/max overload set: (float,float)->float with fminf/fmaxf
 *     NaN behaviour, mixed (float,double)/(double,float)->double,
 *     (double,double)->double, (int,int)->int.
 *   - exp/sqrt on float resolve to expf/sqrtf (std:: overloads).
 *   - atomicAdd(float*) : `omp atomic` so the backward can be run in
 *     parallel for the timing baseline; serial runs are deterministic.
 */
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <math.h>

#define __global__
#define __device__
#define __host__
#define __restrict__
#define __forceinline__ inline

struct uint3_shim { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern thread_local uint3_shim blockIdx, threadIdx;
extern thread_local dim3 blockDim, gridDim;

typedef float float32;

using std::exp;
using std::sqrt;
using std::isnan;

static inline float  min(float a, float b)   { return fminf(a, b); }
static inline float  max(float a, float b)   { return fmaxf(a, b); }
static inline double min(double a, double b) { return fmin(a, b); }
static inline double max(double a, double b) { return fmax(a, b); }
static inline double min(float a, double b)  { return fmin((double)a, b); }
static inline double max(float a, double b)  { return fmax((double)a, b); }
static inline double min(double a, float b)  { return fmin(a, (double)b); }
static inline double max(double a, float b)  { return fmax(a, (double)b); }
static inline int    min(int a, int b)       { return a < b ? a : b; }
static inline int    max(int a, int b)       { return a > b ? a : b; }

/* Optional double-precision SHADOW of up to two float accumulator arrays (ref_softras_backward_exactsum): every
 * atomicAdd into a registered array is also added, in double, to its shadow.  The shadow is the exact sum (to double
 * rounding) of the float terms the reference's kernel produces - the reference's result without the order noise of its
 * float atomics. */
struct atomic_shadow { float* base; size_t len; double* sum; };
extern atomic_shadow g_atomic_shadow[2];
static inline float atomicAdd(float* addr, float v) {
    for (int i = 0; i < 2; i++) {
        const atomic_shadow& s = g_atomic_shadow[i];
        if (s.sum && addr >= s.base && addr < s.base + s.len) {
            double* d = s.sum + (addr - s.base);
#pragma omp atomic
            *d += (double)v;
        }
    }
    float old;
#pragma omp atomic capture
    { old = *addr; *addr += v; }
    return old;
}
/* the reference's backward kernel is a template over scalar_t: its double instantiation (ref_softras_backward_f64,
 * the truth both float implementations are measured against) needs the double overload CUDA has since sm_60 */
static inline double atomicAdd(double* addr, double v) {
    double old;
#pragma omp atomic capture
    { old = *addr; *addr += v; }
    return old;
}

/* Names the reference's coarse-to-fine kernels use on top of the above (soft_rasterize_coarse_to_fine.py: BitMask over dynamic
 * shared memory, RasterizeCoarseCudaKernel).  oracle/ref_c2f_driver.cpp launches that kernel as <<<1, 1>>> - a legal launch of its
 * block- and grid-stride loops in which __syncthreads() has nothing to wait for and the chunks are taken in ascending order. */
#include <stdint.h>
#define __shared__
static inline void __syncthreads() {}
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline unsigned atomicOr(unsigned* addr, unsigned v)  { unsigned old; { old = *addr; *addr |= v; } return old; }
static inline unsigned atomicAnd(unsigned* addr, unsigned v) { unsigned old; { old = *addr; *addr &= v; } return old; }
static inline int atomicAdd(int* addr, int v) { int old; { old = *addr; *addr += v; } return old; }
