// SoftRas per-(pixel, face) arithmetic for gfx950 — shared by the forward and backward kernels.
//
// The values computed here are the ones the reference's kernels compute
// (jrender/renderer/dr/softras/cuda/soft_rasterize.py, "SRK"), in the same
// precision and association order, because the per-pixel face-index buffer has
// to match bit for bit.  This translation unit MUST be
// built with -ffp-contract=off (no FMA contraction) and without fast-math.
//
// What is NOT the reference's: the data organisation and how the IEEE quotients are obtained.
//  * Every face gets one packed 176-byte record (FaceGeo) written once per forward by the setup
//    kernel: border box incl. cull radius, face_inv, vertices, edge-difference vectors of the
//    Gram matrix, obtuse vertex, facing, colour, and the correctly rounded reciprocals of the
//    per-face divisors.  The raster kernels copy the records a wavefront needs into LDS and every
//    lane reads the record of ITS current face — no dynamic register indexing, no scattered
//    global gathers in the inner loop.
//  * A float quotient a/b whose divisor is a per-face or per-call constant is computed from
//    y = RN(1/b) (one true division, done once) with the two-step FMA refinement
//        q0 = a*y; r0 = fma(-b,q0,a); q1 = fma(r0,y,q0); r1 = fma(-b,q1,a); q = fma(r1,y,q1)
//    which is the tail of the compiler's own correctly-rounded division expansion and returns
//    RN(a/b) — the SAME bits as `a / b` — whenever no intermediate leaves the normal range.
//    Faces / constants outside a conservative range are flagged and take the plain division.
//    tests/test_gpu_parity.py::test_fast_division_identity checks the identity on 2^31 operands.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jr_tuning.h"

namespace jr {

constexpr int TILE = 8;          // pixels per side of a wavefront's tile (8x8 = 64 lanes)
constexpr int TILE_LOG2 = 3;
// Pixels per side of a binning cell: a RUN-TIME value since round 5 (RasterParams::bin_log2 = 3, 4 or 5: bins of 1, 2x2
// or 4x4 tiles), chosen per launch from the image size or by the caller (jr_softras_set_bin_size; the reference exposes
// the same choice as `bin_size`, soft_rasterize.py:85-99, C2F:16-18).  Results never depend on it - every pixel sees its
// faces in ascending order whatever the cell size - only how precisely the lists, the launch order and the heavy-tile
// classification follow the tiles.  32 is the largest cell: a list entry carries a 16-bit tile mask.
constexpr int BIN_LOG2_MIN = 3, BIN_LOG2_MAX = 5;
constexpr int CHUNK = 64;        // faces staged per pass = one per lane
constexpr int MAX_IMAGE = 4096;

struct RasterParams {
    int B, NF, T, R, IS, K;
    float near_, far_, eps, sigma, dist_eps, gamma;
    float thr;   // dist_eps * sigma           (SRK:289)
    float rad;   // sqrtf(thr)                 (SRK:316)
    int dist, rgb, alpha, tex, double_side;
    float bg[3];
    int bins_x, bins_y;   // ceil(IS / bin size)
    int bin_log2;         // log2 of the bin's side in pixels: 3, 4 or 5
    int sub_log2;         // bin_log2 - 3: the bin is (1 << sub_log2)^2 tiles; tile `sub` of a bin = (ty << sub_log2) | tx = its bit in a list entry's mask
    // correctly rounded reciprocals of the per-call divisors + "they are in the safe range"
    float far_minus_near, near_minus_far;
    float r_sigma, r_gamma, r_far_minus_near, r_near_minus_far;
    float rs_log2e, rg_log2e;   // log2(e) / sigma, log2(e) / gamma: exp(x / sigma) = exp2(x * rs_log2e) on the colour path
    int consts_safe;
};

__host__ __device__ inline int bin_log2_of(const RasterParams& p) { return p.bin_log2; }
__host__ __device__ inline int sub_log2_of(const RasterParams& p) { return p.sub_log2; }

constexpr int FLAG_FRONT = 1;    // check_face_frontside (SRK:37-40)
constexpr int FLAG_SAFE = 2;     // per-face divisors/operands inside the fast-division range
constexpr int META_ID_BITS = 28; // FaceGeo::meta = id | flags << 28 | (obtuse vertex + 1) << 30
constexpr int MAX_FACES_PER_IMAGE = (1 << META_ID_BITS) - 1;

// Packed per-face record, 44 floats = 11 x 16 B (an ODD number of 16-byte slots: lanes that read
// records of different faces from LDS spread over all bank groups).
struct FaceGeo {
    float xlo, xhi, ylo, yhi;      // border box incl. cull radius        (SRK:28-34, :316)
    float inv[9];                  // face_inv                            (SRK:205-217)
    float z[3];
    float x0, y0, x1, y1, x2, y2;
    float A[9];                    // A[e] = sym[e] - sym[e+1]            (SRK:77-79)
    float Dn[3];                   // A[e][e] - A[e][e+1]                 (SRK:81 denominator)
    float spare[3];                // (kept: 11 x 16 B is an odd number of 16-byte lanes; was RN(1/Dn), tools/ablate/patches/)
    float rz[3];                   // RN(1/z[k])
    int meta;                      // face index inside its image | FLAG_* << 28 | (obtuse vertex + 1) << 30 (SRK:227-235)
    float col[3];                  // surface colour when T == 1
};
static_assert(sizeof(FaceGeo) == 176, "FaceGeo layout");
__device__ inline int face_id(int meta) { return meta & MAX_FACES_PER_IMAGE; }
__device__ inline bool face_front(int meta) { return (meta >> META_ID_BITS) & FLAG_FRONT; }
__device__ inline bool face_safe(int meta) { return (meta >> META_ID_BITS) & FLAG_SAFE; }
__device__ inline int face_obtuse(int meta) { return (int)((unsigned)meta >> 30) - 1; }
typedef FaceGeo FaceRec;

// ---- exact division by a divisor whose correctly rounded reciprocal is known ------------------
template <bool FAST>
__device__ inline float div_known(float a, float b, float rb) {
    if (FAST) {
        const float q0 = a * rb;
        const float r0 = __builtin_fmaf(-b, q0, a);
        const float q1 = __builtin_fmaf(r0, rb, q0);
        const float r1 = __builtin_fmaf(-b, q1, a);
        return __builtin_fmaf(r1, rb, q1);
    }
    return a / b;
}

__device__ inline bool in_fast_range(float v) {       // finite, 2^-40 <= |v| <= 2^40
    const float a = fabsf(v);
    return a >= 9.094947017729282e-13f && a <= 1.099511627776e12f;
}
__device__ inline bool zero_or_in_fast_range(float v) { return v == 0.f || in_fast_range(v); }

// Per-face preprocessing = faces_info of the reference (SRK:176-236).
__device__ inline void face_setup(const float* __restrict__ f, float* __restrict__ info) {
    const float x0 = f[0], y0 = f[1], x1 = f[3], y1 = f[4], x2 = f[6], y2 = f[7];
    float star[9];
    star[0] = y1 - y2; star[1] = x2 - x1; star[2] = x1 * y2 - x2 * y1;
    star[3] = y2 - y0; star[4] = x0 - x2; star[5] = x2 * y0 - x0 * y2;
    star[6] = y0 - y1; star[7] = x1 - x0; star[8] = x0 * y1 - x1 * y0;
    float det = (x2 * (y0 - y1) + x0 * (y1 - y2)) + x1 * (y2 - y0);
    det = det > 0 ? (float)fmax((double)det, 1e-10) : (float)fmin((double)det, -1e-10);
#pragma unroll
    for (int k = 0; k < 9; k++) info[k] = star[k] / det;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int k = 0; k < 3; k++)
            info[9 + 3 * j + k] = (f[3 * j] * f[3 * k] + f[3 * j + 1] * f[3 * k + 1]) + 1.f;
    const float d0 = (x1 - x0) * (x2 - x0) + (y1 - y0) * (y2 - y0);
    const float d1 = (x2 - x1) * (x0 - x1) + (y2 - y1) * (y0 - y1);
    const float d2 = (x0 - x2) * (x1 - x2) + (y0 - y2) * (y1 - y2);
    const int obt = d0 < 0 ? 0 : (d1 < 0 ? 1 : (d2 < 0 ? 2 : -1));
    info[18] = obt == 0 ? 1.f : 0.f;
    info[19] = obt == 1 ? 1.f : 0.f;
    info[20] = obt == 2 ? 1.f : 0.f;
#pragma unroll
    for (int k = 21; k < 27; k++) info[k] = 0.f;
}

// Record from the face, its faces_info and (T == 1) its colour.
__device__ inline void build_face_geo(FaceGeo& r, const float* __restrict__ f,
                                      const float* __restrict__ fi, float rad, int id) {
    const float x0 = f[0], y0 = f[1], x1 = f[3], y1 = f[4], x2 = f[6], y2 = f[7];
    r.xhi = fmaxf(fmaxf(x0, x1), x2) + rad;
    r.xlo = fminf(fminf(x0, x1), x2) - rad;
    r.yhi = fmaxf(fmaxf(y0, y1), y2) + rad;
    r.ylo = fminf(fminf(y0, y1), y2) - rad;
    bool safe = true;
#pragma unroll
    for (int k = 0; k < 9; k++) { r.inv[k] = fi[k]; safe = safe && zero_or_in_fast_range(fi[k]); }
    r.z[0] = f[2]; r.z[1] = f[5]; r.z[2] = f[8];
    r.x0 = x0; r.y0 = y0; r.x1 = x1; r.y1 = y1; r.x2 = x2; r.y2 = y2;
    const float* sym = fi + 9;
#pragma unroll
    for (int e = 0; e < 3; e++) {
        const int e1 = (e + 1) % 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            r.A[3 * e + c] = sym[3 * e + c] - sym[3 * e1 + c];
            safe = safe && zero_or_in_fast_range(r.A[3 * e + c]);
        }
        r.Dn[e] = r.A[3 * e + e] - r.A[3 * e + e1];
        r.spare[e] = 0.f;
        r.rz[e] = 1.0f / r.z[e];
        safe = safe && in_fast_range(r.Dn[e]) && in_fast_range(r.z[e]);
    }
    safe = safe && fabsf(x0) <= 1024.f && fabsf(y0) <= 1024.f && fabsf(x1) <= 1024.f &&
           fabsf(y1) <= 1024.f && fabsf(x2) <= 1024.f && fabsf(y2) <= 1024.f;
    const int obt = fi[18] == 1.f ? 0 : (fi[19] == 1.f ? 1 : (fi[20] == 1.f ? 2 : -1));
    const int flags = (((y2 - y0) * (x1 - x0) < (y1 - y0) * (x2 - x0)) ? FLAG_FRONT : 0) | (safe ? FLAG_SAFE : 0);
    r.meta = id | (flags << META_ID_BITS) | ((obt + 1) << 30);
    r.col[0] = 0.f; r.col[1] = 0.f; r.col[2] = 0.f;

}

// pixel centre in NDC: (2*i + 1 - IS) / IS evaluated in double, rounded once (SRK:280-283)
// Both integers are exact floats (|2i+1-IS| < 2^24), and rounding the double quotient to float
// equals the correctly rounded float quotient (double rounding is innocuous for division at
// 53 >= 2*24+2 bits), so one IEEE float division yields the reference's bits.
__device__ inline float pixel_centre(int i, int is) {
    return (float)(2 * i + 1 - is) / (float)is;
}

struct Bary { float w0, w1, w2; };

__device__ inline Bary barycentric(const FaceGeo& r, float x, float y) {             // SRK:20-25
    Bary b;
    b.w0 = (r.inv[0] * x + r.inv[1] * y) + r.inv[2];
    b.w1 = (r.inv[3] * x + r.inv[4] * y) + r.inv[5];
    b.w2 = (r.inv[6] * x + r.inv[7] * y) + r.inv[8];
    return b;
}

__device__ inline bool pixel_inside(const Bary& b) {                                 // SRK:43-46
    return b.w0 <= 1 && b.w0 >= 0 && b.w1 <= 1 && b.w1 >= 0 && b.w2 <= 1 && b.w2 >= 0;
}

// Instruction costs measured on MI355X (tools/ubench/valu_rates.hip, cycles per wave-instruction):
// v_mul/v_add/v_fma 2 | v_cmp, v_cndmask, v_min, v_max 4 | f64 4-5 | v_rcp/v_exp 8 | IEEE a/b 45.
// The code below is written against those prices: no select chains where an LDS-indexed read
// does the job, v_med3 for clamps, Newton-refined v_rcp where it is PROVEN equal to the IEEE
// reciprocal, refinement quotients where the divisor's exact reciprocal is known.

// IEEE 1.0f/x from v_rcp_f32 + one Newton step.  Verified EXHAUSTIVELY equal to the correctly
// rounded quotient for every float with exponent in [-40, 40] (tools/ubench/rcp_exact.hip and
// jr_selftest_reciprocal: 1.36e9 values, 0 mismatches).  12 cycles instead of 45.
__device__ inline float recip_exact(float x) {
    const float y = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, y, 1.f), y, y);
}

// max(min(v, 1.), 0.) / min(max(v, 0.), 1.): the reference evaluates these in double because of
// the bare literals; both are exact selections, so the float form returns the same value.  For
// non-NaN input both equal the median of (v, 0, 1) = one v_med3_f32; NaN input (only possible for
// degenerate faces, which are never flagged FAST) keeps the two-instruction form, which matches
// CUDA's fmin/fmax NaN behaviour.  Only the sign of a zero may differ, which nothing observes.
template <bool FAST> __device__ inline float clamp01(float v) {                          // SRK:51
    return FAST ? __builtin_amdgcn_fmed3f(v, 0.f, 1.f) : fmaxf(fminf(v, 1.f), 0.f);
}
template <bool FAST> __device__ inline float clamp01_maxfirst(float v) {                 // SRK:138
    return FAST ? __builtin_amdgcn_fmed3f(v, 0.f, 1.f) : fminf(fmaxf(v, 0.f), 1.f);
}

template <bool FAST>
__device__ inline Bary barycentric_clip(Bary b) {                                    // SRK:49-54
    b.w0 = clamp01<FAST>(b.w0); b.w1 = clamp01<FAST>(b.w1); b.w2 = clamp01<FAST>(b.w2);
    // max(w_sum, 1e-5) compares in double and stores (float)1e-5 when clamped == fmaxf(s, 1e-5f)
    const float s = fmaxf((b.w0 + b.w1) + b.w2, 1e-5f);
    // three quotients by the same divisor s in [1e-5, 3]: one exact reciprocal, three refinements
    const float rs = FAST ? recip_exact(s) : 0.f;
    b.w0 = div_known<FAST>(b.w0, s, rs);
    b.w1 = div_known<FAST>(b.w1, s, rs);
    b.w2 = div_known<FAST>(b.w2, s, rs);
    return b;
}

// depth of the clipped barycentric point, 1./(sum w/z) (SRK:364, :1296).  The reference divides in
// double and rounds to float; for a float divisor that equals the correctly rounded float quotient
// (53 >= 2*24+2 bits: double rounding is innocuous for division), i.e. IEEE 1.0f/s.
template <bool FAST>
__device__ inline float depth_of(const FaceGeo& r, const Bary& c) {
    const float s = (div_known<FAST>(c.w0, r.z[0], r.rz[0]) + div_known<FAST>(c.w1, r.z[1], r.rz[1])) +
                    div_known<FAST>(c.w2, r.z[2], r.rz[2]);
    if (FAST && in_fast_range(s)) return recip_exact(s);     // s == 0 (all weights clipped) etc. -> IEEE path
    return 1.0f / s;
}

struct Dist {
    float sign, dx, dy;   // sign: +1 inside / -1 outside; (dx,dy) = nearest point - pixel
    float t0, t1, t2;     // nearest-point barycentric minus w   (SRK:98-100, :139)
};

// Projection of the pixel onto edge e (vertices e, e+1): t along the edge, the point's barycentric
// coordinates minus w, the offset to the pixel and its squared length (SRK:73-92 / :123-144).
// `clamp` selects the outside-the-triangle variant (t clamped to the segment, SRK:137-140).
// `r` refers to an LDS record: a per-lane edge index is an LDS address, not a select chain.
struct EdgeCand { float u0, u1, u2, ex, ey, dd; };

// TV selects how the quotient of SRK:81 / :132 is formed:
//   TV_IEEE   plain IEEE division (~50 cycles).  (The same bits from a reciprocal-refinement quotient with RN(1/Dn)
//             stored in the record were built twice and cost what the hardware's v_div_scale / v_div_fmas /
//             v_div_fixup sequence costs: tools/ablate/patches/dead_switches_r03.patch.)
//   TV_RCP    reciprocal multiply (<= 2 ulp off): only where nothing is decided from the result
constexpr int TV_IEEE = 0, TV_RCP = 2;

template <bool FAST, int TV = TV_IEEE>
__device__ inline EdgeCand edge_candidate(const FaceGeo& r, const Bary& b, int e, bool clamp) {
    const int e1 = e == 2 ? 0 : e + 1;
    const float a0 = r.A[3 * e], a1 = r.A[3 * e + 1], a2 = r.A[3 * e + 2];
    const float av1 = r.A[3 * e + e1];
    const float dn = r.Dn[e];
    const float num = ((b.w0 * a0 + b.w1 * a1) + b.w2 * a2) - av1;                       // SRK:81 / :132
    float tv;
    if (FAST && TV == TV_RCP) tv = num * __builtin_amdgcn_rcpf(dn);
    else tv = num / dn;
    const float tn = 1 - tv;
    // t[e] = tv, t[e+1] = 1 - tv, t[e+2] = 0 (indices mod 3)
    float u0 = e == 0 ? tv : (e == 2 ? tn : 0.f);
    float u1 = e == 1 ? tv : (e == 0 ? tn : 0.f);
    float u2 = e == 2 ? tv : (e == 1 ? tn : 0.f);
    if (clamp) { u0 = clamp01_maxfirst<FAST>(u0); u1 = clamp01_maxfirst<FAST>(u1); u2 = clamp01_maxfirst<FAST>(u2); }
    EdgeCand c;
    c.u0 = u0 - b.w0; c.u1 = u1 - b.w1; c.u2 = u2 - b.w2;
    c.ex = (c.u0 * r.x0 + c.u1 * r.x1) + c.u2 * r.x2;
    c.ey = (c.u0 * r.y0 + c.u1 * r.y1) + c.u2 * r.y2;
    c.dd = c.ex * c.ex + c.ey * c.ey;
    return c;
}

// region of an outside pixel (SRK:107-121): two non-positive weights -> the vertex region `corner` (edge =
// corner, unless the vertex is the obtuse one and the pixel lies beyond it: then the previous edge); one ->
// that edge.  Returns the edge index, or -1 where the reference indexes t[-1] (undefined behaviour: some
// w >= 1 by rounding while none is <= 0).
__device__ inline int outside_edge(const FaceGeo& r, int obt, const Bary& b, float xp, float yp) {
    const bool n0 = b.w0 <= 0, n1 = b.w1 <= 0, n2 = b.w2 <= 0;
    int v0 = -1, corner = -1;
    if (n1 && n2) corner = 0;
    else if (n2 && n0) corner = 1;
    else if (n0 && n1) corner = 2;
    else if (n0) v0 = 1;
    else if (n1) v0 = 2;
    else if (n2) v0 = 0;
    if (corner >= 0) {
        v0 = corner;
        if (obt == corner) {                       // rare: obtuse vertex region
            const int other = corner == 0 ? 2 : corner - 1;
            const float* xy = &r.x0;
            const float xc = xy[2 * corner], yc = xy[2 * corner + 1];
            if ((xp - xc) * (xy[2 * other] - xc) + (yp - yc) * (xy[2 * other + 1] - yc) > 0) v0 = other;
        }
    }
    return v0;
}

__device__ inline bool strictly_inside(const Bary& b) {
    return b.w0 > 0 && b.w1 > 0 && b.w2 > 0 && b.w0 < 1 && b.w1 < 1 && b.w2 < 1;
}
// the same for finite weights (FLAG_SAFE faces): v_min3 / v_max3 and two compares instead of six compares
template <bool FAST>
__device__ inline bool strictly_inside_t(const Bary& b) {
    if (FAST) return fminf(fminf(b.w0, b.w1), b.w2) > 0 && fmaxf(fmaxf(b.w0, b.w1), b.w2) < 1;
    return strictly_inside(b);
}

// squared-distance machinery, euclidean mode (SRK:57-147).  The inside case (three edge
// projections, keep the nearest) and the outside case (one edge chosen from the sign pattern of
// w, clamped) share the first projection so that a wavefront with both kinds of pixels does not
// execute two separate code paths; the two further projections run only if some lane is inside.
// (Projecting only the edge the weights name as nearest was built, proven exact and measured slower:
// tools/ablate/patches/, profiles/r02_ab_inside_select.log.)
template <bool FAST, int TV = TV_IEEE>
__device__ inline Dist euclidean_p2f(const FaceGeo& r, int meta, const Bary& b, float xp, float yp) {
    const bool inside = (JR_TUNE_DIAG & 4) ? false : strictly_inside_t<FAST>(b);    // (diagnostic bit 2: what do the inside pairs' extra projections cost?)
    const int v0 = outside_edge(r, face_obtuse(meta), b, xp, yp);
    Dist d;
    const EdgeCand c = edge_candidate<FAST, TV>(r, b, inside ? 0 : (v0 < 0 ? 0 : v0), !inside);
    if (inside) {
        // SRK:68-105: dis_min starts at 1e8, strict '<' keeps the first of equal candidates
        float best = 100000000.f;
        d.dx = 0.f; d.dy = 0.f; d.t0 = 0.f; d.t1 = 0.f; d.t2 = 0.f;
        if (c.dd < best) { best = c.dd; d.dx = c.ex; d.dy = c.ey; d.t0 = c.u0; d.t1 = c.u1; d.t2 = c.u2; }
        const EdgeCand c1 = edge_candidate<FAST, TV>(r, b, 1, false);
        if (c1.dd < best) { best = c1.dd; d.dx = c1.ex; d.dy = c1.ey; d.t0 = c1.u0; d.t1 = c1.u1; d.t2 = c1.u2; }
        const EdgeCand c2 = edge_candidate<FAST, TV>(r, b, 2, false);
        if (c2.dd < best) { best = c2.dd; d.dx = c2.ex; d.dy = c2.ey; d.t0 = c2.u0; d.t1 = c2.u1; d.t2 = c2.u2; }
        d.sign = 1.f;
    } else if (v0 < 0) {
        // Reference indexes t[-1]/a0[-1] here (undefined behaviour; only reachable when some
        // w >= 1 by rounding while none is <= 0).  Defined like the oracle: distance 0.
        d.sign = -1.f; d.dx = 0.f; d.dy = 0.f; d.t0 = 0.f - b.w0; d.t1 = 0.f - b.w1; d.t2 = 0.f - b.w2;
    } else {
        d.sign = -1.f; d.dx = c.ex; d.dy = c.ey; d.t0 = c.u0; d.t1 = c.u1; d.t2 = c.u2;
    }
    return d;
}

// What the FORWARD needs of it: the sign and the squared distance dx*dx + dy*dy (SRK:341-342), which is
// bit for bit the candidate's own dd = ex*ex + ey*ey; the nearest point and its barycentric offsets (six
// selects per candidate) are only used by the backward.  An inside pixel is never culled by distance, so
// its three projections only feed the coverage sigmoid (colour path, 1e-4): the 2nd and 3rd use a
// reciprocal multiply - UNLESS the alpha aggregation is 'hard' (EXACT_INSIDE: those launches run their own kernel
// instantiations, DIST = 3; a wave-uniform run-time test in this place cost the other modes 2 % of the forward): D > 0.5
// is then a decision taken from this very number (alpha_accumulate), and for a pixel centre within float noise of an
// edge the distance IS noise - the reference's noise has to be reproduced bit for bit (fuzz seed 61 case 141, round 4:
// true x / sigma = -9e-9, the reference's float arithmetic makes it < -9e-8, a 2-ulp quotient made it something else).
template <bool FAST, bool EXACT_INSIDE>
__device__ inline void euclidean_sign_dis(const FaceGeo& r, int meta, const Bary& b, float xp, float yp, float& sign, float& dis) {
    const bool inside = strictly_inside_t<FAST>(b);
    const int v0 = outside_edge(r, face_obtuse(meta), b, xp, yp);
    const EdgeCand c = edge_candidate<FAST>(r, b, inside ? 0 : (v0 < 0 ? 0 : v0), !inside);
    if (inside) {
        float best = 100000000.f;                    // SRK:68: candidates that are not < 1e8 (NaN) leave dis = 0
        if (c.dd < best) best = c.dd;
        constexpr int TVI = (tune::fwd_inside_rcp && !EXACT_INSIDE) ? TV_RCP : TV_IEEE;
        const float d1 = edge_candidate<FAST, TVI>(r, b, 1, false).dd;
        const float d2 = edge_candidate<FAST, TVI>(r, b, 2, false).dd;
        if (d1 < best) best = d1;
        if (d2 < best) best = d2;
        sign = 1.f;
        dis = best < 100000000.f ? best : 0.f;
    } else {
        sign = -1.f;
        dis = v0 < 0 ? 0.f : c.dd;
    }
}

// The two halves of euclidean_sign_dis for callers that know the class of the pair (the heavy tile's evaluate passes):
// an OUTSIDE pixel's squared distance (SRK:107-146; sign = -1), decided from exactly ...
template <bool FAST>
__device__ inline float euclidean_outside_dis(const FaceGeo& r, int meta, const Bary& b, float xp, float yp) {
    const int v0 = outside_edge(r, face_obtuse(meta), b, xp, yp);
    const float dd = edge_candidate<FAST>(r, b, v0 < 0 ? 0 : v0, true).dd;
    return v0 < 0 ? 0.f : dd;
}
// ... and an INSIDE pixel's (SRK:68-105; sign = +1): nearest of the three edge projections, strict '<' from 1e8.
// Only the coverage sigmoid reads it (colour path): the 2nd / 3rd projection use the reciprocal multiply.
template <bool FAST, bool EXACT_INSIDE>
__device__ inline float euclidean_inside_dis(const FaceGeo& r, const Bary& b) {
    float best = 100000000.f;                        // SRK:68: candidates that are not < 1e8 (NaN) leave dis = 0
    const float d0 = edge_candidate<FAST>(r, b, 0, false).dd;
    if (d0 < best) best = d0;
    constexpr int TVI = (tune::fwd_inside_rcp && !EXACT_INSIDE) ? TV_RCP : TV_IEEE;   // 'hard' alpha decides D > 0.5 from this distance: the reference's bits (see euclidean_sign_dis)
    const float d1 = edge_candidate<FAST, TVI>(r, b, 1, false).dd;
    const float d2 = edge_candidate<FAST, TVI>(r, b, 2, false).dd;
    if (d1 < best) best = d1;
    if (d2 < best) best = d2;
    return best < 100000000.f ? best : 0.f;
}

__device__ inline float barycentric_dist(const Bary& b) {                             // SRK:150-154
    const float m = b.w0 > b.w1 ? (b.w1 > b.w2 ? b.w2 : b.w1) : (b.w0 > b.w2 ? b.w2 : b.w0);
    return m > 0 ? m * m : -m * m;
}

// ---- colour path (1e-4 tolerance, not bit-critical: nothing here feeds the face-index buffer) ----
// exp(x) as v_exp_f32(x * log2 e): relative error ~ 1.2e-7 * (1 + |x|), i.e. < 3e-6 for every
// argument whose result is not negligible (|x| < 20); libm's expf costs 3x more.
__device__ inline float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
// x / sigma as a reciprocal multiply (<= 1 ulp off the IEEE quotient)
__device__ inline float over_sigma(float x, const RasterParams& p) {
    return p.consts_safe ? x * p.r_sigma : x / p.sigma;
}
// exp(x / gamma) of the softmax weights, x = zn - smax <= 0 (SRK:401-411, :1308).  Colour path: the weights only
// have to be good to 1e-4 where they are not negligible (|x / gamma| < ~20), so the quotient and the change of
// base are ONE multiply by the precomputed log2(e) / gamma (argument error <= 2 ulp: relative error of the weight
// 1.2e-7 * |x / gamma|).  zn itself keeps the reference's exact bits (div_known at the call sites): differences of
// it are divided by gamma, an ulp there would be 6e-4 of the weight.
// EX (tune::fwd_exact / tune::bwd_exact, bit 1): the reference's own form - IEEE quotient, libm exponential - for the
// A/B that prices what the approximation costs in gradient parity (tools/grad_parity.py).
template <int EX = 0>
__device__ inline float exp_over_gamma(float x, const RasterParams& p) {
    if (EX & 2) return expf(x / p.gamma);
    return p.consts_safe ? __builtin_amdgcn_exp2f(x * p.rg_log2e) : fast_exp(x / p.gamma);
}
// sigmoid coverage 1/(1+exp(neg_num/sigma)) (SRK:338, :344; the reference adds and divides in double)
// EX bit 0: the reference's own form, (float)(1. / (1. + (double)expf(x / sigma)))
template <int EX = 0>
__device__ inline float coverage_fast(float neg_num, const RasterParams& p) {
    if (EX & 1) return (float)(1.0 / (1.0 + (double)expf(neg_num / p.sigma)));
    const float e = p.consts_safe ? __builtin_amdgcn_exp2f(neg_num * p.rs_log2e) : fast_exp(neg_num / p.sigma);
    return __builtin_amdgcn_rcpf(1.0f + e);
}

// The same coverage for the BACKWARD, where D * (1 - D) / sigma scales the whole distance gradient (SRK:1336): the plain
// form rounds 1.0f + e first, and for e << 1 that quantises 1 - D to the ulp ABOVE 1 - twice the ulp below it, on which
// the reference's D = (float)(1. / (1. + e)) lives.  With e = 1.58e-7 the reference holds 1 - D = 3 ulp, the plain form
// 2 ulp: that pair's gradient comes out at 2/3 of the reference's (tests/golden/regress_saturated_coverage.npz: a 17^2
// image with sigma 1e-6 has no unsaturated pair to hide it behind, 1.1e-3 of the largest gradient).  For e < 2^-10 the
// quotient is 1 - e (1 - e) + O(e^3) with e^3 < 1e-9, far below the rounding step: ONE rounding of that value is the
// reference's double quotient rounded to float.  Above 2^-10, 1 - D >= 1e-3 and an ulp of D is 1e-4 of it or less.
template <int EX = 0>
__device__ inline float coverage_backward(float neg_num, const RasterParams& p) {
    if (EX & 1) return (float)(1.0 / (1.0 + (double)expf(neg_num / p.sigma)));
    const float e = p.consts_safe ? __builtin_amdgcn_exp2f(neg_num * p.rs_log2e) : fast_exp(neg_num / p.sigma);
    const float far_ = __builtin_amdgcn_rcpf(1.0f + e);
    const float near_ = __builtin_fmaf(-e, 1.0f - e, 1.0f);
    return e < 0x1p-10f ? near_ : far_;
}

// 'surface' sampler texel choice (SRK:159-166, identical in SRK:1138-1145)
__device__ inline int surface_texel(const Bary& c, int R) {
    const int wx = (int)fminf(c.w0 * R, (float)(R - 1));
    const int wy = (int)fminf(c.w1 * R, (float)(R - 1));
    if (((c.w0 + c.w1) * R - wx) - wy <= 1) return wy * R + wx;
    return (R - 1 - wy) * R + (R - 1 - wx);
}

// ---- instrumented builds only (tune::profile_sections): wall-clock (s_memtime) totals per kernel section,
// summed over wavefronts into BinWorkspace::counters[4 + section] ---------------------------------
struct SectionClock {
    unsigned long long t, acc[8];
    __device__ inline void start() {
        if (tune::profile_sections) {
            t = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = 0;
        }
    }
    __device__ inline void lap(int section) {
        if (tune::profile_sections) {
            const unsigned long long n = __builtin_amdgcn_s_memtime();
            acc[section] += n - t;
            t = n;
        }
    }
    __device__ inline void flush0(unsigned long long* out, int base) {       // the caller picks the lane
        if (tune::profile_sections) {
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (acc[i]) atomicAdd(out + base + i, acc[i]);
        }
    }
    __device__ inline void flush(unsigned long long* out, int base) {
        if (tune::profile_sections && threadIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (acc[i]) atomicAdd(out + base + i, acc[i]);
        }
    }
};

// ---- wavefront helpers (64 lanes) ----------------------------------------------------------
__device__ inline unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// Value of a 64-bit wave-uniform table entry selected by a small per-lane index (0..7): the
// compiler lowers this to v_cndmask chains on SGPR operands.
__device__ inline unsigned long long select8(const unsigned long long (&tab)[8], int idx) {
    unsigned long long m = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) m = idx == c ? tab[c] : m;
    return m;
}

}  // namespace jr
