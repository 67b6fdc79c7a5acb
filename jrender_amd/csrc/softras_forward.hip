// SoftRas forward rasteriser for gfx950 (MI355X).
//
// Replaces forward_soft_rasterize_cuda_kernel (SRK:243-456).  The reference runs one thread per
// pixel over ALL faces.  Here ONE WAVEFRONT owns an 8x8-pixel tile and walks its 32x32 bin's ascending
// face list with the roles of the lanes switched between phases:
//
//   cull   (lane = list entry)  64 entries at a time: drop an entry unless its tile mask has this tile's
//                         bit and its border box can touch the tile; survivors copy their packed record
//                         into LDS, COMPACTED across list chunks (ascending order kept) until 64 slots are
//                         full — the raster loop's trip count is the maximum over the pixels, and max/mean
//                         shrinks with the batch.
//   ballot (lane = slot)  the reference's border test (SRK:28-34) against the tile's 8 column and 8 row
//                         pixel centres.  Every compare is a v_cmp whose 64-bit result IS the wavefront
//                         ballot over the 64 slots: 32 compares decide 64 faces x 64 pixels.
//   raster (lane = pixel) each pixel ANDs its column ballot with its row ballot: a private bitmask of the
//                         faces that pass ITS border test.  It pops its own bits in ascending face order
//                         and runs the per-(pixel,face) arithmetic on the LDS record of ITS face — lanes
//                         stay busy although neighbouring pixels see different face subsets.
//
// Tiles are visited heaviest bin first (k_bin_alloc_schedule), bins dealt round-robin to the XCDs: the list
// length varies from 1 to >1000 faces and the launch would otherwise end with a long tail.
//
// The per-pixel state machine (alpha, online softmax over depth, K-nearest buffer) lives in VGPRs;
// it is sequential in face order, which is why the lists are sorted.  No MFMA: no dense contraction.
//
// Exactness: everything that decides the face-index buffer (border test, barycentrics, distance,
// distance cull, clipped depth, depth cull, K-buffer) reproduces the reference bit for bit.  The
// colour path (coverage sigmoid, softmax weights) only has to stay within 1e-4 and uses
// reciprocal multiplies where the reference divides by sigma / gamma (error <= 1-2 ulp).
#include <stdlib.h>

#include <type_traits>
#include "jr_kernels.h"

// This file is compiled TWICE (round 5): as it is -> namespace jr, the product's forward; and through
// softras_forward_precise.hip with JR_TUNE_FWD_EXACT = 3 -> namespace jr_precise, the same kernels with the colour path in
// the reference's own arithmetic (libm expf, IEEE quotients, the double-precision sigmoid: SRK:344, :401-411) for callers
// that need the saved colours - and through them the gradients - within 1e-4 ELEMENT-WISE of the reference
// (jr_softras_set_precise_colour; +15 % forward time).  Every name of this file lives in that namespace, so
// the two sets of kernels are distinct symbols.
#ifndef JR_FWD_NAMESPACE
#define JR_FWD_NAMESPACE jr
#endif
namespace JR_FWD_NAMESPACE {
using namespace jr;

// The K-buffer's face indices never feed a decision, so they do not live in registers: every insert stores the face
// index straight into its slot plane of faces_id_buffer (the output), the slots that were never filled get their -1
// at the end.  Same-address stores of one wavefront complete in program order, so the plane ends up with the slot's
// LAST face, exactly what a register K-buffer would have held.  (In registers the ids cost KCAP v_cndmask per insert
// AND - being live beyond the divergent raster loop - a second register set with KCAP v_mov to copy them back: seen
// in the ISA.  tune::fwd_ids_global = 0 rebuilds that form for the A/B.)
template <int KCAP>
constexpr bool ids_in_global() { return tune::fwd_ids_global; }

template <int KCAP>
struct KBuffer {
    static constexpr bool IDS_GLOBAL = ids_in_global<KCAP>();
    // tune::fwd_fill_shift.  71 % of the pairs that reach the K-buffer are APPENDS (slot = size, a per-lane index:
    // KCAP v_cmp + KCAP v_cndmask at 4.3 cycles each), 7.5 % replace the largest depth (oracle statistics of the
    // headline scene).  Appends therefore shift: z[k] = z[k-1], z[0] = zp - KCAP v_mov at 2.7 cycles, no compares -
    // and slot s lives in register size-1-s; once the buffer is full the map is the constant K-1-s, which is all
    // the replace path needs.  (The ids are not in registers, so nothing else has to follow the shift.)
    static constexpr bool SHIFT = tune::fwd_fill_shift && IDS_GLOBAL;
    int id[IDS_GLOBAL ? 1 : KCAP];
    int32_t* gplane;               // IDS_GLOBAL: slot plane 0 of this pixel's image in faces_id_buffer
    unsigned goff, gstride;        //             element offset of the pixel, elements per slot plane
    float z[KCAP];
    int size;
    float max_z;
    int max_slot;

    // Slots >= K are never written; their depth is -inf so that the rescan (which starts from -1) can
    // run over all KCAP registers without a per-slot "k < K" predicate (16 SGPR pairs otherwise).
    __device__ inline void init(int K, int32_t* plane0, unsigned pixel, unsigned stride) {
        gplane = plane0; goff = pixel; gstride = stride;
#pragma unroll
        for (int k = 0; k < KCAP; k++) {
            if (IDS_GLOBAL) { if (k == 0) id[0] = -1; }
            else id[k] = -1;
            z[k] = (k < K && !SHIFT) ? 0.f : -__builtin_inff();   // SHIFT: K appends move the first K initial values beyond register K-1
        }
        size = 0; max_z = -1.f; max_slot = -1;
    }
    __device__ inline int id_of(int k) const { return id[IDS_GLOBAL ? 0 : k]; }
    // largest depth by a v_max3 tree, then the FIRST slot that holds it; equals the reference's strict '>'
    // scan from -1 (SRK:379-385), NaN depths are skipped by both.  No serial compare-select chain.
    __device__ inline void rescan(int K) {
        float t[KCAP];
#pragma unroll
        for (int k = 0; k < KCAP; k++) t[k] = z[k];
#pragma unroll
        for (int w = KCAP; w > 1; w = (w + 2) / 3) {
#pragma unroll
            for (int i = 0; i * 3 < w; i++) {
                const float a = t[3 * i], b = 3 * i + 1 < w ? t[3 * i + 1] : a, c = 3 * i + 2 < w ? t[3 * i + 2] : a;
                t[i] = fmaxf(fmaxf(a, b), c);
            }
        }
        const float m = fmaxf(t[0], -1.f);
        int ms = max_slot;
        if (m > -1.f) {
            if (SHIFT) {           // slot s sits in register K-1-s: the first slot is the LAST register that holds m
                int mr = 0;
#pragma unroll
                for (int k = 0; k < KCAP; k++) mr = z[k] == m ? k : mr;
                ms = K - 1 - mr;
            } else {               // descending selects, so the smallest k is written last
#pragma unroll
                for (int k = KCAP - 1; k >= 0; k--) ms = z[k] == m ? k : ms;
            }
        }
        max_z = m; max_slot = ms;
    }
    // K-nearest insert with the reference's slot semantics (SRK:369-385): append while not
    // full (tracking the first largest depth), afterwards overwrite the largest-depth slot
    // when strictly nearer and rescan, first maximum wins.
    // -> the slot the face went to, -1 when it was not nearer than the largest buffered depth
    __device__ inline int insert(int fn, float zp, int K) {
        if (JR_TUNE_DIAG & 2) return -1;   // diagnostic builds only (empty index buffer): what does the K-buffer cost?
        const bool filling = size < K;
        if (!filling && !(zp < max_z)) return -1;
        const int slot = filling ? size : max_slot;
        if (IDS_GLOBAL && !(JR_TUNE_DIAG & 64)) gplane[(unsigned)slot * gstride + goff] = fn;     // (diagnostic bit 6: what do the per-insert id stores cost?)
        if (SHIFT) {
            if (filling) {
#pragma unroll
                for (int k = KCAP - 1; k > 0; k--) z[k] = z[k - 1];
                z[0] = zp;
                if (zp > max_z) { max_z = zp; max_slot = size; }
                size++;
            } else {
                const int reg = K - 1 - slot;
#pragma unroll
                for (int k = 0; k < KCAP; k++) z[k] = k == reg ? zp : z[k];
                rescan(K);
            }
            return slot;
        }
#pragma unroll
        for (int k = 0; k < KCAP; k++) {
            const bool hit = k == slot;
            if (!IDS_GLOBAL) id[k] = hit ? fn : id[k];
            z[k] = hit ? zp : z[k];
        }
        if (filling) {
            if (zp > max_z) { max_z = zp; max_slot = size; }
            size++;
        } else rescan(K);
        return slot;
    }
};

template <int KCAP>
struct PixelState {                 // SRK:291-309
    float c0, c1, c2, alpha, ssum, smax, depth_min;
    int face_min;
    KBuffer<KCAP> q;
};

// colour of the face at the clipped barycentric point (SRK:156-173)
template <bool FAST>
__device__ inline void sample_colour(const RasterParams& p, const FaceRec& r, const float* vc,
                                     const float* __restrict__ tbase, const Bary& wc, float zp,
                                     float& k0, float& k1, float& k2) {
    if (p.tex == 0) {
        if (p.T == 1) { k0 = r.col[0]; k1 = r.col[1]; k2 = r.col[2]; }
        else {
            const float* tx_ = tbase + ((size_t)face_id(r.meta) * p.T + surface_texel(wc, p.R)) * 3;
            k0 = tx_[0]; k1 = tx_[1]; k2 = tx_[2];
        asm volatile("" : "+v"(k0), "+v"(k1), "+v"(k2));          // (the load's wait stays in this branch: see sample_colour)
            // Round 6: consume the load HERE.  Left to the compiler, the wait for it lands at the join with the single-texel path as
            // `s_waitcnt vmcnt(0)` - and on gfx950 stores and atomics count on the same in-order counter, so EVERY trip of the raster loop,
            // T = 1 launches included, waited there for the id store it had just issued (the backward: for its gradient atomics).
            asm volatile("" : "+v"(k0), "+v"(k1), "+v"(k2));
        }
    } else {                                                                   // SRK:168-171
        k0 = ((wc.w0 * vc[0] / r.z[0] + wc.w1 * vc[3] / r.z[1]) + wc.w2 * vc[6] / r.z[2]) * zp;
        k1 = ((wc.w0 * vc[1] / r.z[0] + wc.w1 * vc[4] / r.z[1]) + wc.w2 * vc[7] / r.z[2]) * zp;
        k2 = ((wc.w0 * vc[2] / r.z[0] + wc.w1 * vc[5] / r.z[1]) + wc.w2 * vc[8] / r.z[2]) * zp;
    }
}

// online softmax over the normalised depth (SRK:399-419) and the colour sums it weights
template <bool FAST, int KCAP>
__device__ inline void softmax_accumulate(const RasterParams& p, const FaceRec& r, const float* vc,
                                          const float* __restrict__ tbase, const Bary& wc, float zp, float D,
                                          PixelState<KCAP>& s) {
    if (JR_TUNE_DIAG & 1) return;          // diagnostic builds only (wrong colours): what does the softmax update cost?
    // zn must carry the reference's exact bits: the softmax divides differences of it by gamma
    const float zn = div_known<FAST>(p.far_ - zp, p.far_minus_near, p.r_far_minus_near);
    float ed, ez;
    if (tune::fwd_exp1) {
        // one of the reference's two exponentials is always exp(0) = 1 (after "smax = zn" the second argument is 0):
        // ONE v_exp of -|zn - smax| and two selects give the same two values, bit for bit
        const float x = zn - s.smax;
        const bool up = x > 0.f;                       // zn > smax (NaN: false, like the reference's compare)
        const float e = exp_over_gamma<tune::fwd_exact>(-fabsf(x), p);  // (neg / abs are operand modifiers)
        ed = up ? e : 1.f;
        ez = up ? 1.f : e;
        s.smax = fmaxf(s.smax, zn);                    // (a NaN zn leaves smax alone, like the compare)
    } else {
        ed = 1.f;
        if (zn > s.smax) { ed = exp_over_gamma<tune::fwd_exact>(s.smax - zn, p); s.smax = zn; }
        ez = exp_over_gamma<tune::fwd_exact>(zn - s.smax, p);
    }
    float k0, k1, k2;
    sample_colour<FAST>(p, r, vc, tbase, wc, zp, k0, k1, k2);
    if (tune::fwd_exp1) {                              // colour path (1e-4): fused multiply-adds
        const float t = ez * D;
        s.ssum = __builtin_fmaf(ed, s.ssum, t);
        s.c0 = __builtin_fmaf(ed, s.c0, t * k0);
        s.c1 = __builtin_fmaf(ed, s.c1, t * k1);
        s.c2 = __builtin_fmaf(ed, s.c2, t * k2);
    } else {
        s.ssum = ed * s.ssum + ez * D;
        s.c0 = ed * s.c0 + ez * D * k0;
        s.c1 = ed * s.c1 + ez * D * k1;
        s.c2 = ed * s.c2 + ez * D * k2;
    }
}

// alpha aggregation (SRK:350-358); neg_num = the sigmoid's numerator -sign*dis (any negative value for 'hard' distance)
template <int DIST, bool FAST, int KCAP>
__device__ inline void alpha_accumulate(const RasterParams& p, float neg_num, float D, PixelState<KCAP>& s) {
    if (p.alpha == 0) {
        // 'hard' alpha is a DECISION (D > 0.5), so it must not ride on the approximate sigmoid.  With the
        // reference's arithmetic, D = (float)(1/(1 + (double)expf(x))), x = neg_num/sigma (float division),
        // D > 0.5 holds exactly when x < -1.5 * 2^-24 (monotone; checked float by float around the boundary).
        const float x = (DIST == 0) ? -1.f
                      : ((neg_num == 0.f || in_fast_range(neg_num)) ? div_known<FAST>(neg_num, p.sigma, p.r_sigma)
                                                                     : neg_num / p.sigma);
        if (x < -8.940696716308594e-08f) s.alpha = 1.f;
    }
    else if (p.alpha == 1) s.alpha += D;
    else {
        // SRK:357 multiplies in double and rounds to float: alpha * (1 - D) with 1 - D exact.  One float fma
        // returns the correctly rounded alpha - alpha * D — the same value up to the (rare, half-ulp) double rounding of
        // the reference; alpha only has to meet 1e-4.
        s.alpha = __builtin_fmaf(-s.alpha, D, s.alpha);
    }
}

// Instrumented build JR_TUNE_COUNT_PATHS (tools/sim/min_valu.py --measure): how many TRIPS of the raster loop execute each
// region of forward_pair and how many LANES need it, per launch - the dynamic half of the VALU model.  Regions: 0 every pair,
// 1 inside (two extra edge projections), 2 live (passed the distance cull), 3 K-buffer insert, 4 append, 5 replace + rescan,
// 6 softmax update, 7 the SLOW (non-FAST face) copy of the loop.  Dead code in every other build.
struct PathCount {
    unsigned t[8], l[8], batches, chunks;
    __device__ inline void clear() {
#pragma unroll
        for (int i = 0; i < 8; i++) { t[i] = 0; l[i] = 0; }
        batches = 0; chunks = 0;
    }
    __device__ inline void hit(int i) {          // called by every lane that executes region i in this trip
        if (!tune::count_paths) return;
        const unsigned long long m = ballot(true);
        const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        l[i] += 1;
        if (lane == (int)__builtin_ctzll(m)) t[i] += 1;
    }
    __device__ inline void flush(unsigned long long* counters, int lane) {     // all lanes
        if (!tune::count_paths) return;
        auto wsum = [](unsigned v) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
            return v;
        };
#pragma unroll
        for (int i = 0; i < 7; i++) {
            const unsigned a = wsum(t[i]), b = wsum(l[i]);
            if (lane == 0) { atomicAdd(counters + 4 + 2 * i, (unsigned long long)a); atomicAdd(counters + 5 + 2 * i, (unsigned long long)b); }
        }
        const unsigned a = wsum(t[7]), b = wsum(l[7]);
        if (lane == 0) {
            atomicAdd(counters + 18, (unsigned long long)batches); atomicAdd(counters + 19, (unsigned long long)chunks);
            atomicAdd(counters + 20, 1ull); atomicAdd(counters + 21, (unsigned long long)a); atomicAdd(counters + 22, (unsigned long long)b);
        }
    }
};

// One (pixel, face) pair of the raster loop (SRK:316-419).  (Handling the pairs whose pixel lies strictly inside the face -
// three edge projections instead of one, 8 % of the pairs but present in nearly every trip - in a second loop per batch
// was built and measured: VALU instructions -6.7 %, time +2.6 %; tools/ablate/patches/dead_switches_r04.patch.)
template <int DIST, int RGB, bool FAST, int KCAP>
__device__ inline void forward_pair(const RasterParams& p, const FaceRec& r, const float* vc,
                                    const float* __restrict__ tbase, float xp, float yp,
                                    PixelState<KCAP>& s, PathCount& pc) {
    const Bary w = barycentric(r, xp, yp);
    const int meta = r.meta;
    float D = 1.f, neg_num = -1.f;
    pc.hit(0);
    if (tune::count_paths && !FAST) pc.hit(7);
    if (tune::count_paths && DIST >= 2 && strictly_inside_t<FAST>(w)) pc.hit(1);
    if (DIST == 0) {                                                           // SRK:331-333
        if (!pixel_inside(w)) return;
    } else if (DIST == 1) {                                                    // SRK:335-338
        const float dis = barycentric_dist(w);
        if (-dis >= p.thr) return;
        neg_num = -dis;
        D = coverage_fast<tune::fwd_exact>(neg_num, p);
    } else {                                                                   // SRK:340-344
        float sign, dis;
        if (tune::fwd_dis_only) euclidean_sign_dis<FAST, DIST == 3>(r, meta, w, xp, yp, sign, dis);
        else {
            const Dist dd = euclidean_p2f<FAST>(r, meta, w, xp, yp);
            sign = dd.sign;
            dis = dd.dx * dd.dx + dd.dy * dd.dy;
        }
        if (sign < 0 && dis >= p.thr) return;
        neg_num = -sign * dis;
        D = coverage_fast<tune::fwd_exact>(neg_num, p);
    }
    // alpha aggregation happens before the depth cull (SRK:350-358)
    pc.hit(2);
    alpha_accumulate<DIST, FAST>(p, neg_num, D, s);

    const Bary wc = barycentric_clip<FAST>(w);
    const float zp = depth_of<FAST>(r, wc);
    if (zp < p.near_ || zp > p.far_) return;                                  // SRK:365
    const int fn = face_id(meta);
    if (tune::count_paths) {
        const bool was_filling = s.q.size < p.K;
        if (s.q.insert(fn, zp, p.K) >= 0) { pc.hit(3); if (was_filling) pc.hit(4); else pc.hit(5); }
    } else s.q.insert(fn, zp, p.K);

    if (RGB == 0) {                                                            // SRK:390-397
        if (zp < s.depth_min && pixel_inside(w) && (p.double_side || face_front(meta))) {
            s.depth_min = zp; s.face_min = fn;
            sample_colour<FAST>(p, r, vc, tbase, wc, zp, s.c0, s.c1, s.c2);
        }
    } else if (RGB == 1) {                                                     // SRK:399-419
        if (face_front(meta) || p.double_side) { pc.hit(6); softmax_accumulate<FAST>(p, r, vc, tbase, wc, zp, D, s); }
    }
}

// wavefronts per SIMD asked of the register allocator: K <= 16: the single-wavefront kernel fits 96 VGPRs (5), the
// four-wavefront one 128 (4); K <= 32: 168 (3); K <= 64: 256 (2) - all without scratch
constexpr int fwd_waves(int kcap, bool mixed, int rgb = 1, int dist = 2) {
    // ('hard' rgb at K <= 64 keeps depth_min / face_min / the colour next to 64 K-buffer depths: 12 - 16 B of scratch at three
    //  wavefronts per SIMD, none at two - VERDICT r4 next #6)
    // (and the all-'hard' kernel <0,0,16> carried 8 B at five wavefronts per SIMD: four)
    return kcap <= 16 ? (mixed ? 4 : (rgb == 0 && dist == 0 && tune::fwd_waves16 > 4 ? 4 : tune::fwd_waves16))
                      : (kcap <= 32 ? (mixed ? 3 : tune::fwd_waves32) : (mixed || rgb == 0 ? 2 : tune::fwd_waves64));   // (the four-wavefront kernel spills at one more)
}

// LDS hand-over inside ONE wavefront (writes by some lanes, reads by others): LDS instructions of a wavefront
// execute in order, so only the compiler has to be kept from moving accesses across this point.  WAVE_IS_WG: the
// workgroup is a single wavefront and __syncthreads() is the same thing (the round-2 kernel).
template <bool WAVE_IS_WG>
__device__ inline void wave_sync() {
    if (WAVE_IS_WG) __syncthreads();
    else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// What a tile's wavefront knows about its place: the bin, the image, the pixel of this lane.
struct TileGeom {
    int b, bin, sub, n;                 // image, bin, tile of the bin (0..15), listed faces of the bin
    int col, row; bool valid;           // this lane's pixel
    float xp, yp;                       // its centre in NDC (SRK:280-283)
};

__device__ inline bool tile_geom(const RasterParams& p, int bin, int sub, int n, int lane, TileGeom& t, int bl) {      // bl = log2 of the bin size (a constant in the headline kernel)
    const int bins_per_img = p.bins_x * p.bins_y;
    t.bin = bin; t.sub = sub; t.n = n;
    t.b = bin / bins_per_img;
    const int bb = bin - t.b * bins_per_img;
    const int by = bb / p.bins_x, bx = bb - by * p.bins_x;
    const int sl = bl - TILE_LOG2;
    const int col0 = (bx << bl) + ((sub & ((1 << sl) - 1)) << TILE_LOG2);
    const int row0 = (by << bl) + ((sub >> sl) << TILE_LOG2);
    if (col0 >= p.IS || row0 >= p.IS) return false;      // tile lies outside the image
    t.col = col0 + (lane & 7); t.row = row0 + (lane >> 3);
    t.valid = t.col < p.IS && t.row < p.IS;
    t.xp = pixel_centre(t.col, p.IS);
    t.yp = pixel_centre(p.IS - 1 - t.row, p.IS);
    return true;
}

template <int RGB, int KCAP>
__device__ inline void init_colour_state(const RasterParams& p, PixelState<KCAP>& s) {
    s.c0 = 1.f; s.c1 = 1.f; s.c2 = 1.f;                                       // SRK:291-309
    s.alpha = p.alpha == 2 ? 1.f : 0.f;
    s.ssum = expf(p.eps / p.gamma); s.smax = p.eps;
    if (RGB == 0) { s.c0 = p.bg[0]; s.c1 = p.bg[1]; s.c2 = p.bg[2]; }
    else if (RGB == 1) { s.c0 = p.bg[0] * s.ssum; s.c1 = p.bg[1] * s.ssum; s.c2 = p.bg[2] * s.ssum; }
    s.depth_min = 10000000.f;
    s.face_min = -1;
}
template <class KB>
__device__ inline void init_kbuffer(const RasterParams& p, const TileGeom& t, int32_t* __restrict__ ids, KB& q) {
    q.init(p.K, ids + (size_t)t.b * p.K * p.IS * p.IS, t.valid ? (unsigned)(t.row * p.IS + t.col) : 0u, (unsigned)(p.IS * p.IS));
}
template <int RGB, int KCAP>
__device__ inline void init_pixel_state(const RasterParams& p, const TileGeom& t, int32_t* __restrict__ ids, PixelState<KCAP>& s) {
    init_colour_state<RGB>(p, s);
    init_kbuffer(p, t, ids, s.q);
}

// ---- finalise (SRK:426-455): colour planes / K-buffer planes (two wavefronts hold the two halves of a heavy tile's state) ----
template <int RGB, int KCAP>
__device__ inline void final_colour(const RasterParams& p, const PixelState<KCAP>& s, float (&o)[6]) {    // r g b a, aggrs_info[2]
    if (p.alpha == 0) o[3] = s.alpha;
    else if (p.alpha == 1) o[3] = s.alpha / p.NF;
    else o[3] = (float)(1. - (double)s.alpha);
    o[0] = p.bg[0]; o[1] = p.bg[1]; o[2] = p.bg[2]; o[4] = 0.f; o[5] = 0.f;
    if (RGB == 0) {
        if (s.face_min != -1) { o[0] = s.c0; o[1] = s.c1; o[2] = s.c2; }
        o[4] = s.depth_min; o[5] = (float)s.face_min;
    } else if (RGB == 1) {
        o[0] = s.c0 / s.ssum; o[1] = s.c1 / s.ssum; o[2] = s.c2 / s.ssum;
        o[4] = s.ssum; o[5] = s.smax;
    }
}
template <int RGB, int KCAP>
__device__ inline void store_colour(const RasterParams& p, const TileGeom& t, const PixelState<KCAP>& s,
                                    float* __restrict__ aggrs, float* __restrict__ rgba) {
    if (!t.valid) return;
    const size_t pp = (size_t)p.IS * p.IS;
    const size_t pn = (size_t)t.row * p.IS + t.col;
    float o[6];
    final_colour<RGB>(p, s, o);
    float* out = rgba + (size_t)t.b * 4 * pp + pn;
    out[0] = o[0]; out[pp] = o[1]; out[2 * pp] = o[2]; out[3 * pp] = o[3];
    float* ag = aggrs + (size_t)t.b * 2 * pp + pn;
    ag[0] = o[4]; ag[pp] = o[5];
}
// An EMPTY bin (no face reaches its pixels; more than half of the 32x32 bins of the headline batch): ONE wavefront writes
// the initial state's outputs for all its tiles with 16-byte stores of whole bin rows - the same values the tiles'
// own wavefronts would have stored (final_colour of the untouched state), 88 wide stores instead of 16 x 22 narrow ones
// for a 32x32 bin.  Only when image rows are 16-byte aligned and bins are whole (IS % bin == 0); other sizes take the
// per-tile path.
template <int RGB, int KCAP, int BL>          // BL = log2 of the bin size: compile-time trip counts (as run-time loops the headline forward - 60 % empty bins - lost 5 %)
__device__ inline void store_empty_bin_t(const RasterParams& p, int bin, int lane,
                                         float* __restrict__ aggrs, float* __restrict__ rgba, int32_t* __restrict__ ids) {
    const int bins_per_img = p.bins_x * p.bins_y;
    const int b = bin / bins_per_img, bb = bin - b * bins_per_img;
    const int by = bb / p.bins_x, bx = bb - by * p.bins_x;
    PixelState<KCAP> s;
    init_colour_state<RGB>(p, s);
    float o[6];
    final_colour<RGB>(p, s, o);
    const size_t pp = (size_t)p.IS * p.IS;
    // a bin row is (bin / 4) lanes x 16 B (32-pixel bins: 8 lanes = one 128-byte row); the wavefront covers 64 / that many rows per pass
    constexpr int LSH = BL - 2, BINP = 1 << BL, ROWS_PP = 64 >> LSH, PASSES = BINP >= ROWS_PP ? BINP / ROWS_PP : 1;   // 32: 4 passes, 16: 1, 8: 1 (lanes beyond the bin's 8 rows idle)
    const int lrow = lane >> LSH;
    if (lrow >= BINP) return;
    const size_t base = ((size_t)((by << BL) + lrow)) * p.IS + (bx << BL) + (lane & ((1 << LSH) - 1)) * 4;
    const size_t rstep = (size_t)ROWS_PP * p.IS;
#pragma unroll
    for (int c = 0; c < 6; c++) {
        float* pl = (c < 4 ? rgba + ((size_t)b * 4 + c) * pp : aggrs + ((size_t)b * 2 + (c - 4)) * pp) + base;
        const float4 v = make_float4(o[c], o[c], o[c], o[c]);
#pragma unroll
        for (int i = 0; i < PASSES; i++) *reinterpret_cast<float4*>(pl + i * rstep) = v;
    }
    const int4 m1 = make_int4(-1, -1, -1, -1);
    for (int k = 0; k < p.K; k++) {
        int32_t* pl = ids + ((size_t)b * p.K + k) * pp + base;
#pragma unroll
        for (int i = 0; i < PASSES; i++) *reinterpret_cast<int4*>(pl + i * rstep) = m1;
    }
}
template <int RGB, int KCAP>
__device__ inline void store_empty_bin(const RasterParams& p, int bin, int lane, int bl,
                                       float* __restrict__ aggrs, float* __restrict__ rgba, int32_t* __restrict__ ids) {
    if (bl == 5) store_empty_bin_t<RGB, KCAP, 5>(p, bin, lane, aggrs, rgba, ids);
    else if (bl == 4) store_empty_bin_t<RGB, KCAP, 4>(p, bin, lane, aggrs, rgba, ids);
    else store_empty_bin_t<RGB, KCAP, 3>(p, bin, lane, aggrs, rgba, ids);
}
template <int KCAP, bool WRITTEN_THROUGH, class KB>
__device__ inline void store_ids(const RasterParams& p, const TileGeom& t, const KB& q, int32_t* __restrict__ ids) {
    if (!t.valid) return;
    const size_t pp = (size_t)p.IS * p.IS;
    int32_t* io = ids + (size_t)t.b * p.K * pp + (size_t)t.row * p.IS + t.col;
#pragma unroll
    for (int k = 0; k < KCAP; k++)
        if (k < p.K) {
            if (!WRITTEN_THROUGH) io[(size_t)k * pp] = q.id_of(k);
            else if (k >= q.size) io[(size_t)k * pp] = -1;           // the filled slots were stored when they were filled
        }
}
template <int RGB, int KCAP>
__device__ inline void store_pixel(const RasterParams& p, const TileGeom& t, const PixelState<KCAP>& s,
                                   float* __restrict__ aggrs, float* __restrict__ rgba, int32_t* __restrict__ ids) {
    store_colour<RGB>(p, t, s, aggrs, rgba);
    store_ids<KCAP, ids_in_global<KCAP>()>(p, t, s.q, ids);
}

// ---- cull + stage: lane = list entry ------------------------------------------------------------------------
// The bin's list is walked 64 entries at a time, but only a fraction of them concerns THIS tile.
// Survivors are compacted (ascending order kept) into the LDS record slots across list chunks and the
// raster only runs on a full batch: its trip count is the MAXIMUM number of faces any pixel
// needs, and max/mean over 64 lanes shrinks with the batch (measured: 17 survivors per list chunk
// -> 56 per batch), and the per-batch ballots are paid 4x less often.
template <int DEPTH>                   // list chunks in flight ahead of the one in use (1: rounds 1-2; the pipelined heavy tile's lone walker: more)
struct ListWalkerT {
    const unsigned long long* seg;
    const FaceGeo* gbase;
    const float* tbase;
    int n, sub, s0, cnt, rank;
    bool pending, keep;
    const FaceGeo* gp;
    unsigned long long e_q[DEPTH];      // the list is read DEPTH chunks AHEAD of its use (head of a chain of dependent loads)

    __device__ inline void start(const unsigned long long* seg_, const FaceGeo* gbase_, const float* tbase_, int n_, int sub_, int lane) {
        seg = seg_; gbase = gbase_; tbase = tbase_; n = n_; sub = sub_;
        s0 = 0; cnt = 0; rank = 0; pending = false; keep = false; gp = gbase_;
#pragma unroll
        for (int d = 0; d < DEPTH; d++) e_q[d] = d * CHUNK + lane < n ? seg[d * CHUNK + lane] : 0ull;
    }
    // the next chunk's entry of this lane; the chunk DEPTH ahead is requested
    __device__ inline unsigned long long next_chunk(int lane) {
        const unsigned long long e = e_q[0];
        s0 += CHUNK;
#pragma unroll
        for (int d = 0; d + 1 < DEPTH; d++) e_q[d] = e_q[d + 1];
        const int at = s0 + (DEPTH - 1) * CHUNK + lane;
        e_q[DEPTH - 1] = at < n ? seg[at] : 0ull;
        return e;
    }
    // fills record slots [0, fill) of the next batch; 0 = the list is exhausted
    template <int BATCH>
    __device__ inline int stage(const RasterParams& p, FaceRec* s_rec, float* s_vcol, int lane) {
        int fill = 0;
        while (pending || s0 < n) {
            if (!pending) {
                const unsigned long long e = next_chunk(lane);
                // The entry's tile mask is exact per axis (binning.hip: pixel_range), i.e. the face's border box
                // reaches a pixel column AND a pixel row of this tile: no box load, no second test here.
                keep = (e >> sub) & 1ull;
                const unsigned long long surv = ballot(keep);
                if (!surv) continue;                // no face of this chunk touches this tile
                gp = gbase + (int)(e >> 32);
                cnt = __builtin_popcountll(surv);
                rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(surv >> 32),
                                                      __builtin_amdgcn_mbcnt_lo((unsigned)surv, 0u));
            }
            // take as many of the chunk's survivors as the batch still has room for (ascending order kept);
            // the rest stays pending for the next batch
            const int take = min(cnt, BATCH - fill);
            if (keep && rank < take) {
                const int slot = fill + rank;
                const float4* src = reinterpret_cast<const float4*>(gp);
                float4* dst = reinterpret_cast<float4*>(&s_rec[slot]);
#pragma unroll
                for (int k = 0; k < 11; k++) dst[k] = src[k];
                if (p.tex == 1) {
                    const float* tx_ = tbase + (size_t)(gp - gbase) * p.T * 3;
#pragma unroll
                    for (int k = 0; k < 9; k++) s_vcol[slot * 9 + k] = tx_[k];
                }
            }
            fill += take;
            pending = take < cnt;
            if (pending) {
                keep = keep && rank >= take;
                rank -= take;
                cnt -= take;
                break;
            }
        }
        return fill;
    }
    // The same with the record copies DEFERRED to one round per batch (heavy tiles): the walk only notes which face
    // goes to which slot (LDS), then lane = slot copies its record - one exposed global-load latency per batch
    // instead of one per list chunk with survivors.  (Measured +-0 on the single-wavefront kernel at full load, where
    // other wavefronts hide the latency; a heavy tile's wavefront 0 walks 23 chunks for 9 batches on its own: -5 %.)
    template <int BATCH>
    __device__ inline int stage_deferred(const RasterParams& p, FaceRec* s_rec, int* s_slot, int lane) {
        int fill = 0;
        while (pending || s0 < n) {
            if (!pending) {
                const unsigned long long e = next_chunk(lane);
                keep = (e >> sub) & 1ull;
                const unsigned long long surv = ballot(keep);
                if (!surv) continue;
                gp = gbase + (int)(e >> 32);
                cnt = __builtin_popcountll(surv);
                rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(surv >> 32),
                                                      __builtin_amdgcn_mbcnt_lo((unsigned)surv, 0u));
            }
            const int take = min(cnt, BATCH - fill);
            if (keep && rank < take) s_slot[fill + rank] = (int)(gp - gbase);
            fill += take;
            pending = take < cnt;
            if (pending) {
                keep = keep && rank >= take;
                rank -= take;
                cnt -= take;
                break;
            }
        }
        wave_sync<false>();
        if (lane < fill) {
            const float4* src = reinterpret_cast<const float4*>(gbase + s_slot[lane]);
            float4* dst = reinterpret_cast<float4*>(&s_rec[lane]);
#pragma unroll
            for (int k = 0; k < 11; k++) dst[k] = src[k];
        }
        return fill;
    }
};

using ListWalker = ListWalkerT<tune::fwd_list_depth>;

// ---- ballots + pre-cull: lane = slot -> per-pixel masks of the batch's faces ---------------------------------
// Rows [R0, R1) of the tile: the lanes of those rows get the mask of THEIR pixel (the other lanes' result is
// meaningless when the pre-cull runs); a single wavefront asks for all 8 rows, the wavefronts of a heavy tile
// for two rows each.
template <int DIST, int R0, int R1>
__device__ inline unsigned long long pixel_masks(const RasterParams& p, const FaceRec* s_rec, int fill, int lane,
                                                 float xp, float yp) {
    // the tile's 8 column / 8 row centres are the xp of lanes 0..7 and the yp of lanes 0,8,..,56:
    // read them with v_readlane where they are used instead of pinning 16 SGPRs over the raster loop
    auto xc = [&](int c) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xp), c)); };
    auto yc = [&](int c) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yp), 8 * c)); };
    float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool have = lane < fill;
    if (have) box = *reinterpret_cast<const float4*>(&s_rec[lane]);
    constexpr bool PRE = DIST >= 2 && tune::fwd_prepass;
    unsigned long long cx[8], ry[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        // check_border (SRK:28-34, :316): a pixel is culled when strictly outside the grown box
        cx[c] = ballot(have && !(xc(c) > box.y) && !(xc(c) < box.x));
        if (!PRE || (c >= R0 && c < R1)) ry[c] = ballot(have && !(yc(c) > box.w) && !(yc(c) < box.z));
    }
    if (!PRE) return select8(cx, lane & 7) & select8(ry, lane >> 3);
    // Conservative pre-cull, lane = slot.  A pixel that lies more than the cull radius beyond the
    // LINE of one edge (on its outer side) is farther than the radius from the triangle, and the
    // reference culls that pair by distance (SRK:341-342) before it touches any state.  With w_k
    // the barycentric of vertex k (affine in the pixel, |grad w_k| = g_k = 1 / altitude_k) the test is
    //     w_k(pixel) < -g_k * (rad + margin)     for some k,
    // evaluated for the face of this lane at the pixel centres: 5 VALU per pixel, and the v_cmp
    // result IS the 64-face reject mask of that pixel.
    // The margin covers what the reference's float arithmetic can make of the distance, which this
    // test does not reproduce ("pre-cull margin"; measured with tools/sim/precull_noise.py
    // on 10^7 kept pairs: the reference's distance falls short of the geometric one by at most 0.91 E):
    //   E1: face_inv is star/det with ONE rounded det, so the three w_k sum to 1 + delta instead of 1 and
    //       the reference effectively measures from the pixel displaced by (a x + b y + c - 1) * position,
    //       a, b, c = column sums of face_inv;
    //   E2: rounding of w_k itself (3 eps S_k per weight, S_k = |inv0| X + |inv1| Y + |inv2|) times the
    //       vertex positions, and of the three products that form the offset vector.
    // margin = 2.5 (E1 + E2) + 1e-4 rad; a face whose margin would exceed rad / 2, or that is outside the
    // fast-arithmetic range, never rejects.  Survivors run the exact arithmetic as before: the pre-cull
    // can only remove work, never change a result.
    float gx[3], gy[3], cc[3];
    {
        const FaceRec& me = s_rec[have ? lane : 0];
        constexpr float EPS = 5.9604645e-08f;     // 2^-24
        const float X = fmaxf(fabsf(box.x), fabsf(box.y)), Y = fmaxf(fabsf(box.z), fabsf(box.w));
        const float pos = __builtin_sqrtf(__builtin_fmaf(X, X, Y * Y));
        const float ext = (box.y - box.x) + (box.w - box.z);
        const float* vx = &me.x0;
        float g[3], gmax = 0.f, ssum = 0.f, sv = 0.f;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            gx[q] = me.inv[3 * q]; gy[q] = me.inv[3 * q + 1];
            g[q] = __builtin_sqrtf(__builtin_fmaf(gx[q], gx[q], gy[q] * gy[q]));
            gmax = fmaxf(gmax, g[q]);
            const float S = __builtin_fmaf(fabsf(gx[q]), X, __builtin_fmaf(fabsf(gy[q]), Y, fabsf(me.inv[3 * q + 2])));
            ssum += S;
            sv = __builtin_fmaf(S, __builtin_sqrtf(__builtin_fmaf(vx[2 * q], vx[2 * q], vx[2 * q + 1] * vx[2 * q + 1])), sv);
        }
        const float ca = fabsf((gx[0] + gx[1]) + gx[2]), cb = fabsf((gy[0] + gy[1]) + gy[2]);
        const float cd = fabsf(((me.inv[2] + me.inv[5]) + me.inv[8]) - 1.f);
        const float e1 = (__builtin_fmaf(ca, X, __builtin_fmaf(cb, Y, cd)) + 4.f * EPS * ssum) * pos;
        const float e2 = 3.f * EPS * sv + 4.f * EPS * __builtin_fmaf(gmax, ext, 1.f) * pos;
        const float margin = __builtin_fmaf(2.5f, e1 + e2, 1.0001f * p.rad);
        // NaN anywhere makes the comparison false -> never rejects
        const bool ok = have && face_safe(me.meta) && p.consts_safe && (margin <= 1.5f * p.rad);
#pragma unroll
        for (int q = 0; q < 3; q++) cc[q] = __builtin_fmaf(margin, g[q], me.inv[3 * q + 2]);
        if (!ok) {
#pragma unroll
            for (int q = 0; q < 3; q++) { gx[q] = 0.f; gy[q] = 0.f; cc[q] = 1.f; }
        }
    }
    int mlo = 0, mhi = 0;
#pragma unroll
    for (int rr = R0; rr < R1; rr++) {
        const float yq = yc(rr);
        const float b0 = __builtin_fmaf(gy[0], yq, cc[0]), b1 = __builtin_fmaf(gy[1], yq, cc[1]),
                    b2 = __builtin_fmaf(gy[2], yq, cc[2]);
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const float xq = xc(c);
            const float smin = __builtin_fminf(__builtin_fminf(__builtin_fmaf(gx[0], xq, b0), __builtin_fmaf(gx[1], xq, b1)),
                                               __builtin_fmaf(gx[2], xq, b2));
            const unsigned long long keep = cx[c] & ry[rr] & ~ballot(smin < 0.f);
            // v_writelane_b32: the wave-uniform mask goes into lane (rr, c) of the mask registers
            // (clang has no builtin for it)
            asm("v_writelane_b32 %0, %1, %2" : "+v"(mlo) : "s"((int)(unsigned)keep), "n"(rr * 8 + c));
            asm("v_writelane_b32 %0, %1, %2" : "+v"(mhi) : "s"((int)(unsigned)(keep >> 32)), "n"(rr * 8 + c));
        }
    }
    return ((unsigned long long)(unsigned)mhi << 32) | (unsigned)mlo;
}

// ---- one tile by ONE wavefront: lane = pixel pops the bits of its private face mask (round 1 / 2 organisation) ----
template <int DIST, int RGB, int KCAP, int BATCH, bool WAVE_IS_WG>
__device__ inline void tile_single(const RasterParams& p, const TileGeom& t, int lane, float4* s_mem,
                                   const float* __restrict__ textures, const FaceGeo* __restrict__ geo,
                                   const unsigned long long* __restrict__ seg, unsigned long long* __restrict__ counters,
                                   float* __restrict__ aggrs, float* __restrict__ rgba, int32_t* __restrict__ ids) {
    FaceRec* s_rec = reinterpret_cast<FaceRec*>(s_mem);                        // [BATCH] record slots of this wavefront
    float* s_vcol = reinterpret_cast<float*>(s_rec + BATCH);                   // [BATCH*9] iff vertex colours
    SectionClock clk;            // instrumented builds only: 0 set-up, 1 cull + stage, 2 ballots + pre-cull, 3 raster loop, 4 stores
    clk.start();
    const float xp = t.xp, yp = t.yp;
    PixelState<KCAP> s;
    init_pixel_state<RGB>(p, t, ids, s);
    const float* tbase = textures + (size_t)t.b * p.NF * p.T * 3;
    ListWalker lw;
    lw.start(seg, geo + (size_t)t.b * p.NF, tbase, t.n, t.sub, lane);
    PathCount pc;
    pc.clear();
    clk.lap(0);
    for (;;) {
        const int fill = lw.stage<BATCH>(p, s_rec, s_vcol, lane);
        if (fill == 0) break;
        if (tune::count_paths) pc.batches++;
        wave_sync<WAVE_IS_WG>();
        clk.lap(1);
        // ---- raster: lane = slot for the ballots, then lane = pixel ----
        unsigned long long M = pixel_masks<DIST, 0, 8>(p, s_rec, fill, lane, xp, yp);   // the faces that pass this pixel's border test (and pre-cull)
        if (!t.valid) M = 0ull;
        clk.lap(2);
        while (M) {
            const int j = __builtin_ctzll(M);
            M &= M - 1;
            const FaceRec& r = s_rec[j];
            const float* vc = s_vcol + j * 9;
            if (face_safe(r.meta) && p.consts_safe) forward_pair<DIST, RGB, true, KCAP>(p, r, vc, tbase, xp, yp, s, pc);
            else forward_pair<DIST, RGB, false, KCAP>(p, r, vc, tbase, xp, yp, s, pc);
        }
        wave_sync<WAVE_IS_WG>();                // readers are done with s_rec before it is refilled
        clk.lap(3);
    }
    clk.lap(1);
    store_pixel<RGB>(p, t, s, aggrs, rgba, ids);
    clk.lap(4);
    if (JR_TUNE_PROFILE_SECTIONS == 1) clk.flush(counters, 4);
    if (tune::count_paths) { pc.chunks = (unsigned)((t.n + CHUNK - 1) / CHUNK); pc.flush(counters, lane); }   // (the walk visits every chunk of the bin's list)
}

// =====================================================================================================================
// A HEAVY tile by FOUR wavefronts (round 3).  The per-pixel state machine (alpha, online softmax, K-buffer) is
// sequential in face order, and a tile at the sphere's limb has pixels that need > 300 faces: one wavefront walking
// them pinned the whole kernel (one 39k-face view alone 0.70 ms, eight views 0.88 ms).  But only alpha / softmax /
// K-buffer carry state; barycentrics -> distance -> cull -> coverage -> clip -> depth -> zn are STATELESS per pair.  So:
//
//   stage     wavefront 0 walks the bin list and stages a batch of records (as above)
//   masks     every wavefront runs the ballots + pre-cull for two of the tile's eight pixel rows
//   list      wavefront 0 (lane = pixel): prefix sum of the pixels' pair counts, then every lane writes ITS pairs
//             (slot, pixel) into a compact list, pixel-major, ascending face inside a pixel
//   evaluate  ALL wavefronts, lane = any pair of the list, 64 per trip at full lanes: the stateless arithmetic
//             -> one 16-byte cell {zp, meta, D, slot | flags} per pair; pairs that lie strictly INSIDE their face
//             (three edge projections instead of one) are pushed to a second list ...
//   inside    ... and get their coverage in dense trips of their own
//   apply     lane = pixel walks ITS contiguous cells in order, in TWO wavefronts at once: wavefront 0 owns the
//             K-buffer (it needs zp and the face id only), wavefront 1 the colour state (alpha, online softmax /
//             hard rgb: D, the depth, the colour) - the two halves of the state machine do not talk to each other,
//             and a lone wavefront issues a dependent instruction stream at ~9 clocks per instruction (measured:
//             apply was 60 % of the heaviest tile's time with one wavefront doing both)
//
// A batch whose pairs do not fit the cell buffer is cut into rounds of face-slot ranges (halved until they fit).
// Results are those of the single-wavefront path: the same device functions on the same operands, the order
// dependent steps in the same order; only the commutative alpha / softmax sums of inside pairs are unaffected here
// (they stay in face order).
// =====================================================================================================================
constexpr unsigned CELL_SLOT = 63u, CELL_LIVE = 64u, CELL_DEPTH = 128u, CELL_AHARD = 256u, CELL_INCLOSED = 512u, CELL_FRONT = 1024u;
constexpr int CELL_TEXEL_SHIFT = 12;
constexpr int HEAVY_BATCH = tune::fwd_batch_mixed;                                              // record slots of a heavy tile = of each of the four tiles of a lighter workgroup
constexpr int HEAVY_LDS_BYTES = 4 * (int)sizeof(FaceRec) * HEAVY_BATCH;                         // = what four single-wavefront tiles use
constexpr int HEAVY_FIXED_BYTES = (int)sizeof(FaceRec) * HEAVY_BATCH + 64 * 8 + 64 * 8 + 64 + 2 * 64 * 8 + 2 * HEAVY_BATCH * 16;   // records, pixel centres, masks, scalars, per-pixel (first cell, cells) of two rounds, colours + meta words of two batches
constexpr int HEAVY_CAP = ((HEAVY_LDS_BYTES - HEAVY_FIXED_BYTES) / 20) & ~63;                   // 16 B cell + 2 B pair + 2 B inside entry per pair
static_assert(HEAVY_CAP >= 256 && HEAVY_CAP <= 4096, "cell buffer of the heavy-tile path");

// stateless arithmetic of one pair -> cell (zp, zn, D, aux); `deferred`: coverage still to come (inside pair).
// Round 5: the cell carries what the COLOUR wavefront's chain used to derive per cell - the normalised depth zn (five
// dependent FMAs + the face's FAST test) and the facing test - computed here by lanes that run dense (64 pairs per trip); the
// face's meta word left the cell (the K-buffer wavefront reads the id from the batch's meta table, off its critical path:
// it only feeds the id store).
template <int DIST, int RGB, bool FAST>
__device__ inline float4 evaluate_pair(const RasterParams& p, const FaceRec& r, float xp, float yp, unsigned slot, bool& deferred) {
    const Bary w = barycentric(r, xp, yp);
    const int meta = r.meta;
    float D = 1.f, neg_num = -1.f, zp = 0.f, zn = 0.f;
    bool live;
    deferred = false;
    if (DIST == 0) live = pixel_inside(w);                                     // SRK:331-333
    else if (DIST == 1) {                                                      // SRK:335-338
        const float dis = barycentric_dist(w);
        live = !(-dis >= p.thr);
        neg_num = -dis;
        D = coverage_fast<tune::fwd_exact>(neg_num, p);
    } else {                                                                   // SRK:340-344
        deferred = strictly_inside_t<FAST>(w);
        live = true;
        if (!deferred) {
            const float dis = euclidean_outside_dis<FAST>(r, meta, w, xp, yp);
            live = !(dis >= p.thr);
            neg_num = dis;
            D = coverage_fast<tune::fwd_exact>(neg_num, p);
        }
    }
    unsigned aux = slot;
    if (live) {
        aux |= CELL_LIVE;
        if (p.alpha == 0 && !deferred) {           // 'hard' alpha: the decision of alpha_accumulate, taken here
            const float x = (DIST == 0) ? -1.f
                          : ((neg_num == 0.f || in_fast_range(neg_num)) ? div_known<FAST>(neg_num, p.sigma, p.r_sigma)
                                                                         : neg_num / p.sigma);
            if (x < -8.940696716308594e-08f) aux |= CELL_AHARD;
        }
        const Bary wc = barycentric_clip<FAST>(w);
        zp = depth_of<FAST>(r, wc);
        if (!(zp < p.near_ || zp > p.far_)) {                                  // SRK:365
            aux |= CELL_DEPTH;
            if (RGB == 0 && pixel_inside(w)) aux |= CELL_INCLOSED;
            if (RGB != 2 && p.T != 1) aux |= (unsigned)surface_texel(wc, p.R) << CELL_TEXEL_SHIFT;
            if (face_front(meta) || p.double_side) aux |= CELL_FRONT;
            // zn must carry the reference's exact bits (see softmax_accumulate): the softmax divides differences of it by gamma
            if (RGB == 1) zn = div_known<FAST>(p.far_ - zp, p.far_minus_near, p.r_far_minus_near);
        }
    } else deferred = false;
    return make_float4(zp, zn, D, __builtin_bit_cast(float, aux));
}

// coverage of a deferred (inside) pair -> (D, aux) of its cell
template <bool FAST, bool EXACT_INSIDE>
__device__ inline float2 evaluate_inside(const RasterParams& p, const FaceRec& r, float xp, float yp, unsigned aux) {
    const Bary w = barycentric(r, xp, yp);
    const float neg_num = -euclidean_inside_dis<FAST, EXACT_INSIDE>(r, w);
    const float D = coverage_fast<tune::fwd_exact>(neg_num, p);
    if (p.alpha == 0) {
        const float x = (neg_num == 0.f || in_fast_range(neg_num)) ? div_known<FAST>(neg_num, p.sigma, p.r_sigma) : neg_num / p.sigma;
        if (x < -8.940696716308594e-08f) aux |= CELL_AHARD;
    }
    return make_float2(D, __builtin_bit_cast(float, aux));
}

// the K-buffer half of the state machine of one cell (lane = pixel, wavefront 0)
template <class KB>
__device__ inline void apply_kbuf(const RasterParams& p, const float4 cell, const int* s_meta, KB& q) {
    const unsigned aux = __builtin_bit_cast(unsigned, cell.w);
    if ((aux & (CELL_LIVE | CELL_DEPTH)) != (CELL_LIVE | CELL_DEPTH)) return;
    q.insert(face_id(s_meta[aux & CELL_SLOT]), cell.x, p.K);
}

// the colour half (lane = pixel, wavefront 1): alpha (SRK:350-358), hard rgb (SRK:390-397) or online softmax (SRK:399-419)
template <int RGB, int KCAP>
__device__ inline void apply_colour(const RasterParams& p, const float4 cell, const float* s_col, const int* s_meta,
                                    const float* __restrict__ tbase, PixelState<KCAP>& s) {
    const unsigned aux = __builtin_bit_cast(unsigned, cell.w);
    if (!(aux & CELL_LIVE)) return;
    const float zp = cell.x, D = cell.z;
    if (p.alpha == 0) { if (aux & CELL_AHARD) s.alpha = 1.f; }
    else if (p.alpha == 1) s.alpha += D;
    else s.alpha = __builtin_fmaf(-s.alpha, D, s.alpha);
    if (RGB == 2 || (aux & (CELL_DEPTH | CELL_FRONT)) != (CELL_DEPTH | CELL_FRONT)) return;    // (FRONT is only set together with DEPTH)
    if (RGB == 0) { if (!(zp < s.depth_min && (aux & CELL_INCLOSED))) return; }
    float k0, k1, k2;
    int fn = 0;
    if (p.T == 1) { const float* col = s_col + (aux & CELL_SLOT) * 3; k0 = col[0]; k1 = col[1]; k2 = col[2]; }
    else {
        fn = face_id(s_meta[aux & CELL_SLOT]);
        const float* tx_ = tbase + ((size_t)fn * p.T + (aux >> CELL_TEXEL_SHIFT)) * 3;
        k0 = tx_[0]; k1 = tx_[1]; k2 = tx_[2];
    }
    if (RGB == 0) { s.depth_min = zp; s.face_min = p.T == 1 ? face_id(s_meta[aux & CELL_SLOT]) : fn; s.c0 = k0; s.c1 = k1; s.c2 = k2; }
    else {
        const float zn = cell.y;                 // evaluate_pair: the reference's exact bits
        const float x = zn - s.smax;
        const bool up = x > 0.f;
        const float e = exp_over_gamma<tune::fwd_exact>(-fabsf(x), p);
        const float ed = up ? e : 1.f, ez = up ? 1.f : e;
        s.smax = fmaxf(s.smax, zn);
        const float t = ez * D;
        s.ssum = __builtin_fmaf(ed, s.ssum, t);
        s.c0 = __builtin_fmaf(ed, s.c0, t * k0);
        s.c1 = __builtin_fmaf(ed, s.c1, t * k1);
        s.c2 = __builtin_fmaf(ed, s.c2, t * k2);
    }
}

template <int DIST, int RGB, int KCAP>
__device__ inline void tile_heavy(const RasterParams& p, const TileGeom& t, int wid, int lane, float4* s_mem,
                                  const float* __restrict__ textures, const FaceGeo* __restrict__ geo,
                                  const unsigned long long* __restrict__ seg, unsigned long long* __restrict__ counters,
                                  float* __restrict__ aggrs, float* __restrict__ rgba, int32_t* __restrict__ ids) {
    constexpr int BATCH = HEAVY_BATCH, CAP = HEAVY_CAP;
    // tune::fwd_heavy_overlap: the apply pass keeps two wavefronts busy for half of the tile's time while the other two
    // wait.  Wavefront 3 therefore owns the list walk and stages the NEXT batch's records during the apply of a batch's
    // LAST round (the records are not read after the inside pass: the colours apply needs were copied to s_col), and
    // wavefront 2 owns the pixels' face masks and writes the NEXT round's pair list during the apply of the current one.
    constexpr bool OVERLAP = tune::fwd_heavy_overlap;
    constexpr int STAGER = OVERLAP ? 3 : 0, LISTER = OVERLAP ? 2 : 0;
    FaceRec* s_rec = reinterpret_cast<FaceRec*>(s_mem);                                        // [BATCH]
    float4* s_cell = reinterpret_cast<float4*>(s_rec + BATCH);                                 // [CAP]
    float2* s_pix = reinterpret_cast<float2*>(s_cell + CAP);                                   // [64] pixel centres
    unsigned long long* s_M = reinterpret_cast<unsigned long long*>(s_pix + 64);               // [64] face masks of the batch
    int* s_misc = reinterpret_cast<int*>(s_M + 64);                                            // [16] fill, total, j1, inside count
    int2* s_span = reinterpret_cast<int2*>(s_misc + 16);                                       // [2][64] a pixel's first cell, number of cells (by round parity)
    float* s_col = reinterpret_cast<float*>(s_span + 128);                                     // [2][BATCH][3] single-texel colours of the batch (by batch parity)
    int* s_meta = reinterpret_cast<int*>(s_col + 2 * BATCH * 3);                               // [2][BATCH] the faces' meta words (id, facing), by batch parity
    unsigned short* s_pair = reinterpret_cast<unsigned short*>(s_meta + 2 * BATCH);             // [CAP] slot | pixel << 6
    unsigned short* s_in = s_pair + CAP;                                                       // [CAP] cells of inside pairs
    const float xp = t.xp, yp = t.yp;
    const float* tbase = textures + (size_t)t.b * p.NF * p.T * 3;
    PixelState<KCAP> s;                                  // the K-buffer lives in wavefront 0, the colour state in wavefront 1
    ListWalker lw;
    SectionClock clk;            // instrumented builds only (wavefront 0): 0 stage (wait), 1 masks, 2 pair list (wait), 3 evaluate, 5 inside, 6 apply, 7 stores
    clk.start();
    if (wid == 1) init_colour_state<RGB>(p, s);
    if (wid == 0) {
        init_kbuffer(p, t, ids, s.q);
        s_pix[lane] = make_float2(xp, yp);
    }
    // stage batch `nb` (wavefront STAGER): records, their colours, the fill count
    auto stage_batch = [&](int nb) {
        const int f = tune::fwd_heavy_defer_copy ? lw.stage_deferred<BATCH>(p, s_rec, reinterpret_cast<int*>(s_M), lane)   // (s_M is free: its owner holds the masks in registers)
                                                 : lw.stage<BATCH>(p, s_rec, nullptr, lane);
        wave_sync<false>();
        if (lane < f) {
            float* c = s_col + ((nb & 1) * BATCH + lane) * 3;
            c[0] = s_rec[lane].col[0]; c[1] = s_rec[lane].col[1]; c[2] = s_rec[lane].col[2];
            s_meta[(nb & 1) * BATCH + lane] = s_rec[lane].meta;
        }
        if (lane == 0) s_misc[0] = f;
    };
    // pair list of the round that starts at slot j0 (wavefront LISTER, lane = pixel): the widest slot range [j0, j1)
    // whose pairs fit the cell buffer (a face has <= 64 pairs); pixel-major, ascending face inside a pixel
    unsigned long long M = 0ull;
    auto build_list = [&](int j0, int fill, int round) {
        int j1 = fill, total, cnt, base;
        unsigned long long Mr;
        for (;;) {
            const unsigned long long range = (j1 >= 64 ? ~0ull : ((1ull << j1) - 1ull)) & ~((1ull << j0) - 1ull);
            Mr = M & range;
            cnt = __builtin_popcountll(Mr);
            int incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            total = __builtin_amdgcn_readlane(incl, 63);
            base = incl - cnt;
            if (total <= CAP || j1 - j0 <= 1) break;
            j1 = j0 + ((j1 - j0 + 1) >> 1);
        }
        if (lane == 0) { s_misc[1] = total; s_misc[2] = j1; s_misc[3] = 0; }
        s_span[(round & 1) * 64 + lane] = make_int2(base, cnt);   // (the apply pass may still be reading the previous round's)
        int a = base;
        while (Mr) {
            const int j = __builtin_ctzll(Mr);
            Mr &= Mr - 1;
            s_pair[a++] = (unsigned short)(j | (lane << 6));
        }
    };
    if (wid == STAGER) {
        lw.start(seg, geo + (size_t)t.b * p.NF, tbase, t.n, t.sub, lane);
        if (OVERLAP) stage_batch(0);
    }
    for (int nb = 0;; nb++) {
        if (!OVERLAP && wid == STAGER) stage_batch(nb);
        __syncthreads();                                                       // A: records staged (and the previous batch applied)
        const int fill = s_misc[0];
        if (wid == 0) clk.lap(0);
        if (fill == 0) break;
        {   // masks: two pixel rows per wavefront
            unsigned long long Mw;
            switch (wid) {
                case 0: Mw = pixel_masks<DIST, 0, 2>(p, s_rec, fill, lane, xp, yp); break;
                case 1: Mw = pixel_masks<DIST, 2, 4>(p, s_rec, fill, lane, xp, yp); break;
                case 2: Mw = pixel_masks<DIST, 4, 6>(p, s_rec, fill, lane, xp, yp); break;
                default: Mw = pixel_masks<DIST, 6, 8>(p, s_rec, fill, lane, xp, yp); break;
            }
            if ((lane >> 4) == wid) s_M[lane] = t.valid ? Mw : 0ull;
        }
        __syncthreads();                                                       // B: masks of all 64 pixels
        if (wid == 0) clk.lap(1);
        if (wid == LISTER) { M = s_M[lane]; build_list(0, fill, 0); }
        if (wid == 0) clk.lap(2);
        for (int round = 0;; round++) {                                        // rounds of the batch
            __syncthreads();                                                   // C: pair list (and the previous round applied)
            const int total = s_misc[1], j1 = s_misc[2];
            const bool last = j1 >= fill;
            if (wid == 0) clk.lap(2);
            for (int q0 = wid * 64; q0 < total; q0 += 256) {                   // ---- evaluate: lane = pair ----
                const int q = q0 + lane;
                const bool act = q < total;
                bool deferred = false;
                if (act) {
                    const unsigned pr = s_pair[q];
                    const FaceRec& r = s_rec[pr & 63u];
                    const float2 c = s_pix[pr >> 6];
                    float4 cell;
                    if (face_safe(r.meta) && p.consts_safe) cell = evaluate_pair<DIST, RGB, true>(p, r, c.x, c.y, pr & 63u, deferred);
                    else cell = evaluate_pair<DIST, RGB, false>(p, r, c.x, c.y, pr & 63u, deferred);
                    s_cell[q] = cell;
                }
                if (DIST >= 2) {
                    const unsigned long long im = ballot(deferred);
                    if (im) {
                        int at = 0;
                        if (lane == __builtin_ctzll(im)) at = atomicAdd(&s_misc[3], __builtin_popcountll(im));
                        at = __builtin_amdgcn_readlane(at, __builtin_ctzll(im));
                        const int rk = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(im >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)im, 0u));
                        if (deferred) s_in[at + rk] = (unsigned short)q;
                    }
                }
            }
            __syncthreads();                                                   // D: cells, inside list
            if (wid == 0) clk.lap(3);
            if (DIST >= 2) {
                const int nin = s_misc[3];
                for (int i0 = wid * 64; i0 < nin; i0 += 256) {                 // ---- inside pairs: lane = pair ----
                    const int i = i0 + lane;
                    if (i < nin) {
                        const int q = s_in[i];
                        const unsigned pr = s_pair[q];
                        const FaceRec& r = s_rec[pr & 63u];
                        const float2 c = s_pix[pr >> 6];
                        const unsigned aux = __builtin_bit_cast(unsigned, s_cell[q].w);
                        float2 da;
                        if (face_safe(r.meta) && p.consts_safe) da = evaluate_inside<true, DIST == 3>(p, r, c.x, c.y, aux);
                        else da = evaluate_inside<false, DIST == 3>(p, r, c.x, c.y, aux);
                        *reinterpret_cast<float2*>(&s_cell[q].z) = da;
                    }
                }
                __syncthreads();                                               // E: coverage of the inside pairs; records, pair list and scalars are free
                if (wid == 0) clk.lap(5);
            }
            if (wid <= 1) {                                                    // ---- apply: lane = pixel, K-buffer | colour ----
                const int2 span = s_span[(round & 1) * 64 + lane];
                const float* colb = s_col + (nb & 1) * BATCH * 3;
                const int* metab = s_meta + (nb & 1) * BATCH;
                // (cells are read one ahead of their use; beyond the pixel's last cell: cell 0 with aux = 0, "not live")
                float4 cur = s_cell[span.y > 0 ? span.x : 0];
                if (!(span.y > 0)) cur.w = 0.f;
                for (int k = 0; ballot(k < span.y) != 0ull; k++) {
                    float4 nxt = s_cell[k + 1 < span.y ? span.x + k + 1 : 0];
                    if (!(k + 1 < span.y)) nxt.w = 0.f;
                    if (wid == 0) apply_kbuf(p, cur, metab, s.q);
                    else apply_colour<RGB, KCAP>(p, cur, colb, metab, tbase, s);
                    cur = nxt;
                }
                if (wid == 0) clk.lap(6);
            }
            if (OVERLAP) {
                if (wid == LISTER && !last) build_list(j1, fill, round + 1);   // ... meanwhile: the next round's pair list,
                if (wid == STAGER && last) stage_batch(nb + 1);                // the next batch's records
            }
            if (last) break;
            if (!OVERLAP && wid == LISTER) build_list(j1, fill, round + 1);
        }
    }
    if (wid == 1) store_colour<RGB>(p, t, s, aggrs, rgba);
    if (wid == 0) {
        store_ids<KCAP, ids_in_global<KCAP>()>(p, t, s.q, ids);
        clk.lap(7);
        if (JR_TUNE_PROFILE_SECTIONS == 2 && t.n == (int)counters[2]) clk.flush(counters, 4);   // the 16 tiles of the heaviest bin
    }
}

// =====================================================================================================================
// The same heavy tile as a PIPELINE (tune::fwd_heavy_pipe, round 3).  In tile_heavy the apply pass is 55 % of the tile's
// clocks and two of the four wavefronts wait through it; the evaluate pass of the NEXT round needs nothing from it.  So
// the rounds are cut smaller (PIPE_CAP cells, PIPE_BATCH records), cells / pair lists / records are double-buffered, and a
// STEP is: wavefronts 0 / 1 apply round n-1 while everything else of round n (and of the batch after) happens around them:
//
//   pinned     wavefront 3 owns the list walk: it stages the batch AFTER the one being evaluated (records of two batches
//              are resident), and writes the pair list of the round to be evaluated in the next step
//   claimable  64-pair evaluate chunks of round n, and the four 2-row mask tasks of a freshly staged batch: every wavefront
//              (0 / 1 once their apply is done) takes the next one from an LDS counter
//   inside     a wavefront collects the inside pairs of ITS chunks and runs them in dense trips of its own
//
// One barrier per step; all decisions that shape the control flow are taken by wavefront 3 and published in the step's
// state block (two blocks, by step parity), so every wavefront takes the same branches.  Results: the same device
// functions on the same operands in the same per-pixel order as tile_heavy / tile_single.
// =====================================================================================================================
// NW wavefronts per workgroup (tune::fwd_heavy_waves): 4, or 8 - two apply, one stages / lists, five only take tasks; the
// workgroup's LDS is what NW single-wavefront tiles use, so eight wavefronts also get longer rounds and batches.
constexpr int PIPE_IN = 128;
constexpr int pipe_batch(int nw) { return nw >= 8 ? tune::fwd_pipe8_batch : 40; }
constexpr int pipe_cap(int nw) { return nw >= 8 ? tune::fwd_pipe8_cap : 512; }
constexpr int pipe_lds_bytes(int nw) {
    return 2 * pipe_batch(nw) * (int)sizeof(FaceRec) + 2 * pipe_cap(nw) * 16 + 64 * 8 + 2 * 64 * 8 + 3 * 64 * 8
           + 4 * pipe_batch(nw) * 16 + 64 * 4 + pipe_batch(nw) * 4 + 2 * pipe_cap(nw) * 2 + nw * PIPE_IN * 2;
}
// dynamic LDS of a workgroup: what nw single-wavefront tiles use, or the pipelined heavy tile if that is (a few hundred bytes) more
constexpr int mixed_lds_bytes(int nw) {
    return nw * (int)sizeof(FaceRec) * HEAVY_BATCH > pipe_lds_bytes(nw) ? nw * (int)sizeof(FaceRec) * HEAVY_BATCH : pipe_lds_bytes(nw);
}
static_assert(4 * mixed_lds_bytes(4) <= 160 * 1024 && 2 * mixed_lds_bytes(8) <= 160 * 1024, "four / two workgroups per CU");
enum { PS_VALID = 0, PS_BATCH, PS_J0, PS_J1, PS_TOTAL, PS_LAST, PS_MASKS, PS_MBATCH, PS_MFILL, PS_CLAIM, PS_DONE, PS_WORDS = 16 };

template <int DIST, int RGB, int KCAP, int NW>
__device__ inline void tile_heavy_pipe(const RasterParams& p, const TileGeom& t, int wid, int lane, float4* s_mem,
                                       const float* __restrict__ textures, const FaceGeo* __restrict__ geo,
                                       const unsigned long long* __restrict__ seg, unsigned long long* __restrict__ counters,
                                       float* __restrict__ aggrs, float* __restrict__ rgba, int32_t* __restrict__ ids) {
    constexpr int BATCH = pipe_batch(NW), CAP = pipe_cap(NW);
    FaceRec* s_rec = reinterpret_cast<FaceRec*>(s_mem);                                        // [2][BATCH] by batch parity
    float4* s_cell = reinterpret_cast<float4*>(s_rec + 2 * BATCH);                             // [2][CAP]   by step parity
    float2* s_pix = reinterpret_cast<float2*>(s_cell + 2 * CAP);                               // [64] pixel centres
    unsigned long long* s_M = reinterpret_cast<unsigned long long*>(s_pix + 64);               // [2][64] face masks, by batch parity
    int2* s_span = reinterpret_cast<int2*>(s_M + 2 * 64);                                      // [3][64] first cell, cells of a pixel, by step % 3
    float* s_col = reinterpret_cast<float*>(s_span + 3 * 64);                                  // [4][BATCH][3] colours, by batch & 3
    int* s_meta = reinterpret_cast<int*>(s_col + 4 * BATCH * 3);                               // [4][BATCH] meta words (id, facing), by batch & 3
    int* s_state = s_meta + 4 * BATCH;                                                         // [2][PS_WORDS] by step parity
    int* s_slot = s_state + 64;                                                                // [BATCH] staging scratch
    unsigned short* s_pair = reinterpret_cast<unsigned short*>(s_slot + BATCH);                 // [2][CAP] slot | pixel << 6, by step parity
    unsigned short* s_in = s_pair + 2 * CAP + wid * PIPE_IN;                                   // [PIPE_IN] this wavefront's inside pairs
    const float xp = t.xp, yp = t.yp;
    const float* tbase = textures + (size_t)t.b * p.NF * p.T * 3;
    PixelState<KCAP> s;                                  // the K-buffer lives in wavefront 0, the colour state in wavefront 1
    ListWalkerT<tune::fwd_pipe_list_depth> lw;   // wavefront 3 walks alone: several list chunks in flight
    SectionClock clk;            // instrumented builds only (wavefront tune::sections_wave): 0 barrier wait, 3 claimed tasks (wavefront 3: + staging / lists), 6 apply, 7 stores
    clk.start();
    if (wid == 1) init_colour_state<RGB>(p, s);
    if (wid == 0) {
        init_kbuffer(p, t, ids, s.q);
        s_pix[lane] = make_float2(xp, yp);
    }
    // ---- wavefront 3: staging and pair lists ----
    auto stage_batch = [&](int nb) -> int {
        FaceRec* rec = s_rec + (nb & 1) * BATCH;
        const int f = lw.stage_deferred<BATCH>(p, rec, s_slot, lane);
        wave_sync<false>();
        if (lane < f) {
            float* c = s_col + ((nb & 3) * BATCH + lane) * 3;
            c[0] = rec[lane].col[0]; c[1] = rec[lane].col[1]; c[2] = rec[lane].col[2];
            s_meta[(nb & 3) * BATCH + lane] = rec[lane].meta;
        }
        return f;
    };
    // pair list of the round of batch `nb` that starts at slot j0, for the step `ns`: the widest slot range [j0, j1) whose
    // pairs fit the cell buffer (a face has <= 64 pairs); pixel-major, ascending face inside a pixel
    auto build_list = [&](int nb, int j0, int fill, int ns, int* st) {
        const unsigned long long M = s_M[(nb & 1) * 64 + lane];
        int j1 = fill, total, cnt, base;
        unsigned long long Mr;
        for (;;) {
            const unsigned long long range = (j1 >= 64 ? ~0ull : ((1ull << j1) - 1ull)) & ~((1ull << j0) - 1ull);
            Mr = M & range;
            cnt = __builtin_popcountll(Mr);
            int incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            total = __builtin_amdgcn_readlane(incl, 63);
            base = incl - cnt;
            if (total <= CAP || j1 - j0 <= 1) break;
            j1 = j0 + ((j1 - j0 + 1) >> 1);
        }
        if (lane == 0) { st[PS_VALID] = 1; st[PS_BATCH] = nb; st[PS_J0] = j0; st[PS_J1] = j1; st[PS_TOTAL] = total; st[PS_LAST] = j1 >= fill; }
        s_span[(ns % 3) * 64 + lane] = make_int2(base, cnt);
        unsigned short* pl = s_pair + (ns & 1) * CAP;
        int a = base;
        while (Mr) {
            const int j = __builtin_ctzll(Mr);
            Mr &= Mr - 1;
            pl[a++] = (unsigned short)(j | (lane << 6));
        }
    };
    // the four 2-row mask tasks of batch nb
    auto mask_task = [&](int k, int nb, int fill) {
        const FaceRec* rec = s_rec + (nb & 1) * BATCH;
        unsigned long long Mw;
        switch (k) {
            case 0: Mw = pixel_masks<DIST, 0, 2>(p, rec, fill, lane, xp, yp); break;
            case 1: Mw = pixel_masks<DIST, 2, 4>(p, rec, fill, lane, xp, yp); break;
            case 2: Mw = pixel_masks<DIST, 4, 6>(p, rec, fill, lane, xp, yp); break;
            default: Mw = pixel_masks<DIST, 6, 8>(p, rec, fill, lane, xp, yp); break;
        }
        if ((lane >> 4) == k) s_M[(nb & 1) * 64 + lane] = t.valid ? Mw : 0ull;
    };

    // ---- prologue: batch 0 staged, masked, its first round listed (serial, as in tile_heavy) ----
    int w3_fill = 0;                                     // wavefront 3: records of the batch it lists from
    if (wid == 3) {
        lw.start(seg, geo + (size_t)t.b * p.NF, tbase, t.n, t.sub, lane);
        w3_fill = stage_batch(0);
        if (lane == 0) s_state[PS_MFILL] = w3_fill;
    }
    __syncthreads();
    const int fill0 = s_state[PS_MFILL];
    if (fill0 > 0) {
        if (wid < 4) mask_task(wid, 0, fill0);
        __syncthreads();
        if (wid == 3) {
            build_list(0, 0, fill0, 0, s_state);
            if (lane == 0) { s_state[PS_MASKS] = 0; s_state[PS_CLAIM] = 0; s_state[PS_DONE] = 0; }
        }
        __syncthreads();
        // wavefront 3's private view of the pipeline
        int cur_batch = 0;               // batch of the latest listed round
        bool staged = false, masked = false, walker_done = false;
        bool offer_next = false, offer_now = false;      // the staged batch's mask tasks run in the next / in this step
        int staged_fill = 0;
        // every wavefront: the round applied in this step = the round evaluated in the previous one
        bool a_valid = false;
        int a_batch = 0;
        for (int step = 0;; step++) {
            const int* st = s_state + (step & 1) * PS_WORDS;
            int* nx = s_state + ((step + 1) & 1) * PS_WORDS;
            const bool e_valid = st[PS_VALID] != 0;
            const int e_batch = st[PS_BATCH], e_total = st[PS_TOTAL];
            const bool offer = st[PS_MASKS] != 0;
            if (st[PS_DONE] && !a_valid) break;
            if (wid == tune::sections_wave) clk.lap(0);
            // ---- apply round step-1: lane = pixel, K-buffer | colour ----
            if (wid <= 1 && a_valid) {
                const int2 span = s_span[((step + 2) % 3) * 64 + lane];                        // (step - 1) % 3
                const float4* cells = s_cell + ((step + 1) & 1) * CAP;
                const float* colb = s_col + (a_batch & 3) * BATCH * 3;
                const int* metab = s_meta + (a_batch & 3) * BATCH;
                float4 cur = cells[span.y > 0 ? span.x : 0];
                if (!(span.y > 0)) cur.w = 0.f;
                for (int k = 0; ballot(k < span.y) != 0ull; k++) {
                    float4 nxt = cells[k + 1 < span.y ? span.x + k + 1 : 0];
                    if (!(k + 1 < span.y)) nxt.w = 0.f;
                    if (wid == 0) apply_kbuf(p, cur, metab, s.q);
                    else apply_colour<RGB, KCAP>(p, cur, colb, metab, tbase, s);
                    cur = nxt;
                }
                if (wid == tune::sections_wave) clk.lap(6);
            }
            // ---- wavefront 3: the batch after, the next round's list, the next step's state ----
            if (wid == 3) {
                if (offer_now) masked = true;                // the mask tasks that ran in the previous step are complete
                offer_now = offer_next; offer_next = false;
                int nmasks = 0;
                if (!staged && !walker_done && !st[PS_DONE]) {
                    const int f = stage_batch(cur_batch + 1);
                    if (f == 0) walker_done = true;
                    else { staged = true; staged_fill = f; nmasks = 1; offer_next = true; }
                }
                bool listed = false;
                if (e_valid && !st[PS_LAST]) { build_list(e_batch, st[PS_J1], w3_fill, step + 1, nx); listed = true; }
                else if (masked) {
                    cur_batch++; w3_fill = staged_fill;
                    staged = false; masked = false;
                    build_list(cur_batch, 0, w3_fill, step + 1, nx);
                    listed = true;
                }
                if (lane == 0) {
                    if (!listed) nx[PS_VALID] = 0;
                    nx[PS_MASKS] = nmasks; nx[PS_MBATCH] = cur_batch + 1; nx[PS_MFILL] = staged_fill;
                    nx[PS_CLAIM] = 0;
                    nx[PS_DONE] = (!listed && walker_done && !staged) ? 1 : 0;
                }
            }
            // ---- claimable tasks: evaluate chunks of this step's round, mask tasks of a freshly staged batch ----
            // (tune::fwd_pipe_consumer_tasks: bit 0 / 1 = wavefront 0 / 1 joins after its apply)
            if (wid >= 2 || ((tune::fwd_pipe_consumer_tasks >> wid) & 1)) {
                const int nchunks = e_valid ? (e_total + 63) >> 6 : 0;
                const int ntasks = nchunks + (offer ? 4 : 0);
                const FaceRec* recE = s_rec + (e_batch & 1) * BATCH;
                const unsigned short* pl = s_pair + (step & 1) * CAP;
                float4* cells = s_cell + (step & 1) * CAP;
                int n_in = 0;
                auto run_inside = [&](int base, int count) {
                    wave_sync<false>();
                    if (lane < count) {
                        const int q = s_in[base + lane];
                        const unsigned pr = pl[q];
                        const FaceRec& r = recE[pr & 63u];
                        const float2 c = s_pix[pr >> 6];
                        const unsigned aux = __builtin_bit_cast(unsigned, cells[q].w);
                        float2 da;
                        if (face_safe(r.meta) && p.consts_safe) da = evaluate_inside<true, DIST == 3>(p, r, c.x, c.y, aux);
                        else da = evaluate_inside<false, DIST == 3>(p, r, c.x, c.y, aux);
                        *reinterpret_cast<float2*>(&cells[q].z) = da;
                    }
                };
                for (;;) {
                    int k = 0;
                    if (lane == 0) k = atomicAdd(const_cast<int*>(&st[PS_CLAIM]), 1);
                    k = __builtin_amdgcn_readfirstlane(k);
                    if (k >= ntasks) break;
                    if (k >= nchunks) { mask_task(k - nchunks, st[PS_MBATCH], st[PS_MFILL]); continue; }
                    const int q = k * 64 + lane;
                    bool deferred = false;
                    if (q < e_total) {
                        const unsigned pr = pl[q];
                        const FaceRec& r = recE[pr & 63u];
                        const float2 c = s_pix[pr >> 6];
                        float4 cell;
                        if (face_safe(r.meta) && p.consts_safe) cell = evaluate_pair<DIST, RGB, true>(p, r, c.x, c.y, pr & 63u, deferred);
                        else cell = evaluate_pair<DIST, RGB, false>(p, r, c.x, c.y, pr & 63u, deferred);
                        cells[q] = cell;
                    }
                    if (DIST >= 2) {
                        const unsigned long long im = ballot(deferred);
                        if (im) {
                            const int rk = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(im >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)im, 0u));
                            if (deferred) s_in[n_in + rk] = (unsigned short)q;
                            n_in += __builtin_popcountll(im);
                            if (n_in >= 64) { n_in -= 64; run_inside(n_in, 64); }      // a dense trip of the newest 64
                        }
                    }
                }
                if (DIST >= 2 && n_in > 0) run_inside(0, n_in);
                if (wid == tune::sections_wave) clk.lap(3);
            }
            a_valid = e_valid; a_batch = e_batch;
            __syncthreads();
        }
    }
    if (wid == 1) store_colour<RGB>(p, t, s, aggrs, rgba);
    if (wid == 0) store_ids<KCAP, ids_in_global<KCAP>()>(p, t, s.q, ids);
    if (wid == tune::sections_wave) {
        clk.lap(7);
        if (JR_TUNE_PROFILE_SECTIONS == 2 && t.n == (int)counters[2] && lane == 0) clk.flush0(counters, 4);   // the 16 tiles of the heaviest bin
    }
}

// ---- kernels -----------------------------------------------------------------------------------------------------------
// One wavefront per workgroup, one tile per wavefront (rounds 1-2; tune::fwd_heavy = 0, and vertex colours).
// BL: log2 of the bin size as a COMPILE-TIME constant (5: the headline batch and every launch above 4 Mpixels under the
// automatic policy), or 0 = read it from the launch parameters.  Round 5 measured what the run-time value costs this kernel:
// forward 0.841 -> 0.878 ms on the headline batch (same-box A/B against a build with the constant, profiles/r05_experiments.md
// call 5) - two more scalars live across a kernel that already spills SGPRs into VGPR lanes.
template <int DIST, int RGB, int KCAP, int BL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(fwd_waves(KCAP, false, RGB, DIST)))) void k_softras_forward(
    RasterParams p, int ntiles_total, const float* __restrict__ textures,
    const FaceGeo* __restrict__ geo, const int* __restrict__ bin_order, const int* __restrict__ bin_count,
    const int* __restrict__ bin_base, const unsigned long long* __restrict__ pool,
    unsigned long long* __restrict__ counters, unsigned long long pool_cap,
    float* __restrict__ aggrs, float* __restrict__ rgba, int32_t* __restrict__ ids) {
    extern __shared__ float4 s_dyn[];
    if (counters[0] > pool_cap) return;     // lists were not built (pool too small): the host launches again
    // XCD-aware order: consecutive workgroup ids land on different XCDs (id % 8); the tiles of a
    // bin (same list, same records) go to ONE XCD so that they share its L2.
    const int k = blockIdx.x >> 3;                       // k-th workgroup of XCD (blockIdx.x & 7)
    const int bl = BL ? BL : bin_log2_of(p);
    const int tl = 2 * (bl - TILE_LOG2), tmask = (1 << tl) - 1; // a bin has 1 << tl tiles
    const int brank = (k >> tl) * 8 + (blockIdx.x & 7);  // bins are dealt round-robin to the XCDs ...
    if ((brank << tl) >= ntiles_total) return;
    const int bin = bin_order[brank];                    // ... heaviest first (k_bin_alloc_schedule)
    const int n = bin_count[bin];
    if (tune::fwd_empty_bins && n == 0 && (p.IS & ((1 << bl) - 1)) == 0) {    // empty bin: tile 0's wavefront writes all its tiles
        if ((k & tmask) == 0 && !(JR_TUNE_DIAG & 2048)) store_empty_bin<RGB, KCAP>(p, bin, threadIdx.x, bl, aggrs, rgba, ids);   // (diagnostic bit 11, WRONG images: what do the empty bins' stores cost the launch?)
        return;
    }
    TileGeom t;
    if (!tile_geom(p, bin, k & tmask, n, threadIdx.x, t, bl)) return;
    tile_single<DIST, RGB, KCAP, tune::fwd_batch_for(KCAP), !tune::light_sync>(p, t, threadIdx.x, s_dyn, textures, geo, pool + bin_base[bin], counters, aggrs, rgba, ids);
}

// NW = four or eight wavefronts per workgroup (round 3).  The launch order of the bins is heaviest first
// (k_bin_alloc_schedule) and its first counters[3] bins are HEAVY (list longer than tune::fwd_heavy): a workgroup takes
// ONE tile of a heavy bin with all its wavefronts together (tile_heavy_pipe; tile_heavy with four), or NW tiles of a
// lighter bin, one per wavefront (tile_single, each with its share of the workgroup's LDS).  Per XCD (workgroup id % 8)
// the heavy tiles come first.
template <int DIST, int RGB, int KCAP, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(fwd_waves(KCAP, true)))) void k_softras_forward_mixed(
    RasterParams p, int nbins, int heavy_cap, const float* __restrict__ textures,
    const FaceGeo* __restrict__ geo, const int* __restrict__ bin_order, const int* __restrict__ bin_count,
    const int* __restrict__ bin_base, const unsigned long long* __restrict__ pool,
    unsigned long long* __restrict__ counters, unsigned long long pool_cap,
    float* __restrict__ aggrs, float* __restrict__ rgba, int32_t* __restrict__ ids) {
    extern __shared__ float4 s_dyn[];
    if (counters[0] > pool_cap) return;     // lists were not built (pool too small): the host launches again
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nheavy = min((int)counters[3], heavy_cap);
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int hx = (nheavy - xcd + 7) >> 3;              // heavy bins dealt to this XCD (launch ranks xcd, xcd + 8, ...)
    const int bl = bin_log2_of(p);
    const int tl = 2 * (bl - TILE_LOG2), tmask = (1 << tl) - 1; // a bin has 1 << tl tiles
    const bool heavy = k < (hx << tl);
    int brank, sub;
    if (heavy) { brank = (k >> tl) * 8 + xcd; sub = k & tmask; }
    else { const int s = (k - (hx << tl)) * NW + wid; brank = (hx + (s >> tl)) * 8 + xcd; sub = s & tmask; }   // NW consecutive tiles of this XCD's lighter bins per workgroup (a quarter of a 4x4 bin, a 2x2 bin, four one-tile bins ...)
    if (brank >= nbins) return;
    const int bin = bin_order[brank];
    const int n = bin_count[bin];
    if (tune::fwd_empty_bins && n == 0 && (p.IS & ((1 << bl) - 1)) == 0) {    // empty bin (never heavy): the wavefront that holds its tile 0
        if (sub == 0) store_empty_bin<RGB, KCAP>(p, bin, lane, bl, aggrs, rgba, ids);
        return;
    }
    TileGeom t;
    if (!tile_geom(p, bin, sub, n, lane, t, bl)) return;     // (a heavy tile: uniform for the workgroup)
    const unsigned long long* seg = pool + bin_base[bin];
    if ((JR_TUNE_DIAG & 512) && !heavy) return;          // (diagnostic bits 9 / 10, WRONG images: the makespan of the heavy / of the light tiles alone)
    if ((JR_TUNE_DIAG & 1024) && heavy) return;
    if (heavy) {
        if (tune::fwd_heavy_pipe) tile_heavy_pipe<DIST, RGB, KCAP, NW>(p, t, wid, lane, s_dyn, textures, geo, seg, counters, aggrs, rgba, ids);
        else if (NW == 4) tile_heavy<DIST, RGB, KCAP>(p, t, wid, lane, s_dyn, textures, geo, seg, counters, aggrs, rgba, ids);
    }
    else tile_single<DIST, RGB, KCAP, HEAVY_BATCH, false>(p, t, lane, s_dyn + wid * (sizeof(FaceRec) * HEAVY_BATCH / sizeof(float4)),
                                             textures, geo, seg, counters, aggrs, rgba, ids);
}

#ifndef JR_FWD_PRECISE          // (one definition: the policy does not depend on the arithmetic)
}  // namespace JR_FWD_NAMESPACE
namespace jr {
bool forward_uses_heavy_path(const RasterParams& p, const BinWorkspace& ws) {
    return ws.heavy_min > 0 && p.tex == 0 && (long)p.B * p.IS * p.IS <= (long)tune::fwd_heavy_pixels;
}
}  // namespace jr
namespace JR_FWD_NAMESPACE {
#endif

template <int DIST, int RGB, int KCAP>
static void launch_kk(hipStream_t st, const RasterParams& p, const float* textures,
                      const BinWorkspace& ws, float* aggrs, float* rgba, int32_t* ids) {
    const int nbins = p.B * p.bins_x * p.bins_y;
    const int tl = 2 * sub_log2_of(p);                   // a bin has 1 << tl tiles
    const int ntiles = nbins << tl;
    // Four wavefronts per heavy tile cut the critical path of a launch (one 39k-face view: 0.69 -> 0.38 ms), but the
    // waiting wavefronts hold slots that a full GPU has better uses for (eight views: 0.89 -> 0.98 ms even when only the
    // bins above 1024 faces are heavy): the four-wavefront kernel takes launches of up to fwd_heavy_pixels pixels.
    // Heavy tiles need single-texel or per-texel surface colours (the cell has no room for three vertex colours).
    if (forward_uses_heavy_path(p, ws)) {
        // upper bound of the heavy bins the device will find (their lists hold more than fwd_heavy_floor() entries each; the
        // host tightens it with what it knows about the shape): the grid has a workgroup per tile of that many bins
        const int heavy_cap = heavy_bins_cap(ws, nbins);
        auto launch = [&](auto nw_tag) {
            constexpr int NW = decltype(nw_tag)::value;
            const int per_xcd = (((heavy_cap + 7) / 8) << tl) + ((((nbins + 7) / 8) << tl) + NW - 1) / NW;
            k_softras_forward_mixed<DIST, RGB, KCAP, NW><<<8 * per_xcd, 64 * NW, mixed_lds_bytes(NW), st>>>(
                p, nbins, heavy_cap, textures, ws.geo, ws.bin_order, ws.bin_count, ws.bin_base, ws.pool, ws.counters, ws.pool_cap, aggrs, rgba, ids);
        };
        // (the sequential heavy tile is written for four wavefronts; eight only where the host's policy asks for them)
        bool eight = tune::fwd_heavy_pipe && tune::fwd_heavy_waves == 8 && ws.heavy_waves == 8;
        if (eight && mixed_lds_bytes(8) > 65536) {
            // more than 64 KB of dynamic LDS per workgroup is an opt-in PER DEVICE and per kernel instantiation: asked once
            // per context (= device) and instantiation; a refusal falls back to four wavefronts instead of a failed launch
            const unsigned long long bit = 1ull << ((DIST * 3 + RGB) * 3 + (KCAP <= 16 ? 0 : (KCAP <= 32 ? 1 : 2)));
#ifdef JR_FWD_PRECISE
            constexpr int SET = 1;
#else
            constexpr int SET = 0;
#endif
            if (!(ws.lds_optin_tried[SET] & bit)) {
                ws.lds_optin_tried[SET] |= bit;
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_softras_forward_mixed<DIST, RGB, KCAP, 8>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, mixed_lds_bytes(8)) == hipSuccess)
                    ws.lds_optin_ok[SET] |= bit;
                else (void)hipGetLastError();
            }
            eight = (ws.lds_optin_ok[SET] & bit) != 0;
        }
        ws.heavy_waves_used = eight ? 8 : 4;
        if (eight) launch(std::integral_constant<int, 8>());
        else launch(std::integral_constant<int, 4>());
        return;
    }
    ws.heavy_waves_used = 1;
    const int per = 8 << tl;
    const int grid = ((ntiles + per - 1) / per) * per;   // whole bins per XCD slot
    // JR_FWD_LDS_PAD (bytes, diagnostics only): more dynamic LDS per wavefront = fewer wavefronts per CU
    static const size_t pad = getenv("JR_FWD_LDS_PAD") ? (size_t)atol(getenv("JR_FWD_LDS_PAD")) : 0;
    constexpr int BATCH1 = tune::fwd_batch_for(KCAP);
    const size_t smem = sizeof(FaceRec) * BATCH1 + (p.tex == 1 ? sizeof(float) * 9 * BATCH1 : 0) + pad;
    if (bin_log2_of(p) == 5)
        k_softras_forward<DIST, RGB, KCAP, 5><<<grid, 64, smem, st>>>(
            p, ntiles, textures, ws.geo, ws.bin_order, ws.bin_count, ws.bin_base, ws.pool, ws.counters, ws.pool_cap, aggrs, rgba, ids);
    else
        k_softras_forward<DIST, RGB, KCAP, 0><<<grid, 64, smem, st>>>(
            p, ntiles, textures, ws.geo, ws.bin_order, ws.bin_count, ws.bin_base, ws.pool, ws.counters, ws.pool_cap, aggrs, rgba, ids);
}

template <int DIST, int RGB>
static void launch_k(hipStream_t st, const RasterParams& p, const float* textures,
                     const BinWorkspace& ws, float* aggrs, float* rgba, int32_t* ids) {
    // K-buffer capacity: 16 (the default K), 32 (K = 17..32), 64
    if (p.K <= 16) launch_kk<DIST, RGB, 16>(st, p, textures, ws, aggrs, rgba, ids);
    else if (p.K <= 32) launch_kk<DIST, RGB, 32>(st, p, textures, ws, aggrs, rgba, ids);
    else launch_kk<DIST, RGB, 64>(st, p, textures, ws, aggrs, rgba, ids);
}

void launch_softras_forward(hipStream_t st, const RasterParams& p, const float* textures,
                            const BinWorkspace& ws, float* aggrs, float* rgba, int32_t* ids) {
#define JR_FWD(D, R) launch_k<D, R>(st, p, textures, ws, aggrs, rgba, ids)
    const int rgb = p.rgb == 0 ? 0 : (p.rgb == 1 ? 1 : 2);
    // DIST = 3: euclidean distance under 'hard' alpha - the inside distance keeps the reference's IEEE quotients because
    // D > 0.5 is decided from it (softras_device.h: euclidean_sign_dis); its own instantiations, so the other modes pay nothing
    // ... and (round 5) under a sigma below tune::fwd_exact_inside_sigma: the reciprocal-multiply projection of an inside pixel is <= 2 ulp
    // off in t, i.e. ~1e-7 of the edge length in the nearest point; at sigma = 1e-6 the coverage sigmoid's knee sits at distances of
    // 1e-3, where that is 2e-4 of d^2 / sigma: a 7 788-face soup at 324^2 showed softmax_sum 1.03e-4 off at ONE pixel
    // (tools/ablate/cases/fuzz_fail_89_238.npz; with IEEE quotients 3.5e-7).  The default sigma 1e-5 and anything softer keep the fast path.
    const int dist = (p.dist == 2 && ((p.alpha == 0 && tune::fwd_hard_exact) || p.sigma < tune::fwd_exact_inside_sigma)) ? 3 : p.dist;
    switch (dist * 3 + rgb) {
        case 0: JR_FWD(0, 0); break;
        case 1: JR_FWD(0, 1); break;
        case 2: JR_FWD(0, 2); break;
        case 3: JR_FWD(1, 0); break;
        case 4: JR_FWD(1, 1); break;
        case 5: JR_FWD(1, 2); break;
        case 6: JR_FWD(2, 0); break;
        case 7: JR_FWD(2, 1); break;
        case 8: JR_FWD(2, 2); break;
        case 9: JR_FWD(3, 0); break;
        case 10: JR_FWD(3, 1); break;
        default: JR_FWD(3, 2); break;
    }
#undef JR_FWD
}

}  // namespace JR_FWD_NAMESPACE
