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
// Tiles are visited heaviest bin first (k_bin_schedule), bins dealt round-robin to the XCDs: the list
// length varies from 1 to >1000 faces and the launch would otherwise end with a long tail.
//
// The per-pixel state machine (alpha, online softmax over depth, K-nearest buffer) lives in VGPRs;
// it is sequential in face order, which is why the lists are sorted.  No MFMA: no dense contraction.
//
// Exactness: everything that decides the face-index buffer (border test, barycentrics, distance,
// distance cull, clipped depth, depth cull, K-buffer) reproduces the reference bit for bit.  The
// colour path (coverage sigmoid, softmax weights) only has to stay within 1e-4 and uses
// reciprocal multiplies where the reference divides by sigma / gamma (error <= 1-2 ulp).
#include "jr_kernels.h"

namespace jr {

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
    __device__ inline void insert(int fn, float zp, int K) {
        const bool filling = size < K;
        if (!filling && !(zp < max_z)) return;
        const int slot = filling ? size : max_slot;
        if (IDS_GLOBAL) gplane[(unsigned)slot * gstride + goff] = fn;
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
            return;
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
    // zn must carry the reference's exact bits: the softmax divides differences of it by gamma
    const float zn = div_known<FAST>(p.far_ - zp, p.far_minus_near, p.r_far_minus_near);
    float ed, ez;
    if (tune::fwd_exp1) {
        // one of the reference's two exponentials is always exp(0) = 1 (after "smax = zn" the second argument is 0):
        // ONE v_exp of -|zn - smax| and two selects give the same two values, bit for bit
        const float x = zn - s.smax;
        const bool up = x > 0.f;                       // zn > smax (NaN: false, like the reference's compare)
        const float e = exp_over_gamma(up ? -x : x, p);
        ed = up ? e : 1.f;
        ez = up ? 1.f : e;
        s.smax = up ? zn : s.smax;
    } else {
        ed = 1.f;
        if (zn > s.smax) { ed = exp_over_gamma(s.smax - zn, p); s.smax = zn; }
        ez = exp_over_gamma(zn - s.smax, p);
    }
    s.ssum = ed * s.ssum + ez * D;
    float k0, k1, k2;
    sample_colour<FAST>(p, r, vc, tbase, wc, zp, k0, k1, k2);
    s.c0 = ed * s.c0 + ez * D * k0;
    s.c1 = ed * s.c1 + ez * D * k1;
    s.c2 = ed * s.c2 + ez * D * k2;
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

// One (pixel, face) pair of the raster loop.  Returns true when the pair was DEFERRED (tune::fwd_defer_inside,
// euclidean distance only): the pixel lies strictly inside the face.  Such a pair is never culled by distance and its
// coverage needs three edge projections instead of one; 8 % of the pairs are inside, but with 64 lanes on 64
// different faces nearly every trip of the loop had one, so the whole wavefront paid the 60 extra instructions
// at 8 % lane use.  Everything that is ORDER dependent (depth cull, K-buffer insert, 'hard' rgb) does not
// depend on the coverage and happens here, in face order; what depends on it — the alpha product and the softmax
// sums — commutes (1e-4 colour path) and is added by forward_pair_inside in a second loop over the batch's
// deferred pairs, where all active lanes are inside.
template <int DIST, int RGB, bool FAST, int KCAP>
__device__ inline bool forward_pair(const RasterParams& p, const FaceRec& r, const float* vc,
                                    const float* __restrict__ tbase, float xp, float yp,
                                    PixelState<KCAP>& s) {
    const Bary w = barycentric(r, xp, yp);
    const int meta = r.meta;
    float D = 1.f, neg_num = -1.f;
    bool deferred = false;
    if (DIST == 0) {                                                           // SRK:331-333
        if (!pixel_inside(w)) return false;
    } else if (DIST == 1) {                                                    // SRK:335-338
        const float dis = barycentric_dist(w);
        if (-dis >= p.thr) return false;
        neg_num = -dis;
        D = coverage_fast(neg_num, p);
    } else if (tune::fwd_defer_inside) {                                       // SRK:340-344
        deferred = strictly_inside_t<FAST>(w);
        if (!deferred) {
            const float dis = euclidean_outside_dis<FAST>(r, meta, w, xp, yp);
            if (dis >= p.thr) return false;
            neg_num = dis;
            D = coverage_fast(neg_num, p);
        }
    } else {
        float sign, dis;
        if (tune::fwd_dis_only) euclidean_sign_dis<FAST>(r, meta, w, xp, yp, sign, dis);
        else {
            const Dist dd = euclidean_p2f<FAST>(r, meta, w, xp, yp);
            sign = dd.sign;
            dis = dd.dx * dd.dx + dd.dy * dd.dy;
        }
        if (sign < 0 && dis >= p.thr) return false;
        neg_num = -sign * dis;
        D = coverage_fast(neg_num, p);
    }
    // alpha aggregation happens before the depth cull (SRK:350-358)
    if (!deferred) alpha_accumulate<DIST, FAST>(p, neg_num, D, s);

    const Bary wc = barycentric_clip<FAST>(w);
    const float zp = depth_of<FAST>(r, wc);
    if (zp < p.near_ || zp > p.far_) return deferred;                         // SRK:365
    const int fn = face_id(meta);
    s.q.insert(fn, zp, p.K);

    if (RGB == 0) {                                                            // SRK:390-397
        if (zp < s.depth_min && pixel_inside(w) && (p.double_side || face_front(meta))) {
            s.depth_min = zp; s.face_min = fn;
            sample_colour<FAST>(p, r, vc, tbase, wc, zp, s.c0, s.c1, s.c2);
        }
    } else if (RGB == 1) {                                                     // SRK:399-419
        if (!deferred && (face_front(meta) || p.double_side)) softmax_accumulate<FAST>(p, r, vc, tbase, wc, zp, D, s);
    }
    return deferred;
}

// The coverage-dependent part of a deferred inside pair: three edge projections -> coverage -> alpha, softmax.
template <int RGB, bool FAST, int KCAP>
__device__ inline void forward_pair_inside(const RasterParams& p, const FaceRec& r, const float* vc,
                                           const float* __restrict__ tbase, float xp, float yp,
                                           PixelState<KCAP>& s) {
    const Bary w = barycentric(r, xp, yp);
    const float neg_num = -euclidean_inside_dis<FAST>(r, w);
    const float D = coverage_fast(neg_num, p);
    alpha_accumulate<2, FAST>(p, neg_num, D, s);
    if (RGB == 1 && (face_front(r.meta) || p.double_side)) {
        const Bary wc = barycentric_clip<FAST>(w);
        const float zp = depth_of<FAST>(r, wc);
        if (zp < p.near_ || zp > p.far_) return;                              // SRK:365
        softmax_accumulate<FAST>(p, r, vc, tbase, wc, zp, D, s);
    }
}

// wavefronts per SIMD asked of the register allocator: K <= 16 and K <= 32 fit 128 VGPRs (4), K <= 64 fits 168 (3)
constexpr int fwd_waves(int kcap) { return JR_TUNE_FWD_OCC4 > 1 ? JR_TUNE_FWD_OCC4 : (JR_TUNE_FWD_OCC4 ? (kcap <= 32 ? 4 : 3) : 1); }
template <int DIST, int RGB, int KCAP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(fwd_waves(KCAP)))) void k_softras_forward(
    RasterParams p, int ntiles_total, const float* __restrict__ textures,
    const FaceGeo* __restrict__ geo, const int* __restrict__ bin_order, const int* __restrict__ bin_count,
    const int* __restrict__ bin_base, const unsigned long long* __restrict__ pool,
    unsigned long long* __restrict__ counters, unsigned long long pool_cap,
    float* __restrict__ aggrs, float* __restrict__ rgba, int32_t* __restrict__ ids) {
    extern __shared__ float4 s_dyn[];
    if (counters[0] > pool_cap) return;     // lists were not built (pool too small): the host launches again
    constexpr int BATCH = tune::fwd_batch;       // record slots per wavefront (64 x 176 B cap a CU at 14 wavefronts)
    FaceRec* s_rec = reinterpret_cast<FaceRec*>(s_dyn);                        // [BATCH]
    float* s_vcol = reinterpret_cast<float*>(s_rec + BATCH);                   // [BATCH*9] iff vertex colours

    // XCD-aware order: consecutive workgroup ids land on different XCDs (id % 8); the 16 tiles of a
    // bin (same list, same records) go to ONE XCD so that they share its L2.
    const int k = blockIdx.x >> 3;                       // k-th workgroup of XCD (blockIdx.x & 7)
    const int brank = (k >> 4) * 8 + (blockIdx.x & 7);   // bins are dealt round-robin to the XCDs ...
    if (brank * 16 >= ntiles_total) return;
    const int bin = bin_order[brank];                    // ... heaviest first (k_bin_schedule)
    const int sub = k & 15;                              // tile of the bin
    const int bins_per_img = p.bins_x * p.bins_y;
    const int b = bin / bins_per_img;
    const int bb = bin - b * bins_per_img;
    const int by = bb / p.bins_x, bx = bb - by * p.bins_x;
    const int n = bin_count[bin];
    // tune::fwd_prio: the wavefronts of the heaviest bins are the kernel's critical path (one 39k-face view alone takes
    // 0.70 ms, eight take 0.88): they get issue priority over the lighter wavefronts they share a SIMD with
    if (tune::fwd_prio > 0 && n > tune::fwd_prio) __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x, lx = lane & 7, ly = lane >> 3;
    SectionClock clk;            // instrumented builds only: 0 set-up, 1 cull + stage, 2 ballots + pre-cull, 3 raster loop, 4 stores
    clk.start();
    const int col0 = bx * BIN + (sub & 3) * TILE, row0 = by * BIN + (sub >> 2) * TILE;
    if (col0 >= p.IS || row0 >= p.IS) return;            // tile lies outside the image
    const int col = col0 + lx, row = row0 + ly;
    const bool valid = col < p.IS && row < p.IS;
    const float xp = pixel_centre(col, p.IS);
    const float yp = pixel_centre(p.IS - 1 - row, p.IS);                      // SRK:280-283
    // the tile's 8 column / 8 row centres are the xp of lanes 0..7 and the yp of lanes 0,8,..,56:
    // read them with v_readlane where they are used instead of pinning 16 SGPRs over the raster loop
    auto xc = [&](int c) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xp), c)); };
    auto yc = [&](int c) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yp), 8 * c)); };

    PixelState<KCAP> s;
    s.c0 = 1.f; s.c1 = 1.f; s.c2 = 1.f;
    s.alpha = p.alpha == 2 ? 1.f : 0.f;
    s.ssum = expf(p.eps / p.gamma); s.smax = p.eps;
    if (RGB == 0) { s.c0 = p.bg[0]; s.c1 = p.bg[1]; s.c2 = p.bg[2]; }
    else if (RGB == 1) { s.c0 = p.bg[0] * s.ssum; s.c1 = p.bg[1] * s.ssum; s.c2 = p.bg[2] * s.ssum; }
    s.depth_min = 10000000.f;
    s.face_min = -1;
    s.q.init(p.K, ids + (size_t)b * p.K * p.IS * p.IS, valid ? (unsigned)(row * p.IS + col) : 0u, (unsigned)(p.IS * p.IS));

    const unsigned long long* seg = pool + bin_base[bin];
    const FaceGeo* gbase = geo + (size_t)b * p.NF;
    const float* tbase = textures + (size_t)b * p.NF * p.T * 3;

    // The bin's list is walked 64 entries at a time, but only a fraction of them concerns THIS tile.
    // Survivors are compacted (ascending order kept) into the 64 LDS slots across list chunks and the
    // raster loop only runs on a full batch: its trip count is the MAXIMUM number of faces any pixel
    // needs, and max/mean over 64 lanes shrinks with the batch (measured: 17 survivors per list chunk
    // -> 64 per batch), and the per-batch ballots are paid 4x less often.
    clk.lap(0);
    int s0 = 0, fill = 0, cnt = 0;
    bool pending = false, keep = false;
    const FaceGeo* gp = gbase;
    int rank = 0;                                   // of this lane's entry among the chunk's survivors
    // the list is read one chunk AHEAD of its use (the entry load is the head of a chain of dependent loads)
    unsigned long long e_next = lane < n ? seg[lane] : 0ull;
    for (;;) {
        // ---- cull + stage: lane = list entry ----
        while (pending || s0 < n) {
            if (!pending) {
                const unsigned long long e = e_next;
                s0 += CHUNK;
                e_next = s0 + lane < n ? seg[s0 + lane] : 0ull;
                // The entry's tile mask is exact per axis (binning.hip: pixel_range), i.e. the face's border box
                // reaches a pixel column AND a pixel row of this tile: no box load, no second test here.
                keep = (e >> sub) & 1ull;
                if (!ballot(keep)) continue;        // no face of this chunk touches this tile
                gp = gbase + (int)(e >> 32);
                const unsigned long long surv = ballot(keep);
                if (!surv) continue;
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
        if (fill == 0) break;
        __syncthreads();
        clk.lap(1);

        // ---- raster: lane = slot for the ballots, then lane = pixel ----
        {
            float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
            const bool have = lane < fill;
            if (have) box = *reinterpret_cast<const float4*>(&s_rec[lane]);
            unsigned long long cx[8], ry[8];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                // check_border (SRK:28-34, :316): a pixel is culled when strictly outside the grown box
                cx[c] = ballot(have && !(xc(c) > box.y) && !(xc(c) < box.x));
                ry[c] = ballot(have && !(yc(c) > box.w) && !(yc(c) < box.z));
            }
            // private mask of the faces that pass this pixel's border test
            unsigned long long M;
            if (DIST == 2 && tune::fwd_prepass) {
                // Conservative pre-cull, lane = slot.  A pixel that lies more than the cull radius beyond the
                // LINE of one edge (on its outer side) is farther than the radius from the triangle, and the
                // reference culls that pair by distance (SRK:341-342) before it touches any state.  With w_k
                // the barycentric of vertex k (affine in the pixel, |grad w_k| = g_k = 1 / altitude_k) the test is
                //     w_k(pixel) < -g_k * (rad + margin)     for some k,
                // evaluated for the face of this lane at all 64 pixel centres: 5 VALU per pixel, and the v_cmp
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
                for (int rr = 0; rr < 8; rr++) {
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
                M = valid ? (((unsigned long long)(unsigned)mhi << 32) | (unsigned)mlo) : 0ull;
            } else {
                M = valid ? (select8(cx, lx) & select8(ry, ly)) : 0ull;
            }
            clk.lap(2);
            unsigned long long Mdef = 0ull;        // this pixel's deferred (inside) pairs of the batch
            while (M) {
                const int j = __builtin_ctzll(M);
                const unsigned long long rest = M & (M - 1);
                const FaceRec& r = s_rec[j];
                const float* vc = s_vcol + j * 9;
                bool deferred;
                if (face_safe(r.meta) && p.consts_safe)
                    deferred = forward_pair<DIST, RGB, true, KCAP>(p, r, vc, tbase, xp, yp, s);
                else
                    deferred = forward_pair<DIST, RGB, false, KCAP>(p, r, vc, tbase, xp, yp, s);
                if (DIST == 2 && tune::fwd_defer_inside && deferred) Mdef |= M ^ rest;
                M = rest;
            }
            if (DIST == 2 && tune::fwd_defer_inside) {
                while (Mdef) {
                    const int j = __builtin_ctzll(Mdef);
                    Mdef &= Mdef - 1;
                    const FaceRec& r = s_rec[j];
                    const float* vc = s_vcol + j * 9;
                    if (face_safe(r.meta) && p.consts_safe)
                        forward_pair_inside<RGB, true, KCAP>(p, r, vc, tbase, xp, yp, s);
                    else
                        forward_pair_inside<RGB, false, KCAP>(p, r, vc, tbase, xp, yp, s);
                }
            }
        }
        fill = 0;
        __syncthreads();                        // readers are done with s_rec before it is refilled
        clk.lap(3);
    }
    clk.lap(1);

    if (!valid) return;
    // ---- finalise (SRK:426-455) ----
    const size_t pp = (size_t)p.IS * p.IS;
    const size_t pn = (size_t)row * p.IS + col;
    float a_out;
    if (p.alpha == 0) a_out = s.alpha;
    else if (p.alpha == 1) a_out = s.alpha / p.NF;
    else a_out = (float)(1. - (double)s.alpha);
    float o0 = p.bg[0], o1 = p.bg[1], o2 = p.bg[2], g0 = 0.f, g1 = 0.f;
    if (RGB == 0) {
        if (s.face_min != -1) { o0 = s.c0; o1 = s.c1; o2 = s.c2; }
        g0 = s.depth_min; g1 = (float)s.face_min;
    } else if (RGB == 1) {
        o0 = s.c0 / s.ssum; o1 = s.c1 / s.ssum; o2 = s.c2 / s.ssum;
        g0 = s.ssum; g1 = s.smax;
    }
    float* out = rgba + (size_t)b * 4 * pp + pn;
    out[0] = o0; out[pp] = o1; out[2 * pp] = o2; out[3 * pp] = a_out;
    float* ag = aggrs + (size_t)b * 2 * pp + pn;
    ag[0] = g0; ag[pp] = g1;
    int32_t* io = ids + (size_t)b * p.K * pp + pn;
#pragma unroll
    for (int k = 0; k < KCAP; k++)
        if (k < p.K) {
            if (!ids_in_global<KCAP>()) io[(size_t)k * pp] = s.q.id_of(k);
            else if (k >= s.q.size) io[(size_t)k * pp] = -1;         // the filled slots were stored when they were filled
        }
    clk.lap(4);
    clk.flush(counters, 4);
}

template <int DIST, int RGB>
static void launch_k(hipStream_t st, const RasterParams& p, int ntiles, const float* textures,
                     const BinWorkspace& ws, float* aggrs, float* rgba, int32_t* ids) {
    const int grid = ((ntiles + 127) / 128) * 128;   // whole bins (16 tiles) per XCD slot
    const size_t smem = sizeof(FaceRec) * tune::fwd_batch + (p.tex == 1 ? sizeof(float) * 9 * tune::fwd_batch : 0);
    // K-buffer capacity: 16 (the default K), 32 (K = 17..32), 64
    if (p.K <= 16)
        k_softras_forward<DIST, RGB, 16><<<grid, 64, smem, st>>>(
            p, ntiles, textures, ws.geo, ws.bin_order, ws.bin_count, ws.bin_base, ws.pool, ws.counters, ws.pool_cap, aggrs, rgba, ids);
    else if (p.K <= 32)
        k_softras_forward<DIST, RGB, 32><<<grid, 64, smem, st>>>(
            p, ntiles, textures, ws.geo, ws.bin_order, ws.bin_count, ws.bin_base, ws.pool, ws.counters, ws.pool_cap, aggrs, rgba, ids);
    else
        k_softras_forward<DIST, RGB, 64><<<grid, 64, smem, st>>>(
            p, ntiles, textures, ws.geo, ws.bin_order, ws.bin_count, ws.bin_base, ws.pool, ws.counters, ws.pool_cap, aggrs, rgba, ids);
}

void launch_softras_forward(hipStream_t st, const RasterParams& p, const float* textures,
                            const BinWorkspace& ws, float* aggrs, float* rgba, int32_t* ids) {
    const int ntiles = p.B * p.bins_x * p.bins_y * SUBS * SUBS;
#define JR_FWD(D, R) launch_k<D, R>(st, p, ntiles, textures, ws, aggrs, rgba, ids)
    const int rgb = p.rgb == 0 ? 0 : (p.rgb == 1 ? 1 : 2);
    switch (p.dist * 3 + rgb) {
        case 0: JR_FWD(0, 0); break;
        case 1: JR_FWD(0, 1); break;
        case 2: JR_FWD(0, 2); break;
        case 3: JR_FWD(1, 0); break;
        case 4: JR_FWD(1, 1); break;
        case 5: JR_FWD(1, 2); break;
        case 6: JR_FWD(2, 0); break;
        case 7: JR_FWD(2, 1); break;
        default: JR_FWD(2, 2); break;
    }
#undef JR_FWD
}

}  // namespace jr
