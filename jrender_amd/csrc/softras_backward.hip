// SoftRas backward for gfx950 (MI355X).
//
// Replaces backward_soft_rasterize_cuda_kernel (SRK:1177-1360).  The reference runs one thread
// per pixel that gathers its <=K buffered faces from global memory (63 scattered floats per
// pixel-face pair) and issues 9+3T float atomics per pair.  Here one wavefront owns an 8x8 tile:
//
//   * every lane (= pixel) loads its K buffered ids and sorts them in registers;
//   * the faces the tile needs are the UNION of those ids: the wavefront repeatedly extracts the
//     smallest pending id (DPP min), the ballot of the lanes whose head equals it is that face's holder
//     mask, those lanes advance.  Up to 64 distinct faces form a batch (a tile of the headline workload
//     needs ~40), their packed records are staged into LDS (lane = slot).  The bin lists are only used
//     for the launch order and the empty-bin exit;
//   * the batch is cut into WORK ITEMS = (face, up to 16 of the pixels that hold it): the pixel x slot
//     bit matrix is transposed with ballots, a prefix sum numbers the items, and every 16-lane DPP row
//     takes one item per trip.  The lanes of the row pick "their" pixel (n-th set bit of the face's
//     holder mask) and GATHER that pixel's state with ds_bpermute — lanes are no longer tied to pixels,
//     so a row is full unless the face has fewer than 16 holders left (lane utilisation 45 % -> 74 %);
//   * the 12 gradient components are reduced with a row-local butterfly transpose-reduction
//     (row_mirror / row_half_mirror / quad_perm, payload halving every step) that leaves component k
//     in lane k, and ONE atomic instruction per row covers all components of the item (instead of one
//     atomic per pixel and component).
//
// (Measured on MI355X, tools/: LDS float atomics — ds_add_f32 — retire about one lane per clock, so a
// per-lane scatter into LDS accumulators is 2x slower; a whole-wavefront reduction per face 1.5x slower;
// rows tied to fixed 4x4 pixel blocks leave 55 % of the lanes idle.)
#include "jr_kernels.h"

namespace jr {

// ---- DPP row (16 lanes) primitives; every lane of the wavefront must be active ----------------
template <int CTRL>
__device__ inline float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ inline unsigned dpp_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

// Sum of 16 per-lane values over the 16 lanes of a row, "transposed": lane i of the row ends up
// with the row total of v[i].  Butterfly with halving payload (8+4+2+1 exchanges instead of
// 16 x 4): partners are row_mirror, row_half_mirror, quad_perm[3,2,1,0], quad_perm[1,0,3,2].
// (Bank-masked v_add_f32_dpp instead of the selects: +3 %, tools/ablate/patches/dead_switches_r03.patch.)
__device__ inline float row_transpose_reduce(const float (&v)[16], int li) {
    float a[8], b[4], c[2];
    const bool h8 = li & 8, h4 = li & 4, h2 = li & 2, h1 = li & 1;
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = (h8 ? v[j + 8] : v[j]) + dpp_f<0x140>(h8 ? v[j] : v[j + 8]);
#pragma unroll
    for (int j = 0; j < 4; j++) b[j] = (h4 ? a[j + 4] : a[j]) + dpp_f<0x141>(h4 ? a[j] : a[j + 4]);
#pragma unroll
    for (int j = 0; j < 2; j++) c[j] = (h2 ? b[j + 2] : b[j]) + dpp_f<0x1B>(h2 ? b[j] : b[j + 2]);
    return (h1 ? c[1] : c[0]) + dpp_f<0xB1>(h1 ? c[0] : c[1]);
}

// smallest value over the wavefront, uniform (all lanes active)
__device__ inline int wave_min(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));   // row_half_mirror
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));   // row_mirror
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// position of the n-th (0-based) set bit of m; n < popcount(m)
__device__ inline int select_bit(unsigned long long m, int n) {
    const unsigned lo = (unsigned)m, hi = (unsigned)(m >> 32);
    const int cl = __builtin_popcount(lo);
    const bool up64 = n >= cl;
    unsigned w = up64 ? hi : lo;
    int base = up64 ? 32 : 0;
    n = up64 ? n - cl : n;
#pragma unroll
    for (int s = 16; s >= 1; s >>= 1) {
        const int c = __builtin_popcount(w & ((1u << s) - 1u));
        const bool up = n >= c;
        n = up ? n - c : n;
        base += up ? s : 0;
        w = up ? (w >> s) : w;
    }
    return base;
}

// value of v in lane src (all lanes active)
__device__ inline float gather(float v, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, v)));
}

template <int N>
__device__ inline void sort_ascending(int (&s)[N]) {
#pragma unroll
    for (int k = 2; k <= N; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const int a = s[i], b = s[l];
                    const int lo = min(a, b), hi = max(a, b);
                    s[i] = up ? lo : hi;
                    s[l] = up ? hi : lo;
                }
            }
}

struct PixelGrad {           // per-pixel inputs of the backward (SRK:1230-1231, :1281, :1315, :1323)
    float g0, g1, g2, g3;    // upstream gradient of r g b a
    float o0, o1, o2, o3;    // forward outputs r g b a
    float ssum, smax;        // aggrs_info
    float r_ssum;            // 1/ssum (gradient-only quantity: reciprocal multiply, <= 1 ulp)
};

// x / c for gradient-only quantities: reciprocal multiply when the constant is in the safe range
__device__ inline float gdiv(float x, float c, float rc, const RasterParams& p) {
    return (p.consts_safe && !(tune::bwd_exact & 8)) ? x * rc : x / c;
}

// Contribution of one (pixel, face) pair: gv = d/d(x0 y0 z0 x1 y1 z1 x2 y2 z2); the colour gradient is tgs * upstream
// (times the clipped weights wcw for vertex colours) and is formed by the caller right where it is reduced: a gt[9]
// array carried through this function cost two scratch slots with a load on the critical path of every trip.
// (3 values for a single-texel surface, 9 for vertex colours).  Returns the sampled texel.
// The forward quantities it re-derives (w, distance, coverage, clipped depth, normalised depth) go
// through the SAME device functions as the forward kernel, so they carry the forward's exact bits
// (the softmax weight exp((zn - smax)/gamma) is extremely sensitive to zn).  Everything that only
// feeds gradients uses reciprocal multiplies: float atomics already make the sums order dependent.
template <int DIST, int RGB, bool FAST>
__device__ inline int backward_pair(const RasterParams& p, const FaceRec& r, const float* vc, const float* tex_block,
                                    const PixelGrad& px, float xp, float yp,
                                    const float* __restrict__ tbase, float (&gv)[9], float (&wcw)[3],
                                    float& tgs, bool& tex_on, unsigned long long* __restrict__ counters) {
    const int meta = r.meta;
    const int fn = face_id(meta);
    const Bary w = barycentric(r, xp, yp);
    float D, dis = 0.f;
    Dist dd;
    dd.sign = 0.f; dd.dx = 0.f; dd.dy = 0.f; dd.t0 = 0.f; dd.t1 = 0.f; dd.t2 = 0.f;
    if (DIST == 0) D = 1.f;                                               // SRK:1258-1270
    else if (DIST == 1) { dis = barycentric_dist(w); D = coverage_backward<tune::bwd_exact>(-dis, p); }
    else {
        // nothing is decided from the projection parameter here (sign and region come from the exact w)
        dd = euclidean_p2f<FAST, tune::bwd_tv_rcp ? TV_RCP : TV_IEEE>(r, meta, w, xp, yp);
        dis = dd.dx * dd.dx + dd.dy * dd.dy;
        D = coverage_backward<tune::bwd_exact>(-dd.sign * dis, p);
    }
    float ca = px.g3;                                                     // SRK:1281-1291
    if (p.alpha == 1) ca /= p.NF;
    else if (p.alpha == 2) ca = (tune::bwd_exact & 4) ? ca * ((1 - px.o3) / fmaxf(1 - D, 1e-6f))
                                                      : ca * (1 - px.o3) * __builtin_amdgcn_rcpf(fmaxf(1 - D, 1e-6f));
    float cxy = ca;
    const Bary wc = barycentric_clip<FAST>(w);                            // SRK:1294-1296
    const float zp = depth_of<FAST>(r, wc);
    const int texel = p.tex == 0 ? surface_texel(wc, p.R) : 0;
    tgs = 0.f;
    tex_on = false;
    if (RGB == 0) {                                                       // SRK:1299-1306
        if ((float)fn == px.smax) { tgs = 1.f; tex_on = true; }
    } else if (RGB == 1) {                                                // SRK:1308-1332
        const float zn = div_known<FAST>(p.far_ - zp, p.far_minus_near, p.r_far_minus_near);
        const float zs = (tune::bwd_exact & 4) ? D * exp_over_gamma<tune::bwd_exact>(zn - px.smax, p) / px.ssum
                                               : D * exp_over_gamma<tune::bwd_exact>(zn - px.smax, p) * px.r_ssum;
        tgs = zs; tex_on = true;
        float k0, k1, k2;
        if (p.tex == 0) {
            if (p.T == 1) { k0 = r.col[0]; k1 = r.col[1]; k2 = r.col[2]; }
            else {
                const float* tx_ = (tex_block ? tex_block : tbase + (size_t)fn * p.T * 3) + texel * 3;      // (the staged LDS copy of the face's texels, or global)
                k0 = tx_[0]; k1 = tx_[1]; k2 = tx_[2];
                if (!tex_block) asm volatile("" : "+v"(k0), "+v"(k1), "+v"(k2));     // (a global load: its wait stays in this branch, not at the join with the T = 1 path - softras_forward.hip: sample_colour)
            }
        } else {                                                           // SRK:1147-1149 (affine)
            k0 = (wc.w0 * vc[0] + wc.w1 * vc[3]) + wc.w2 * vc[6];
            k1 = (wc.w0 * vc[1] + wc.w1 * vc[4]) + wc.w2 * vc[7];
            k2 = (wc.w0 * vc[2] + wc.w1 * vc[5]) + wc.w2 * vc[8];
        }
        float crgb = 0.f;
        crgb += px.g0 * (k0 - px.o0);
        crgb += px.g1 * (k1 - px.o1);
        crgb += px.g2 * (k2 - px.o2);
        crgb *= zs;
        cxy += (tune::bwd_exact & 4) ? crgb / D : crgb * __builtin_amdgcn_rcpf(D);
        const float cz = gdiv(gdiv(crgb, p.gamma, p.r_gamma, p), p.near_minus_far, p.r_near_minus_far, p) * zp * zp;
        gv[2] = cz * wc.w0 * r.rz[0] * r.rz[0];
        gv[5] = cz * wc.w1 * r.rz[1] * r.rz[1];
        gv[8] = cz * wc.w2 * r.rz[2] * r.rz[2];
    }
    wcw[0] = wc.w0; wcw[1] = wc.w1; wcw[2] = wc.w2;      // backward_sample_texture vertex: w[j]*grad (SRK:1170-1172)
    cxy *= gdiv(D * (1 - D), p.sigma, p.r_sigma, p);                      // SRK:1336
    if (DIST == 1) {                                                      // SRK:1118-1132
        const int q = w.w0 > w.w1 ? (w.w1 > w.w2 ? 2 : 1) : (w.w0 > w.w2 ? 2 : 0);
        const float mul = 2.f * sqrtf(fabsf(dis));
#pragma unroll
        for (int l = 0; l < 2; l++) {
            const float ql = q == 0 ? r.inv[l] : (q == 1 ? r.inv[3 + l] : r.inv[6 + l]);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float s = 0.f;
                s += -ql * r.inv[3 * k + 0] * xp;
                s += -ql * r.inv[3 * k + 1] * yp;
                s += -ql * r.inv[3 * k + 2] * 1.f;
                gv[3 * k + l] = s * cxy * mul;
            }
        }
    } else if (DIST == 2) {                                               // SRK:1341-1347
        const float w0s[3] = {w.w0, w.w1, w.w2};
        const float ts[3] = {dd.t0, dd.t1, dd.t2};
        const float cx2 = 2 * dd.sign * cxy;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float c = cx2 * (ts[k] + w0s[k]);
            gv[3 * k + 0] = c * dd.dx;
            gv[3 * k + 1] = c * dd.dy;
        }
    }
    return texel;
}


// wavefronts per SIMD asked of the register allocator: every instantiation without scratch (tools/kernel_resources.sh).  The
// 'barycentric' distance keeps three more live values through the pair loop than 'euclidean': <1,0,16> and <1,1,16> carried
// 20 / 32 B at five wavefronts, <1,1,32> with the LDS texel sums 16 B at four - one wavefront fewer for those three.
constexpr int bwd_waves(int dist, int rgb, int kcap, bool texlds) {
    const int w = texlds ? (kcap <= 32 ? 4 : 3) : (kcap <= 16 ? JR_TUNE_BWD_WAVES : (kcap <= 32 ? 4 : JR_TUNE_BWD_WAVES64));
    const bool spills = dist == 1 && rgb != 2 && ((kcap <= 16 && !texlds) || (kcap == 32 && texlds) || (kcap > 32 && !texlds));   // (K = 64 'barycentric': 32 - 48 B at four wavefronts)
    return spills && w > 3 ? w - 1 : w;
}

// LDS hand-over inside the kernel's ONE wavefront (writes by some lanes, reads by others).  LDS instructions of a wavefront execute in
// order, so only the compiler has to be kept from moving accesses across this point; __syncthreads() also is a workgroup-scope fence,
// which on gfx950 waits for every outstanding global store and atomic (s_waitcnt vmcnt(0)): the gradient atomics of a whole batch.
__device__ inline void bsync() {
    if (tune::light_sync) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_sched_barrier(0);           // (nothing is scheduled across: with the freedom the lighter fence gives, the allocator spilled 8 - 12 B)
    } else __syncthreads();
}

// open-addressing table of the hashed union (round 6): 256 entries x 16 B over the record slots + 64 compacted entries
constexpr int HT_LOG2 = 8, HT_SIZE = 1 << HT_LOG2, HT_PROBES = 24;
struct HEntry { int key; int pad; unsigned m[2]; };
static_assert(sizeof(HEntry) == 16, "HEntry");

template <int DIST, int RGB, int KCAP, bool TEXLDS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(bwd_waves(DIST, RGB, KCAP, TEXLDS)))) void k_softras_backward(
    RasterParams p, int nbins, int heavy_cap, int split_log2, const float* __restrict__ textures,
    const FaceGeo* __restrict__ geo, const int* __restrict__ bin_order, const int* __restrict__ bin_count,
    const float* __restrict__ rgba, const float* __restrict__ aggrs,
    const int32_t* __restrict__ ids, const float* __restrict__ grad_rgba,
    float* __restrict__ grad_faces, float* __restrict__ grad_textures, unsigned long long* __restrict__ counters) {
    extern __shared__ float4 s_dyn[];
    // faces per batch: a tile of the headline workload needs ~40; 64 slots of 176 B cap a CU at 13 wavefronts
    constexpr int BATCH = TEXLDS ? tune::bwd_batch_for(16) : tune::bwd_batch_for(KCAP);     // (the staged texel blocks are 24 T bytes per slot: no larger batches there)
    FaceRec* s_rec = reinterpret_cast<FaceRec*>(s_dyn);                        // [BATCH]
    static_assert(!tune::bwd_hash_union || (HT_SIZE + 64) * 16 <= BATCH * (int)sizeof(FaceRec), "the hashed union's table lies over the record slots");
    float* s_vcol = reinterpret_cast<float*>(s_rec + BATCH);                   // [BATCH*9] iff vertex colours
    // TEXLDS (round 5; its own instantiations: as a run-time flag it cost the default kernels 32 B of scratch and + 50 % time; 'surface' textures with 1 < T <= BWD_TEX_LDS_MAX texels, small launches): the batch's texture blocks
    // are staged next to the records.  A pair of a T > 1 face reads its texel colour - k0..k2 of SRK:1315 - from global
    // memory IN THE MIDDLE of its arithmetic, and on gfx950 that load's s_waitcnt also waits for every atomic issued
    // before it (one VMEM counter, in order): each trip paid the round trip of the previous trip's texel atomics.  BASELINE
    // configs[1] (spot cow, 25 texels per face, 1024^2, one view): backward 0.28 ms, 4 x the 3 300-face sphere's at T = 1.
    float* s_tex = s_vcol;                                                     // [BATCH][T*3] iff TEXLDS (never together with vertex colours)
    // ... and the texel GRADIENTS of a batch are summed in LDS too: a pair adds its three colour components to the face's texel
    // (ds_add_f32), one global atomic per touched (face, texel, channel) leaves at the end of the batch.  Straight to global
    // (the other path) a face that covers 2 000 pixels of a 1024^2 image takes 6 000 float atomics on the five cache lines of
    // its 25 texels, sixteen lanes of a row often on the SAME address in one instruction.
    float* s_gtex = s_tex + BATCH * (TEXLDS ? p.T * 3 : 0);                    // [BATCH][T*3] iff TEXLDS
    __shared__ unsigned long long s_has[BATCH];     // slot -> pixels (lanes) that hold the face
    __shared__ int s_ioff[CHUNK + 1];                // slot -> first work item (exclusive prefix), [64] = total

    // XCD-aware order as in the forward.  tune::bwd_split (small launches only, heavy_cap > 0): the tiles of the HEAVY
    // bins (the first counters[3] ranks of the order, the forward's definition) are taken by SPLIT wavefronts each -
    // wavefront `part` keeps the buffered ids with id % SPLIT == part and is otherwise a complete tile job.  The
    // gradient is a sum over (pixel, face) pairs, so the parts do not talk to each other; what they share is the
    // critical path of a launch that cannot fill the GPU (one view: the limb tiles' extraction + pair loop).
    // (split_log2: 4 wavefronts per heavy tile, 8 for launches of up to tune::bwd_split8_pixels - the spot cow at 256^2: 0.203 -> 0.182 ms,
    //  while the 1024^2 shapes lose 1 - 5 % with 8, round 5 call 15; a scalar of the prologue only, it is dead before the pair loop)
    static_assert((tune::bwd_split & (tune::bwd_split - 1)) == 0, "JR_TUNE_BWD_SPLIT: a power of two (the parts are id & (split - 1))");
    const int smask = (1 << split_log2) - 1;
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;      // k-th workgroup of its XCD
    const int nheavy = heavy_cap > 0 ? min((int)counters[3], heavy_cap) : 0;
    const int hx = (nheavy - xcd + 7) >> 3;                   // heavy bins dealt to this XCD (launch ranks xcd, xcd + 8, ...)
    const int tl = 2 * sub_log2_of(p), tmask = (1 << tl) - 1;     // a bin has 1 << tl tiles
    int brank, sub, part = -1;
    if (k < ((hx << tl) << split_log2)) { brank = ((k >> split_log2) >> tl) * 8 + xcd; sub = (k >> split_log2) & tmask; part = k & smask; }
    else { const int k2 = k - ((hx << tl) << split_log2); brank = (hx + (k2 >> tl)) * 8 + xcd; sub = k2 & tmask; }   // bins are dealt round-robin to the XCDs ...
    if (brank >= nbins) return;
    const int bin = bin_order[brank];                         // ... heaviest first (k_bin_alloc_schedule)
    const int n = bin_count[bin];
    if (n == 0) return;
    const int bins_per_img = p.bins_x * p.bins_y;
    const int b = bin / bins_per_img;
    const int bb = bin - b * bins_per_img;
    const int by = bb / p.bins_x, bx = bb - by * p.bins_x;
    const int col0 = (bx << bin_log2_of(p)) + ((sub & ((1 << sub_log2_of(p)) - 1)) << TILE_LOG2);
    const int row0 = (by << bin_log2_of(p)) + ((sub >> sub_log2_of(p)) << TILE_LOG2);
    if (col0 >= p.IS || row0 >= p.IS) return;

    // Work items are (face, up to 16 of its holders) and the 16 lanes of a DPP row take one.  (Half rows of 8 were built and
    // measured: trips -13.5 %, lanes in use 72 -> 84 %, time +-0 / +2.4 % - two components per lane after the reduction are
    // two atomic instructions per flush; tools/ablate/patches/dead_switches_r04.patch.)
    constexpr int G = 16, NG = 64 / G, GSH = 4;
    const int lane = threadIdx.x, li = lane & (G - 1), blk = lane >> GSH;
    const int col = col0 + (lane & 7), row = row0 + (lane >> 3);
    const bool valid = col < p.IS && row < p.IS;
    const size_t pp = (size_t)p.IS * p.IS;
    const size_t pn = valid ? (size_t)row * p.IS + col : 0;
    const float xp = pixel_centre(col, p.IS);
    const float yp = pixel_centre(p.IS - 1 - row, p.IS);                      // SRK:1218-1221

    SectionClock clk;            // instrumented builds only: 0 tile state, 1 extraction, 2 staging + items, 3 gather, 4 pair, 5 reduce + atomics
    clk.start();
    // instrumented build JR_TUNE_COUNT_PATHS = 2 (tools/sim/min_valu_bwd.py --measure): the dynamic half of the backward's VALU model -
    // 0 tiles, 1 union passes, 2 batches, 3 trips, 4 lanes with a pair, 5 trips with an inside pair, 6 inside lanes, 7 trips with a
    // non-FAST face, 8 their lanes, 9 distinct faces, 10 work items.  Wave-uniform scalars, flushed by lane 0; dead code otherwise.
    unsigned pcnt[11] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // this pixel's buffered face ids; the reference stops at the first -1 (SRK:1236-1238)
    constexpr int BIG = 0x7fffffff;
    constexpr bool HASHED = tune::bwd_hash_union;
    const int32_t* ip = ids + (size_t)b * p.K * pp + pn;
    int mine[HASHED ? 1 : KCAP];
    int cur = BIG;
    {
        // plane 0 decides whether the tile has anything to do
        const int r0 = valid ? ip[0] : -1;
        if (!ballot(r0 >= 0)) return;           // nothing buffered anywhere in this tile
        if (!HASHED) {
            // the other planes are fetched at once (independent loads) and the reference's early stop is applied in registers
            int raw[KCAP];
            raw[0] = r0;
#pragma unroll
            for (int k = 1; k < KCAP; k++) raw[k] = (valid && k < p.K) ? ip[(size_t)k * pp] : -1;
            bool live = true;
#pragma unroll
            for (int k = 0; k < KCAP; k++) {
                live = live && raw[k] >= 0 && raw[k] < p.NF;   // -1 ends the list (ids outside [0, NF) too)
                mine[HASHED ? 0 : k] = (live && (part < 0 || (raw[k] & smask) == part)) ? raw[k] : BIG;
            }
        }
    }
    if (!HASHED) {
        sort_ascending(mine);
        cur = mine[0];
    }

    PixelGrad px;
    px.g0 = px.g1 = px.g2 = px.g3 = 0.f;
    px.o0 = px.o1 = px.o2 = px.o3 = 0.f;
    px.ssum = 1.f; px.smax = 0.f;
    if (valid) {
        const float* gp = grad_rgba + (size_t)b * 4 * pp + pn;
        const float* op = rgba + (size_t)b * 4 * pp + pn;
        px.g0 = gp[0]; px.g1 = gp[pp]; px.g2 = gp[2 * pp]; px.g3 = gp[3 * pp];
        px.o0 = op[0]; px.o1 = op[pp]; px.o2 = op[2 * pp]; px.o3 = op[3 * pp];
        px.ssum = aggrs[(size_t)b * 2 * pp + pn];
        px.smax = aggrs[(size_t)b * 2 * pp + pp + pn];
    }
    px.r_ssum = 0.f;                                 // (formed after the gather)

    const FaceGeo* gbase = geo + (size_t)b * p.NF;
    const float* tbase = textures + (size_t)b * p.NF * p.T * 3;
    float* gfbase = grad_faces + (size_t)b * p.NF * 9;
    float* gtbase = grad_textures + (size_t)b * p.NF * p.T * 3;
    const int ntex = p.tex == 1 ? 9 : (p.T == 1 ? 3 : 0);   // colour-gradient components shared by all pixels

    // The faces this tile needs are the UNION of its pixels' buffered ids — the bin lists are not needed
    // to find them.  Every lane holds its ids sorted; the wavefront repeatedly extracts the smallest id
    // that is still pending anywhere (DPP min), the ballot of the lanes whose head equals it IS the
    // face's holder mask, and those lanes advance.  Up to 64 distinct faces form a batch (a tile of the
    // headline workload needs ~40): slot = extraction order = ascending id, lane j keeps slot j's id and
    // holder mask.  No list walk, no binary search, no bit-matrix transpose.
    clk.lap(0);
    // ---- one batch: `fill` distinct faces, lane j < fill holds face j's id and holder mask ----
    auto run_batch = [&](const int fill, const int myid, const unsigned long long has) {
        // ---- stage the batch's records (lane = slot) ----
        if (lane < fill) {
            const float4* src = reinterpret_cast<const float4*>(gbase + myid);
            float4* dst = reinterpret_cast<float4*>(&s_rec[lane]);
#pragma unroll
            for (int k = 0; k < 11; k++) dst[k] = src[k];
            if (p.tex == 1) {
                const float* tx_ = tbase + (size_t)myid * p.T * 3;
#pragma unroll
                for (int k = 0; k < 9; k++) s_vcol[lane * 9 + k] = tx_[k];
            }
        }
        if (TEXLDS) {            // 16 lanes per slot, four slots per pass: contiguous T*3 floats per face
            const int t3 = p.T * 3, sub = lane >> 4, l16 = lane & 15;
            for (int s0 = 0; s0 < fill; s0 += 4) {
                const int slot = s0 + sub;
                const int id = __builtin_amdgcn_ds_bpermute((slot < fill ? slot : 0) << 2, myid);    // slot's face id lives in lane `slot`
                if (slot < fill) {
                    const float* src = tbase + (size_t)id * t3;
                    float* dst = s_tex + slot * t3;
                    float* gdst = s_gtex + slot * t3;
                    for (int k = l16; k < t3; k += 16) { dst[k] = src[k]; gdst[k] = 0.f; }
                }
            }
        }

        // ---- work items: (face slot, group of <= 16 of the pixels that hold it) ----
        const int items = (__builtin_popcountll(has) + G - 1) >> GSH;
        int incl = items;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        const int nitems = __builtin_amdgcn_readlane(incl, 63);
        if (tune::count_paths_bwd) { pcnt[2]++; pcnt[9] += (unsigned)fill; pcnt[10] += (unsigned)nitems; }
        if (lane < BATCH) s_has[lane] = has;
        s_ioff[lane] = incl - items;
        if (lane == 0) s_ioff[64] = nitems;
        bsync();

        // ---- each 16-lane DPP row takes one item per trip (four faces in flight per wavefront): gather
        //      the pixels' state, pair arithmetic, row-local transpose-reduction -> lane k of the row
        //      holds component k -> ONE atomic instruction per row and item ----
        clk.lap(2);
        // The item of the NEXT trip is looked up (slot, rank base, holder mask: dependent LDS reads) before the
        // reduction of the current one, so that the latency hides behind the reduction instead of heading the trip.
        // tune::bwd_row_ranges: a row takes a CONTIGUOUS quarter of the items instead of every fourth one.  The items of one
        // face follow each other (2.4 per face and tile on the headline scene), so they now land in the same row, which keeps
        // their sum in a register and issues its atomic only when the face changes: the float atomics were a quarter of the
        // kernel's time (cost probe: no atomics -27 %), and the L2 sees ~2x fewer of them.
        constexpr bool RANGES = tune::bwd_row_ranges;
        const int per_row = RANGES ? (nitems + NG - 1) / NG : 0;
        const int item0 = RANGES ? blk * per_row : blk;
        const int item_end = RANGES ? min(item0 + per_row, nitems) : nitems;
        int j = 0, nth = 64;
        unsigned long long hs = 0ull;
        bool ract = item0 < item_end;                            // uniform within a row
        if (ract) {
            if (RANGES) {                                        // first slot of the row's range: binary search over the prefix
                int lo = 0, hi = 63;                             // largest j with s_ioff[j] <= item0 (s_ioff is non-decreasing, [64] = total)
#pragma unroll
                for (int it = 0; it < 6; it++) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (s_ioff[mid] <= item0) lo = mid; else hi = mid - 1;
                }
                j = lo;
            }
            while (s_ioff[j + 1] <= item0) j++;                  // (skips slots without items)
            nth = (item0 - s_ioff[j]) * G + li;
            hs = s_has[j];
        }
        float acc = 0.f;                                         // component li of the row's current face, not yet in memory
        int acc_fn = -1;
        // component c of face fn: 0..8 vertex coordinates (grad_faces), 9..11 the single-texel colour (grad_textures)
        // tune::bwd_one_atomic: the address is SELECTED per lane, so that the vertex and the colour components of a flush
        // leave in ONE atomic instruction (an if / else over the two output buffers is two)
        auto add_component = [&](int fn, int c, float val) {
            if (tune::bwd_one_atomic) {
                float* at = c < 9 ? gfbase + (size_t)fn * 9 + c : gtbase + (size_t)fn * p.T * 3 + (c - 9);
                if (val != 0.f && c < 9 + (ntex == 3 ? 3 : 0)) { if (JR_TUNE_DIAG & 128) *at = val; else atomicAdd(at, val); }   // (diagnostic bit 7: plain stores - WRONG sums - what does the read-modify-write cost?)
                return;
            }
            if (val == 0.f) return;
            if (c < 9) atomicAdd(gfbase + (size_t)fn * 9 + c, val);
            else if (c < 9 + (ntex == 3 ? 3 : 0)) atomicAdd(gtbase + (size_t)fn * p.T * 3 + (c - 9), val);
        };
        auto flush = [&]() {
            if (acc_fn >= 0) add_component(acc_fn, li, acc);
        };
        const int ntrips = RANGES ? per_row : (nitems + NG - 1) / NG;
        for (int trip = 0; trip < ntrips; trip++) {
            const int i0 = RANGES ? 0 : trip * NG;               // (strided assignment: item = i0 + blk)
            const int jc = j;                                    // slot of this trip's item
            const bool ractc = ract;
            const bool act = nth < __builtin_popcountll(hs);
            const int src = (JR_TUNE_DIAG & 8) ? lane : (act ? select_bit(hs, nth) : lane);    // the pixel (lane) this pair belongs to (diagnostic bit 3: no search)
            PixelGrad q;
            if (JR_TUNE_DIAG & 256) q = px;              // (diagnostic bit 8, WRONG results: what do the 13 ds_bpermute gathers cost?)
            else {
            q.g0 = gather(px.g0, src); q.g1 = gather(px.g1, src); q.g2 = gather(px.g2, src); q.g3 = gather(px.g3, src);
            q.o0 = gather(px.o0, src); q.o1 = gather(px.o1, src); q.o2 = gather(px.o2, src); q.o3 = gather(px.o3, src);
            q.ssum = gather(px.ssum, src); q.smax = gather(px.smax, src);
            q.r_ssum = __builtin_amdgcn_rcpf(q.ssum);       // (the same bits as gathering the pixel's own reciprocal: one v_rcp per trip instead of a register held through the kernel + a 13th ds_bpermute)
            }
            const float qx = (JR_TUNE_DIAG & 256) ? xp : gather(xp, src), qy = (JR_TUNE_DIAG & 256) ? yp : gather(yp, src);
            if (tune::profile_sections) { __builtin_amdgcn_s_waitcnt(0); clk.lap(3); }
            const FaceRec& fr = s_rec[jc];
            float v[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            float tw = 0.f, wcw[3] = {0.f, 0.f, 0.f};               // colour-gradient weight of this pair, clipped weights
            // check_border is repeated by the reference's backward (SRK:1244); one predicate, no
            // short-circuit ladder (every rung would re-materialise the zeroed outputs)
            const float4 box = *reinterpret_cast<const float4*>(&fr);       // xlo xhi ylo yhi
            const bool inb = !(qx > box.y) & !(qx < box.x) & !(qy > box.w) & !(qy < box.z);
            if (tune::count_paths_bwd) {
                const bool on = act & inb, fast = face_safe(fr.meta) && p.consts_safe;
                const bool ins = on && DIST == 2 && strictly_inside(barycentric(fr, qx, qy));
                pcnt[3]++; pcnt[4] += (unsigned)__builtin_popcountll(ballot(on));
                const unsigned long long bi = ballot(ins), bs = ballot(on && !fast);
                pcnt[5] += bi != 0ull; pcnt[6] += (unsigned)__builtin_popcountll(bi);
                pcnt[7] += bs != 0ull; pcnt[8] += (unsigned)__builtin_popcountll(bs);
            }
            if (act & inb) {
                float gv[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // x0 y0 z0 x1 y1 z1 x2 y2 z2
                float tgs;
                bool tex_on;
                const float* vc = s_vcol + jc * 9;
                const float* tb = TEXLDS ? s_tex + jc * (p.T * 3) : nullptr;
                const int texel = (face_safe(fr.meta) && p.consts_safe)
                    ? backward_pair<DIST, RGB, true>(p, fr, vc, tb, q, qx, qy, tbase, gv, wcw, tgs, tex_on, counters)
                    : backward_pair<DIST, RGB, false>(p, fr, vc, tb, q, qx, qy, tbase, gv, wcw, tgs, tex_on, counters);
                if (tex_on && p.tex == 0 && p.T != 1) {      // per-pixel texel: into the batch's LDS sums (TEXLDS) or straight to global
                    float* gtf = TEXLDS ? s_gtex + jc * (p.T * 3) : gtbase + (size_t)face_id(fr.meta) * p.T * 3;
                    const float c0 = tgs * q.g0, c1 = tgs * q.g1, c2 = tgs * q.g2;
                    atomicAdd(gtf + texel * 3 + 0, c0);
                    atomicAdd(gtf + texel * 3 + 1, c1);
                    atomicAdd(gtf + texel * 3 + 2, c2);
                    // The reference adds (texel j sampled ? 1 : 0) * c to EVERY texel j of the face
                    // (SRK:1317-1320): an overflowed weight (inf, NaN) therefore poisons all of them with
                    // 0 * inf = NaN.  Reproduced so that the non-finite pattern of the gradients matches.
                    if (!(isfinite(c0) && isfinite(c1) && isfinite(c2))) {
                        const float qnan = __builtin_nanf("");
                        for (int jt = 0; jt < p.T; jt++) {
                            if (jt == texel) continue;
                            if (!isfinite(c0)) atomicAdd(gtf + jt * 3 + 0, qnan);
                            if (!isfinite(c1)) atomicAdd(gtf + jt * 3 + 1, qnan);
                            if (!isfinite(c2)) atomicAdd(gtf + jt * 3 + 2, qnan);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 9; k++) v[k] = gv[k];
                tw = tex_on ? tgs : 0.f;
                if (ntex == 3) { v[9] = tw * q.g0; v[10] = tw * q.g1; v[11] = tw * q.g2; }
            }
            clk.lap(4);
            const int fn = face_id(fr.meta);
            {                                                    // next trip's item
                const int item = RANGES ? item0 + trip + 1 : i0 + NG + blk;
                ract = item < item_end;
                nth = 64; hs = 0ull;
                if (ract) {
                    while (s_ioff[j + 1] <= item) j++;
                    nth = (item - s_ioff[j]) * G + li;
                    hs = s_has[j];
                }
            }
            float s;                                             // lane li holds component li summed over the row
            if (JR_TUNE_DIAG & 16) s = v[li & 15];               // (diagnostic bit 4: no reduction)
            else s = row_transpose_reduce(v, li);
            if (RANGES) {
                if (ractc) {                                     // SRK:1349-1358 does one atomic per pixel and component
                    if (fn != acc_fn) { if (!(JR_TUNE_DIAG & 32)) flush(); acc = s; acc_fn = fn; }
                    else acc += s;
                }
            } else if (!(JR_TUNE_DIAG & 32) && ractc) add_component(fn, li, s);   // (diagnostic bit 5: no atomics)
            if (ntex == 9) {                                     // vertex colours: 9 more components
                float u[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int jv = 0; jv < 3; jv++) {
                    u[3 * jv + 0] = tw * (wcw[jv] * q.g0);
                    u[3 * jv + 1] = tw * (wcw[jv] * q.g1);
                    u[3 * jv + 2] = tw * (wcw[jv] * q.g2);
                }
                float* gtf = gtbase + (size_t)fn * p.T * 3;
                const float st = row_transpose_reduce(u, li);
                if (ractc && li < 9 && st != 0.f) atomicAdd(gtf + li, st);
            }
            clk.lap(5);
        }
        if (RANGES && !(JR_TUNE_DIAG & 32)) flush();
        if (TEXLDS) {            // the batch's texel sums leave: one global atomic per touched component, 16 lanes per slot as in the staging
            bsync();
            const int t3 = p.T * 3, sub = lane >> 4, l16 = lane & 15;
            for (int s0 = 0; s0 < fill; s0 += 4) {
                const int slot = s0 + sub;
                const int id = __builtin_amdgcn_ds_bpermute((slot < fill ? slot : 0) << 2, myid);
                if (slot < fill) {
                    float* dst = gtbase + (size_t)id * t3;
                    const float* src = s_gtex + slot * t3;
                    for (int k = l16; k < t3; k += 16) {
                        const float v = src[k];
                        if (v != 0.f) atomicAdd(dst + k, v);           // (a NaN sum compares unequal to zero: the poison leaves too)
                    }
                }
            }
        }
        bsync();                        // the batch's records and tables are free again
    };

    if (!HASHED) {
        // Round 2 - 5: every lane holds its ids SORTED; the wavefront repeatedly extracts the smallest id that is still
        // pending anywhere (DPP min), the ballot of the lanes whose head equals it IS the face's holder mask, and those lanes
        // advance: slot = extraction order = ascending id.  ~63 VALU per distinct face + a 16-element bitonic sort per lane:
        // 38 % of the kernel's VALU instructions on the headline batch (round 6 count) - kept as JR_TUNE_BWD_HASH_UNION=0.
        for (;;) {
            int fill = 0, myid = 0;
            unsigned long long has = 0ull;
            while (fill < BATCH) {
                const int m = wave_min(cur);
                if (m == BIG) break;
                const bool hit = cur == m;
                const unsigned long long h = ballot(hit);
                if (lane == fill) { myid = m; has = h; }
                if (hit) {
#pragma unroll
                    for (int k = 0; k + 1 < KCAP; k++) mine[HASHED ? 0 : k] = mine[HASHED ? 0 : k + 1];
                    mine[HASHED ? 0 : KCAP - 1] = BIG;
                    cur = mine[0];
                }
                fill++;
            }
            if (fill == 0) break;
            clk.lap(1);
            run_batch(fill, myid, has);
        }
    } else {
        // Round 6: the union by HASHING.  Gradient sums do not care in which order the faces of a tile are visited, so the
        // ascending extraction order bought nothing.  Every lane inserts its <= K ids into an open-addressing table in LDS
        // (ds_cmpst on the key; the table lies over the record slots, which are free between batches) and ORs its lane bit
        // into the entry's holder mask (ds_or); a compaction pass numbers the used entries, lane j takes entry j.  ~18 VALU
        // per id plane instead of ~63 per distinct face, no sort.
        // A pass holds up to 64 distinct faces (lane j <-> face j).  A tile that needs more is cut into residue CLASSES of the
        // face id - class (level L, value v) = the ids with ((id >> split_log2) & (2^L - 1)) == v - refined only where a class
        // overflows: a depth-first walk over that binary trie, every id in exactly one leaf.
        HEntry* tab = reinterpret_cast<HEntry*>(s_dyn);                       // [HT_SIZE]
        int4* comp = reinterpret_cast<int4*>(tab + HT_SIZE);                   // [64] compacted (id, -, mask lo, mask hi)
        int cl = 0, cv = 0;
        for (;;) {
            // ---- build the class's table ----
            {
                int4 empty = make_int4(-1, 0, 0, 0);
                asm volatile("" : "+v"(empty.x), "+v"(empty.y), "+v"(empty.z), "+v"(empty.w));      // (materialised here: kept live over the class loop it spilled)
#pragma unroll
                for (int e = 0; e < HT_SIZE / 64; e++) reinterpret_cast<int4*>(tab)[e * 64 + lane] = empty;
            }
            int raw[KCAP];
            {                        // wave-uniform plane base + this lane's 32-bit pixel offset (sixteen 64-bit per-lane addresses kept over the class loop spilled)
                // (opaque to the optimiser: hoisted out of the class loop, the sixteen addresses and `k < K` predicates spilled)
                unsigned pn32 = (unsigned)pn;
                int kk = p.K;
                asm volatile("" : "+v"(pn32), "+s"(kk));
                const int32_t* plane0 = ids + (size_t)b * p.K * pp;
#pragma unroll
                for (int k = 0; k < KCAP; k++) raw[k] = (valid && k < kk) ? (plane0 + (size_t)k * pp)[pn32] : -1;
            }
            bsync();
            // Every plane's first probe is issued before any result is looked at (one LDS round trip for the tile's K planes
            // instead of K dependent ones); a lane whose first probe hit another id walks on - at most HT_PROBES entries, a
            // table that full means "too many faces for one pass" anyway.  No short-circuit ladder: the predicates are
            // bitwise, the plane's hash word doubles as its "this lane takes part" flag (< 0: not).
            // (planes in groups of 16: hh / old of a K = 64 pixel would be 128 registers)
            bool lost = false;                                        // this lane found no entry for one of its ids
            {
                const unsigned cmask = (1u << cl) - 1u;
                bool live = true;
                constexpr int PG = KCAP < 16 ? KCAP : 16;
#pragma unroll
                for (int g = 0; g < KCAP; g += PG) {
                    int hh[PG], old[PG];
#pragma unroll
                    for (int k = 0; k < PG; k++) {
                        const int id = raw[g + k];
                        live = live & (id >= 0) & (id < p.NF);            // -1 ends the list (ids outside [0, NF) too)
                        const bool take = live & ((part < 0) | ((id & smask) == part)) & ((((unsigned)id >> split_log2) & cmask) == (unsigned)cv);
                        hh[k] = take ? (int)(((unsigned)id * 2654435761u) >> (32 - HT_LOG2)) : -1;
                    }
#pragma unroll
                    for (int k = 0; k < PG; k++) {
                        old[k] = raw[g + k];
                        if (hh[k] >= 0) old[k] = atomicCAS(&tab[hh[k]].key, -1, raw[g + k]);
                    }
#pragma unroll
                    for (int k = 0; k < PG; k++) {
                        if (hh[k] >= 0) {
                            int h = hh[k];
                            if ((old[k] != -1) & (old[k] != raw[g + k])) {   // first probe hit another id (rare at <= 25 % load)
                                bool placed = false;
                                for (int pr = 0; pr < HT_PROBES && !placed; pr++) {
                                    h = (h + 1) & (HT_SIZE - 1);
                                    const int o = atomicCAS(&tab[h].key, -1, raw[g + k]);
                                    placed = (o == -1) | (o == raw[g + k]);
                                }
                                lost = lost | !placed;
                                if (!placed) h = -1;
                            }
                            if (h >= 0) atomicOr(&tab[h].m[lane >> 5], 1u << (lane & 31));
                        }
                    }
                }
            }
            bsync();
            // ---- compaction: used entries -> comp[0 .. ndist) -> lane j holds face j ----
            int ndist = 0;
#pragma unroll
            for (int e = 0; e < HT_SIZE / 64; e++) {
                const int4 ent = reinterpret_cast<const int4*>(tab)[e * 64 + lane];
                const bool used = ent.x != -1;
                const unsigned long long um = ballot(used);
                const int at = ndist + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(um >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)um, 0u));
                if (used & (at < 64)) comp[at] = ent;
                ndist += __builtin_popcountll(um);
            }
            const bool overflow = ndist > 64 || ballot(lost) != 0ull;
            bsync();
            if (tune::count_paths_bwd) pcnt[1]++;
            if (overflow) { cl++; continue; }                     // refine: (L + 1, v) is the left child; every id in exactly one leaf
            int eid = 0;
            unsigned long long ehas = 0ull;
            if (lane < ndist) { const int4 ent = comp[lane]; eid = ent.x; ehas = ((unsigned long long)(unsigned)ent.w << 32) | (unsigned)ent.z; }
            bsync();                                      // the table's LDS becomes record slots again
            clk.lap(1);
            for (int b0 = 0; b0 < ndist; b0 += BATCH) {
                const int fill = min(BATCH, ndist - b0);
                int myid = eid;
                unsigned long long has = ehas;
                if (b0 > 0) {                                     // faces b0 .. of the pass move down to lanes 0 ..
                    const int from = ((lane + b0) & 63) << 2;
                    myid = __builtin_amdgcn_ds_bpermute(from, eid);
                    has = ((unsigned long long)(unsigned)__builtin_amdgcn_ds_bpermute(from, (int)(unsigned)(ehas >> 32)) << 32) |
                          (unsigned)__builtin_amdgcn_ds_bpermute(from, (int)(unsigned)ehas);
                }
                if (lane >= fill) has = 0ull;
                run_batch(fill, myid, has);
            }
            // ---- next class: right sibling, or up ----
            while (cl > 0 && ((cv >> (cl - 1)) & 1)) { cv &= ~(1 << (cl - 1)); cl--; }
            if (cl == 0) break;
            cv |= 1 << (cl - 1);
        }
    }
    clk.lap(1);
    if (JR_TUNE_PROFILE_SECTIONS == 1) clk.flush(counters, 12);
    if (tune::count_paths_bwd && lane == 0) {
#pragma unroll
        for (int i = 0; i < 11; i++) atomicAdd(counters + 4 + i, (unsigned long long)pcnt[i]);
    }
}

template <int DIST, int RGB>
static void launch_k(hipStream_t st, const RasterParams& p, int ntiles, const float* textures,
                     const BinWorkspace& ws, const float* rgba, const float* aggrs, const int32_t* ids,
                     const float* grad_rgba, float* grad_faces, float* grad_textures) {
    const int tl = 2 * sub_log2_of(p);
    const int nbins = ntiles >> tl;
    // heavy bins' tiles by tune::bwd_split wavefronts each when the launch is too small to fill the GPU anyway
    const int heavy_cap = backward_splits_heavy_tiles(p, ws) ? heavy_bins_cap(ws, nbins) : 0;   // (bound of counters[3], as in the forward)
    const int split = heavy_cap > 0 && (long)p.B * p.IS * p.IS <= (long)tune::bwd_split8_pixels ? 8 : (tune::bwd_split > 1 ? tune::bwd_split : 1);
    const int split_log2 = split >= 8 ? 3 : (split >= 4 ? 2 : (split >= 2 ? 1 : 0));
    const int grid = (8 << tl) * ((1 << split_log2) * ((heavy_cap + 7) / 8) + (nbins + 7) / 8);   // whole bins per XCD slot
    // texture blocks in LDS: 'surface' textures of a few texels in launches that are latency-, not occupancy-bound
    const int tex_lds = (p.tex == 0 && p.T > 1 && p.T <= tune::bwd_tex_lds_max && RGB == 1 &&
                         (long)p.B * p.IS * p.IS <= (long)tune::bwd_tex_lds_pixels) ? 1 : 0;
    const int batch = (p.K <= 16 || tex_lds) ? tune::bwd_batch_for(16) : (p.K <= 32 ? tune::bwd_batch_for(32) : tune::bwd_batch_for(64));
    const size_t smem = sizeof(FaceRec) * batch + (p.tex == 1 ? sizeof(float) * 9 * batch : 0) +
                        (tex_lds ? sizeof(float) * 2 * 3 * p.T * batch : 0);      // texel colours + texel gradient sums
#define JR_BWD_K(KC, TL) \
    k_softras_backward<DIST, RGB, KC, TL><<<grid, 64, smem, st>>>( \
        p, nbins, heavy_cap, split_log2, textures, ws.geo, ws.bin_order, ws.bin_count, rgba, aggrs, ids, grad_rgba, \
        grad_faces, grad_textures, ws.counters)
    if (RGB == 1 && tex_lds) {           // (the staged-texture instantiations exist for the softmax colour path only: nothing else reads texels)
        if (p.K <= 16) JR_BWD_K(16, (RGB == 1));
        else if (p.K <= 32) JR_BWD_K(32, (RGB == 1));
        else JR_BWD_K(64, (RGB == 1));
    } else if (p.K <= 16) JR_BWD_K(16, false);
    else if (p.K <= 32) JR_BWD_K(32, false);
    else JR_BWD_K(64, false);
#undef JR_BWD_K
}

bool backward_splits_heavy_tiles(const RasterParams& p, const BinWorkspace& ws) {
    return tune::bwd_split > 1 && ws.heavy_min > 0 && (long)p.B * p.IS * p.IS <= (long)tune::bwd_split_pixels;
}

// Both gradient outputs cleared by ONE launch (they are separate caller-owned buffers: two memsets were two launches,
// which is what a single view's backward notices).  16-byte stores over the aligned body, scalar head / tail.
__global__ __launch_bounds__(256) void k_zero2(float* __restrict__ a, size_t na, float* __restrict__ b, size_t nb) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (size_t)gridDim.x * blockDim.x;
#pragma unroll
    for (int which = 0; which < 2; which++) {
        float* q = which ? b : a;
        const size_t n = which ? nb : na;
        const size_t head = min(n, (size_t)((16 - ((uintptr_t)q & 15)) & 15) >> 2);   // floats up to the first 16-byte boundary
        const size_t body = (n - head) >> 2;
        float4* q4 = reinterpret_cast<float4*>(q + head);
        for (size_t i = tid; i < body; i += nthreads) q4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < head) q[tid] = 0.f;
        const size_t tail0 = head + (body << 2);
        if (tid < n - tail0) q[tail0 + tid] = 0.f;
    }
}

void launch_softras_backward(hipStream_t st, const RasterParams& p, const float* textures,
                             const float* rgba, const float* aggrs, const int32_t* ids,
                             const float* grad_rgba, const BinWorkspace& ws, float* grad_faces,
                             float* grad_textures) {
    const int ntiles = (p.B * p.bins_x * p.bins_y) << (2 * sub_log2_of(p));
    {                                                                                           // SRK:1374-1375
        const size_t na = (size_t)p.B * p.NF * 9, nb = (size_t)p.B * p.NF * p.T * 3;
        const size_t wgs = ((na + nb) / 4 + 255) / 256;
        k_zero2<<<(unsigned)(wgs < 1 ? 1 : (wgs > 4096 ? 4096 : wgs)), 256, 0, st>>>(grad_faces, na, grad_textures, nb);
    }
#define JR_BWD(D, R) \
    launch_k<D, R>(st, p, ntiles, textures, ws, rgba, aggrs, ids, grad_rgba, grad_faces, grad_textures)
    const int rgb = p.rgb == 0 ? 0 : (p.rgb == 1 ? 1 : 2);
    switch (p.dist * 3 + rgb) {
        case 0: JR_BWD(0, 0); break;
        case 1: JR_BWD(0, 1); break;
        case 2: JR_BWD(0, 2); break;
        case 3: JR_BWD(1, 0); break;
        case 4: JR_BWD(1, 1); break;
        case 5: JR_BWD(1, 2); break;
        case 6: JR_BWD(2, 0); break;
        case 7: JR_BWD(2, 1); break;
        default: JR_BWD(2, 2); break;
    }
#undef JR_BWD
}

}  // namespace jr
