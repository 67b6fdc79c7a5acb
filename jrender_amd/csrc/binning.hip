// Screen binning for the SoftRas kernels (gfx950).
//
// The reference visits ALL faces for every pixel (SRK:311) — O(pixels x faces).  Here the screen
// is cut into bins of 1, 2x2 or 4x4 wavefront tiles (8x8 pixels each; RasterParams::bin_log2, chosen per launch).  Every bin gets the list
// of faces whose border box (triangle bbox grown by the cull radius, SRK:28-34, :316) can touch
// it, in ASCENDING face order — the per-pixel aggregation (alpha product, online softmax,
// K-nearest buffer: SRK:350-419) is order dependent and the face-index buffer must match the
// reference bit for bit.  Each list entry carries a 16-bit mask of the bin's tiles the face can
// touch, so a wavefront discards most of its bin's list with one bit test per face.
//
// Pipeline (all on the context stream; one 32-byte read-back of the totals sizes the pool):
//   k_face_setup : per face -> faces_info (SRK:176-236), packed FaceGeo record, conservative pixel
//                  rectangle, per-bin counts
//   k_bin_alloc_schedule : one workgroup -> segment bases in the pool (placement is irrelevant), totals, launch order
//   k_bin_fill   : per face -> append (id, tile mask) to each touched bin's segment (unordered)
//   k_bin_order  : per bin  -> write the segment in ascending id order.  Ids inside one view are unique
//                  and < NF, so the order is a COUNTING problem, not a comparison sort: set one bit per
//                  id in an LDS bitmap over [0, NF), prefix-popcount the bitmap, rank(id) = bits below id.
//                  Five barriers per bin instead of ~50 bitonic steps.  (Meshes above 262 144 faces do
//                  not fit the bitmap and take k_bin_sort: LDS bitonic, rank sort when a bin is huge.)
// Neighbouring faces of a mesh fall into the same few bins, so the per-bin counters are bumped once
// per (wavefront, bin) with a ballot-matched group instead of once per (face, bin).
// The rectangle is the exact set of pixel columns / rows whose centres pass the border test of the face (per
// axis); the per-pixel test is re-applied in the raster kernels anyway, so results never depend on the binning.
#include "jr_kernels.h"

namespace jr {

// EXACT pixel range [lo, hi] of the centres c(i) = pixel_centre(i) that pass the reference's border test on one
// axis, !(c(i) < vlo) && !(c(i) > vhi) (SRK:28-34): a real-valued estimate with one pixel of slack on both
// sides, then walked inwards with the kernel's own float compare (pixel_centre is monotone in i).  The
// raster kernels re-apply the per-pixel test, so a superset would be harmless; exactness is what lets the
// forward skip its own box test per (tile, face) and keeps faces that touch no pixel centre out of the lists.
// NaN bounds -> full range (the reference's compares are all false for NaN, i.e. the face is NOT culled).
__device__ inline void pixel_range(float vlo, float vhi, int is, int& lo, int& hi) {
    const double a = ((double)vlo * is + is - 1.0) * 0.5;
    const double b = ((double)vhi * is + is - 1.0) * 0.5;
    double flo = floor(a) - 1.0, fhi = ceil(b) + 1.0;
    if (!(vlo == vlo)) flo = 0.0;
    if (!(vhi == vhi)) fhi = (double)(is - 1);
    flo = fmax(flo, 0.0);
    fhi = fmin(fhi, (double)(is - 1));
    if (!(flo <= fhi)) { lo = 1; hi = 0; return; }
    lo = (int)flo;
    hi = (int)fhi;
    for (int it = 0; it < 4 && lo <= hi && pixel_centre(lo, is) < vlo; it++) lo++;
    for (int it = 0; it < 4 && lo <= hi && pixel_centre(hi, is) > vhi; it++) hi--;
}

// Wave-aggregated counter bump: lanes that target the same bin form a group (ballot match against the
// first pending lane, up to 16 rounds; when three rounds have found groups of one the rest - unrelated faces, e.g. a
// triangle soup - are taken as groups of one too), the group's first lane adds the group size, every member gets base + its rank.
// Round 6: 16 rounds instead of 4.  The counters are device-scope atomics that every XCD reaches, and they - not the 328 B of
// stores per face - were what k_face_setup and k_bin_fill waited for: 64 consecutive faces of a mesh fall into ~10 - 20 bins, four
// rounds left 40+ lanes bumping a counter each (headline batch: k_face_setup 52 -> 28 us, k_bin_fill 37 -> 14 us; a round is ~10 ALU instructions).  The matching is ALU only
// and ALL leaders add in ONE atomic instruction: one memory round trip per call (round 3; it was one per group,
// and a single view's k_bin_fill - 610 wavefronts, nothing to hide latency behind - was a chain of ~16 of them).
// -> rank of this lane inside its group, the group's first lane and its size (ALU only)
__device__ inline void wave_bin_match(int tb, int& leader, int& rank, int& cnt) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    rank = 0; leader = lane; cnt = 1;
    int key = tb;
    unsigned long long todo = ballot(tb >= 0);
    int singles = 0;                      // (wave-uniform)
    for (int round = 0; todo != 0 && round < JR_TUNE_BIN_MATCH_ROUNDS; round++) {
        const int l = __builtin_ctzll(todo);
        const int lb = __builtin_amdgcn_readlane(key, l);
        const unsigned long long same = ballot(key == lb);
        singles += (same & (same - 1ull)) == 0ull;
        if (round >= 3 && singles >= 3) break;       // unrelated faces: further rounds would find groups of one as well
        if (key == lb) {
            rank = __builtin_amdgcn_mbcnt_hi((unsigned)(same >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)same, 0u));
            leader = l;
            cnt = __builtin_popcountll(same);
            key = -1;                     // done
        }
        todo &= ~same;
    }
}
template <bool RET>
__device__ inline int wave_bin_add(int* __restrict__ arr, int tb) {
    int rank, leader, cnt;
    wave_bin_match(tb, leader, rank, cnt);
    int base = 0;
    if (tb >= 0 && rank == 0) {
        if (RET) base = atomicAdd(&arr[tb], cnt);
        else (void)atomicAdd(&arr[tb], cnt);
    }
    if (!RET) return 0;
    return __shfl(base, leader) + rank;
}

// Faces per workgroup of k_face_setup.  Round 6: ONE wavefront (11 KB of LDS: 14 per CU, as many wavefronts as before) - the two LDS
// transposes then need no workgroup barrier, and above all no __syncthreads(): its workgroup-scope fence waits for every outstanding global
// store (s_waitcnt vmcnt(0)), i.e. the kernel - 328 B of stores per face, 3 x off the memset rate - stalled at each of its three barriers
// until the stores it had just issued were acknowledged.  LDS instructions of a wavefront execute in order: a compiler fence is enough.
constexpr int SETUP_WG = JR_TUNE_SETUP_WG;
__device__ inline void setup_sync() {
    if (SETUP_WG == 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else __syncthreads();
}
__global__ __launch_bounds__(SETUP_WG) void k_face_setup(RasterParams p, const float* __restrict__ faces,
                                                    const float* __restrict__ textures,
                                                    float* __restrict__ faces_info,
                                                    FaceGeo* __restrict__ geo,
                                                    ushort4* __restrict__ face_rect,
                                                    int* __restrict__ bin_count, unsigned long long* __restrict__ counters) {
    // records leave through LDS so that the global stores are contiguous 16-byte lanes (a thread
    // writing its own 108 B / 176 B record directly touches ~60 cache lines per store instruction)
    __shared__ __align__(16) float s_out[SETUP_WG * 44];
    const int total = p.B * p.NF;
    const int i0 = blockIdx.x * SETUP_WG;
    const int i = i0 + threadIdx.x;
    const bool valid = i < total;
    const int nvalid = min(SETUP_WG, total - i0);
    const int ic = valid ? i : total - 1;
    const float* f = faces + (size_t)ic * 9;
    SectionClock clk;            // instrumented builds only (JR_TUNE_PROFILE_SECTIONS = 3): 0 face_setup (incl. the face load), 1 faces_info store, 2 record + store, 3 pixel ranges, 4 bin counts
    clk.start();
    float info[27];
    face_setup(f, info);
    if (tune::profile_sections) { asm volatile("" :: "v"(info[0]), "v"(info[8]), "v"(info[17])); clk.lap(0); }
    if (faces_info && !(JR_TUNE_DIAG & 4096)) {   // nullptr when the backward only rebuilds the lists  (diagnostic bits 12 - 15, WRONG results: what do the faces_info stores / the record stores / the bin counts / the pixel ranges cost k_face_setup?)
#pragma unroll
        for (int k = 0; k < 27; k++) s_out[threadIdx.x * 27 + k] = info[k];
        setup_sync();
        float* out = faces_info + (size_t)i0 * 27;                  // SETUP_WG*108 B per block: 16 B aligned
        const int nfl = nvalid * 27;
        for (int q = threadIdx.x; q < (nfl >> 2); q += SETUP_WG)
            reinterpret_cast<float4*>(out)[q] = reinterpret_cast<const float4*>(s_out)[q];
        if (threadIdx.x < (nfl & 3)) out[(nfl & ~3) + threadIdx.x] = s_out[(nfl & ~3) + threadIdx.x];
        setup_sync();
    }
    clk.lap(1);
    FaceGeo g;
    build_face_geo(g, f, info, p.rad, ic % p.NF);
    if (p.T == 1) {      // single-texel surface colour travels with the record
        const float* tx = textures + (size_t)ic * 3;
        g.col[0] = tx[0]; g.col[1] = tx[1]; g.col[2] = tx[2];
    }
    *reinterpret_cast<FaceGeo*>(&s_out[threadIdx.x * 44]) = g;
    setup_sync();
    if (!(JR_TUNE_DIAG & 8192)) {
        float4* out = reinterpret_cast<float4*>(geo + i0);
        for (int q = threadIdx.x; q < nvalid * 11; q += SETUP_WG) out[q] = reinterpret_cast<const float4*>(s_out)[q];
    }
    clk.lap(2);

    int px0, px1, py0, py1;
    pixel_range(g.xlo, g.xhi, p.IS, px0, px1);
    pixel_range(g.ylo, g.yhi, p.IS, py0, py1);   // in "yi" space (yi = IS-1-row)
    ushort4 rect = make_ushort4(1, 0, 1, 0);     // empty: x0 > x1
    int bx0 = 0, nbx = 0, by0 = 0, nb = 0;
    if ((JR_TUNE_DIAG & 32768)) { px0 = 1; px1 = 0; asm volatile("" :: "v"(g.xlo), "v"(g.yhi)); }
    if (valid && px0 <= px1 && py0 <= py1) {
        const int row0 = p.IS - 1 - py1, row1 = p.IS - 1 - py0;
        rect = make_ushort4((unsigned short)px0, (unsigned short)px1, (unsigned short)row0,
                            (unsigned short)row1);
        bx0 = px0 >> p.bin_log2; nbx = (px1 >> p.bin_log2) - bx0 + 1;
        by0 = row0 >> p.bin_log2; nb = nbx * ((row1 >> p.bin_log2) - by0 + 1);
    }
    if (valid) face_rect[i] = rect;
    clk.lap(3);
    const int bb = (ic / p.NF) * p.bins_x * p.bins_y;
    int cx = 0, cy = 0;
    if (JR_TUNE_DIAG & 16384) nb = 0;
    for (int it = 0; ballot(it < nb) != 0; it++) {
        const int tb = it < nb ? bb + (by0 + cy) * p.bins_x + bx0 + cx : -1;
        wave_bin_add<false>(bin_count, tb);
        if (++cx == nbx) { cx = 0; cy++; }
    }
    clk.lap(4);
    if (JR_TUNE_PROFILE_SECTIONS == 3 && (threadIdx.x & 63) == 0) clk.flush0(counters, 4);
}

template <int U>
__global__ __launch_bounds__(256) void k_bin_fill(RasterParams p, const ushort4* __restrict__ face_rect,
                                                  const int* __restrict__ bin_base,
                                                  int* __restrict__ bin_cursor,
                                                  unsigned long long* __restrict__ pool,
                                                  const unsigned long long* __restrict__ counters,
                                                  unsigned long long cap) {
    if (counters[0] > cap) return;          // pool too small: the host regrows it and launches again
    SectionClock clk;            // instrumented builds only (JR_TUNE_PROFILE_SECTIONS = 3): 0 rectangle load + set-up, 1 the append loop
    clk.start();
    const int total = p.B * p.NF;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int ic = min(i, total - 1);
    const ushort4 r = face_rect[ic];
    const int b = ic / p.NF, fn = ic - b * p.NF;
    const int bb = b * p.bins_x * p.bins_y;
    const int tx0 = r.x / TILE, tx1 = r.y / TILE, ty0 = r.z / TILE, ty1 = r.w / TILE;
    const int bx0 = r.x >> p.bin_log2, by0 = r.z >> p.bin_log2;
    const int nbx = (r.y >> p.bin_log2) - bx0 + 1;
    const int nb = (i < total && r.x <= r.y) ? nbx * ((r.w >> p.bin_log2) - by0 + 1) : 0;
    const int sl = p.sub_log2, subs = 1 << sl;             // tiles per bin side: 1, 2 or 4
    int cx = 0, cy = 0;
    if (tune::profile_sections) { asm volatile("" :: "v"(nb)); clk.lap(0); }
    // U = 4 bins of every face per pass (round 5; small launches and sub-32-pixel bins): the matching is ALU work, the four cursor bumps and the four segment-base
    // loads are independent memory operations that travel together - a pass costs one round trip instead of four (a single
    // view's launch is ~600 wavefronts with nothing to hide a chain of round trips behind; with 8-pixel bins a face reaches
    // 3 - 9 bins instead of 1 - 2).
    for (int it0 = 0; ballot(it0 < nb) != 0; it0 += U) {
        int t[U], leader[U], rank[U], base[U], seg[U];
        unsigned mask[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int bx = bx0 + cx, by = by0 + cy;
            t[u] = it0 + u < nb ? bb + by * p.bins_x + bx : -1;
            // mask of the bin's tiles overlapped by the face rectangle (bit = (ty << sub_log2) | tx)
            const int sx0 = max(tx0 - (bx << sl), 0), sx1 = min(tx1 - (bx << sl), subs - 1);
            const int sy0 = max(ty0 - (by << sl), 0), sy1 = min(ty1 - (by << sl), subs - 1);
            const unsigned rowbits = ((1u << (sx1 + 1)) - 1u) & ~((1u << sx0) - 1u);
            unsigned m = 0;
            for (int sy = sy0; sy <= sy1; sy++) m |= rowbits << (sy << sl);
            mask[u] = m;
            seg[u] = t[u] >= 0 ? bin_base[t[u]] : 0;
            if (++cx == nbx) { cx = 0; cy++; }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            int cnt;
            wave_bin_match(t[u], leader[u], rank[u], cnt);
            base[u] = 0;
            if (t[u] >= 0 && rank[u] == 0) base[u] = atomicAdd(&bin_cursor[t[u]], cnt);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int pos = __shfl(base[u], leader[u]) + rank[u];
            if (t[u] >= 0) pool[seg[u] + pos] = ((unsigned long long)(unsigned)fn << 32) | mask[u];
        }
    }
    clk.lap(1);
    if (JR_TUNE_PROFILE_SECTIONS == 3 && (threadIdx.x & 63) == 0) clk.flush0(const_cast<unsigned long long*>(counters), 12);
}

// One workgroup per bin: counting order through an LDS bitmap over the face ids (see the file header).
// src = the unordered segments, dst = the same segments in ascending id order.
constexpr int BITMAP_MAX_FACES = 258048;     // 2 x NF/32 words of dynamic LDS + the static tables stay below 64 KB
__global__ __launch_bounds__(256) void k_bin_order(int NF, const int* __restrict__ bin_count,
                                                   const int* __restrict__ bin_base,
                                                   const unsigned long long* __restrict__ src,
                                                   unsigned long long* __restrict__ dst,
                                                   const unsigned long long* __restrict__ counters,
                                                   unsigned long long cap) {
    extern __shared__ unsigned s_dyn[];
    if (counters[0] > cap) return;
    __shared__ int s_wsum[4];
    const int t = blockIdx.x;
    const int n = bin_count[t];
    if (n == 0) return;
    const int base = bin_base[t];
    if (n == 1) {
        if (threadIdx.x == 0) dst[base] = src[base];
        return;
    }
    const int words = (NF + 31) >> 5;
    unsigned* s_bits = s_dyn;
    unsigned* s_pref = s_dyn + words;
    for (int w = threadIdx.x; w < words; w += 256) s_bits[w] = 0u;
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += 256) {
        const unsigned id = (unsigned)(src[base + k] >> 32);
        atomicOr(&s_bits[id >> 5], 1u << (id & 31));
    }
    __syncthreads();
    // exclusive prefix popcount: thread owns W consecutive words
    const int W = (words + 255) >> 8;
    const int w0 = threadIdx.x * W;
    int sum = 0;
    for (int k = 0; k < W; k++)
        if (w0 + k < words) sum += __builtin_popcount(s_bits[w0 + k]);
    int incl = sum;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) run += s_wsum[w];
    for (int k = 0; k < W; k++)
        if (w0 + k < words) {
            s_pref[w0 + k] = (unsigned)run;
            run += __builtin_popcount(s_bits[w0 + k]);
        }
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += 256) {
        const unsigned long long e = src[base + k];
        const unsigned id = (unsigned)(e >> 32);
        const unsigned rank = s_pref[id >> 5] + __builtin_popcount(s_bits[id >> 5] & ((1u << (id & 31)) - 1u));
        dst[base + rank] = e;
    }
}

// Segment bases AND launch order of the bins, one workgroup (round 3: it was two kernels and a memset; the 8 192
// same-address global atomics of the per-bin allocation alone took as long as a launch).
//   bases: wave-level prefix of the counts + ONE LDS atomic per wavefront and pass (placement in the pool is irrelevant);
//   order for the raster kernels: heaviest lists first (longest-processing-time-first keeps the tail of the launch
//   short: the list length of a bin varies from 1 to > 1000 faces), empty bins last - histogram over ~12 buckets per
//   octave of the count, descending prefix, scatter.
// bin_acc = the per-bin counts k_face_setup accumulated: copied to bin_count (what every later kernel reads) and CLEARED here,
// so that the next set-up pass needs no memset in the stream (round 5: one launch less on the one-view critical path).
// counters: [0] = total pairs, [1] = non-empty bins, [2] = max bin count, [3] = heavy bins - WRITTEN here (nothing to
// clear beforehand), and copied to `host_counters` (pinned, device-visible) so that the host's read-back of the pair
// total is a wait on an event, not a copy engine's turn in the stream.
constexpr int SCHED_ITEMS = 8;      // bins per thread and chunk of k_bin_alloc_schedule (8 192 bins = the headline batch in one chunk)
__global__ __launch_bounds__(1024) void k_bin_alloc_schedule(int nbins_total, int* __restrict__ bin_acc, int* __restrict__ bin_count,
                                                             int* __restrict__ bin_base, int* __restrict__ bin_cursor,
                                                             int* __restrict__ bin_order,
                                                             unsigned long long* __restrict__ counters,
                                                             volatile unsigned long long* __restrict__ host_counters,
                                                             int heavy_bucket) {
    __shared__ int s_hist[256], s_start[256];
    __shared__ unsigned long long s_total;
    __shared__ int s_nonempty, s_max, s_heavy;
    auto bucket = [](int n) { return n <= 0 ? 0 : min(255, 1 + (int)(__log2f((float)n) * 12.f)); };
    if (threadIdx.x < 256) s_hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_total = 0ull; s_nonempty = 0; s_max = 0; s_heavy = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // Round 6: a thread owns SCHED_ITEMS bins of a chunk of 1024 x SCHED_ITEMS and loads their counts in ONE go (the loop over
    // 1024-bin slices paid one exposed global-load latency per slice and pass: 16 of them for the headline batch's 8 192 bins),
    // and a launch of up to one chunk - every shape of BASELINE.json - keeps them in registers for the second pass.
    constexpr int CH = 1024 * SCHED_ITEMS;
    const bool one_chunk = nbins_total <= CH;
    int nreg[SCHED_ITEMS];
    for (int c0 = 0; c0 < nbins_total; c0 += CH) {
#pragma unroll
        for (int i = 0; i < SCHED_ITEMS; i++) {
            const int t = c0 + i * 1024 + (int)threadIdx.x;
            nreg[i] = t < nbins_total ? bin_acc[t] : 0;
        }
#pragma unroll
        for (int i = 0; i < SCHED_ITEMS; i++) {
            const int t = c0 + i * 1024 + (int)threadIdx.x;
            if (c0 + i * 1024 >= nbins_total) break;             // (block-uniform)
            const int n = nreg[i];
            // (empty bins are the most frequent bucket by far: one LDS atomic per wavefront for them instead of one per lane)
            const unsigned long long zb = ballot(t < nbins_total && n <= 0);
            if (t < nbins_total && n > 0) atomicAdd(&s_hist[bucket(n)], 1);
            if (zb != 0ull && lane == (int)__builtin_ctzll(zb)) atomicAdd(&s_hist[0], (int)__builtin_popcountll(zb));
            // segment bases: inclusive prefix of the wavefront's counts by DPP (row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31: six adds, no LDS
            // round trip - the __shfl_up ladder of rounds 3 - 5 was six dependent ds_bpermute per slice on the launch's ONE CU) + one LDS
            // atomic per wavefront; the largest count by DPP too.  (One same-address LDS atomic per BIN instead of the scan: 15.8 -> 27.9 us.)
            int incl = n;
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
            const int wtotal = __builtin_amdgcn_readlane(incl, 63);
            const unsigned long long nzb = ballot(n > 0);
            unsigned long long wbase = 0ull;
            if (wtotal > 0) {                                    // wave-uniform
                int mx = n;
                mx = max(mx, __builtin_amdgcn_update_dpp(mx, mx, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
                mx = max(mx, __builtin_amdgcn_update_dpp(mx, mx, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
                mx = max(mx, __builtin_amdgcn_update_dpp(mx, mx, 0x141, 0xf, 0xf, false));   // row_half_mirror
                mx = max(mx, __builtin_amdgcn_update_dpp(mx, mx, 0x140, 0xf, 0xf, false));   // row_mirror
                mx = max(max(__builtin_amdgcn_readlane(mx, 0), __builtin_amdgcn_readlane(mx, 16)),
                         max(__builtin_amdgcn_readlane(mx, 32), __builtin_amdgcn_readlane(mx, 48)));
                if (lane == 0) {
                    wbase = atomicAdd(&s_total, (unsigned long long)wtotal);
                    atomicAdd(&s_nonempty, (int)__builtin_popcountll(nzb));
                    atomicMax(&s_max, mx);
                }
                wbase = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(wbase >> 32), 0) << 32) |
                        (unsigned)__builtin_amdgcn_readlane((int)(unsigned)wbase, 0);
            }
            const int base = n > 0 ? (int)(wbase + (unsigned long long)(incl - n)) : 0;
            if (t < nbins_total) {
                bin_base[t] = base;
                bin_cursor[t] = 0;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 256) {                // first rank of bucket b = number of bins in heavier buckets
        int run = 0;
        for (int b = 255; b > (int)threadIdx.x; b--) run += s_hist[b];
        s_start[threadIdx.x] = run;
        // the bins of buckets >= heavy_bucket are the first `run + own` ranks of the order
        if ((int)threadIdx.x == min(heavy_bucket, 255)) s_heavy = heavy_bucket > 255 ? 0 : run + s_hist[threadIdx.x];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long c[4] = {s_total, (unsigned long long)s_nonempty, (unsigned long long)s_max, (unsigned long long)s_heavy};
#pragma unroll
        for (int k = 0; k < 4; k++) { counters[k] = c[k]; host_counters[k] = c[k]; }
        __threadfence_system();
    }
    for (int c0 = 0; c0 < nbins_total; c0 += CH) {
        if (!one_chunk) {
#pragma unroll
            for (int i = 0; i < SCHED_ITEMS; i++) {
                const int t = c0 + i * 1024 + (int)threadIdx.x;
                nreg[i] = t < nbins_total ? bin_acc[t] : 1;
            }
        }
#pragma unroll
        for (int i = 0; i < SCHED_ITEMS; i++) {
            const int t = c0 + i * 1024 + (int)threadIdx.x;
            if (c0 + i * 1024 >= nbins_total) break;
            const int n = t < nbins_total ? nreg[i] : 1;
            if (t < nbins_total) { bin_count[t] = n; bin_acc[t] = 0; }     // what the other kernels read / the accumulator left clear for the next set-up pass (no memset in the stream)
            const bool empty = t < nbins_total && n <= 0;
            const unsigned long long zb = ballot(empty);
            int zbase = 0;
            if (zb != 0ull) {
                const int leader = (int)__builtin_ctzll(zb);
                if (lane == leader) zbase = atomicAdd(&s_start[0], (int)__builtin_popcountll(zb));
                zbase = __builtin_amdgcn_readlane(zbase, leader);
            }
            if (empty) bin_order[zbase + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(zb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)zb, 0u))] = t;
            else if (t < nbins_total) bin_order[atomicAdd(&s_start[bucket(n)], 1)] = t;
        }
    }
}

constexpr int SORT_LDS = 4096;   // 64-bit entries sortable in LDS by one workgroup (32 KB)

// One workgroup per bin.  Ascending sort of the bin's entries by face id (unique keys).
__global__ __launch_bounds__(256) void k_bin_sort(const int* __restrict__ bin_count,
                                                  const int* __restrict__ bin_base,
                                                  unsigned long long* __restrict__ pool,
                                                  unsigned long long* __restrict__ scratch,
                                                  const unsigned long long* __restrict__ counters,
                                                  unsigned long long cap) {
    __shared__ unsigned long long s[SORT_LDS];
    if (counters[0] > cap) return;
    const int t = blockIdx.x;
    const int n = bin_count[t];
    if (n <= 1) return;
    unsigned long long* seg = pool + bin_base[t];
    if (n <= SORT_LDS) {
        int m = 2;
        while (m < n) m <<= 1;
        for (int i = threadIdx.x; i < m; i += blockDim.x) s[i] = i < n ? seg[i] : ~0ull;
        __syncthreads();
        for (int k = 2; k <= m; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = threadIdx.x; i < m; i += blockDim.x) {
                    const int l = i ^ j;
                    if (l > i) {
                        const unsigned long long a = s[i], b = s[l];
                        const bool up = (i & k) == 0;
                        if ((a > b) == up) { s[i] = b; s[l] = a; }
                    }
                }
                __syncthreads();
            }
        for (int i = threadIdx.x; i < n; i += blockDim.x) seg[i] = s[i];
    } else {
        // Huge segment (a whole mesh inside one bin): rank sort through the scratch copy, staged
        // through LDS in blocks.  O(n^2/256) per bin — correctness path for degenerate inputs.
        unsigned long long* src = scratch + bin_base[t];
        for (int i = threadIdx.x; i < n; i += blockDim.x) src[i] = seg[i];
        __syncthreads();
        for (int i0 = 0; i0 < n; i0 += blockDim.x) {
            const int i = i0 + threadIdx.x;
            const unsigned long long key = i < n ? src[i] : 0;
            int rank = 0;
            for (int c0 = 0; c0 < n; c0 += SORT_LDS) {
                const int cn = min(SORT_LDS, n - c0);
                __syncthreads();
                for (int c = threadIdx.x; c < cn; c += blockDim.x) s[c] = src[c0 + c];
                __syncthreads();
                if (i < n)
                    for (int c = 0; c < cn; c++) rank += s[c] < key;
            }
            if (i < n) seg[rank] = key;
        }
    }
}

void launch_binning(hipStream_t st, const RasterParams& p, const float* faces, const float* textures,
                    float* faces_info, BinWorkspace& ws) {
    const int nfaces = p.B * p.NF;
    const int nbins = p.B * p.bins_x * p.bins_y;
    // (ws.bin_acc is all zero between set-up passes: cleared when allocated, and by k_bin_alloc_schedule after it has read it)
    k_face_setup<<<(nfaces + SETUP_WG - 1) / SETUP_WG, SETUP_WG, 0, st>>>(p, faces, textures, faces_info, ws.geo, ws.face_rect, ws.bin_acc, ws.counters);
    k_bin_alloc_schedule<<<1, 1024, 0, st>>>(nbins, ws.bin_acc, ws.bin_count, ws.bin_base, ws.bin_cursor, ws.bin_order, ws.counters, ws.host_counters, heavy_bucket(ws.heavy_min));
}

// Every kernel here is guarded by "total pairs <= pool capacity" read from device memory, so that the
// host can enqueue them BEFORE it knows the total (no pipeline bubble); see build_bins in jr_api.cpp.
void launch_bin_fill_sort(hipStream_t st, const RasterParams& p, BinWorkspace& ws, bool reset_cursors) {
    const int nfaces = p.B * p.NF;
    const int nbins = p.B * p.bins_x * p.bins_y;
    const unsigned long long cap = ws.pool_cap;
    // k_bin_alloc_schedule left the cursors at zero; only a second attempt (after the pool grew) has to clear them
    if (reset_cursors) (void)hipMemsetAsync(ws.bin_cursor, 0, sizeof(int) * (size_t)nbins, st);
    // Four bins per pass pay where a face reaches many bins (8/16-pixel bins) or the launch is too small to hide a chain of
    // round trips (one view); on the headline batch (312k faces, 32-pixel bins, 1 - 2 bins per face, 4 900 wavefronts that
    // hide each other's latency) the extra slots are pure bookkeeping: 35 -> 54 us measured, so that shape keeps one.
    const bool unroll4 = bin_log2_of(p) < 5 || nfaces <= tune::bin_fill_unroll_max_faces;
    (unroll4 ? k_bin_fill<4> : k_bin_fill<1>)<<<(nfaces + 255) / 256, 256, 0, st>>>(p, ws.face_rect, ws.bin_base, ws.bin_cursor, ws.pool_scratch,
                                                     ws.counters, cap);
    if (p.NF <= BITMAP_MAX_FACES) {
        const size_t lds = sizeof(unsigned) * 2 * (size_t)((p.NF + 31) >> 5);
        k_bin_order<<<nbins, 256, lds, st>>>(p.NF, ws.bin_count, ws.bin_base, ws.pool_scratch, ws.pool, ws.counters, cap);
    } else {
        k_bin_sort<<<nbins, 256, 0, st>>>(ws.bin_count, ws.bin_base, ws.pool_scratch, ws.pool, ws.counters, cap);
        (void)hipMemcpyAsync(ws.pool, ws.pool_scratch, sizeof(unsigned long long) * ws.pool_cap, hipMemcpyDeviceToDevice, st);
    }
}

}  // namespace jr
