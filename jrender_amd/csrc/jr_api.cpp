// C ABI of libjrender_hip.so — see include/jrender_hip.h for the contract and the reference
// interfaces (file:line) each entry point replaces.  Host-side only: argument validation, the
// per-context scratch arena for the bin lists, kernel launches on the context stream.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <new>
#include <unordered_map>
#include <vector>

#include "../../include/jrender_hip.h"
#include "jr_kernels.h"

namespace {

thread_local char g_err[512] = "";

int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

#define JR_HIP(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

}  // namespace

// used by the other host translation units of the library (jr_comm.cpp); not part of the C ABI
extern "C" void jr_set_error_(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg ? msg : ""); }

struct jr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    jr::BinWorkspace ws;
    unsigned long long* h_counters = nullptr;   // pinned, 4 entries
    hipEvent_t ev_counters = nullptr;           // marks the read-back of the pair total
    // Generation of the face records / launch order currently held in ws.  Every set-up pass bumps it;
    // jr_softras_forward hands the value out as a token and jr_softras_backward_ex reuses the records
    // only when the caller presents the CURRENT generation (and the same shapes).  Pointer identity is
    // deliberately not part of the test: allocators recycle addresses.
    uint64_t geo_epoch = 0;
    uint64_t last_forward_token = 0;
    int bins_T = 0;
    int bins_B = 0, bins_NF = 0, bins_IS = 0, bins_bin_log2 = 0;
    // Bin geometry and heavy-bin threshold as the caller set them (jr_softras_set_bin_size / _set_launch_policy); every
    // set-up pass resolves them for ITS launch: bin size 0 = by image size, threshold < 0 = the default of that bin size.
    int bin_size_user = 0;
    int heavy_min_user = -1;
    int precise_colour = 0;                  // jr_softras_set_precise_colour: the forward's colour path in the reference's own arithmetic
    float bins_rad = 0.f;
    int64_t stats[4] = {0, 0, 0, 0};
    int64_t launch_info[4] = {0, 0, 0, 0};  // last forward: multi-wavefront kernel used, heavy bins, wavefronts per workgroup
    // What earlier forwards found, per shape (a small LRU): the number of heavy bins is known on the device only, and the
    // next forward of the same shape sizes its multi-wavefront launch (workgroup size, heavy-tile workgroups) from it while
    // the schedule kernel is still running.  A shape that is NOT in here launches its raster kernel after the host has read
    // this forward's own totals, so that the first call of a shape takes the same path as the second.
    struct ShapeHist { int B, NF, IS, heavy_min, bin_log2; int64_t heavy, pairs; uint64_t stamp; };
    ShapeHist hist[8] = {};
    uint64_t hist_clock = 0;
    int forced_waves = 0;                    // jr_softras_set_launch_policy / JR_FWD_HEAVY_WAVES: 4 or 8 whatever the policy says; 0 = automatic
    unsigned long long* zkey = nullptr;     // n3mr z-buffer keys [B*IS*IS]
    size_t zkey_cap = 0;
    int capturing = 0;                      // between jr_graph_begin and jr_graph_end: launches are recorded, nothing may wait for the GPU or allocate
    size_t zkey_clean = 0;                  // leading entries of zkey known to hold ~0 (k_n3mr_resolve clears what it read)
    unsigned char* n3_scratch = nullptr;    // n3mr backward: packed per-pixel planes in both orientations
    size_t n3_scratch_cap = 0;
    // two-stage sums of the loss / optimiser kernels: [red_cap] double accumulators + [red_cap] tickets, zeroed when
    // allocated and left zeroed by every launch (the last workgroup clears what it publishes)
    double* red_acc = nullptr;
    unsigned* red_ticket = nullptr;
    size_t red_cap = 0;
    bool red_dirty = false;                 // a launch that uses the scratch was issued and not yet known to have been accepted: its tickets may be left non-zero
    // optional per-phase HIP-event timing (jr_profile_*): pairs of events bracketing each phase
    bool prof_on = false;
    std::vector<hipEvent_t> prof_events;     // pool, reused after every collect
    std::vector<int> prof_phase;             // phase id of pair i (events 2i, 2i+1)
    size_t prof_used = 0;                    // events handed out since the last collect
    // Stream-ordered caching allocator behind jr_malloc/jr_free: hipMalloc/hipFree of the
    // 100+ MB image tensors cost milliseconds and synchronise the device, so freed blocks are
    // kept per size class and handed out again.  Safe because every consumer of this context
    // runs on ctx->stream: a block reused by a later call is only touched after all earlier
    // work that used it.  jr_ctx_trim() returns the cache to the driver.
    std::unordered_map<size_t, std::vector<void*>> cache;   // rounded size -> free blocks
    std::unordered_map<void*, size_t> live;                 // block -> rounded size
    size_t cached_bytes = 0;
    // HIP graphs (jr_graph_*) bake device addresses into their nodes.  Every block the allocator hands out or takes back
    // WHILE a capture is open is pinned to that graph: when its owner frees it, it is parked instead of going back to the
    // cache (so no later jr_malloc can be handed memory a replay still reads and writes, and jr_ctx_trim / the out-of-memory
    // retry never hipFree() it) until jr_graph_destroy.  Blocks the captured sequence uses but that were allocated before
    // the capture and are still held by the caller are the caller's to keep alive (Graph.keep()).
    std::unordered_map<void*, int> pinned;                  // block -> 1 while some live graph may address it
    std::unordered_map<void*, size_t> parked;               // freed pinned blocks (rounded size): not in the cache
    std::vector<void*> capture_blocks;                      // the open capture's blocks
    std::unordered_map<size_t, std::vector<void*>> capture_free;   // ... those its sequence has freed again: reusable INSIDE the same capture
                                                            // (a graph replays its nodes in the captured order, as the stream would), parked when it ends
    std::unordered_map<void*, std::vector<void*>> graph_blocks;   // graph exec -> its pinned blocks
    // The library's own scratch (bin arrays, pool, reduction scratch, NMR keys / planes) is addressed by captured kernels too:
    // every reallocation bumps this generation, a graph remembers the one it was captured under, jr_graph_launch refuses older ones.
    uint64_t ws_generation = 0;
    std::unordered_map<void*, uint64_t> graph_generation;
};

namespace {
struct ProfScope {
    jr_ctx* c;
    hipEvent_t stop = nullptr;
    ProfScope(jr_ctx* ctx, int phase) : c(ctx) {
        if (!c->prof_on) return;
        while (c->prof_events.size() < c->prof_used + 2) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            c->prof_events.push_back(e);
        }
        hipEvent_t start = c->prof_events[c->prof_used];
        stop = c->prof_events[c->prof_used + 1];
        c->prof_used += 2;
        c->prof_phase.push_back(phase);
        (void)hipEventRecord(start, c->stream);
    }
    ~ProfScope() {
        if (stop) (void)hipEventRecord(stop, c->stream);
    }
};
}  // namespace

namespace {

// (re)allocation of library scratch: never inside a graph capture (hipMalloc / hipFree / a stream wait would invalidate the
// capture with an opaque HIP error), and every one outdates the graphs captured before it
int scratch_realloc_allowed(jr_ctx* ctx, const char* what) {
    if (ctx->capturing)
        return fail("%s has to grow during a graph capture: run the sequence once outside the graph (the scratch is sized by the "
                    "largest call the context has seen) and capture again", what);
    ctx->ws_generation++;
    return 0;
}

template <typename T>
int grow(jr_ctx* ctx, const char* what, T*& ptr, size_t& cap, size_t need, double slack) {
    if (need <= cap && ptr) return 0;
    if (scratch_realloc_allowed(ctx, what)) return 1;
    if (ptr) JR_HIP(hipFree(ptr));
    ptr = nullptr;
    const size_t ncap = (size_t)(need * slack) + 1024;
    JR_HIP(hipMalloc((void**)&ptr, sizeof(T) * ncap));
    cap = ncap;
    return 0;
}

int ensure_reduction_scratch(jr_ctx* ctx, size_t n) {
    n += 4;                                  // [0..3]: the optimiser kernels' three sums, [4..): one per mesh of a loss launch
    if (n <= ctx->red_cap) return 0;
    if (scratch_realloc_allowed(ctx, "the reduction scratch of the loss / optimiser kernels")) return 1;
    JR_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->red_acc) JR_HIP(hipFree(ctx->red_acc));
    if (ctx->red_ticket) JR_HIP(hipFree(ctx->red_ticket));
    ctx->red_acc = nullptr; ctx->red_ticket = nullptr; ctx->red_cap = 0;
    const size_t cap = n + 60;
    JR_HIP(hipMalloc((void**)&ctx->red_acc, sizeof(double) * cap));
    JR_HIP(hipMalloc((void**)&ctx->red_ticket, sizeof(unsigned) * cap));
    JR_HIP(hipMemsetAsync(ctx->red_acc, 0, sizeof(double) * cap, ctx->stream));
    JR_HIP(hipMemsetAsync(ctx->red_ticket, 0, sizeof(unsigned) * cap, ctx->stream));
    ctx->red_cap = cap;
    return 0;
}

int validate(int B, int NF, int T, int IS, int K, int dist, int rgb, int alpha, int tex) {
    if (B < 1 || NF < 1 || T < 1 || IS < 1) return fail("B, NF, T, image_size must be >= 1 (got %d %d %d %d)", B, NF, T, IS);
    if (IS > jr::MAX_IMAGE) return fail("image_size %d exceeds the supported maximum %d", IS, jr::MAX_IMAGE);
    if (NF > jr::MAX_FACES_PER_IMAGE) return fail("NF = %d faces per image exceed the %d the face records index", NF, jr::MAX_FACES_PER_IMAGE);
    if ((long long)B * NF > 0x7fffffffLL) return fail("B * NF = %lld faces exceed the 2^31 - 1 the kernels index", (long long)B * NF);
    if (K < 1 || K > JR_MAX_FACES_PER_PIXEL)
        return fail("max_faces_per_pixel_for_grad must be in [1,%d] (got %d)", JR_MAX_FACES_PER_PIXEL, K);
    if (dist < 0 || dist > 2) return fail("func_id_dist must be 0 (hard), 1 (barycentric) or 2 (euclidean)");
    if (rgb < 0 || rgb > 2) return fail("func_id_rgb must be 0 (hard), 1 (softmax) or 2 (none)");
    if (alpha < 0 || alpha > 2) return fail("func_id_alpha must be 0 (hard), 1 (sum) or 2 (prod)");
    if (tex < 0 || tex > 1) return fail("texture_sample_type must be 0 (surface) or 1 (vertex)");
    if (tex == 1 && T < 3) return fail("texture_type 'vertex' needs 3 colours per face (T=%d)", T);
    if ((long)NF >= (1L << 24) && rgb == 0)
        return fail("aggr_func_rgb='hard' stores the face index as float: NF must be < 2^24");
    return 0;
}

// Bin size of a launch: the caller's (jr_softras_set_bin_size: the reference's `bin_size` kwarg, soft_rasterize.py:85-99)
// or, by default, one that follows the image - a 32-pixel bin is HALF of a 64^2 image (demo2: every bin lists a third of
// the mesh and every tile walks that list), while on a 1024^2 image it keeps the lists, the ordering kernel and the
// launch order cheap.  Results do not depend on it.
int resolve_bin_log2(const jr_ctx* ctx, int B, int IS, int NF) {
    const int user = ctx->bin_size_user;
    if (user > 0) return user <= 8 ? 3 : (user <= 16 ? 4 : 5);
    // measured (profiles/r05_experiments.md, calls 1 and 4): 64^2 x 64 views 0.48 -> 0.35 ms with 8-pixel bins (the list IS the
    // tile's, half of the tiles are limb-heavy), 256^2 - 12 % with 16; at 1024^2 16-pixel bins pay (- 2 ... - 7 %) while the
    // launch fits the multi-wavefront kernel, the headline batch keeps 32 (+ 2 % otherwise: twice the list entries to order)
    if (IS <= jr::tune::auto_bin8_max_image) return 3;
    // a mesh that is DENSE for its image (39 000 faces at 256^2: 150 per 16-pixel bin on average, lists of up to 1 900) gains
    // nothing from finer bins - every bin is heavy either way - and pays their set-up: 8 views 0.86 ms with 16, 0.70 with 32 (call 6)
    if (IS <= jr::tune::auto_bin16_max_image)
        return (long)NF * 256 > (long)jr::tune::auto_dense_faces_per_bin16 * IS * IS ? 5 : 4;
    return (long)B * IS * IS <= (long)jr::tune::fwd_heavy_pixels ? 4 : 5;
}
// Bins that list more faces than this are HEAVY (a workgroup per tile in the forward, split tiles in the backward): the
// caller's value, or the default of the bin size (a smaller bin lists fewer faces for the same load per tile; with 16-pixel
// bins small meshes - the spot cow's 5 856 faces, a 3 300-face sphere - want the pipeline from 64 listed faces on, the
// 39 000-face sphere from 128: below that too many of its tiles queue for workgroups).
int resolve_heavy_min(const jr_ctx* ctx, int bin_log2, int NF) {
    if (ctx->heavy_min_user >= 0) return ctx->heavy_min_user;
    if (bin_log2 >= 5) return jr::tune::fwd_heavy;
    if (bin_log2 == 4) return NF <= jr::tune::small_mesh_faces ? jr::tune::fwd_heavy16_small_mesh : jr::tune::fwd_heavy16;
    return jr::tune::fwd_heavy8;
}

jr::RasterParams make_params(const jr_ctx* ctx, int B, int NF, int T, int IS, int K, float near_, float far_, float eps,
                             float sigma, int dist, float dist_eps, float gamma, int rgb, int alpha,
                             int tex, int double_side, const float* bg) {
    jr::RasterParams p;
    p.B = B; p.NF = NF; p.T = T; p.R = (int)std::sqrt((double)T); p.IS = IS; p.K = K;      // SRK:475
    p.near_ = near_; p.far_ = far_; p.eps = eps; p.sigma = sigma; p.dist_eps = dist_eps; p.gamma = gamma;
    p.thr = dist_eps * sigma;                                                              // SRK:289 (float multiply)
    p.rad = sqrtf(p.thr);                                                                  // SRK:316
    p.dist = dist; p.rgb = rgb; p.alpha = alpha; p.tex = tex; p.double_side = double_side ? 1 : 0;
    for (int k = 0; k < 3; k++) p.bg[k] = bg ? bg[k] : 0.f;
    p.bin_log2 = resolve_bin_log2(ctx, B, IS, NF);
    p.sub_log2 = p.bin_log2 - jr::TILE_LOG2;
    p.bins_x = (IS + (1 << p.bin_log2) - 1) >> p.bin_log2;
    p.bins_y = p.bins_x;
    // correctly rounded reciprocals of the per-call divisors (float IEEE divisions on the host)
    p.far_minus_near = far_ - near_;
    p.near_minus_far = near_ - far_;
    p.r_sigma = 1.0f / sigma;
    p.r_gamma = 1.0f / gamma;
    p.r_far_minus_near = 1.0f / p.far_minus_near;
    p.r_near_minus_far = 1.0f / p.near_minus_far;
    p.rs_log2e = (float)(1.4426950408889634 / (double)sigma);
    p.rg_log2e = (float)(1.4426950408889634 / (double)gamma);
    auto in_range = [](float v) { const float a = std::fabs(v); return a >= 9.094947017729282e-13f && a <= 1.099511627776e12f; };
    p.consts_safe = in_range(sigma) && in_range(gamma) && in_range(p.far_minus_near) && in_range(near_) &&
                    in_range(far_) && (eps == 0.f || in_range(eps));
    return p;
}

// Per-face records, bin counts and launch order for this geometry (all the backward needs).
// The two-stage sums (loss_kernels.hip, optim_kernels.hip) assume their accumulators and tickets are zero when a launch starts and leave them
// zero when it ends.  A launch the runtime refused (or whose status was never read because an error returned early) may not have: the next
// user of the scratch clears it first instead of publishing early / never.
int begin_reduction(jr_ctx* ctx, size_t n) {
    if (ensure_reduction_scratch(ctx, n)) return 1;
    if (ctx->red_dirty) {
        JR_HIP(hipMemsetAsync(ctx->red_acc, 0, sizeof(double) * ctx->red_cap, ctx->stream));
        JR_HIP(hipMemsetAsync(ctx->red_ticket, 0, sizeof(unsigned) * ctx->red_cap, ctx->stream));
    }
    ctx->red_dirty = true;
    return 0;
}
int end_reduction(jr_ctx* ctx) {
    JR_HIP(hipGetLastError());
    ctx->red_dirty = false;
    return 0;
}

int setup_faces(jr_ctx* ctx, const jr::RasterParams& p, const float* faces, const float* textures,
                float* faces_info) {
    const size_t nfaces = (size_t)p.B * p.NF, nbins = (size_t)p.B * p.bins_x * p.bins_y;
    jr::BinWorkspace& ws = ctx->ws;
    if (nfaces > ws.faces_cap || !ws.geo) {
        size_t c0 = ws.faces_cap, c1 = ws.faces_cap;
        if (grow(ctx, "the face-record array", ws.geo, c0, nfaces, 1.0)) return 1;
        if (grow(ctx, "the face-rectangle array", ws.face_rect, c1, nfaces, 1.0)) return 1;
        ws.faces_cap = c0;
    }
    if (nbins > ws.bins_cap || !ws.bin_count) {
        size_t c0 = ws.bins_cap, c1 = ws.bins_cap, c2 = ws.bins_cap, c3 = ws.bins_cap, c4 = ws.bins_cap;
        if (ctx->capturing) return fail("the bin arrays have to grow during a graph capture: run the sequence once outside the graph and capture again");
        JR_HIP(hipStreamSynchronize(ctx->stream));            // (a set-up pass in flight may still clear the old accumulator)
        if (grow(ctx, "the bin arrays", ws.bin_acc, c4, nbins, 1.0)) return 1;
        JR_HIP(hipMemsetAsync(ws.bin_acc, 0, sizeof(int) * c4, ctx->stream));
        if (grow(ctx, "the bin arrays", ws.bin_count, c0, nbins, 1.0)) return 1;
        if (grow(ctx, "the bin arrays", ws.bin_base, c1, nbins, 1.0)) return 1;
        if (grow(ctx, "the bin arrays", ws.bin_cursor, c2, nbins, 1.0)) return 1;
        if (grow(ctx, "the bin arrays", ws.bin_order, c3, nbins, 1.0)) return 1;
        ws.bins_cap = c0;
    }
    {
        ProfScope ps(ctx, JR_PHASE_BIN_COUNT);
        jr::launch_binning(ctx->stream, p, faces, textures, faces_info, ws);
    }
    ctx->bins_B = p.B; ctx->bins_NF = p.NF; ctx->bins_IS = p.IS; ctx->bins_bin_log2 = p.bin_log2;
    ctx->bins_rad = p.rad; ctx->bins_T = p.T;
    ctx->geo_epoch++;
    return 0;
}

// Forward = setup + per-bin ascending lists + raster.  The pool that holds the lists must fit the total
// number of (bin, face) pairs, which only the device knows after k_bin_alloc_schedule.  Waiting for that number
// before enqueueing the rest costs a host round trip with an idle GPU on every call, so the rest is
// enqueued SPECULATIVELY against the pool we already have (every kernel re-checks "pairs <= capacity" in
// device memory and does nothing otherwise); the host then waits for the 32-byte read-back only, and in
// the rare case that the pool was too small (first call, bigger scene) grows it and enqueues again.
int forward_pipeline(jr_ctx* ctx, const jr::RasterParams& p, const float* faces, const float* textures,
                     float* faces_info, float* aggrs_info, float* soft_colors, int32_t* faces_id_buffer) {
    jr::BinWorkspace& ws = ctx->ws;
    ws.heavy_min = resolve_heavy_min(ctx, p.bin_log2, p.NF);
    const bool heavy_path = jr::forward_uses_heavy_path(p, ws);
    // Workgroup size of the multi-wavefront kernel.  Eight wavefronts per heavy tile cut a lone view's critical path
    // further (one 39k-face view: 0.33 -> 0.28 ms), but they and the eight light tiles per workgroup cost throughput as
    // soon as the launch can fill the GPU (two views: 0.33 -> 0.37 ms): eight when the heavy tiles' wavefronts fit a
    // fraction of the GPU, else four.  How many heavy tiles a launch has is known on the device only: a shape seen before
    // is launched speculatively with what it found then (an optimisation loop renders the same scene again and again), a
    // new shape waits for this forward's own count (the host reads the totals anyway).
    auto waves_for = [&](int64_t heavy_bins, int64_t pairs) {
        if (!(jr::tune::fwd_heavy_pipe && jr::tune::fwd_heavy_waves == 8)) return 4;
        if (ctx->forced_waves == 4 || ctx->forced_waves == 8) return ctx->forced_waves;
        if (heavy_bins <= 0) return 4;            // nothing for the pipeline: four light tiles per workgroup beat eight (3 300 faces at 1024^2: 0.247 against 0.256 ms)
        // eight while the heavy tiles' wavefronts fit a budget: generous for launches that cannot fill the GPU anyway (one 39k
        // view: 1 152 tiles of 16-pixel bins at eight wavefronts 0.489 ms, at four 0.571), tight for the others (four views)
        // a launch whose bins ALL carry long lists (39 000 faces at 256^2: 740 per bin on average) is bounded by its heavy tiles whatever
        // its size: eight wavefronts per tile 0.70 ms against 0.81 with four (8 views, call 6); a batch of 64^2 views has as many
        // heavy bins, but of 90 faces each, and loses 13 % with eight
        if (pairs >= (int64_t)jr::tune::fwd_waves8_mean_list * p.B * p.bins_x * p.bins_y) return 8;
        const int64_t tiles = heavy_bins << (2 * p.sub_log2);
        const long budget = (long)p.B * p.IS * p.IS <= jr::tune::fwd_waves8_small_pixels ? jr::tune::fwd_heavy_waves8_budget_small
                                                                                          : jr::tune::fwd_heavy_waves8_budget;
        return tiles * 8 <= budget ? 8 : 4;
    };
    jr_ctx::ShapeHist* hist = nullptr;
    for (auto& h : ctx->hist)
        if (h.stamp && h.B == p.B && h.NF == p.NF && h.IS == p.IS && h.heavy_min == ws.heavy_min && h.bin_log2 == p.bin_log2) hist = &h;
    if (setup_faces(ctx, p, faces, textures, faces_info)) return 1;
    if (ctx->capturing) {
        // Recorded into a HIP graph (jr_graph_begin): nothing here may wait for the GPU.  The launches are the speculative ones of the
        // normal path - lists against the pool we have, raster with what this shape found last time - and the kernels re-check the
        // pair total on the device at every replay; jr_graph_check compares it with the pool afterwards.
        if (!ws.pool || !ws.pool_cap || (heavy_path && !hist))
            return fail("graph capture of a forward needs one earlier forward of the same shape on this context (pool and launch history)");
        ProfScope ps(ctx, JR_PHASE_BIN_FILL_SORT);
        jr::launch_bin_fill_sort(ctx->stream, p, ws, false);
        const int64_t hv = hist ? hist->heavy : 0, pr = hist ? hist->pairs : 0;
        {
            // (waves_for / enqueue_raster are defined below for the normal path; the same decisions, inline)
            int waves = 4;
            if (jr::tune::fwd_heavy_pipe && jr::tune::fwd_heavy_waves == 8) {
                if (ctx->forced_waves == 4 || ctx->forced_waves == 8) waves = ctx->forced_waves;
                else if (hv > 0) {
                    const int64_t tiles = hv << (2 * p.sub_log2);
                    const long budget = (long)p.B * p.IS * p.IS <= jr::tune::fwd_waves8_small_pixels ? jr::tune::fwd_heavy_waves8_budget_small
                                                                                                      : jr::tune::fwd_heavy_waves8_budget;
                    waves = (pr >= (int64_t)jr::tune::fwd_waves8_mean_list * p.B * p.bins_x * p.bins_y || tiles * 8 <= budget) ? 8 : 4;
                }
            }
            ws.heavy_waves = waves;
            ws.heavy_bound = (long)(hv + hv / 4 + 16);
            if (ctx->precise_colour) jr_precise::launch_softras_forward(ctx->stream, p, textures, ws, aggrs_info, soft_colors, faces_id_buffer);
            else jr::launch_softras_forward(ctx->stream, p, textures, ws, aggrs_info, soft_colors, faces_id_buffer);
        }
        ctx->launch_info[0] = heavy_path ? 1 : 0;
        ctx->launch_info[2] = ws.heavy_waves_used;
        ctx->launch_info[3] = ws.heavy_min;
        JR_HIP(hipGetLastError());
        return 0;
    }
    JR_HIP(hipEventRecord(ctx->ev_counters, ctx->stream));     // k_bin_alloc_schedule has written the totals to h_counters
    auto enqueue_lists = [&](bool again) {
        ProfScope ps(ctx, JR_PHASE_BIN_FILL_SORT);
        jr::launch_bin_fill_sort(ctx->stream, p, ws, again);
    };
    auto enqueue_raster = [&](int64_t heavy_bins, int64_t pairs, bool exact) {
        ws.heavy_waves = waves_for(heavy_bins, pairs);
        ws.heavy_bound = exact ? (long)heavy_bins : (long)(heavy_bins + heavy_bins / 4 + 16);
        ProfScope ps(ctx, JR_PHASE_FWD_RASTER);
        if (ctx->precise_colour) jr_precise::launch_softras_forward(ctx->stream, p, textures, ws, aggrs_info, soft_colors, faces_id_buffer);
        else jr::launch_softras_forward(ctx->stream, p, textures, ws, aggrs_info, soft_colors, faces_id_buffer);
    };
    const bool spec_lists = ws.pool != nullptr && ws.pool_cap > 0;
    const bool spec_raster = spec_lists && (!heavy_path || hist != nullptr);
    if (spec_lists) enqueue_lists(false);
    if (spec_raster) enqueue_raster(hist ? hist->heavy : 0, hist ? hist->pairs : 0, false);
    JR_HIP(hipEventSynchronize(ctx->ev_counters));
    const size_t pairs = (size_t)ctx->h_counters[0];
    const int64_t heavy_now = (int64_t)ctx->h_counters[3];
    ctx->stats[0] = (int64_t)pairs;
    ctx->stats[1] = (int64_t)ctx->h_counters[1];
    ctx->stats[2] = (int64_t)ctx->h_counters[2];
    ctx->stats[3] = (int64_t)p.bins_x * p.bins_y;
    if (heavy_path) {                                            // remember what this shape found (least recently used slot)
        if (!hist) {
            hist = &ctx->hist[0];
            for (auto& h : ctx->hist) if (h.stamp < hist->stamp) hist = &h;
            hist->B = p.B; hist->NF = p.NF; hist->IS = p.IS; hist->heavy_min = ws.heavy_min; hist->bin_log2 = p.bin_log2;
        }
        hist->heavy = heavy_now; hist->pairs = (int64_t)pairs; hist->stamp = ++ctx->hist_clock;
    }
    if (pairs > 0x7fffffffULL)       // segment bases are 32-bit
        return fail("%zu (bin, face) pairs exceed the 2^31 - 1 the bin lists index: render fewer views per call", pairs);
    if (!spec_lists || pairs > ws.pool_cap) {
        if (pairs > ws.pool_cap || !ws.pool) {
            if (ctx->capturing) return fail("the pair pool has to grow during a graph capture: run the sequence once outside the graph and capture again");
            JR_HIP(hipStreamSynchronize(ctx->stream));      // nobody may still read the old pool
            size_t c0 = ws.pool_cap, c1 = ws.pool_cap;
            if (grow(ctx, "the pair pool", ws.pool, c0, pairs > 0 ? pairs : 1, 1.25)) return 1;
            if (grow(ctx, "the pair pool", ws.pool_scratch, c1, pairs > 0 ? pairs : 1, 1.25)) return 1;
            ws.pool_cap = c0;
        }
        enqueue_lists(spec_lists);
        enqueue_raster(heavy_now, (int64_t)pairs, true);
    } else if (!spec_raster) enqueue_raster(heavy_now, (int64_t)pairs, true);
    ctx->launch_info[0] = heavy_path ? 1 : 0;
    ctx->launch_info[1] = heavy_path ? std::min<int64_t>(heavy_now, jr::heavy_bins_cap(ws, p.B * p.bins_x * p.bins_y)) : 0;
    ctx->launch_info[2] = ws.heavy_waves_used;
    ctx->launch_info[3] = ws.heavy_min;
    ws.heavy_bound = heavy_now;      // the backward of this forward (token reuse) knows the exact count
    JR_HIP(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

const char* jr_last_error(void) { return g_err; }
const char* jr_version(void) { return "jrender_hip 0.1 (gfx950)"; }

int jr_device_count(int* count) {
    JR_HIP(hipGetDeviceCount(count));
    return 0;
}

int jr_ctx_create(int device, jr_ctx** out) {
    if (!out) return fail("jr_ctx_create: out is NULL");
    int n = 0;
    JR_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail("jr_ctx_create: device %d out of range (%d visible)", device, n);
    JR_HIP(hipSetDevice(device));
    jr_ctx* c = new (std::nothrow) jr_ctx();
    if (!c) return fail("out of host memory");
    c->device = device;
    JR_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    JR_HIP(hipHostMalloc((void**)&c->h_counters, sizeof(unsigned long long) * 4, hipHostMallocMapped | hipHostMallocCoherent));
    JR_HIP(hipHostGetDevicePointer((void**)&c->ws.host_counters, c->h_counters, 0));
    JR_HIP(hipMalloc((void**)&c->ws.counters, sizeof(unsigned long long) * 32));   // [0..3] bin totals (live from a forward to its backward), [4..23] section clocks (instrumented builds), [24..] scratch of the self-tests
    JR_HIP(hipMemset(c->ws.counters, 0, sizeof(unsigned long long) * 32));
    // JR_FWD_HEAVY_MIN / JR_FWD_HEAVY_WAVES (tests, diagnostics): initial launch policy, see jr_softras_set_launch_policy
    if (const char* e = getenv("JR_FWD_HEAVY_MIN")) c->heavy_min_user = atoi(e) > 0 ? atoi(e) : 0;
    if (const char* e = getenv("JR_BIN_SIZE")) c->bin_size_user = atoi(e) > 0 ? atoi(e) : 0;
    if (const char* e = getenv("JR_FWD_HEAVY_WAVES")) c->forced_waves = (atoi(e) == 4 || atoi(e) == 8) ? atoi(e) : 0;
    JR_HIP(hipEventCreateWithFlags(&c->ev_counters, hipEventDisableTiming));
    *out = c;
    return 0;
}

int jr_ctx_destroy(jr_ctx* ctx) {
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->cache)
        for (void* p : kv.second) (void)hipFree(p);
    for (auto& kv : ctx->live) (void)hipFree(kv.first);
    for (auto& kv : ctx->parked) (void)hipFree(kv.first);
    for (auto& kv : ctx->capture_free)
        for (void* p : kv.second) (void)hipFree(p);
    jr::BinWorkspace& ws = ctx->ws;
    (void)hipFree(ctx->zkey); (void)hipFree(ctx->n3_scratch); (void)hipFree(ctx->red_acc); (void)hipFree(ctx->red_ticket);
    (void)hipFree(ws.geo); (void)hipFree(ws.face_rect); (void)hipFree(ws.bin_acc); (void)hipFree(ws.bin_count); (void)hipFree(ws.bin_base); (void)hipFree(ws.bin_cursor); (void)hipFree(ws.bin_order);
    (void)hipFree(ws.counters); (void)hipFree(ws.pool); (void)hipFree(ws.pool_scratch);
    for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
    (void)hipHostFree(ctx->h_counters);
    if (ctx->ev_counters) (void)hipEventDestroy(ctx->ev_counters);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

int jr_ctx_device(const jr_ctx* ctx) { return ctx ? ctx->device : -1; }
void* jr_ctx_stream(const jr_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int jr_malloc(jr_ctx* ctx, size_t bytes, void** dptr) {
    if (!ctx || !dptr) return fail("jr_malloc: NULL argument");
    JR_HIP(hipSetDevice(ctx->device));
    const size_t sz = ((bytes ? bytes : 1) + 255) & ~(size_t)255;
    auto it = ctx->cache.find(sz);
    auto cf = ctx->capturing ? ctx->capture_free.find(sz) : ctx->capture_free.end();
    if (cf != ctx->capture_free.end() && !cf->second.empty()) {          // a block this capture's own sequence has freed
        *dptr = cf->second.back();
        cf->second.pop_back();
    } else if (it != ctx->cache.end() && !it->second.empty()) {
        *dptr = it->second.back();
        it->second.pop_back();
        ctx->cached_bytes -= sz;
    } else {
        if (ctx->capturing)
            return fail("jr_malloc(%zu) during graph capture found no cached block: run the sequence once (twice) before capturing it", sz);
        hipError_t e = hipMalloc(dptr, sz);
        if (e != hipSuccess && ctx->cached_bytes) {      // out of memory: drop the cache and retry
            (void)hipGetLastError();
            if (jr_ctx_trim(ctx)) return 1;
            e = hipMalloc(dptr, sz);
        }
        if (e != hipSuccess) return fail("hipMalloc(%zu) failed: %s", sz, hipGetErrorString(e));
    }
    ctx->live[*dptr] = sz;
    if (ctx->capturing && !ctx->pinned.count(*dptr)) { ctx->pinned[*dptr] = 1; ctx->capture_blocks.push_back(*dptr); }
    return 0;
}
int jr_free(jr_ctx* ctx, void* dptr) {
    if (!ctx) return fail("jr_free: NULL context");
    if (!dptr) return 0;
    auto it = ctx->live.find(dptr);
    if (it == ctx->live.end()) return fail("jr_free: pointer %p was not allocated by this context", dptr);
    if (ctx->capturing && !ctx->pinned.count(dptr)) { ctx->pinned[dptr] = 1; ctx->capture_blocks.push_back(dptr); }   // (a captured kernel may have used it)
    if (ctx->capturing) ctx->capture_free[it->second].push_back(dptr);       // reusable by the rest of THIS capture only
    else if (ctx->pinned.count(dptr)) ctx->parked[dptr] = it->second;        // a graph addresses it: out of circulation until jr_graph_destroy
    else {
        ctx->cache[it->second].push_back(dptr);
        ctx->cached_bytes += it->second;
    }
    ctx->live.erase(it);
    return 0;
}
namespace {
// the blocks of a destroyed / failed / aborted graph go back into circulation
void unpin_blocks(jr_ctx* ctx, const std::vector<void*>& blocks) {
    for (void* b : blocks) {
        ctx->pinned.erase(b);
        auto pk = ctx->parked.find(b);
        if (pk != ctx->parked.end()) {
            ctx->cache[pk->second].push_back(b);
            ctx->cached_bytes += pk->second;
            ctx->parked.erase(pk);
        }
    }
}
}  // namespace
int jr_ctx_trim(jr_ctx* ctx) {
    if (!ctx) return fail("NULL context");
    if (ctx->capturing) return fail("jr_ctx_trim during graph capture");
    JR_HIP(hipSetDevice(ctx->device));
    JR_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& kv : ctx->cache)
        for (void* p : kv.second) JR_HIP(hipFree(p));
    ctx->cache.clear();
    ctx->cached_bytes = 0;
    return 0;
}
int jr_memcpy_h2d(jr_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx) return fail("NULL context");
    JR_HIP(hipSetDevice(ctx->device));
    JR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    JR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}
int jr_memcpy_d2h(jr_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx) return fail("NULL context");
    JR_HIP(hipSetDevice(ctx->device));
    JR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    JR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}
int jr_memcpy_d2d(jr_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx) return fail("NULL context");
    JR_HIP(hipSetDevice(ctx->device));
    JR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}
int jr_memcpy2d_d2d(jr_ctx* ctx, void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width,
                    size_t height) {
    if (!ctx) return fail("NULL context");
    if (width > dst_pitch || width > src_pitch) return fail("jr_memcpy2d_d2d: width exceeds a pitch");
    if (!width || !height) return 0;
    JR_HIP(hipSetDevice(ctx->device));
    JR_HIP(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width, height, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}
int jr_memset(jr_ctx* ctx, void* dptr, int value, size_t bytes) {
    if (!ctx) return fail("NULL context");
    JR_HIP(hipSetDevice(ctx->device));
    JR_HIP(hipMemsetAsync(dptr, value, bytes, ctx->stream));
    return 0;
}
int jr_synchronize(jr_ctx* ctx) {
    if (!ctx) return fail("NULL context");
    JR_HIP(hipSetDevice(ctx->device));
    JR_HIP(hipStreamSynchronize(ctx->stream));
    JR_HIP(hipGetLastError());
    return 0;
}

int jr_event_create(jr_ctx* ctx, void** event) {
    if (!ctx || !event) return fail("NULL argument");
    JR_HIP(hipSetDevice(ctx->device));
    hipEvent_t e;
    JR_HIP(hipEventCreate(&e));
    *event = (void*)e;
    return 0;
}
int jr_event_destroy(jr_ctx* ctx, void* event) {
    if (!ctx) return fail("NULL context");
    JR_HIP(hipEventDestroy((hipEvent_t)event));
    return 0;
}
int jr_event_record(jr_ctx* ctx, void* event) {
    if (!ctx) return fail("NULL context");
    JR_HIP(hipEventRecord((hipEvent_t)event, ctx->stream));
    return 0;
}
int jr_event_elapsed_ms(jr_ctx* ctx, void* start, void* stop, float* ms) {
    if (!ctx || !ms) return fail("NULL argument");
    JR_HIP(hipEventSynchronize((hipEvent_t)stop));
    JR_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return 0;
}

int jr_softras_forward(jr_ctx* ctx, const float* face_vertices, const float* textures,
                       float* faces_info, float* aggrs_info, float* soft_colors,
                       int32_t* faces_id_buffer, int B, int NF, int T, int IS, int K,
                       float near_, float far_, float eps, float sigma_val, int func_id_dist,
                       float dist_eps, float gamma_val, int func_id_rgb, int func_id_alpha,
                       int texture_sample_type, int double_side, const float* background_rgb) {
    if (!ctx) return fail("jr_softras_forward: NULL context");
    if (!face_vertices || !textures || !faces_info || !aggrs_info || !soft_colors || !faces_id_buffer)
        return fail("jr_softras_forward: NULL tensor pointer");
    if (validate(B, NF, T, IS, K, func_id_dist, func_id_rgb, func_id_alpha, texture_sample_type)) return 1;
    JR_HIP(hipSetDevice(ctx->device));
    const jr::RasterParams p = make_params(ctx, B, NF, T, IS, K, near_, far_, eps, sigma_val, func_id_dist,
                                           dist_eps, gamma_val, func_id_rgb, func_id_alpha,
                                           texture_sample_type, double_side, background_rgb);
    ctx->last_forward_token = 0;
    if (forward_pipeline(ctx, p, face_vertices, textures, faces_info, aggrs_info, soft_colors, faces_id_buffer))
        return 1;
    ctx->last_forward_token = ctx->geo_epoch;
    return 0;
}

uint64_t jr_softras_forward_token(const jr_ctx* ctx) { return ctx ? ctx->last_forward_token : 0; }

int jr_softras_backward_ex(jr_ctx* ctx, const float* face_vertices, const float* textures,
                           const float* soft_colors, const float* faces_info,
                           const float* aggrs_info, const int32_t* faces_id_buffer,
                           const float* grad_soft_colors, float* grad_faces, float* grad_textures,
                           int B, int NF, int T, int IS, int K, float near_, float far_, float eps,
                           float sigma_val, int func_id_dist, float dist_eps, float gamma_val,
                           int func_id_rgb, int func_id_alpha, int texture_sample_type,
                           int double_side, uint64_t forward_token) {
    if (!ctx) return fail("jr_softras_backward: NULL context");
    if (!face_vertices || !textures || !soft_colors || !faces_info || !aggrs_info || !faces_id_buffer ||
        !grad_soft_colors || !grad_faces || !grad_textures)
        return fail("jr_softras_backward: NULL tensor pointer");
    if (validate(B, NF, T, IS, K, func_id_dist, func_id_rgb, func_id_alpha, texture_sample_type)) return 1;
    JR_HIP(hipSetDevice(ctx->device));
    const jr::RasterParams p = make_params(ctx, B, NF, T, IS, K, near_, far_, eps, sigma_val, func_id_dist,
                                           dist_eps, gamma_val, func_id_rgb, func_id_alpha,
                                           texture_sample_type, double_side, nullptr);
    // The face records / launch order of the matching forward are reused only when the caller proves
    // that nothing rebuilt them since: its token must be the context's current generation.  Anything
    // else rebuilds them from face_vertices / textures as passed (0.06 ms on the headline workload; no
    // faces_info write, no lists: the backward finds its faces through the id buffer).
    const bool reuse = forward_token != 0 && forward_token == ctx->geo_epoch && ctx->bins_B == B &&
                       ctx->bins_NF == NF && ctx->bins_IS == IS && ctx->bins_rad == p.rad && ctx->bins_T == T &&
                       ctx->bins_bin_log2 == p.bin_log2 && ctx->ws.heavy_min == resolve_heavy_min(ctx, p.bin_log2, NF);
    if (!reuse) {
        ctx->ws.heavy_min = resolve_heavy_min(ctx, p.bin_log2, NF);
        if (setup_faces(ctx, p, face_vertices, textures, nullptr)) return 1;
        ctx->ws.heavy_bound = -1;           // nobody read this schedule's totals: the pool-capacity bound
    }
    {
        ProfScope ps(ctx, JR_PHASE_BWD_RASTER);
        jr::launch_softras_backward(ctx->stream, p, textures, soft_colors, aggrs_info, faces_id_buffer,
                                    grad_soft_colors, ctx->ws, grad_faces, grad_textures);
    }
    JR_HIP(hipGetLastError());
    return 0;
}

int jr_softras_backward(jr_ctx* ctx, const float* face_vertices, const float* textures,
                        const float* soft_colors, const float* faces_info,
                        const float* aggrs_info, const int32_t* faces_id_buffer,
                        const float* grad_soft_colors, float* grad_faces, float* grad_textures,
                        int B, int NF, int T, int IS, int K, float near_, float far_, float eps,
                        float sigma_val, int func_id_dist, float dist_eps, float gamma_val,
                        int func_id_rgb, int func_id_alpha, int texture_sample_type,
                        int double_side) {
    return jr_softras_backward_ex(ctx, face_vertices, textures, soft_colors, faces_info, aggrs_info,
                                  faces_id_buffer, grad_soft_colors, grad_faces, grad_textures, B, NF, T, IS, K,
                                  near_, far_, eps, sigma_val, func_id_dist, dist_eps, gamma_val, func_id_rgb,
                                  func_id_alpha, texture_sample_type, double_side, 0);
}

int jr_face_vertices_forward(jr_ctx* ctx, const float* vertices, const int32_t* faces,
                             float* face_vertices, int B, int NV, int NF) {
    if (!ctx || !vertices || !faces || !face_vertices) return fail("jr_face_vertices_forward: NULL argument");
    if (B < 1 || NV < 1 || NF < 1) return fail("jr_face_vertices_forward: bad sizes");
    JR_HIP(hipSetDevice(ctx->device));
    jr::launch_face_vertices_forward(ctx->stream, vertices, faces, face_vertices, B, NV, NF);
    JR_HIP(hipGetLastError());
    return 0;
}
int jr_face_vertices_backward(jr_ctx* ctx, const float* grad_face_vertices, const int32_t* faces,
                              float* grad_vertices, int B, int NV, int NF) {
    if (!ctx || !grad_face_vertices || !faces || !grad_vertices) return fail("jr_face_vertices_backward: NULL argument");
    if (B < 1 || NV < 1 || NF < 1) return fail("jr_face_vertices_backward: bad sizes");
    JR_HIP(hipSetDevice(ctx->device));
    jr::launch_face_vertices_backward(ctx->stream, grad_face_vertices, faces, grad_vertices, B, NV, NF);
    JR_HIP(hipGetLastError());
    return 0;
}
int jr_face_vertices_backward_shared(jr_ctx* ctx, const float* grad_face_vertices, const int32_t* faces,
                                     float* grad_vertices, int B, int NV, int NF) {
    if (!ctx || !grad_face_vertices || !faces || !grad_vertices) return fail("jr_face_vertices_backward_shared: NULL argument");
    if (B < 0 || NV < 1 || NF < 1) return fail("jr_face_vertices_backward_shared: bad sizes");
    JR_HIP(hipSetDevice(ctx->device));
    jr::launch_face_vertices_backward_shared(ctx->stream, grad_face_vertices, faces, grad_vertices, B, NV, NF);
    JR_HIP(hipGetLastError());
    return 0;
}
int jr_avgpool2x2_forward(jr_ctx* ctx, const float* in, float* out, int planes, int H, int W) {
    if (!ctx || !in || !out) return fail("jr_avgpool2x2_forward: NULL argument");
    if (planes < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) return fail("jr_avgpool2x2: H and W must be even");
    JR_HIP(hipSetDevice(ctx->device));
    jr::launch_avgpool2x2_forward(ctx->stream, in, out, planes, H, W);
    JR_HIP(hipGetLastError());
    return 0;
}
int jr_avgpool2x2_backward(jr_ctx* ctx, const float* grad_out, float* grad_in, int planes, int H, int W) {
    if (!ctx || !grad_out || !grad_in) return fail("jr_avgpool2x2_backward: NULL argument");
    if (planes < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) return fail("jr_avgpool2x2: H and W must be even");
    JR_HIP(hipSetDevice(ctx->device));
    jr::launch_avgpool2x2_backward(ctx->stream, grad_out, grad_in, planes, H, W);
    JR_HIP(hipGetLastError());
    return 0;
}

int jr_camera_forward(jr_ctx* ctx, const float* vertices, const float* eye, const float* rot, float* out,
                      int B, int VB, int NV, int kind, float param) {
    if (!ctx || !vertices || !eye || !rot || !out) return fail("jr_camera_forward: NULL argument");
    if (B < 1 || NV < 1 || (VB != 1 && VB != B)) return fail("jr_camera_forward: vertices must be [1 or B, NV, 3]");
    if (kind < 0 || kind > 2) return fail("jr_camera_forward: kind must be 0 (none), 1 (perspective) or 2 (orthogonal)");
    JR_HIP(hipSetDevice(ctx->device));
    jr::launch_camera_forward(ctx->stream, vertices, eye, rot, out, B, VB, NV, kind, param);
    JR_HIP(hipGetLastError());
    return 0;
}
int jr_camera_backward(jr_ctx* ctx, const float* grad_out, const float* vertices, const float* eye,
                       const float* rot, float* grad_vertices, int B, int VB, int NV, int kind, float param) {
    if (!ctx || !grad_out || !vertices || !eye || !rot || !grad_vertices) return fail("jr_camera_backward: NULL argument");
    if (B < 1 || NV < 1 || (VB != 1 && VB != B)) return fail("jr_camera_backward: vertices must be [1 or B, NV, 3]");
    if (kind < 0 || kind > 2) return fail("jr_camera_backward: kind must be 0 (none), 1 (perspective) or 2 (orthogonal)");
    JR_HIP(hipSetDevice(ctx->device));
    jr::launch_camera_backward(ctx->stream, grad_out, vertices, eye, rot, grad_vertices, B, VB, NV, kind, param);
    JR_HIP(hipGetLastError());
    return 0;
}
int jr_face_camera_backward_shared(jr_ctx* ctx, const float* grad_face_vertices, const int32_t* faces,
                                   const float* vertices, const float* eye, const float* rot, float* grad_vertices,
                                   int B, int NV, int NF, int kind, float param) {
    if (!ctx || !grad_face_vertices || !faces || !vertices || !eye || !rot || !grad_vertices)
        return fail("jr_face_camera_backward_shared: NULL argument");
    if (B < 0 || NV < 1 || NF < 1) return fail("jr_face_camera_backward_shared: bad sizes");
    if (kind < 0 || kind > 2) return fail("jr_face_camera_backward_shared: kind must be 0, 1 or 2");
    JR_HIP(hipSetDevice(ctx->device));
    jr::launch_face_camera_backward_shared(ctx->stream, grad_face_vertices, faces, vertices, eye, rot, grad_vertices, B,
                                           NV, NF, kind, param);
    JR_HIP(hipGetLastError());
    return 0;
}
int jr_neg_iou_loss(jr_ctx* ctx, const float* predict, const float* target, float* iou, float* grad_predict,
                    int B, int n, float divisor) {
    if (!ctx || !predict || !target || !iou) return fail("jr_neg_iou_loss: NULL argument");
    if (B < 1 || n < 1) return fail("jr_neg_iou_loss: bad sizes");
    if (grad_predict && !(divisor > 0.f)) return fail("jr_neg_iou_loss: divisor must be > 0");
    JR_HIP(hipSetDevice(ctx->device));
    jr::launch_neg_iou_loss(ctx->stream, predict, target, iou, grad_predict, B, n, divisor);
    JR_HIP(hipGetLastError());
    return 0;
}

int jr_laplacian_loss(jr_ctx* ctx, const int32_t* rowptr, const int32_t* col, const float* val,
                      const int32_t* rowptr_t, const int32_t* col_t, const float* val_t, const float* vertices,
                      float* scratch, float* loss, float* grad_vertices, int B, int NV, float grad_scale) {
    if (!ctx || !rowptr || !col || !val || !vertices || !scratch || !loss) return fail("jr_laplacian_loss: NULL argument");
    if (grad_vertices && (!rowptr_t || !col_t || !val_t)) return fail("jr_laplacian_loss: the gradient needs the transposed matrix");
    if (B < 1 || NV < 1) return fail("jr_laplacian_loss: bad sizes");
    if (B > 65535) return fail("jr_laplacian_loss: at most 65 535 meshes per call (the batch is the launch grid's second dimension), got %d", B);
    JR_HIP(hipSetDevice(ctx->device));
    if (begin_reduction(ctx, (size_t)B)) return 1;
    jr::launch_laplacian_loss(ctx->stream, rowptr, col, val, rowptr_t, col_t, val_t, vertices, scratch, loss, grad_vertices,
                              ctx->red_acc + 4, ctx->red_ticket + 4, B, NV, grad_scale);
    return end_reduction(ctx);
}
int jr_flatten_loss(jr_ctx* ctx, const int32_t* v0s, const int32_t* v1s, const int32_t* v2s, const int32_t* v3s,
                    const float* vertices, float* loss, float* grad_vertices, int B, int NV, int NE, float eps,
                    float grad_scale) {
    if (!ctx || !v0s || !v1s || !v2s || !v3s || !vertices || !loss) return fail("jr_flatten_loss: NULL argument");
    if (B < 1 || NV < 1 || NE < 0) return fail("jr_flatten_loss: bad sizes");
    if (B > 65535) return fail("jr_flatten_loss: at most 65 535 meshes per call (the batch is the launch grid's second dimension), got %d", B);
    JR_HIP(hipSetDevice(ctx->device));
    if (begin_reduction(ctx, (size_t)B)) return 1;
    jr::launch_flatten_loss(ctx->stream, v0s, v1s, v2s, v3s, vertices, loss, grad_vertices, ctx->red_acc + 4,
                            ctx->red_ticket + 4, B, NV, NE, eps, grad_scale);
    return end_reduction(ctx);
}

int jr_deform_vertices_forward(jr_ctx* ctx, const float* template_vertices, const float* displace, const float* center,
                               float* vertices, int NV) {
    if (!ctx || !template_vertices || !displace || !center || !vertices) return fail("jr_deform_vertices_forward: NULL argument");
    if (NV < 1 || NV > (1 << 28)) return fail("jr_deform_vertices_forward: bad vertex count %d", NV);
    JR_HIP(hipSetDevice(ctx->device));
    jr::launch_deform_forward(ctx->stream, template_vertices, displace, center, vertices, NV);
    JR_HIP(hipGetLastError());
    return 0;
}
int jr_deform_vertices_backward(jr_ctx* ctx, const float* template_vertices, const float* displace, const float* center,
                                const float* grad0, float w0, const float* grad1, float w1, const float* grad2, float w2,
                                float* grad_displace, float* grad_center, int NV) {
    if (!ctx || !template_vertices || !displace || !center || !grad0 || !grad_displace || !grad_center)
        return fail("jr_deform_vertices_backward: NULL argument");
    if (NV < 1 || NV > (1 << 28)) return fail("jr_deform_vertices_backward: bad vertex count %d", NV);
    JR_HIP(hipSetDevice(ctx->device));
    if (begin_reduction(ctx, 0)) return 1;
    jr::launch_deform_backward(ctx->stream, template_vertices, displace, center, grad0, w0, grad1, w1, grad2, w2,
                               grad_displace, grad_center, ctx->red_acc, ctx->red_ticket, NV);
    return end_reduction(ctx);
}
int jr_adam_step(jr_ctx* ctx, float* param, const float* grad, float* m, float* v, size_t n, double lr, double beta0,
                 double beta1, double eps, double weight_decay, int step) {
    if (!ctx || !param || !grad || !m || !v) return fail("jr_adam_step: NULL argument");
    if (n == 0) return 0;
    if (step < 1) return fail("jr_adam_step: step counts from 1 (got %d)", step);
    if (!(beta0 >= 0.0 && beta0 < 1.0 && beta1 >= 0.0 && beta1 < 1.0)) return fail("jr_adam_step: betas must be in [0, 1)");
    JR_HIP(hipSetDevice(ctx->device));
    // the mirror's Python scalars: formed in double, rounded to float where NumPy multiplies them into a float32 array
    const double c0 = 1.0 - std::pow(beta0, step), c1 = 1.0 - std::pow(beta1, step);
    jr::launch_adam_step(ctx->stream, param, grad, m, v, n, (float)(lr / c0), (float)beta0, (float)(1.0 - beta0), (float)beta1,
                         (float)(1.0 - beta1), (float)c1, (float)eps, (float)weight_decay, nullptr, lr, beta0, beta1);
    JR_HIP(hipGetLastError());
    return 0;
}
int jr_adam_step_counted(jr_ctx* ctx, float* param, const float* grad, float* m, float* v, size_t n, double lr, double beta0,
                         double beta1, double eps, double weight_decay, const int32_t* iteration) {
    if (!ctx || !param || !grad || !m || !v || !iteration) return fail("jr_adam_step_counted: NULL argument");
    if (n == 0) return 0;
    if (!(beta0 >= 0.0 && beta0 < 1.0 && beta1 >= 0.0 && beta1 < 1.0)) return fail("jr_adam_step_counted: betas must be in [0, 1)");
    JR_HIP(hipSetDevice(ctx->device));
    jr::launch_adam_step(ctx->stream, param, grad, m, v, n, 0.f, (float)beta0, (float)(1.0 - beta0), (float)beta1,
                         (float)(1.0 - beta1), 1.f, (float)eps, (float)weight_decay, iteration, lr, beta0, beta1);
    JR_HIP(hipGetLastError());
    return 0;
}
int jr_scalar_accumulate(jr_ctx* ctx, float* dst, const float* src, int n, float scale, float bias, int accumulate) {
    if (!ctx || !dst || (n > 0 && !src)) return fail("jr_scalar_accumulate: NULL argument");
    if (n < 0) return fail("jr_scalar_accumulate: n must be >= 0");
    JR_HIP(hipSetDevice(ctx->device));
    jr::launch_scalar_accumulate(ctx->stream, dst, src, n, scale, bias, accumulate, nullptr, 0);
    JR_HIP(hipGetLastError());
    return 0;
}
int jr_scalar_accumulate_at(jr_ctx* ctx, float* dst, int stride, const int32_t* iteration, const float* src, int n, float scale,
                            float bias, int accumulate) {
    if (!ctx || !dst || !iteration || (n > 0 && !src)) return fail("jr_scalar_accumulate_at: NULL argument");
    if (n < 0 || stride < 0) return fail("jr_scalar_accumulate_at: n and stride must be >= 0");
    JR_HIP(hipSetDevice(ctx->device));
    jr::launch_scalar_accumulate(ctx->stream, dst, src, n, scale, bias, accumulate, iteration, stride);
    JR_HIP(hipGetLastError());
    return 0;
}
int jr_counter_add(jr_ctx* ctx, int32_t* counter, int delta) {
    if (!ctx || !counter) return fail("jr_counter_add: NULL argument");
    JR_HIP(hipSetDevice(ctx->device));
    jr::launch_counter_add(ctx->stream, counter, delta);
    JR_HIP(hipGetLastError());
    return 0;
}

// ---- a fixed launch sequence as ONE HIP graph (round 5) ----------------------------------------------------------------------
int jr_graph_begin(jr_ctx* ctx) {
    if (!ctx) return fail("jr_graph_begin: NULL context");
    if (ctx->capturing) return fail("jr_graph_begin: a capture is already open on this context");
    if (ctx->prof_on) return fail("jr_graph_begin: switch phase profiling off first (its events are host-timed)");
    JR_HIP(hipSetDevice(ctx->device));
    JR_HIP(hipStreamSynchronize(ctx->stream));
    JR_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
    ctx->capturing = 1;
    ctx->capture_blocks.clear();
    return 0;
}
int jr_graph_end(jr_ctx* ctx, void** graph_exec) {
    if (!ctx || !graph_exec) return fail("jr_graph_end: NULL argument");
    if (!ctx->capturing) return fail("jr_graph_end: no capture is open");
    ctx->capturing = 0;
    for (auto& kv : ctx->capture_free)
        for (void* b : kv.second) ctx->parked[b] = kv.first;
    ctx->capture_free.clear();
    std::vector<void*> blocks;
    blocks.swap(ctx->capture_blocks);
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
    if (e != hipSuccess) { unpin_blocks(ctx, blocks); return fail("hipStreamEndCapture: %s", hipGetErrorString(e)); }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) { unpin_blocks(ctx, blocks); return fail("hipGraphInstantiate: %s", hipGetErrorString(e)); }
    ctx->graph_blocks[exec] = std::move(blocks);
    ctx->graph_generation[exec] = ctx->ws_generation;
    *graph_exec = exec;
    return 0;
}
int jr_graph_abort(jr_ctx* ctx) {
    if (!ctx) return fail("jr_graph_abort: NULL context");
    if (!ctx->capturing) return 0;
    ctx->capturing = 0;
    {
        for (auto& kv : ctx->capture_free)
            for (void* b : kv.second) ctx->parked[b] = kv.first;
        ctx->capture_free.clear();
        std::vector<void*> blocks;
        blocks.swap(ctx->capture_blocks);
        unpin_blocks(ctx, blocks);
    }
    hipGraph_t graph = nullptr;
    (void)hipStreamEndCapture(ctx->stream, &graph);
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    return 0;
}
int jr_graph_launch(jr_ctx* ctx, void* graph_exec) {
    if (!ctx || !graph_exec) return fail("jr_graph_launch: NULL argument");
    if (ctx->capturing) return fail("jr_graph_launch during capture");
    {
        auto g = ctx->graph_generation.find(graph_exec);
        if (g == ctx->graph_generation.end()) return fail("jr_graph_launch: not a graph of this context");
        if (g->second != ctx->ws_generation)
            return fail("jr_graph_launch: the context's scratch was reallocated after this graph was captured (a larger call outside the "
                        "graph): its nodes address freed memory - destroy it and capture again");
    }
    // (sticky "pool too small": what the last COMPLETED schedule kernel of a replayed forward wrote to pinned memory - no wait
    //  here, so it may lag a replay or two; jr_graph_check is the synchronous form)
    if (ctx->h_counters && ctx->ws.pool_cap && ctx->h_counters[0] > ctx->ws.pool_cap)
        return fail("jr_graph_launch: a replayed forward found %llu (bin, face) pairs, the captured pool holds %zu: its raster kernels did "
                    "nothing - run the sequence outside the graph once (the pool grows) and capture again",
                    (unsigned long long)ctx->h_counters[0], ctx->ws.pool_cap);
    JR_HIP(hipSetDevice(ctx->device));
    JR_HIP(hipGraphLaunch(static_cast<hipGraphExec_t>(graph_exec), ctx->stream));
    return 0;
}
int jr_graph_check(jr_ctx* ctx) {
    if (!ctx) return fail("jr_graph_check: NULL context");
    JR_HIP(hipSetDevice(ctx->device));
    JR_HIP(hipStreamSynchronize(ctx->stream));
    // the last replayed forward's pair total (k_bin_alloc_schedule writes it to pinned memory) against the pool the graph was
    // captured with: beyond it the list kernels and both raster kernels do nothing (they re-check on the device)
    if (ctx->h_counters && ctx->ws.pool_cap && ctx->h_counters[0] > ctx->ws.pool_cap)
        return fail("a replayed forward found %llu (bin, face) pairs, the captured pool holds %zu: run the sequence outside the graph once "
                    "(the pool grows) and capture again", (unsigned long long)ctx->h_counters[0], ctx->ws.pool_cap);
    return 0;
}
int jr_graph_destroy(jr_ctx* ctx, void* graph_exec) {
    if (!ctx) return fail("jr_graph_destroy: NULL context");
    if (!graph_exec) return 0;
    JR_HIP(hipSetDevice(ctx->device));
    JR_HIP(hipStreamSynchronize(ctx->stream));            // a replay in flight still uses the pinned blocks
    JR_HIP(hipGraphExecDestroy(static_cast<hipGraphExec_t>(graph_exec)));
    auto gb = ctx->graph_blocks.find(graph_exec);
    if (gb != ctx->graph_blocks.end()) { unpin_blocks(ctx, gb->second); ctx->graph_blocks.erase(gb); }
    ctx->graph_generation.erase(graph_exec);
    return 0;
}

int jr_n3mr_forward(jr_ctx* ctx, const float* faces, const float* textures, float* faces_inv,
                    int32_t* face_index_map, float* weight_map, float* depth_map, float* face_inv_map,
                    float* rgb_map, float* alpha_map, int32_t* sampling_index_map,
                    float* sampling_weight_map, int B, int NF, int TS, int IS, float near_, float far_,
                    float eps, const float* background_rgb, int return_rgb, int return_alpha,
                    int return_depth) {
    if (!ctx) return fail("jr_n3mr_forward: NULL context");
    if (!faces || !faces_inv || !face_index_map || !weight_map || !depth_map)
        return fail("jr_n3mr_forward: NULL tensor pointer");
    if (B < 1 || NF < 1 || IS < 1) return fail("jr_n3mr_forward: B, NF, image_size must be >= 1");
    if (IS > jr::MAX_IMAGE) return fail("jr_n3mr_forward: image_size %d exceeds the supported maximum %d", IS, jr::MAX_IMAGE);
    if (return_rgb && (!textures || !rgb_map || !sampling_index_map || !sampling_weight_map || TS < 2))
        return fail("jr_n3mr_forward: return_rgb needs textures (texture_size >= 2), rgb_map and sampling maps");
    if (return_alpha && !alpha_map) return fail("jr_n3mr_forward: return_alpha needs alpha_map");
    if (return_depth && !face_inv_map) return fail("jr_n3mr_forward: return_depth needs face_inv_map");
    if (!(near_ >= 0.f)) return fail("jr_n3mr_forward: near must be >= 0 (depth keys are ordered by bit pattern)");
    JR_HIP(hipSetDevice(ctx->device));
    const size_t P = (size_t)B * IS * IS;
    if (P > ctx->zkey_cap || !ctx->zkey) ctx->zkey_clean = 0;
    if (grow(ctx, "the NMR z-buffer keys", ctx->zkey, ctx->zkey_cap, P, 1.0)) return 1;
    const size_t was_clean = ctx->zkey_clean;
    const bool clean = P <= was_clean;
    ctx->zkey_clean = 0;                     // (until the launches below are known to be in the stream)
    jr::launch_n3mr_forward(ctx->stream, faces, textures, faces_inv, ctx->zkey, face_index_map, weight_map,
                            depth_map, face_inv_map, rgb_map, alpha_map, sampling_index_map,
                            sampling_weight_map, B, NF, TS, IS, near_, far_, eps, background_rgb, return_rgb,
                            return_alpha, return_depth, clean);
    JR_HIP(hipGetLastError());
    ctx->zkey_clean = clean ? was_clean : P;
    return 0;
}

int jr_n3mr_backward(jr_ctx* ctx, const float* faces, const int32_t* face_index_map,
                     const float* weight_map, const float* depth_map, const float* face_inv_map,
                     const float* rgb_map, const float* alpha_map, const float* sampling_weight_map,
                     const int32_t* sampling_index_map, const float* grad_rgb_map,
                     const float* grad_alpha_map, const float* grad_depth_map, float* grad_faces,
                     float* grad_textures, int B, int NF, int TS, int IS, float eps, int return_rgb,
                     int return_alpha, int return_depth) {
    if (!ctx) return fail("jr_n3mr_backward: NULL context");
    if (!faces || !face_index_map || !grad_faces) return fail("jr_n3mr_backward: NULL tensor pointer");
    if (return_rgb && (!rgb_map || !grad_rgb_map || !grad_textures || !sampling_weight_map || !sampling_index_map))
        return fail("jr_n3mr_backward: return_rgb needs rgb_map, grad_rgb_map, sampling maps, grad_textures");
    if (return_alpha && (!alpha_map || !grad_alpha_map)) return fail("jr_n3mr_backward: return_alpha needs alpha maps");
    if (return_depth && (!depth_map || !face_inv_map || !weight_map || !grad_depth_map))
        return fail("jr_n3mr_backward: return_depth needs depth / face_inv / weight maps and grad_depth_map");
    if (B < 1 || NF < 1 || IS < 1 || IS > jr::MAX_IMAGE) return fail("jr_n3mr_backward: bad sizes (image_size <= %d)", jr::MAX_IMAGE);
    JR_HIP(hipSetDevice(ctx->device));
    if (return_rgb || return_alpha) {
        const size_t need = jr::n3mr_backward_scratch_bytes(B, IS);
        if (need > ctx->n3_scratch_cap && grow(ctx, "the NMR backward's scratch planes", ctx->n3_scratch, ctx->n3_scratch_cap, need, 1.0)) return 1;
    }
    jr::launch_n3mr_backward(ctx->stream, faces, face_index_map, weight_map, depth_map, face_inv_map, rgb_map,
                             alpha_map, sampling_weight_map, sampling_index_map, grad_rgb_map, grad_alpha_map,
                             grad_depth_map, grad_faces, grad_textures, ctx->n3_scratch, B, NF, TS, IS, eps, return_rgb,
                             return_alpha, return_depth);
    JR_HIP(hipGetLastError());
    return 0;
}

int jr_n3mr_image_forward(jr_ctx* ctx, const float* in_nhwc, float* out_nchw, int B, int H, int W, int C, int pool) {
    if (!ctx || !in_nhwc || !out_nchw) return fail("jr_n3mr_image_forward: NULL argument");
    if (B < 1 || H < 1 || W < 1 || C < 1 || (pool != 1 && pool != 2) || (pool == 2 && ((H | W) & 1)))
        return fail("jr_n3mr_image_forward: bad sizes (pool must be 1 or 2, H and W even when pooling)");
    JR_HIP(hipSetDevice(ctx->device));
    jr::launch_n3mr_image_forward(ctx->stream, in_nhwc, out_nchw, B, H, W, C, pool);
    JR_HIP(hipGetLastError());
    return 0;
}
int jr_n3mr_image_backward(jr_ctx* ctx, const float* grad_out_nchw, float* grad_in_nhwc, int B, int H, int W, int C, int pool) {
    if (!ctx || !grad_out_nchw || !grad_in_nhwc) return fail("jr_n3mr_image_backward: NULL argument");
    if (B < 1 || H < 1 || W < 1 || C < 1 || (pool != 1 && pool != 2) || (pool == 2 && ((H | W) & 1)))
        return fail("jr_n3mr_image_backward: bad sizes (pool must be 1 or 2, H and W even when pooling)");
    JR_HIP(hipSetDevice(ctx->device));
    jr::launch_n3mr_image_backward(ctx->stream, grad_out_nchw, grad_in_nhwc, B, H, W, C, pool);
    JR_HIP(hipGetLastError());
    return 0;
}

int jr_selftest_division(jr_ctx* ctx, uint64_t n, uint32_t seed, uint64_t* mismatches) {
    if (!ctx || !mismatches) return fail("NULL argument");
    JR_HIP(hipSetDevice(ctx->device));
    // (own scratch word: counters[0..3] hold the schedule a token-reusing backward still reads)
    unsigned long long* scratch = ctx->ws.counters + 24;
    unsigned long long h = 0;
    JR_HIP(hipMemsetAsync(scratch, 0, sizeof(unsigned long long), ctx->stream));
    jr::launch_selftest_div(ctx->stream, n, seed, scratch);
    JR_HIP(hipMemcpyAsync(&h, scratch, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    JR_HIP(hipStreamSynchronize(ctx->stream));
    *mismatches = h;
    return 0;
}

int jr_selftest_reciprocal(jr_ctx* ctx, uint64_t* mismatches) {
    if (!ctx || !mismatches) return fail("NULL argument");
    JR_HIP(hipSetDevice(ctx->device));
    unsigned long long* scratch = ctx->ws.counters + 24;
    unsigned long long h = 0;
    JR_HIP(hipMemsetAsync(scratch, 0, sizeof(unsigned long long), ctx->stream));
    jr::launch_selftest_rcp(ctx->stream, scratch);
    JR_HIP(hipMemcpyAsync(&h, scratch, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    JR_HIP(hipStreamSynchronize(ctx->stream));
    *mismatches = h;
    return 0;
}

int jr_profile_enable(jr_ctx* ctx, int on) {
    if (!ctx) return fail("NULL context");
    ctx->prof_on = on != 0;
    return 0;
}

int jr_profile_collect(jr_ctx* ctx, double ms[JR_NUM_PHASES], int64_t launches[JR_NUM_PHASES]) {
    if (!ctx || !ms || !launches) return fail("NULL argument");
    JR_HIP(hipSetDevice(ctx->device));
    JR_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < JR_NUM_PHASES; i++) { ms[i] = 0.0; launches[i] = 0; }
    for (size_t i = 0; i < ctx->prof_phase.size(); i++) {
        float t = 0.f;
        JR_HIP(hipEventElapsedTime(&t, ctx->prof_events[2 * i], ctx->prof_events[2 * i + 1]));
        ms[ctx->prof_phase[i]] += t;
        launches[ctx->prof_phase[i]] += 1;
    }
    ctx->prof_phase.clear();
    ctx->prof_used = 0;
    return 0;
}

int jr_debug_section_clocks(jr_ctx* ctx, uint64_t clocks[20]) {
    if (!ctx || !clocks) return fail("NULL argument");
    JR_HIP(hipSetDevice(ctx->device));
    JR_HIP(hipStreamSynchronize(ctx->stream));
    JR_HIP(hipMemcpy(clocks, ctx->ws.counters + 4, sizeof(uint64_t) * 20, hipMemcpyDeviceToHost));
    JR_HIP(hipMemset(ctx->ws.counters + 4, 0, sizeof(uint64_t) * 20));
    return 0;
}

int jr_softras_last_stats(jr_ctx* ctx, int64_t stats[4]) {
    if (!ctx || !stats) return fail("NULL argument");
    memcpy(stats, ctx->stats, sizeof(ctx->stats));
    return 0;
}

int jr_softras_set_launch_policy(jr_ctx* ctx, int heavy_min_faces, int heavy_waves) {
    if (!ctx) return fail("NULL context");
    if (heavy_waves != 0 && heavy_waves != 4 && heavy_waves != 8) return fail("jr_softras_set_launch_policy: heavy_waves must be 0 (automatic), 4 or 8");
    if (heavy_min_faces > (1 << 24)) return fail("jr_softras_set_launch_policy: heavy_min_faces out of range");
    ctx->heavy_min_user = heavy_min_faces < 0 ? -1 : heavy_min_faces;
    ctx->forced_waves = heavy_waves;
    ctx->ws.heavy_bound = -1;
    ctx->geo_epoch++;                        // the schedule in the workspace was built under the old threshold: no backward may reuse it
    for (auto& h : ctx->hist) h.stamp = 0;
    return 0;
}

int jr_softras_set_bin_size(jr_ctx* ctx, int bin_size) {
    if (!ctx) return fail("NULL context");
    if (bin_size < 0 || bin_size > jr::MAX_IMAGE) return fail("jr_softras_set_bin_size: bin_size must be 0 (automatic) or a pixel count (got %d)", bin_size);
    // (no generation bump: a backward reuses the forward's records only when ITS resolved bin size is the one they were built with)
    ctx->bin_size_user = bin_size;
    return 0;
}

int jr_softras_set_precise_colour(jr_ctx* ctx, int on) {
    if (!ctx) return fail("NULL context");
    ctx->precise_colour = on ? 1 : 0;
    return 0;
}

int jr_softras_bin_size(const jr_ctx* ctx, int image_size, int batch, int num_faces) {
    if (!ctx) return -1;
    if (image_size <= 0) return ctx->bins_bin_log2 ? 1 << ctx->bins_bin_log2 : 0;     // of the set-up pass the workspace holds
    return 1 << resolve_bin_log2(ctx, batch > 0 ? batch : 1, image_size, num_faces > 0 ? num_faces : 0);
}

int jr_softras_last_launch(jr_ctx* ctx, int64_t info[4]) {
    if (!ctx || !info) return fail("NULL argument");
    memcpy(info, ctx->launch_info, sizeof(ctx->launch_info));
    return 0;
}

}  // extern "C"
