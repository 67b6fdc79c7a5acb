// Tuning switches of the raster kernels.  The product is built with the defaults below; tools/ablate/
// builds variants with -DJR_TUNE_<X>=0|1 (python -m jrender_amd._build --variant NAME -D...) to reproduce
// the A/B tables of DESIGN.md on one GPU box.  Every switch selects between two EXACTNESS-EQUIVALENT
// implementations (same face-index buffer bits, colours / gradients within tolerance; tools/ablate/run.py runs the
// parity tests on every variant) - except JR_TUNE_BWD_TV_RCP, kept to reproduce the measurement that rules it out.
// Switches that were measured dead in round 2 (K-buffer ids in LDS, SALU slot masks, tile box test, several tiles
// per wavefront, refinement quotient for the projection parameter, bank-masked DPP reduction) left the kernels:
// tools/ablate/patches/dead_switches_r03.patch restores them; round 3's dead ones (inside pairs in a second loop,
// s_setprio for heavy bins, 8-lane backward work items): dead_switches_r04.patch.
#pragma once

#ifndef JR_TUNE_FWD_DIS_ONLY     // forward: carry only (sign, dis) out of the distance machinery
#define JR_TUNE_FWD_DIS_ONLY 1
#endif
#ifndef JR_TUNE_FWD_PREPASS      // forward: conservative half-plane pre-cull of (pixel, face) pairs, lane = face
#define JR_TUNE_FWD_PREPASS 1
#endif
#ifndef JR_TUNE_FWD_INSIDE_RCP   // forward: 2nd / 3rd edge projection of INSIDE pixels (colour path only) by reciprocal multiply
#define JR_TUNE_FWD_INSIDE_RCP 1
#endif
#ifndef JR_TUNE_FWD_BATCH        // forward, one wavefront per tile: faces per batch (LDS record slots per wavefront), <= 64; 46 x 176 B = 8 KB -> 20 wavefronts per CU
#define JR_TUNE_FWD_BATCH 46
#endif
#ifndef JR_TUNE_FWD_BATCH_MIXED  // forward, four-wavefront workgroups: record slots per tile (56 x 176 B x 4 = 38.5 KB per workgroup -> 16 wavefronts per CU)
#define JR_TUNE_FWD_BATCH_MIXED 56
#endif
#ifndef JR_TUNE_FWD_IDS_GLOBAL   // forward: K-buffer ids are stored straight into faces_id_buffer at every insert instead of living in registers
#define JR_TUNE_FWD_IDS_GLOBAL 1
#endif
#ifndef JR_TUNE_FWD_WAVES16      // forward, one wavefront per tile, K <= 16: wavefronts per SIMD asked of the register allocator (5 -> 96 VGPRs, no scratch)
#define JR_TUNE_FWD_WAVES16 5
#endif
#ifndef JR_TUNE_FWD_WAVES32      // the same for 16 < K <= 32 (3 -> up to 168 VGPRs; the allocator uses 131)
#define JR_TUNE_FWD_WAVES32 4
#endif
#ifndef JR_TUNE_FWD_WAVES64      // and for 32 < K <= 64 (2 -> up to 256 VGPRs; 187 used)
#define JR_TUNE_FWD_WAVES64 3
#endif
#ifndef JR_TUNE_FWD_EMPTY_BINS   // forward: one wavefront writes the outputs of all 16 tiles of an empty bin with 16-byte stores
#define JR_TUNE_FWD_EMPTY_BINS 1
#endif
#ifndef JR_TUNE_FWD_FILL_SHIFT   // forward: K-buffer appends shift the depth registers (KCAP v_mov) instead of writing a per-lane slot (KCAP v_cmp + v_cndmask)
#define JR_TUNE_FWD_FILL_SHIFT 1
#endif
#ifndef JR_TUNE_FWD_EXP1         // forward: one v_exp per softmax update instead of two (the other one is exp(0))
#define JR_TUNE_FWD_EXP1 1
#endif
#ifndef JR_TUNE_FWD_HEAVY        // forward: bins whose list is longer than this get FOUR wavefronts per tile (evaluate / apply split); 0 = one wavefront per tile everywhere
#define JR_TUNE_FWD_HEAVY 512
#endif
#ifndef JR_TUNE_FWD_HEAVY16      // the same threshold for 16-pixel bins (2x2 tiles): meshes above / up to JR_TUNE_SMALL_MESH_FACES faces
#define JR_TUNE_FWD_HEAVY16 192
#endif
#ifndef JR_TUNE_FWD_HEAVY16_SMALL_MESH
#define JR_TUNE_FWD_HEAVY16_SMALL_MESH 64
#endif
#ifndef JR_TUNE_AUTO_DENSE_FACES_PER_BIN16   // automatic bin size: images of up to 512^2 whose mesh would put more than this many faces into a 16-pixel bin ON AVERAGE (faces x 256 / image_size^2) keep 32-pixel bins
#define JR_TUNE_AUTO_DENSE_FACES_PER_BIN16 100
#endif
#ifndef JR_TUNE_FWD_WAVES8_MEAN_LIST      // workgroup size: launches whose bins list this many faces on average (pairs / bins) take eight wavefronts whatever their size
#define JR_TUNE_FWD_WAVES8_MEAN_LIST 512
#endif
#ifndef JR_TUNE_BIN_FILL_UNROLL_MAX_FACES   // k_bin_fill: launches of up to this many faces (batch x mesh) append 4 bins per pass at 32-pixel bins too
#define JR_TUNE_BIN_FILL_UNROLL_MAX_FACES 65536
#endif
#ifndef JR_TUNE_SMALL_MESH_FACES
#define JR_TUNE_SMALL_MESH_FACES 10000
#endif
#ifndef JR_TUNE_FWD_HEAVY8       // ... and for 8-pixel bins (one tile: the list IS the tile's)
#define JR_TUNE_FWD_HEAVY8 48
#endif
#ifndef JR_TUNE_AUTO_BIN8_MAX_IMAGE    // bin size by image size when the caller does not choose (jr_softras_set_bin_size): 8-pixel bins up to this image size,
#define JR_TUNE_AUTO_BIN8_MAX_IMAGE 128
#endif
#ifndef JR_TUNE_AUTO_BIN16_MAX_IMAGE   // 16-pixel bins up to this one; above it 16 while the launch fits the multi-wavefront kernel (JR_TUNE_FWD_HEAVY_PIXELS), else 32
#define JR_TUNE_AUTO_BIN16_MAX_IMAGE 512
#endif
#ifndef JR_TUNE_FWD_HEAVY_WAVES8_BUDGET_SMALL   // the eight-wavefront budget of launches of up to JR_TUNE_FWD_WAVES8_SMALL_PIXELS pixels
#define JR_TUNE_FWD_HEAVY_WAVES8_BUDGET_SMALL 10240
#endif
#ifndef JR_TUNE_FWD_WAVES8_SMALL_PIXELS
#define JR_TUNE_FWD_WAVES8_SMALL_PIXELS 2621440
#endif
#ifndef JR_TUNE_FWD_HEAVY_DEFER_COPY // forward, heavy tiles: the record copies of a batch in one round after the list walk
#define JR_TUNE_FWD_HEAVY_DEFER_COPY 1
#endif
#ifndef JR_TUNE_FWD_HEAVY_OVERLAP // heavy tiles: wavefront 3 stages the next batch and wavefront 2 writes the next round's pair list WHILE wavefronts 0 / 1 apply (0: wavefront 0 does both between the passes)
#define JR_TUNE_FWD_HEAVY_OVERLAP 1
#endif
#ifndef JR_TUNE_FWD_HEAVY_PIPE    // heavy tiles as a pipeline: wavefronts 0 / 1 apply round n-1 while round n is evaluated around them (0: tile_heavy, passes in sequence)
#define JR_TUNE_FWD_HEAVY_PIPE 1
#endif
#ifndef JR_TUNE_FWD_HEAVY_WAVES   // wavefronts per workgroup of the multi-wavefront kernel when the heavy tiles are pipelined: 4, or 8 = eight where the host's policy says so (jr_api.cpp)
#define JR_TUNE_FWD_HEAVY_WAVES 8
#endif
#ifndef JR_TUNE_FWD_HEAVY_WAVES8_BUDGET   // eight wavefronts per heavy tile while (heavy tiles of the previous forward) x 8 wavefronts stay below this (the GPU holds 4096 at 16 per CU)
#define JR_TUNE_FWD_HEAVY_WAVES8_BUDGET 2048
#endif
#ifndef JR_TUNE_FWD_LIST_DEPTH    // forward, one wavefront per tile (and the sequential heavy tile): list chunks requested ahead of the one being culled
#define JR_TUNE_FWD_LIST_DEPTH 2
#endif
#ifndef JR_TUNE_FWD_PIPE_LIST_DEPTH   // pipelined heavy tile: list chunks requested ahead of the one wavefront 3 is culling (1 = as the other kernels)
#define JR_TUNE_FWD_PIPE_LIST_DEPTH 2
#endif
#ifndef JR_TUNE_FWD_PIPE8_CAP     // pipelined heavy tile with eight wavefronts: cells per round / record slots per batch (four wavefronts: 512 / 40 - what their LDS holds)
#define JR_TUNE_FWD_PIPE8_CAP 768
#endif
#ifndef JR_TUNE_FWD_PIPE8_BATCH
#define JR_TUNE_FWD_PIPE8_BATCH 64
#endif
#ifndef JR_TUNE_FWD_PIPE_CONSUMER_TASKS   // pipelined heavy tile: which applying wavefronts also take evaluate / mask tasks once their apply is done (bit 0: the K-buffer wavefront, bit 1: the colour wavefront)
#define JR_TUNE_FWD_PIPE_CONSUMER_TASKS 3
#endif
#ifndef JR_TUNE_FWD_HEAVY_PIXELS // forward: launches of up to this many pixels (B x IS x IS) use the four-wavefront kernel, larger ones one wavefront per tile
#define JR_TUNE_FWD_HEAVY_PIXELS 4194304
#endif
#ifndef JR_TUNE_DIAG             // diagnostic builds (WRONG results): bit 0 skips the forward's softmax update, bit 1 the K-buffer insert;
                                 // backward: bit 2 no three-projection path, bit 3 no n-th-holder search, bit 4 no row reduction, bit 5 no atomics, bit 7 atomics as plain stores, bit 8 no gathers;
                                 // multi-wavefront forward: bit 9 heavy tiles only, bit 10 light tiles only
#define JR_TUNE_DIAG 0
#endif
#ifndef JR_TUNE_BWD_ROW_RANGES   // backward: a row takes a contiguous quarter of the work items and adds up consecutive items of one face before its atomic
#define JR_TUNE_BWD_ROW_RANGES 1
#endif
#ifndef JR_TUNE_BWD_HASH_UNION   // backward (round 6): the faces a tile needs by hashing the pixels' buffered ids into an LDS table (0: per-lane sort + min-extraction, rounds 2 - 5)
#define JR_TUNE_BWD_HASH_UNION 1
#endif
#ifndef JR_TUNE_LIGHT_SYNC       // one-wavefront raster kernels (round 6): LDS hand-overs inside the wavefront by a compiler fence (LDS instructions of a wavefront
                                 // execute in order) instead of __syncthreads(), whose workgroup-scope fence waits for every outstanding store / atomic (vmcnt(0))
#define JR_TUNE_LIGHT_SYNC 1
#endif
#ifndef JR_TUNE_BIN_MATCH_ROUNDS  // wave_bin_match: ballot-matching rounds before the remaining lanes bump their counters one by one (4 = rounds 3 - 5)
#define JR_TUNE_BIN_MATCH_ROUNDS 16
#endif
#ifndef JR_TUNE_SETUP_WG         // k_face_setup: faces per workgroup (64 = one wavefront, LDS hand-overs without __syncthreads(); 128 = rounds 1 - 5)
#define JR_TUNE_SETUP_WG 64
#endif
#ifndef JR_TUNE_BWD_TV_RCP       // backward: edge-projection parameter by reciprocal multiply (gradient-only use)
#define JR_TUNE_BWD_TV_RCP 0
#endif

#ifndef JR_TUNE_FWD_EXACT         // forward colour path in the reference's own (slow) arithmetic: bit 0 coverage sigmoid, bit 1 softmax exponential (A/B for tools/grad_parity.py)
#define JR_TUNE_FWD_EXACT 0
#endif
#ifndef JR_TUNE_FWD_HARD_EXACT    // 'hard' alpha: the inside distance with the reference's IEEE quotients (its D > 0.5 decision rides on it); 0 = round 3's behaviour, for the A/B of what the uniform branch costs the other modes
#define JR_TUNE_FWD_HARD_EXACT 1
#endif
#ifndef JR_TUNE_BWD_EXACT         // backward: bit 0 coverage sigmoid, bit 1 softmax exponential, bit 2 IEEE divisions by ssum / D / (1 - D), bit 3 IEEE divisions by sigma / gamma / (near - far)
#define JR_TUNE_BWD_EXACT 0
#endif
#ifndef JR_TUNE_BWD_BATCH         // backward: faces per batch (LDS record slots per wavefront), <= 64; 40 slots + tables = 7.6 KB -> 20 wavefronts per CU
#define JR_TUNE_BWD_BATCH 40
#endif
#ifndef JR_TUNE_BWD_ONE_ATOMIC    // backward: grad_faces and grad_textures components of a flush in one atomic instruction (per-lane selected address)
#define JR_TUNE_BWD_ONE_ATOMIC 1
#endif
#ifndef JR_TUNE_BWD_WAVES64       // backward at 32 < K <= 64: wavefronts per SIMD (rounds 3 - 5: 3 -> 158 VGPRs, 4 spilled 52 B; round 6: 4 -> 128 VGPRs, euclidean 'softmax' keeps 36 B outside the pair loop)
#define JR_TUNE_BWD_WAVES64 4
#endif
#ifndef JR_TUNE_BWD_SPLIT         // backward: wavefronts per tile of a HEAVY bin in launches of up to BWD_SPLIT_PIXELS pixels (each keeps the ids with id % SPLIT == its part); 0 / 1 = off
#define JR_TUNE_BWD_SPLIT 4
#endif
#ifndef JR_TUNE_BWD_SPLIT8_PIXELS  // ... EIGHT wavefronts per heavy tile for launches of up to this many pixels (round 5)
#define JR_TUNE_BWD_SPLIT8_PIXELS 131072
#endif
#ifndef JR_TUNE_BWD_SPLIT_PIXELS
#define JR_TUNE_BWD_SPLIT_PIXELS 1572864
#endif
#ifndef JR_TUNE_BWD_TEX_LDS_MAX   // backward, 'surface' textures: faces of up to this many texels get their texture block staged in LDS with the record (0: never)
#define JR_TUNE_BWD_TEX_LDS_MAX 36
#endif
#ifndef JR_TUNE_BWD_TEX_LDS_PIXELS // ... in launches of up to this many pixels (the staging costs LDS: 40 x T x 12 B per wavefront)
#define JR_TUNE_BWD_TEX_LDS_PIXELS 4194304
#endif
#ifndef JR_TUNE_BWD_WAVES         // backward: wavefronts per SIMD asked of the register allocator at K <= 16 (5 -> 96 VGPRs, still 32 B of scratch;
                                  // 6 and 7 spill 64-112 B and are 1.5-2.2x slower: profiles/r02_ablation_sweep.log)
#define JR_TUNE_BWD_WAVES 5
#endif


#ifndef JR_TUNE_N3_PIXMAP_ALL     // NMR backward: all six (edge, axis) passes of a face in the lanes at once, out-walks four at a time (0: one pass after the other, one out-walk at a time)
#define JR_TUNE_N3_PIXMAP_ALL 1
#endif

#ifndef JR_TUNE_N3_PIXMAP_WAVES   // NMR pixel-map gradient: wavefronts per SIMD asked of the register allocator
#define JR_TUNE_N3_PIXMAP_WAVES 7
#endif
#ifndef JR_TUNE_N3_WALKS          // NMR pixel-map gradient: out-walks of a face taken at a time (their loads are independent)
#define JR_TUNE_N3_WALKS 4
#endif
#ifndef JR_TUNE_N3_LINE_WALKS     // NMR pixel-map gradient: the out-walks regrouped by scan line and run from an LDS copy of the line (0: every face walks its own lines through the L2s)
#define JR_TUNE_N3_LINE_WALKS 1
#endif
#ifndef JR_TUNE_N3_ZBUF_GROUP     // NMR z-buffer pass (round 5): faces per wavefront, set up in the lanes, walked one after the other (0: one wavefront per face, round 1 - 4)
#define JR_TUNE_N3_ZBUF_GROUP 8
#endif
#ifndef JR_TUNE_N3_FACE_FAST      // NMR per-face depth / texture gradient (round 5): reciprocal pixel mapping, transposing reductions (0: round 4)
#define JR_TUNE_N3_FACE_FAST 1
#endif
#ifndef JR_TUNE_N3_LINE_FAST      // NMR line walks (round 5): per-crossing sign of the eps, uniform trip counts, four groups of 64 pixels per trip (0: the round-4 loop)
#define JR_TUNE_N3_LINE_FAST 1
#endif
#ifndef JR_TUNE_N3_LINE_PARTS     // NMR line walks: sub-lists (= workgroups of the walk kernel) per scan line; power of two, 1024 crossings per line in total
#define JR_TUNE_N3_LINE_PARTS 8
#endif
#ifndef JR_TUNE_N3_XCD_GROUP      // NMR backward: runs of this many consecutive workgroups (4 faces each) go to one XCD (0: round-robin, the hardware's order)
#define JR_TUNE_N3_XCD_GROUP 32
#endif
#ifndef JR_TUNE_PROFILE_SECTIONS  // instrumented build: per-section shader-clock totals of the raster kernels (tools/ablate)
#define JR_TUNE_PROFILE_SECTIONS 0
#endif

#ifndef JR_TUNE_COUNT_PATHS       // instrumented builds: 1 = trips and lanes per region of the forward's raster loop (tools/sim/min_valu.py --measure), 2 = the backward's (min_valu_bwd.py)
#define JR_TUNE_COUNT_PATHS 0
#endif
#ifndef JR_TUNE_SECTIONS_WAVE     // instrumented build 2: which wavefront of the pipelined heavy tile's workgroup keeps the section clocks (0 K-buffer, 1 colour, 2 tasks, 3 stager)
#define JR_TUNE_SECTIONS_WAVE 0
#endif

namespace jr {
namespace tune {
constexpr int sections_wave = JR_TUNE_SECTIONS_WAVE;
constexpr bool profile_sections = JR_TUNE_PROFILE_SECTIONS != 0;
constexpr bool count_paths = JR_TUNE_COUNT_PATHS == 1;        // the forward's raster loop
constexpr bool count_paths_bwd = JR_TUNE_COUNT_PATHS == 2;    // the backward: tiles, passes, batches, trips, lanes, inside / slow trips (tools/sim/min_valu_bwd.py --measure)
constexpr bool n3_pixmap_all = JR_TUNE_N3_PIXMAP_ALL != 0;
constexpr int n3_xcd_group = JR_TUNE_N3_XCD_GROUP;
constexpr int n3_walks = JR_TUNE_N3_WALKS;
constexpr bool n3_line_walks = JR_TUNE_N3_LINE_WALKS != 0;
constexpr int n3_line_parts = JR_TUNE_N3_LINE_PARTS;
constexpr bool n3_line_fast = JR_TUNE_N3_LINE_FAST != 0;
constexpr int n3_zbuf_group = JR_TUNE_N3_ZBUF_GROUP;
constexpr bool n3_face_fast = JR_TUNE_N3_FACE_FAST != 0;
constexpr int bwd_batch = JR_TUNE_BWD_BATCH;
constexpr int fwd_exact = JR_TUNE_FWD_EXACT, bwd_exact = JR_TUNE_BWD_EXACT;
constexpr bool fwd_hard_exact = JR_TUNE_FWD_HARD_EXACT != 0;
#ifndef JR_TUNE_FWD_EXACT_INSIDE_SIGMA   // forward, euclidean distance: sigma_val below this runs the instantiations with IEEE quotients for inside pixels (DIST = 3)
#define JR_TUNE_FWD_EXACT_INSIDE_SIGMA 5e-6f
#endif
constexpr float fwd_exact_inside_sigma = JR_TUNE_FWD_EXACT_INSIDE_SIGMA;
constexpr bool fwd_dis_only = JR_TUNE_FWD_DIS_ONLY != 0;
constexpr bool fwd_prepass = JR_TUNE_FWD_PREPASS != 0;
constexpr bool fwd_inside_rcp = JR_TUNE_FWD_INSIDE_RCP != 0;
constexpr int fwd_batch = JR_TUNE_FWD_BATCH;
// record slots per wavefront by K capacity: at K <= 64 the registers cap both raster kernels at 3 wavefronts per SIMD (12 per CU), so 64
// slots (11.3 KB) cost no occupancy and save batch turn-arounds: forward 1.444 -> 1.393 ms, backward 1.548 -> 1.529 (round 5 call 17);
// at K <= 32 (4 per SIMD) 56 / 64 slots measured slower in both kernels
constexpr int fwd_batch_for(int kcap) { return kcap > 32 ? 64 : JR_TUNE_FWD_BATCH; }
#ifndef JR_TUNE_BWD_BATCH64      // backward at K > 32: record slots per wavefront.  Round 6: with the hashed union the K = 64 kernels fit 128 VGPRs, so 4 wavefronts
                                 // per SIMD pay when the LDS allows 16 per CU: 52 slots (9.7 KB) instead of 64 (11.3 KB: 13 per CU) - backward 1.536 -> 1.365 ms on the headline batch at K = 64
#define JR_TUNE_BWD_BATCH64 52
#endif
constexpr int bwd_batch_for(int kcap) { return kcap > 32 ? JR_TUNE_BWD_BATCH64 : JR_TUNE_BWD_BATCH; }
constexpr int fwd_batch_mixed = JR_TUNE_FWD_BATCH_MIXED;
constexpr int fwd_waves16 = JR_TUNE_FWD_WAVES16;
constexpr bool fwd_heavy_overlap = JR_TUNE_FWD_HEAVY_OVERLAP != 0;
constexpr bool fwd_heavy_pipe = JR_TUNE_FWD_HEAVY_PIPE != 0;
constexpr int fwd_pipe_consumer_tasks = JR_TUNE_FWD_PIPE_CONSUMER_TASKS;
constexpr int fwd_heavy_waves = JR_TUNE_FWD_HEAVY_WAVES;
constexpr int fwd_pipe8_cap = JR_TUNE_FWD_PIPE8_CAP, fwd_pipe8_batch = JR_TUNE_FWD_PIPE8_BATCH;
constexpr int fwd_pipe_list_depth = JR_TUNE_FWD_PIPE_LIST_DEPTH;
constexpr int fwd_list_depth = JR_TUNE_FWD_LIST_DEPTH;
constexpr long fwd_heavy_waves8_budget = JR_TUNE_FWD_HEAVY_WAVES8_BUDGET;
static_assert(fwd_heavy_waves == 4 || fwd_heavy_waves == 8, "JR_TUNE_FWD_HEAVY_WAVES");
constexpr int bwd_split = JR_TUNE_BWD_SPLIT;
constexpr int bwd_tex_lds_max = JR_TUNE_BWD_TEX_LDS_MAX;
constexpr long bwd_tex_lds_pixels = JR_TUNE_BWD_TEX_LDS_PIXELS;
constexpr bool bwd_one_atomic = JR_TUNE_BWD_ONE_ATOMIC != 0;
constexpr long bwd_split_pixels = JR_TUNE_BWD_SPLIT_PIXELS, bwd_split8_pixels = JR_TUNE_BWD_SPLIT8_PIXELS;
constexpr int fwd_waves32 = JR_TUNE_FWD_WAVES32;
constexpr int fwd_waves64 = JR_TUNE_FWD_WAVES64;
constexpr long fwd_heavy_pixels = JR_TUNE_FWD_HEAVY_PIXELS;
constexpr bool fwd_ids_global = JR_TUNE_FWD_IDS_GLOBAL != 0;
constexpr bool fwd_fill_shift = JR_TUNE_FWD_FILL_SHIFT != 0;
constexpr bool fwd_empty_bins = JR_TUNE_FWD_EMPTY_BINS != 0;
constexpr bool fwd_exp1 = JR_TUNE_FWD_EXP1 != 0;
constexpr int fwd_heavy = JR_TUNE_FWD_HEAVY, fwd_heavy16 = JR_TUNE_FWD_HEAVY16, fwd_heavy8 = JR_TUNE_FWD_HEAVY8;
constexpr int bin_fill_unroll_max_faces = JR_TUNE_BIN_FILL_UNROLL_MAX_FACES;
constexpr int auto_dense_faces_per_bin16 = JR_TUNE_AUTO_DENSE_FACES_PER_BIN16, fwd_waves8_mean_list = JR_TUNE_FWD_WAVES8_MEAN_LIST;
constexpr int fwd_heavy16_small_mesh = JR_TUNE_FWD_HEAVY16_SMALL_MESH, small_mesh_faces = JR_TUNE_SMALL_MESH_FACES;
constexpr long fwd_heavy_waves8_budget_small = JR_TUNE_FWD_HEAVY_WAVES8_BUDGET_SMALL, fwd_waves8_small_pixels = JR_TUNE_FWD_WAVES8_SMALL_PIXELS;
constexpr int auto_bin8_max_image = JR_TUNE_AUTO_BIN8_MAX_IMAGE, auto_bin16_max_image = JR_TUNE_AUTO_BIN16_MAX_IMAGE;
constexpr bool fwd_heavy_defer_copy = JR_TUNE_FWD_HEAVY_DEFER_COPY != 0;
constexpr bool bwd_tv_rcp = JR_TUNE_BWD_TV_RCP != 0;
constexpr bool bwd_hash_union = JR_TUNE_BWD_HASH_UNION != 0;
constexpr bool light_sync = JR_TUNE_LIGHT_SYNC != 0;
constexpr bool bwd_row_ranges = JR_TUNE_BWD_ROW_RANGES != 0;
}  // namespace tune
}  // namespace jr
