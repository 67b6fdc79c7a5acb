"""Ablation variants of libjrender_hip.so: name -> JR_TUNE_* defines (csrc/jr_tuning.h).
`python tools/ablate/build.py [names...]` builds them next to the product library (CPU box, hipcc);
`python tools/ablate/run.py [names...]` (GPU box) checks parity and times each one in ONE process
sequence on ONE box (boxes differ by a few per cent: only numbers from one call are comparable)."""
# (the round-1 kernels also tested every listed face's box against the tile: that switch left the tree, patches/dead_switches_r03.patch)
OFF = ["JR_TUNE_FWD_BATCH=64", "JR_TUNE_BWD_BATCH=64", "JR_TUNE_FWD_PREPASS=0", "JR_TUNE_FWD_DIS_ONLY=0", "JR_TUNE_FWD_WAVES16=1", "JR_TUNE_FWD_INSIDE_RCP=0", "JR_TUNE_FWD_IDS_GLOBAL=0",
       "JR_TUNE_FWD_FILL_SHIFT=0", "JR_TUNE_FWD_EXP1=0", "JR_TUNE_FWD_HEAVY=0"]
VARIANTS = {
    "product": [],                                           # the defaults of jr_tuning.h
    "r1": OFF,                                               # every switch off = round-1 kernels
    "r1_occ4": OFF[:4] + ["JR_TUNE_FWD_WAVES16=4"] + OFF[5:],      # + the 4-waves-per-SIMD request alone
    "no_prepass": ["JR_TUNE_FWD_PREPASS=0", "JR_TUNE_FWD_HEAVY=0"],
    "no_dis": ["JR_TUNE_FWD_DIS_ONLY=0"],
    "bwd_rcp": ["JR_TUNE_BWD_TV_RCP=1"],                     # dead: breaks the 1e-4 gradient bar
    "no_inside_rcp": ["JR_TUNE_FWD_INSIDE_RCP=0"],
    "bwd64": ["JR_TUNE_BWD_BATCH=64"],                       # 13 wavefronts per CU (round 1)
    "bwd48": ["JR_TUNE_BWD_BATCH=48"],
    "bwd56": ["JR_TUNE_BWD_BATCH=56"],
    "bwd44": ["JR_TUNE_BWD_BATCH=44"],
    "bwd40": ["JR_TUNE_BWD_BATCH=40"],
    "bwd44w5": ["JR_TUNE_BWD_BATCH=44", "JR_TUNE_BWD_WAVES=5"],
    "bwd40w5": ["JR_TUNE_BWD_BATCH=40", "JR_TUNE_BWD_WAVES=5"],
    "bwd36w6": ["JR_TUNE_BWD_BATCH=36", "JR_TUNE_BWD_WAVES=6"],
    "bwd32w6": ["JR_TUNE_BWD_BATCH=32", "JR_TUNE_BWD_WAVES=6"],
    "bwd36w5": ["JR_TUNE_BWD_BATCH=36", "JR_TUNE_BWD_WAVES=5"],
    "bwd32w5": ["JR_TUNE_BWD_BATCH=32", "JR_TUNE_BWD_WAVES=5"],
    "bwd28w7": ["JR_TUNE_BWD_BATCH=28", "JR_TUNE_BWD_WAVES=7"],
    "bwd52w4": ["JR_TUNE_BWD_BATCH=52", "JR_TUNE_BWD_WAVES=4"],   # round-2 start
    "bwd34w6": ["JR_TUNE_BWD_BATCH=34", "JR_TUNE_BWD_WAVES=6"],
    "bwd48w5": ["JR_TUNE_BWD_BATCH=48", "JR_TUNE_BWD_WAVES=5"],
    "ids_regs": ["JR_TUNE_FWD_IDS_GLOBAL=0"],                # K-buffer ids in registers (round-2 start)
    "no_shift": ["JR_TUNE_FWD_FILL_SHIFT=0"],                # round 3: K-buffer appends by per-lane slot select instead of the register shift
    "no_exp1": ["JR_TUNE_FWD_EXP1=0"],                       # round 3: two v_exp per softmax update
    "h128": ["JR_TUNE_FWD_HEAVY=128"], "h256": ["JR_TUNE_FWD_HEAVY=256"], "h384": ["JR_TUNE_FWD_HEAVY=384"],
    "h64": ["JR_TUNE_FWD_HEAVY=64"], "h768": ["JR_TUNE_FWD_HEAVY=768"], "h1024": ["JR_TUNE_FWD_HEAVY=1024"],
    "h1024all": ["JR_TUNE_FWD_HEAVY=1024", "JR_TUNE_FWD_HEAVY_PIXELS=1000000000"],
    "diag_nosoftmax": ["JR_TUNE_DIAG=1", "JR_TUNE_FWD_HEAVY=0"],   # WRONG results: cost probes of the single-wavefront path
    "diag_nokbuf": ["JR_TUNE_DIAG=2", "JR_TUNE_FWD_HEAVY=0"],
    "diag_neither": ["JR_TUNE_DIAG=3", "JR_TUNE_FWD_HEAVY=0"],
    "sh_w1": ["JR_TUNE_PROFILE_SECTIONS=2", "JR_TUNE_SECTIONS_WAVE=1"], "sh_w2": ["JR_TUNE_PROFILE_SECTIONS=2", "JR_TUNE_SECTIONS_WAVE=2"], "sh_w3": ["JR_TUNE_PROFILE_SECTIONS=2", "JR_TUNE_SECTIONS_WAVE=3"],   # ... of the colour / a task / the staging wavefront
    "diag_heavy_only": ["JR_TUNE_DIAG=512"], "diag_light_only": ["JR_TUNE_DIAG=1024"],   # WRONG images: the multi-wavefront forward with only its heavy / only its light tiles (their makespans)
    "sections_heavy": ["JR_TUNE_PROFILE_SECTIONS=2"],       # instrumented: wavefront 0 of the heaviest bin's tiles (tools/ablate/sections.py --heavy)
    "r2fwd": ["JR_TUNE_FWD_FILL_SHIFT=0", "JR_TUNE_FWD_EXP1=0", "JR_TUNE_FWD_HEAVY=0", "JR_TUNE_FWD_BATCH=56", "JR_TUNE_FWD_WAVES16=4"],   # the round-2 forward
    "h0": ["JR_TUNE_FWD_HEAVY=0"],                           # round 3: one wavefront per tile whatever the size of the launch
    "hall": ["JR_TUNE_FWD_HEAVY_PIXELS=1000000000"],         # round 3: the four-wavefront kernel whatever the size of the launch
    "hnone": ["JR_TUNE_FWD_HEAVY=1000000", "JR_TUNE_FWD_HEAVY_PIXELS=1000000000"],   # four tiles per workgroup, no heavy bin at all: what does that organisation cost?
    "w4b56": ["JR_TUNE_FWD_WAVES16=4", "JR_TUNE_FWD_BATCH=56", "JR_TUNE_FWD_HEAVY=0"],   # single-wavefront forward at 4 wavefronts per SIMD (round 2) / 5 with other batch sizes / 6
    "w5b44": ["JR_TUNE_FWD_WAVES16=5", "JR_TUNE_FWD_BATCH=44", "JR_TUNE_FWD_HEAVY=0"],
    "w6b36": ["JR_TUNE_FWD_WAVES16=6", "JR_TUNE_FWD_BATCH=36", "JR_TUNE_FWD_HEAVY=0"],
    "bdiag_noinside": ["JR_TUNE_DIAG=4"], "bdiag_nosearch": ["JR_TUNE_DIAG=8"], "bdiag_noreduce": ["JR_TUNE_DIAG=16"], "bdiag_noatomics": ["JR_TUNE_DIAG=32"],   # WRONG results: cost probes of the backward
    "no_ranges": ["JR_TUNE_BWD_ROW_RANGES=0"],              # round 3: backward rows take every fourth item, one atomic per item
    "diag_nostore": ["JR_TUNE_DIAG=64", "JR_TUNE_FWD_HEAVY=0"],   # WRONG results: forward without the per-insert id stores
    "no_defer_copy": ["JR_TUNE_FWD_HEAVY_DEFER_COPY=0"],     # round 3: heavy tiles copy their records chunk by chunk while walking the list
    "no_empty_bins": ["JR_TUNE_FWD_EMPTY_BINS=0"],           # round 3: the 16 tiles of an empty bin store their own outputs
    "fwd_k32w3_k64w2": ["JR_TUNE_FWD_WAVES32=3", "JR_TUNE_FWD_WAVES64=2"],   # round 3: K > 16 forward at the wavefront counts it had before (131 / 187 VGPRs)
    "no_bwd_split": ["JR_TUNE_BWD_SPLIT=0"],                 # round 3: one wavefront per backward tile whatever the launch
    "bwd_split2": ["JR_TUNE_BWD_SPLIT=2"], "bwd_split8": ["JR_TUNE_BWD_SPLIT=8"],
    "no_heavy_overlap": ["JR_TUNE_FWD_HEAVY_OVERLAP=0"],     # round 3: heavy tiles stage / list between the passes (wavefront 0) instead of during the apply (wavefronts 3 / 2)
    "bwd_k64w4": ["JR_TUNE_BWD_WAVES64=4"],                  # round 3: backward at K = 64 with 4 wavefronts per SIMD (52 B of scratch)
    "sections_sorted": ["JR_TUNE_PROFILE_SECTIONS=1", "JR_TUNE_FWD_HEAVY=0", "JR_TUNE_BWD_HASH_UNION=0"],
    "hashv1": [],                                           # (round 6 call 2: a saved build of the first hashed union - one dependent LDS round trip per id plane; not reproducible from a define)
    "count_paths_bwd": ["JR_TUNE_COUNT_PATHS=2"],           # instrumented: tools/sim/min_valu_bwd.py --measure
    "bwd_k64w4b52": ["JR_TUNE_BWD_WAVES64=4", "JR_TUNE_BWD_BATCH64=52"],   # round 6: K = 64 backward fits 128 VGPRs with the hashed union: 4 wavefronts per SIMD need <= 10 KB of LDS
    "prev": [],                                             # (round 6 call 6: a saved build from before the texel-load wait fix / light sync)
    "heavy_sync": ["JR_TUNE_LIGHT_SYNC=0"],                 # round 6: __syncthreads() (workgroup-scope fence: vmcnt(0)) for the LDS hand-overs of the one-wavefront raster kernels
    "setup128": ["JR_TUNE_SETUP_WG=128"],                   # round 6: k_face_setup with 128-face workgroups and __syncthreads() (rounds 1 - 5)
    "diag_noempty": ["JR_TUNE_DIAG=2048"],                  # WRONG images: the headline forward without the empty bins' output stores (what do they cost the launch?)
    "match4": ["JR_TUNE_BIN_MATCH_ROUNDS=4"], "match8": ["JR_TUNE_BIN_MATCH_ROUNDS=8"], "match32": ["JR_TUNE_BIN_MATCH_ROUNDS=32"], "match64": ["JR_TUNE_BIN_MATCH_ROUNDS=64"],   # round 6 calls 16 / 17: ballot-matching rounds of wave_bin_match (product 16; rounds 3 - 5: 4) - fewer device-scope atomics
    "sd_noinfo": ["JR_TUNE_DIAG=4096"], "sd_nogeo": ["JR_TUNE_DIAG=8192"], "sd_nocount": ["JR_TUNE_DIAG=16384"], "sd_norange": ["JR_TUNE_DIAG=49152"], "sd_compute": ["JR_TUNE_DIAG=61440"],   # WRONG results: cost probes of k_face_setup (round 6 call 15)
    "bwd_sorted": ["JR_TUNE_BWD_HASH_UNION=0"],             # round 6: the backward's face union by per-lane sort + min-extraction (rounds 2 - 5) instead of the LDS hash table
    "no_heavy_pipe": ["JR_TUNE_FWD_HEAVY_PIPE=0"],           # round 3: heavy tiles with the passes in sequence (tile_heavy) instead of the pipeline
    "pipe_ct2": ["JR_TUNE_FWD_PIPE_CONSUMER_TASKS=2"], "pipe_ct0": ["JR_TUNE_FWD_PIPE_CONSUMER_TASKS=0"],   # round 3: the K-buffer wavefront / both applying wavefronts take no evaluate tasks
    "pipe_nw4": ["JR_TUNE_FWD_HEAVY_WAVES=4"],               # round 3: the pipelined heavy tile with four wavefronts per workgroup instead of eight
    "pipe_ld1": ["JR_TUNE_FWD_PIPE_LIST_DEPTH=1"], "pipe_ld4": ["JR_TUNE_FWD_PIPE_LIST_DEPTH=4"],   # round 3: list chunks in flight ahead of wavefront 3's cull (2 in the product)
    "p8_c512b40": ["JR_TUNE_FWD_PIPE8_CAP=512", "JR_TUNE_FWD_PIPE8_BATCH=40"], "p8_c1024": ["JR_TUNE_FWD_PIPE8_CAP=1024"], "p8_c640": ["JR_TUNE_FWD_PIPE8_CAP=640"], "p8_c896": ["JR_TUNE_FWD_PIPE8_CAP=896"], "p8_b48": ["JR_TUNE_FWD_PIPE8_BATCH=48"], "p8_b56": ["JR_TUNE_FWD_PIPE8_BATCH=56"],   # round 3: round / batch sizes of the eight-wavefront pipeline
    "bwd_two_atomics": ["JR_TUNE_BWD_ONE_ATOMIC=0"],         # round 3: one atomic instruction per output buffer and flush
    "bdiag_stores": ["JR_TUNE_DIAG=128"],                    # WRONG results: the backward's atomics as plain stores
    "fwd_ld1": ["JR_TUNE_FWD_LIST_DEPTH=1"], "fwd_ld3": ["JR_TUNE_FWD_LIST_DEPTH=3"],   # round 3: list chunks in flight ahead of a single-wavefront tile's cull (2 in the product)
    "w5b40": ["JR_TUNE_FWD_BATCH=40"], "w5b44": ["JR_TUNE_FWD_BATCH=44"], "w4b56h0": ["JR_TUNE_FWD_WAVES16=4", "JR_TUNE_FWD_BATCH=56"],   # round 3, after the list prefetch: batch sizes around the product's 46 slots / 5 wavefronts per SIMD
    # round 4: what do the colour-path / gradient-only approximations cost in gradient parity (tools/grad_parity.py)?  The reference's own arithmetic instead:
    "bx1": ["JR_TUNE_BWD_EXACT=1"], "bx2": ["JR_TUNE_BWD_EXACT=2"], "bx4": ["JR_TUNE_BWD_EXACT=4"], "bx8": ["JR_TUNE_BWD_EXACT=8"], "bx15": ["JR_TUNE_BWD_EXACT=15"],
    "fx1": ["JR_TUNE_FWD_EXACT=1"], "fx2": ["JR_TUNE_FWD_EXACT=2"], "fx3": ["JR_TUNE_FWD_EXACT=3"], "xall": ["JR_TUNE_FWD_EXACT=3", "JR_TUNE_BWD_EXACT=15"],
    "n3_seq": ["JR_TUNE_N3_PIXMAP_ALL=0"],                  # round 4: NMR pixel-map gradient one (edge, axis) pass and one out-walk at a time (rounds 1-3)
    "bdiag_nogather": ["JR_TUNE_DIAG=256"], "bdiag_nogather_nosearch": ["JR_TUNE_DIAG=264"],   # WRONG results: the backward without its 13 ds_bpermute gathers (and without the n-th-holder search)
    "n3_rr": ["JR_TUNE_N3_XCD_GROUP=0"], "n3_g8": ["JR_TUNE_N3_XCD_GROUP=8"], "n3_g128": ["JR_TUNE_N3_XCD_GROUP=128"], "n3_g512": ["JR_TUNE_N3_XCD_GROUP=512"],   # round 4: NMR pixel-map gradient: runs of G workgroups per XCD (0 = round-robin)
    "n3_w5": ["JR_TUNE_N3_PIXMAP_WAVES=5"], "n3_w6": ["JR_TUNE_N3_PIXMAP_WAVES=6"],   # round 4: NMR pixel-map gradient at 5 / 6 wavefronts per SIMD (12 / 72 B of scratch; product: 4, none)
    "n3_walks2": ["JR_TUNE_N3_WALKS=2"], "n3_walks8": ["JR_TUNE_N3_WALKS=8"], "n3_walks1": ["JR_TUNE_N3_WALKS=1"],   # round 4: NMR out-walks in flight per wavefront (product: 4)
    "n3diag_noout": ["JR_TUNE_DIAG=2048"], "n3diag_noin": ["JR_TUNE_DIAG=4096"], "n3diag_nowalks": ["JR_TUNE_DIAG=6144"],   # WRONG gradients: the NMR pixel-map kernel without its out / in walks
    "n3_face_r4": ["JR_TUNE_N3_FACE_FAST=0"],           # round 5: NMR per-face depth / texture kernel with 64-bit pixel divisions and 33 separate wave reductions (round 4)
    "n3_zb_r4": ["JR_TUNE_N3_ZBUF_GROUP=0"], "n3_zb4": ["JR_TUNE_N3_ZBUF_GROUP=4"], "n3_zb16": ["JR_TUNE_N3_ZBUF_GROUP=16"], "n3_zb32": ["JR_TUNE_N3_ZBUF_GROUP=32"],   # round 5: NMR z-buffer pass with one wavefront per face (rounds 1 - 4) / 4 / 16 / 32 faces per wavefront (product: 8)
    "fwd52": ["JR_TUNE_FWD_BATCH=52"], "fwd56": ["JR_TUNE_FWD_BATCH=56"], "fwd64": ["JR_TUNE_FWD_BATCH=64"],   # round 5: record slots per wavefront of the one-wavefront forward for K > 16, where the VGPRs - not LDS - bound the occupancy (time with --K 32 / --K 64; at K = 16 these cost occupancy)
    "n3_line_r4": ["JR_TUNE_N3_LINE_FAST=0"],             # round 5: the round-4 walk loop of k_n3mr_backward_line_walks (per-pixel eps sign, per-lane trip count)
    "n3_face_walks": ["JR_TUNE_N3_LINE_WALKS=0", "JR_TUNE_N3_PIXMAP_WAVES=4"], "n3_lp1": ["JR_TUNE_N3_LINE_PARTS=1"], "n3_lp4": ["JR_TUNE_N3_LINE_PARTS=4"], "n3_lp16": ["JR_TUNE_N3_LINE_PARTS=16"], "n3_lw4": ["JR_TUNE_N3_PIXMAP_WAVES=4"],           # round 4: NMR out-walks by the per-face kernel through the L2s (before the per-line regrouping)
    "hard_exact_off": ["JR_TUNE_FWD_HARD_EXACT=0"],          # round 4: what does the uniform 'hard alpha -> IEEE inside distance' branch cost the default modes?
    # round 5
    "count_paths": ["JR_TUNE_COUNT_PATHS=1", "JR_TUNE_FWD_HEAVY=0"],   # instrumented: trips / lanes per region of the raster loop (tools/sim/min_valu.py --measure)
    "sections_setup": ["JR_TUNE_PROFILE_SECTIONS=3"],     # instrumented: k_face_setup / k_bin_fill section clocks (tools/ablate/sections.py --setup)
    "base": [],                                              # a library built from another commit, copied to libjrender_hip_base.so by hand
    "sections": ["JR_TUNE_PROFILE_SECTIONS=1", "JR_TUNE_FWD_HEAVY=0"],              # instrumented: tools/ablate/sections.py
}
