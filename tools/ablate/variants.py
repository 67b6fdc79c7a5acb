"""Ablation variants of libjrender_hip.so: name -> JR_TUNE_* defines (csrc/jr_tuning.h).
`python tools/ablate/build.py [names...]` builds them next to the product library (CPU box, hipcc);
`python tools/ablate/run.py [names...]` (GPU box) checks parity and times each one in ONE process
sequence on ONE box (boxes differ by a few per cent: only numbers from one call are comparable)."""
# (the round-1 kernels also tested every listed face's box against the tile: that switch left the tree, patches/dead_switches_r03.patch)
OFF = ["JR_TUNE_FWD_BATCH=64", "JR_TUNE_BWD_BATCH=64", "JR_TUNE_FWD_PREPASS=0", "JR_TUNE_FWD_DIS_ONLY=0", "JR_TUNE_FWD_OCC4=0", "JR_TUNE_FWD_INSIDE_RCP=0", "JR_TUNE_FWD_IDS_GLOBAL=0"]
VARIANTS = {
    "product": [],                                           # the defaults of jr_tuning.h
    "r1": OFF,                                               # every switch off = round-1 kernels
    "r1_occ4": OFF[:4] + OFF[5:],                            # + the 4-waves-per-SIMD request alone
    "no_prepass": ["JR_TUNE_FWD_PREPASS=0", "JR_TUNE_FWD_OCC4=0"],
    "no_dis": ["JR_TUNE_FWD_DIS_ONLY=0"],
    "bwd_rcp": ["JR_TUNE_BWD_TV_RCP=1"],                     # dead: breaks the 1e-4 gradient bar
    "no_inside_rcp": ["JR_TUNE_FWD_INSIDE_RCP=0"],
    "fwd64": ["JR_TUNE_FWD_BATCH=64"],                       # 64 record slots: 14 wavefronts per CU (round 1)
    "fwd48": ["JR_TUNE_FWD_BATCH=48"],
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
    "fwd44w5": ["JR_TUNE_FWD_BATCH=44", "JR_TUNE_FWD_OCC4=5"],
    "fwd52": ["JR_TUNE_FWD_BATCH=52"],
    "ids_regs": ["JR_TUNE_FWD_IDS_GLOBAL=0"],                # K-buffer ids in registers (round-2 start)
    "fwd44w5g": ["JR_TUNE_FWD_BATCH=44", "JR_TUNE_FWD_OCC4=5"],  # with the ids out of the registers: 96 VGPRs, 20 B of scratch outside the trip loop
    "no_shift": ["JR_TUNE_FWD_FILL_SHIFT=0"],                # round 3: K-buffer appends by per-lane slot select instead of the register shift
    "no_defer": ["JR_TUNE_FWD_DEFER_INSIDE=0"],              # round 3: inside pairs evaluated in the main raster loop
    "no_exp1": ["JR_TUNE_FWD_EXP1=0"],                       # round 3: two v_exp per softmax update
    "r2fwd": ["JR_TUNE_FWD_FILL_SHIFT=0", "JR_TUNE_FWD_DEFER_INSIDE=0", "JR_TUNE_FWD_EXP1=0"],   # the round-2 forward
    "prio256": ["JR_TUNE_FWD_PRIO=256", "JR_TUNE_FWD_DEFER_INSIDE=0"],
    "prio600": ["JR_TUNE_FWD_PRIO=600", "JR_TUNE_FWD_DEFER_INSIDE=0"],
    "occ3": ["JR_TUNE_FWD_OCC4=3", "JR_TUNE_FWD_DEFER_INSIDE=0"],       # diagnostic: 3 / 2 wavefronts per SIMD in the forward
    "occ2": ["JR_TUNE_FWD_OCC4=2", "JR_TUNE_FWD_DEFER_INSIDE=0"],
    "base": [],                                              # a library built from another commit, copied to libjrender_hip_base.so by hand
    "sections": ["JR_TUNE_PROFILE_SECTIONS=1", "JR_TUNE_FWD_DEFER_INSIDE=0"],              # instrumented: tools/ablate/sections.py
}
