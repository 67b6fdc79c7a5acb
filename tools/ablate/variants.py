"""Ablation variants of libjrender_hip.so: name -> JR_TUNE_* defines (csrc/jr_tuning.h).
`python tools/ablate/build.py [names...]` builds them next to the product library (CPU box, hipcc);
`python tools/ablate/run.py [names...]` (GPU box) checks parity and times each one in ONE process
sequence on ONE box (boxes differ by a few per cent: only numbers from one call are comparable)."""
OFF = ["JR_TUNE_FWD_BATCH=64", "JR_TUNE_BWD_BATCH=64", "JR_TUNE_FWD_TILE_BOXTEST=1", "JR_TUNE_FWD_PREPASS=0", "JR_TUNE_FWD_DIS_ONLY=0", "JR_TUNE_FWD_OCC4=0", "JR_TUNE_FWD_IDS_LDS_BIGK=0", "JR_TUNE_FWD_INSIDE_RCP=0"]
VARIANTS = {
    "product": [],                                           # the defaults of jr_tuning.h
    "r1": OFF,                                               # every switch off = round-1 kernels
    "r1_occ4": OFF[:5] + OFF[6:],                            # + the 4-waves-per-SIMD request alone
    "no_prepass": ["JR_TUNE_FWD_PREPASS=0", "JR_TUNE_FWD_OCC4=0"],
    "no_dis": ["JR_TUNE_FWD_DIS_ONLY=0"],
    "tv": ["JR_TUNE_TV_DIVKNOWN=1"],                         # dead: refinement quotient with the stored reciprocal for tv (fwd +0.5 %, bwd +1.2 %)
    "ids_lds": ["JR_TUNE_FWD_IDS_LDS=1"],                    # dead at K <= 16: K-buffer ids in LDS
    "bigk_regs": ["JR_TUNE_FWD_IDS_LDS_BIGK=0"],             # K > 16 with ids in registers (round 1)
    "bwd_rcp": ["JR_TUNE_BWD_TV_RCP=1"],                     # dead: breaks the 1e-4 gradient bar
    "no_inside_rcp": ["JR_TUNE_FWD_INSIDE_RCP=0"],
    "boxtest": ["JR_TUNE_FWD_TILE_BOXTEST=1"],
    "tpw2": ["JR_TUNE_FWD_TPW=2"],
    "tpw4": ["JR_TUNE_FWD_TPW=4"],
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
    "kbuf_salu": ["JR_TUNE_FWD_KBUF_SALU=1"],
    "bankmask": ["JR_TUNE_BWD_REDUCE_BANKMASK=1"],          # dead (+2 % bwd): bank-masked DPP adds instead of selects — v_add_f32_dpp costs what v_cndmask costs
    "base": [],                                              # a library built from another commit, copied to libjrender_hip_base.so by hand
    "sections": ["JR_TUNE_PROFILE_SECTIONS=1"],              # instrumented: tools/ablate/sections.py
}
