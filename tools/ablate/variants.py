"""Ablation variants of libjrender_hip.so: name -> JR_TUNE_* defines (csrc/jr_tuning.h).
`python tools/ablate/build.py [names...]` builds them next to the product library (CPU box, hipcc);
`python tools/ablate/run.py [names...]` (GPU box) checks parity and times each one in ONE process
sequence on ONE box (boxes differ by a few per cent: only numbers from one call are comparable)."""
VARIANTS = {
    "base": [],
    "base_noocc": ["JR_TUNE_FWD_OCC4=0"],                    # without the waves-per-SIMD request (round-1 register allocation)
    "pre_dis": ["JR_TUNE_FWD_PREPASS=1", "JR_TUNE_FWD_DIS_ONLY=1"],
    "pre_noocc": ["JR_TUNE_FWD_PREPASS=1", "JR_TUNE_FWD_OCC4=0"],                                              # every switch off = round-1 kernels
    "tv": ["JR_TUNE_TV_DIVKNOWN=1"],
    "dis": ["JR_TUNE_FWD_DIS_ONLY=1"],
    "tvdis": ["JR_TUNE_TV_DIVKNOWN=1", "JR_TUNE_FWD_DIS_ONLY=1"],
    "pre": ["JR_TUNE_FWD_PREPASS=1"],
    "idslds": ["JR_TUNE_FWD_IDS_LDS=1"],
    "bwdrcp": ["JR_TUNE_BWD_TV_RCP=1"],
    "all": ["JR_TUNE_TV_DIVKNOWN=1", "JR_TUNE_FWD_DIS_ONLY=1", "JR_TUNE_FWD_PREPASS=1", "JR_TUNE_BWD_TV_RCP=1"],
    "all_ids": ["JR_TUNE_TV_DIVKNOWN=1", "JR_TUNE_FWD_DIS_ONLY=1", "JR_TUNE_FWD_PREPASS=1", "JR_TUNE_BWD_TV_RCP=1",
                "JR_TUNE_FWD_IDS_LDS=1"],
}
