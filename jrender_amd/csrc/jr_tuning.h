// Tuning switches of the raster kernels.  The product is built with the defaults below; tools/ablate/
// builds variants with -DJR_TUNE_<X>=0|1 (python -m jrender_amd._build --variant NAME -D...) to reproduce
// the A/B tables of DESIGN.md on one GPU box.  Every switch selects between two EXACTNESS-EQUIVALENT
// implementations (same face-index buffer bits, colours / gradients within tolerance).
#pragma once

#ifndef JR_TUNE_TV_DIVKNOWN      // edge-projection parameter: refinement quotient with the record's RN(1/Dn) instead of IEEE '/' (same bits).
                                 // DEAD twice: +3 % with the reciprocal formed on the fly, +0.5 % / +1.2 % with it stored in the record
#define JR_TUNE_TV_DIVKNOWN 0
#endif
#ifndef JR_TUNE_FWD_DIS_ONLY     // forward: carry only (sign, dis) out of the distance machinery
#define JR_TUNE_FWD_DIS_ONLY 1
#endif
#ifndef JR_TUNE_FWD_PREPASS      // forward: conservative half-plane pre-cull of (pixel, face) pairs, lane = face
#define JR_TUNE_FWD_PREPASS 1
#endif
#ifndef JR_TUNE_FWD_IDS_LDS      // forward: K-buffer ids live in LDS (one ds_write per insert), depths stay in VGPRs
#define JR_TUNE_FWD_IDS_LDS 0
#endif
#ifndef JR_TUNE_FWD_IDS_LDS_BIGK // forward: the same for K > 16 only (where the registers no longer hold ids + depths)
#define JR_TUNE_FWD_IDS_LDS_BIGK 1
#endif
#ifndef JR_TUNE_FWD_INSIDE_RCP   // forward: 2nd / 3rd edge projection of INSIDE pixels (colour path only) by reciprocal multiply
#define JR_TUNE_FWD_INSIDE_RCP 1
#endif
#ifndef JR_TUNE_FWD_TPW          // forward: tiles of a bin rendered by one wavefront, one after the other (1, 2, 4, 8, 16)
#define JR_TUNE_FWD_TPW 1
#endif
#ifndef JR_TUNE_FWD_TILE_BOXTEST // forward: load every listed face's box and test it against the tile before staging (round 1)
#define JR_TUNE_FWD_TILE_BOXTEST 0
#endif
#ifndef JR_TUNE_FWD_BATCH        // forward: faces per batch (LDS record slots per wavefront), <= 64; 56 x 176 B = 9.6 KB -> 16 wavefronts per CU
#define JR_TUNE_FWD_BATCH 56
#endif
#ifndef JR_TUNE_FWD_KBUF_SALU    // forward: K-buffer slot masks from 4 bit ballots + scalar logic instead of 16 v_cmp
#define JR_TUNE_FWD_KBUF_SALU 0
#endif
#ifndef JR_TUNE_FWD_IDS_GLOBAL   // forward, K <= 16: K-buffer ids are stored straight into faces_id_buffer at every insert instead of living in registers
#define JR_TUNE_FWD_IDS_GLOBAL 1
#endif
#ifndef JR_TUNE_FWD_OCC4         // forward: ask the register allocator for 4 wavefronts per SIMD at K <= 16 (128 VGPRs)
#define JR_TUNE_FWD_OCC4 1
#endif
#ifndef JR_TUNE_BWD_TV_RCP       // backward: edge-projection parameter by reciprocal multiply (gradient-only use)
#define JR_TUNE_BWD_TV_RCP 0
#endif

#ifndef JR_TUNE_BWD_BATCH         // backward: faces per batch (LDS record slots per wavefront), <= 64; 40 slots + tables = 7.6 KB -> 20 wavefronts per CU
#define JR_TUNE_BWD_BATCH 40
#endif
#ifndef JR_TUNE_BWD_WAVES         // backward: wavefronts per SIMD asked of the register allocator at K <= 16 (5 -> 96 VGPRs, still 32 B of scratch;
                                  // 6 and 7 spill 64-112 B and are 1.5-2.2x slower: profiles/r02_ablation_sweep.log)
#define JR_TUNE_BWD_WAVES 5
#endif

#ifndef JR_TUNE_BWD_REDUCE_BANKMASK // backward: row transpose-reduction with bank-masked DPP adds instead of selects
#define JR_TUNE_BWD_REDUCE_BANKMASK 0
#endif

#ifndef JR_TUNE_PROFILE_SECTIONS  // instrumented build: per-section shader-clock totals of the raster kernels (tools/ablate)
#define JR_TUNE_PROFILE_SECTIONS 0
#endif

namespace jr {
namespace tune {
constexpr bool profile_sections = JR_TUNE_PROFILE_SECTIONS != 0;
constexpr bool bwd_reduce_bankmask = JR_TUNE_BWD_REDUCE_BANKMASK != 0;
constexpr int bwd_batch = JR_TUNE_BWD_BATCH;
constexpr bool tv_divknown = JR_TUNE_TV_DIVKNOWN != 0;
constexpr bool fwd_dis_only = JR_TUNE_FWD_DIS_ONLY != 0;
constexpr bool fwd_prepass = JR_TUNE_FWD_PREPASS != 0;
constexpr bool fwd_ids_lds = JR_TUNE_FWD_IDS_LDS != 0;
constexpr bool fwd_ids_lds_bigk = JR_TUNE_FWD_IDS_LDS_BIGK != 0;
constexpr bool fwd_inside_rcp = JR_TUNE_FWD_INSIDE_RCP != 0;
constexpr int fwd_tiles_per_wave = JR_TUNE_FWD_TPW;
constexpr int fwd_batch = JR_TUNE_FWD_BATCH;
constexpr bool fwd_ids_global = JR_TUNE_FWD_IDS_GLOBAL != 0;
constexpr bool fwd_kbuf_salu = JR_TUNE_FWD_KBUF_SALU != 0;
constexpr bool fwd_tile_boxtest = JR_TUNE_FWD_TILE_BOXTEST != 0;
constexpr bool bwd_tv_rcp = JR_TUNE_BWD_TV_RCP != 0;
}  // namespace tune
}  // namespace jr
