// Internal declarations shared by the HIP translation units of libjrender_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "softras_device.h"

namespace jr {

// Per-context scratch for the bin lists (owned by jr_ctx, grown lazily, never freed per call).
struct BinWorkspace {
    FaceGeo* geo = nullptr;                    // [B*NF] packed geometry records
    ushort4* face_rect = nullptr;              // [B*NF] conservative pixel rectangle (x0,x1,row0,row1)
    int* bin_acc = nullptr;                    // [B*bins] counts as k_face_setup accumulates them; zero between set-up passes (k_bin_alloc_schedule clears what it read)
    int* bin_count = nullptr;                  // [B*bins] listed faces per bin
    int* bin_base = nullptr;                   // [B*bins] segment start in pool
    int* bin_cursor = nullptr;                 // [B*bins]
    int* bin_order = nullptr;                  // [B*bins] launch rank -> bin, heaviest list first
    unsigned long long* counters = nullptr;    // [4] device: total pairs, non-empty bins, max count, heavy bins (a prefix of bin_order)
    unsigned long long* host_counters = nullptr;   // the same four in pinned host memory (device address), written by k_bin_alloc_schedule
    unsigned long long* pool = nullptr;        // [pool_cap] (face id << 32 | tile mask), per bin ascending
    unsigned long long* pool_scratch = nullptr;// [pool_cap] the same segments as filled (unordered)
    size_t faces_cap = 0, bins_cap = 0, pool_cap = 0;
    int heavy_waves = 4;                        // wavefronts per workgroup of the next forward's four-/eight-wavefront kernel (host policy, jr_api.cpp)
    // Launch policy of the multi-wavefront kernels (jr_softras_set_launch_policy): bins that list more than heavy_min faces
    // are HEAVY (a whole workgroup per tile in the forward, bwd_split wavefronts per tile in the backward); 0 = no heavy
    // tiles.  heavy_bound: upper bound of the heavy bins the NEXT launch can find (what the same shape found before, or
    // the exact count once the host has read the schedule's totals); < 0 = unknown, the pool-capacity bound is used.
    int heavy_min = tune::fwd_heavy;
    long heavy_bound = -1;
    mutable int heavy_waves_used = 0;          // what the last forward launch really used (8 falls back to 4 when the LDS opt-in is refused)
    mutable unsigned long long lds_optin_ok[2] = {0, 0}, lds_optin_tried[2] = {0, 0};   // per (dist, rgb, K class) instantiation of the 8-wavefront kernel ([1]: the precise-colour set): > 64 KB of dynamic LDS granted on this device
};

// Launch order of the bins (k_bin_alloc_schedule): ~12 buckets per octave of the list length, heaviest first.  Bins in
// buckets >= heavy_bucket() are the "heavy" prefix of the order (counters[3]) that the forward gives four
// wavefronts per tile; every one of them lists at least fwd_heavy_floor() faces.
inline int heavy_bucket(int heavy_min) { return heavy_min > 0 ? 1 + (int)(log2f((float)heavy_min) * 12.f) : 1 << 30; }
inline int fwd_heavy_floor(int heavy_min) {
    const int f = (int)(exp2f((float)(heavy_bucket(heavy_min) - 1) / 12.f) * 0.98f);
    return f > 1 ? f : 1;
}
// upper bound of counters[3] for a launch over nbins bins: every heavy bin holds >= fwd_heavy_floor() of the pool's
// entries; tightened by what the host knows about this shape (ws.heavy_bound)
inline int heavy_bins_cap(const BinWorkspace& ws, int nbins) {
    if (ws.heavy_min <= 0) return 0;
    long cap = (long)(ws.pool_cap / (unsigned long long)fwd_heavy_floor(ws.heavy_min)) + 8;
    if (ws.heavy_bound >= 0 && ws.heavy_bound < cap) cap = ws.heavy_bound;
    return (int)(cap < nbins ? cap : nbins);
}

void launch_binning(hipStream_t st, const RasterParams& p, const float* faces, const float* textures,
                    float* faces_info, BinWorkspace& ws);
void launch_bin_fill_sort(hipStream_t st, const RasterParams& p, BinWorkspace& ws, bool reset_cursors);

void launch_softras_forward(hipStream_t st, const RasterParams& p, const float* textures,
                            const BinWorkspace& ws, float* aggrs_info, float* soft_colors,
                            int32_t* faces_id_buffer);
}  // namespace jr
namespace jr_precise {   // the same kernels with the colour path in the reference's own arithmetic (softras_forward_precise.hip)
void launch_softras_forward(hipStream_t st, const jr::RasterParams& p, const float* textures,
                            const jr::BinWorkspace& ws, float* aggrs_info, float* soft_colors,
                            int32_t* faces_id_buffer);
}
namespace jr {
bool forward_uses_heavy_path(const RasterParams& p, const BinWorkspace& ws);   // launches of up to tune::fwd_heavy_pixels pixels: four wavefronts per tile of a heavy bin
bool backward_splits_heavy_tiles(const RasterParams& p, const BinWorkspace& ws);   // launches of up to tune::bwd_split_pixels pixels: tune::bwd_split wavefronts per tile of a heavy bin, each with the face ids of one residue class
void launch_softras_backward(hipStream_t st, const RasterParams& p, const float* textures,
                             const float* soft_colors, const float* aggrs_info,
                             const int32_t* faces_id_buffer, const float* grad_soft_colors,
                             const BinWorkspace& ws, float* grad_faces, float* grad_textures);

void launch_face_vertices_forward(hipStream_t st, const float* vertices, const int32_t* faces,
                                  float* fv, int B, int NV, int NF);
void launch_face_vertices_backward(hipStream_t st, const float* gfv, const int32_t* faces,
                                   float* gv, int B, int NV, int NF);
void launch_face_vertices_backward_shared(hipStream_t st, const float* gfv, const int32_t* faces,
                                          float* gv, int B, int NV, int NF);
void launch_avgpool2x2_forward(hipStream_t st, const float* in, float* out, int planes, int H, int W);
void launch_avgpool2x2_backward(hipStream_t st, const float* gout, float* gin, int planes, int H, int W);
void launch_camera_forward(hipStream_t st, const float* v, const float* eye, const float* rot, float* out, int B,
                           int VB, int NV, int kind, float param);
void launch_camera_backward(hipStream_t st, const float* gout, const float* v, const float* eye, const float* rot,
                            float* gv, int B, int VB, int NV, int kind, float param);
void launch_face_camera_backward_shared(hipStream_t st, const float* gfv, const int32_t* faces, const float* v,
                                        const float* eye, const float* rot, float* gv, int B, int NV, int NF,
                                        int kind, float param);
void launch_neg_iou_loss(hipStream_t st, const float* predict, const float* target, float* iou, float* grad, int B,
                         int n, float divisor);
// (acc [B] doubles and ticket [B] counters: zeroed scratch of the context, left zeroed by every launch)
void launch_laplacian_loss(hipStream_t st, const int* rowptr, const int* col, const float* val, const int* rowptr_t,
                           const int* col_t, const float* val_t, const float* x, float* y, float* loss, float* grad,
                           double* acc, unsigned* ticket, int B, int nv, float scale);
void launch_flatten_loss(hipStream_t st, const int* v0s, const int* v1s, const int* v2s, const int* v3s, const float* x,
                         float* loss, float* grad, double* acc, unsigned* ticket, int B, int nv, int ne, float eps, float scale);
void launch_deform_forward(hipStream_t st, const float* tmpl, const float* displace, const float* center, float* out, int nv);
void launch_deform_backward(hipStream_t st, const float* tmpl, const float* displace, const float* center, const float* g0,
                            float w0, const float* g1, float w1, const float* g2, float w2, float* grad_displace,
                            float* grad_center, double* acc, unsigned* ticket, int nv);
void launch_adam_step(hipStream_t st, float* p, const float* g, float* m, float* v, size_t n, float lr_over_c0, float b0,
                      float one_minus_b0, float b1, float one_minus_b1, float c1, float eps, float weight_decay,
                      const int* iteration, double lr, double beta0, double beta1);
void launch_scalar_accumulate(hipStream_t st, float* dst, const float* src, int n, float scale, float bias, int accumulate,
                              const int* iteration, int stride);
void launch_counter_add(hipStream_t st, int* counter, int delta);
void launch_n3mr_image_forward(hipStream_t st, const float* in, float* out, int B, int H, int W, int C, int pool);
void launch_n3mr_image_backward(hipStream_t st, const float* gout, float* gin, int B, int H, int W, int C, int pool);
void launch_n3mr_forward(hipStream_t st, const float* faces, const float* textures, float* faces_inv,
                         unsigned long long* zkey, int32_t* face_index_map, float* weight_map, float* depth_map,
                         float* face_inv_map, float* rgb_map, float* alpha_map, int32_t* sampling_index_map,
                         float* sampling_weight_map, int B, int NF, int TS, int IS, float near_, float far_,
                         float eps, const float* bg, int rrgb, int ralpha, int rdepth, bool zkey_clean);
void launch_n3mr_backward(hipStream_t st, const float* faces, const int32_t* face_index_map,
                          const float* weight_map, const float* depth_map, const float* face_inv_map,
                          const float* rgb_map, const float* alpha_map, const float* sampling_weight_map,
                          const int32_t* sampling_index_map, const float* grad_rgb_map, const float* grad_alpha_map,
                          const float* grad_depth_map, float* grad_faces, float* grad_textures, void* scratch,
                          int B, int NF, int TS, int IS, float eps, int rrgb, int ralpha, int rdepth);
size_t n3mr_backward_scratch_bytes(int B, int IS);
void launch_selftest_rcp(hipStream_t st, unsigned long long* mismatches);
void launch_selftest_div(hipStream_t st, unsigned long long n, uint32_t seed, unsigned long long* mismatches);

}  // namespace jr
