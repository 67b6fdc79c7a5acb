// Internal declarations shared by the HIP translation units of libjrender_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "softras_device.h"

namespace jr {

// Per-context scratch for the tile lists (owned by jr_ctx, grown lazily, never freed per call).
struct BinWorkspace {
    uint32_t* face_rect = nullptr;             // [B*NF] packed tile rectangle per face
    int* tile_count = nullptr;                 // [B*tiles]
    int* tile_base = nullptr;                  // [B*tiles] segment start in pool
    int* tile_cursor = nullptr;                // [B*tiles]
    unsigned long long* counters = nullptr;    // [4] device: total pairs, non-empty tiles, max count
    int* pool = nullptr;                       // [pool_cap] face ids, per tile ascending
    int* pool_scratch = nullptr;               // [pool_cap] only used by the huge-segment sort
    size_t faces_cap = 0, tiles_cap = 0, pool_cap = 0;
};

void launch_binning(hipStream_t st, const RasterParams& p, const float* faces, float* faces_info,
                    BinWorkspace& ws);
void launch_bin_fill_sort(hipStream_t st, const RasterParams& p, BinWorkspace& ws);

void launch_softras_forward(hipStream_t st, const RasterParams& p, const float* faces,
                            const float* textures, const float* faces_info, const BinWorkspace& ws,
                            float* aggrs_info, float* soft_colors, int32_t* faces_id_buffer);
void launch_softras_backward(hipStream_t st, const RasterParams& p, const float* faces,
                             const float* textures, const float* soft_colors,
                             const float* faces_info, const float* aggrs_info,
                             const int32_t* faces_id_buffer, const float* grad_soft_colors,
                             const BinWorkspace& ws, float* grad_faces, float* grad_textures);

void launch_face_vertices_forward(hipStream_t st, const float* vertices, const int32_t* faces,
                                  float* fv, int B, int NV, int NF);
void launch_face_vertices_backward(hipStream_t st, const float* gfv, const int32_t* faces,
                                   float* gv, int B, int NV, int NF);
void launch_avgpool2x2_forward(hipStream_t st, const float* in, float* out, int planes, int H, int W);
void launch_avgpool2x2_backward(hipStream_t st, const float* gout, float* gin, int planes, int H, int W);

}  // namespace jr
