// Screen-tile binning for the SoftRas kernels (gfx950).
//
// The reference visits ALL faces for every pixel (SRK:311) — O(pixels x faces).  Here every
// 16x16-pixel tile gets the list of faces whose border box (triangle bbox grown by the cull
// radius, SRK:28-34, :316) can touch it, in ASCENDING face order, because the per-pixel
// aggregation (alpha product, online softmax, K-nearest buffer: SRK:350-419) is order dependent
// and the face-index buffer must match the reference bit for bit.
//
// Pipeline (all on the context stream, no host round trip except one 16-byte read of the totals):
//   k_face_setup  : per face -> faces_info (SRK:176-236) + conservative tile rectangle + per-tile counts
//   k_tile_alloc  : per tile -> segment base in the pair pool (atomic bump; placement is irrelevant)
//   k_tile_fill   : per face -> append its id to each touched tile's segment (unordered)
//   k_tile_sort   : per tile -> sort the segment ascending (LDS bitonic; rank sort for huge segments)
// The rectangle is only a conservative superset: the exact per-pixel border test of the
// reference is re-applied in the raster kernels, so results do not depend on the binning.
#include "jr_kernels.h"

namespace jr {

// Conservative pixel range [lo, hi] of centres c(i) = (2i+1-IS)/IS that can satisfy vlo <= c(i) <= vhi.
// One pixel of slack on both sides covers every rounding in this estimate and in the reference's
// float compare.  NaN bounds -> full range (the reference's compares are all false for NaN, i.e.
// the face is NOT culled).
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
}

__global__ __launch_bounds__(256) void k_face_setup(RasterParams p, const float* __restrict__ faces,
                                                    float* __restrict__ faces_info,
                                                    uint32_t* __restrict__ face_rect,
                                                    int* __restrict__ tile_count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.B * p.NF) return;
    const float* f = faces + (size_t)i * 9;
    float info[27];
    face_setup(f, info);
    if (faces_info) {   // nullptr when the backward only rebuilds the tile lists
        float* out = faces_info + (size_t)i * 27;
#pragma unroll
        for (int k = 0; k < 27; k++) out[k] = info[k];
    }

    const float xhi = fmaxf(fmaxf(f[0], f[3]), f[6]) + p.rad;
    const float xlo = fminf(fminf(f[0], f[3]), f[6]) - p.rad;
    const float yhi = fmaxf(fmaxf(f[1], f[4]), f[7]) + p.rad;
    const float ylo = fminf(fminf(f[1], f[4]), f[7]) - p.rad;
    int px0, px1, py0, py1;
    pixel_range(xlo, xhi, p.IS, px0, px1);
    pixel_range(ylo, yhi, p.IS, py0, py1);   // in "yi" space (yi = IS-1-row)
    uint32_t rect = 0xffffffffu;             // empty
    if (px0 <= px1 && py0 <= py1) {
        const int row0 = p.IS - 1 - py1, row1 = p.IS - 1 - py0;
        const int tx0 = px0 / TILE, tx1 = px1 / TILE, ty0 = row0 / TILE, ty1 = row1 / TILE;
        rect = (uint32_t)tx0 | ((uint32_t)ty0 << 8) | ((uint32_t)tx1 << 16) | ((uint32_t)ty1 << 24);
        const int b = i / p.NF;
        int* tc = tile_count + (size_t)b * p.tiles_x * p.tiles_y;
        for (int ty = ty0; ty <= ty1; ty++)
            for (int tx = tx0; tx <= tx1; tx++) atomicAdd(&tc[ty * p.tiles_x + tx], 1);
    }
    face_rect[i] = rect;
}

// counters: [0] = total pairs (bump pointer), [1] = non-empty tiles, [2] = max tile count
__global__ __launch_bounds__(256) void k_tile_alloc(int ntiles_total, const int* __restrict__ tile_count,
                                                    int* __restrict__ tile_base,
                                                    int* __restrict__ tile_cursor,
                                                    unsigned long long* __restrict__ counters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles_total) return;
    const int n = tile_count[t];
    int base = 0;
    if (n > 0) {
        base = (int)atomicAdd(&counters[0], (unsigned long long)n);
        atomicAdd(&counters[1], 1ull);
        atomicMax(&counters[2], (unsigned long long)n);
    }
    tile_base[t] = base;
    tile_cursor[t] = 0;
}

__global__ __launch_bounds__(256) void k_tile_fill(RasterParams p, const uint32_t* __restrict__ face_rect,
                                                   const int* __restrict__ tile_base,
                                                   int* __restrict__ tile_cursor,
                                                   int* __restrict__ pool) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.B * p.NF) return;
    const uint32_t rect = face_rect[i];
    if (rect == 0xffffffffu) return;
    const int tx0 = rect & 255, ty0 = (rect >> 8) & 255, tx1 = (rect >> 16) & 255, ty1 = rect >> 24;
    const int b = i / p.NF, fn = i - b * p.NF;
    const size_t tb = (size_t)b * p.tiles_x * p.tiles_y;
    for (int ty = ty0; ty <= ty1; ty++)
        for (int tx = tx0; tx <= tx1; tx++) {
            const size_t t = tb + ty * p.tiles_x + tx;
            const int pos = atomicAdd(&tile_cursor[t], 1);
            pool[tile_base[t] + pos] = fn;
        }
}

constexpr int SORT_LDS = 4096;   // ints sortable in LDS by one workgroup

// One workgroup per tile.  Ascending sort of the tile's face ids (unique keys).
__global__ __launch_bounds__(256) void k_tile_sort(const int* __restrict__ tile_count,
                                                   const int* __restrict__ tile_base,
                                                   int* __restrict__ pool, int* __restrict__ scratch) {
    __shared__ int s[SORT_LDS];
    const int t = blockIdx.x;
    const int n = tile_count[t];
    if (n <= 1) return;
    int* seg = pool + tile_base[t];
    if (n <= SORT_LDS) {
        int m = 2;
        while (m < n) m <<= 1;
        for (int i = threadIdx.x; i < m; i += blockDim.x) s[i] = i < n ? seg[i] : 0x7fffffff;
        __syncthreads();
        for (int k = 2; k <= m; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = threadIdx.x; i < m; i += blockDim.x) {
                    const int l = i ^ j;
                    if (l > i) {
                        const int a = s[i], b = s[l];
                        const bool up = (i & k) == 0;
                        if ((a > b) == up) { s[i] = b; s[l] = a; }
                    }
                }
                __syncthreads();
            }
        for (int i = threadIdx.x; i < n; i += blockDim.x) seg[i] = s[i];
    } else {
        // Huge segment (many faces inside one tile): rank sort through the scratch copy, staged
        // through LDS in blocks.  O(n^2/256) per tile — correctness path for degenerate inputs.
        int* src = scratch + tile_base[t];
        for (int i = threadIdx.x; i < n; i += blockDim.x) src[i] = seg[i];
        __syncthreads();
        for (int i0 = 0; i0 < n; i0 += blockDim.x) {
            const int i = i0 + threadIdx.x;
            const int key = i < n ? src[i] : 0;
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

void launch_binning(hipStream_t st, const RasterParams& p, const float* faces, float* faces_info,
                    BinWorkspace& ws) {
    const int nfaces = p.B * p.NF;
    const int ntiles = p.B * p.tiles_x * p.tiles_y;
    (void)hipMemsetAsync(ws.tile_count, 0, sizeof(int) * (size_t)ntiles, st);
    (void)hipMemsetAsync(ws.counters, 0, sizeof(unsigned long long) * 4, st);
    k_face_setup<<<(nfaces + 255) / 256, 256, 0, st>>>(p, faces, faces_info, ws.face_rect, ws.tile_count);
    k_tile_alloc<<<(ntiles + 255) / 256, 256, 0, st>>>(ntiles, ws.tile_count, ws.tile_base, ws.tile_cursor,
                                                       ws.counters);
}

void launch_bin_fill_sort(hipStream_t st, const RasterParams& p, BinWorkspace& ws) {
    const int nfaces = p.B * p.NF;
    const int ntiles = p.B * p.tiles_x * p.tiles_y;
    k_tile_fill<<<(nfaces + 255) / 256, 256, 0, st>>>(p, ws.face_rect, ws.tile_base, ws.tile_cursor, ws.pool);
    k_tile_sort<<<ntiles, 256, 0, st>>>(ws.tile_count, ws.tile_base, ws.pool, ws.pool_scratch);
}

}  // namespace jr
