// Screen binning for the SoftRas kernels (gfx950).
//
// The reference visits ALL faces for every pixel (SRK:311) — O(pixels x faces).  Here the screen
// is cut into 32x32-pixel bins of 4x4 wavefront tiles (8x8 pixels each).  Every bin gets the list
// of faces whose border box (triangle bbox grown by the cull radius, SRK:28-34, :316) can touch
// it, in ASCENDING face order — the per-pixel aggregation (alpha product, online softmax,
// K-nearest buffer: SRK:350-419) is order dependent and the face-index buffer must match the
// reference bit for bit.  Each list entry carries a 16-bit mask of the bin's tiles the face can
// touch, so a wavefront discards most of its bin's list with one bit test per face.
//
// Pipeline (all on the context stream; one 32-byte read-back of the totals sizes the pool):
//   k_face_setup : per face -> faces_info (SRK:176-236), packed FaceGeo record, conservative pixel
//                  rectangle, per-bin counts
//   k_bin_alloc  : per bin  -> segment base in the pool (atomic bump; placement is irrelevant)
//   k_bin_fill   : per face -> append (id, tile mask) to each touched bin's segment (unordered)
//   k_bin_sort   : per bin  -> sort the segment ascending by id (LDS bitonic; rank sort if huge)
// The rectangle is only a conservative superset: the exact per-pixel border test of the
// reference is re-applied in the raster kernels, so results never depend on the binning.
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
                                                    const float* __restrict__ textures,
                                                    float* __restrict__ faces_info,
                                                    FaceGeo* __restrict__ geo,
                                                    ushort4* __restrict__ face_rect,
                                                    int* __restrict__ bin_count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.B * p.NF) return;
    const float* f = faces + (size_t)i * 9;
    float info[27];
    face_setup(f, info);
    if (faces_info) {   // nullptr when the backward only rebuilds the lists
        float* out = faces_info + (size_t)i * 27;
#pragma unroll
        for (int k = 0; k < 27; k++) out[k] = info[k];
    }
    FaceGeo g;
    build_face_geo(g, f, info, p.rad, i % p.NF);
    if (p.T == 1) {      // single-texel surface colour travels with the record
        const float* tx = textures + (size_t)i * 3;
        g.col[0] = tx[0]; g.col[1] = tx[1]; g.col[2] = tx[2];
    }
    geo[i] = g;

    int px0, px1, py0, py1;
    pixel_range(g.xlo, g.xhi, p.IS, px0, px1);
    pixel_range(g.ylo, g.yhi, p.IS, py0, py1);   // in "yi" space (yi = IS-1-row)
    ushort4 rect = make_ushort4(1, 0, 1, 0);     // empty: x0 > x1
    if (px0 <= px1 && py0 <= py1) {
        const int row0 = p.IS - 1 - py1, row1 = p.IS - 1 - py0;
        rect = make_ushort4((unsigned short)px0, (unsigned short)px1, (unsigned short)row0,
                            (unsigned short)row1);
        const int b = i / p.NF;
        int* bc = bin_count + (size_t)b * p.bins_x * p.bins_y;
        for (int by = row0 / BIN; by <= row1 / BIN; by++)
            for (int bx = px0 / BIN; bx <= px1 / BIN; bx++) atomicAdd(&bc[by * p.bins_x + bx], 1);
    }
    face_rect[i] = rect;
}

// counters: [0] = total pairs (bump pointer), [1] = non-empty bins, [2] = max bin count
__global__ __launch_bounds__(256) void k_bin_alloc(int nbins_total, const int* __restrict__ bin_count,
                                                   int* __restrict__ bin_base,
                                                   int* __restrict__ bin_cursor,
                                                   unsigned long long* __restrict__ counters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nbins_total) return;
    const int n = bin_count[t];
    int base = 0;
    if (n > 0) {
        base = (int)atomicAdd(&counters[0], (unsigned long long)n);
        atomicAdd(&counters[1], 1ull);
        atomicMax(&counters[2], (unsigned long long)n);
    }
    bin_base[t] = base;
    bin_cursor[t] = 0;
}

__global__ __launch_bounds__(256) void k_bin_fill(RasterParams p, const ushort4* __restrict__ face_rect,
                                                  const int* __restrict__ bin_base,
                                                  int* __restrict__ bin_cursor,
                                                  unsigned long long* __restrict__ pool) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.B * p.NF) return;
    const ushort4 r = face_rect[i];
    if (r.x > r.y) return;
    const int b = i / p.NF, fn = i - b * p.NF;
    const size_t bb = (size_t)b * p.bins_x * p.bins_y;
    const int tx0 = r.x / TILE, tx1 = r.y / TILE, ty0 = r.z / TILE, ty1 = r.w / TILE;
    for (int by = r.z / BIN; by <= r.w / BIN; by++)
        for (int bx = r.x / BIN; bx <= r.y / BIN; bx++) {
            // mask of the bin's 4x4 tiles overlapped by the face rectangle (bit = ty*4 + tx)
            const int sx0 = max(tx0 - bx * SUBS, 0), sx1 = min(tx1 - bx * SUBS, SUBS - 1);
            const int sy0 = max(ty0 - by * SUBS, 0), sy1 = min(ty1 - by * SUBS, SUBS - 1);
            const unsigned rowbits = ((1u << (sx1 + 1)) - 1u) & ~((1u << sx0) - 1u);
            unsigned mask = 0;
            for (int sy = sy0; sy <= sy1; sy++) mask |= rowbits << (sy * SUBS);
            const size_t t = bb + by * p.bins_x + bx;
            const int pos = atomicAdd(&bin_cursor[t], 1);
            pool[bin_base[t] + pos] = ((unsigned long long)(unsigned)fn << 32) | mask;
        }
}

constexpr int SORT_LDS = 4096;   // 64-bit entries sortable in LDS by one workgroup (32 KB)

// One workgroup per bin.  Ascending sort of the bin's entries by face id (unique keys).
__global__ __launch_bounds__(256) void k_bin_sort(const int* __restrict__ bin_count,
                                                  const int* __restrict__ bin_base,
                                                  unsigned long long* __restrict__ pool,
                                                  unsigned long long* __restrict__ scratch) {
    __shared__ unsigned long long s[SORT_LDS];
    const int t = blockIdx.x;
    const int n = bin_count[t];
    if (n <= 1) return;
    unsigned long long* seg = pool + bin_base[t];
    if (n <= SORT_LDS) {
        int m = 2;
        while (m < n) m <<= 1;
        for (int i = threadIdx.x; i < m; i += blockDim.x) s[i] = i < n ? seg[i] : ~0ull;
        __syncthreads();
        for (int k = 2; k <= m; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = threadIdx.x; i < m; i += blockDim.x) {
                    const int l = i ^ j;
                    if (l > i) {
                        const unsigned long long a = s[i], b = s[l];
                        const bool up = (i & k) == 0;
                        if ((a > b) == up) { s[i] = b; s[l] = a; }
                    }
                }
                __syncthreads();
            }
        for (int i = threadIdx.x; i < n; i += blockDim.x) seg[i] = s[i];
    } else {
        // Huge segment (a whole mesh inside one bin): rank sort through the scratch copy, staged
        // through LDS in blocks.  O(n^2/256) per bin — correctness path for degenerate inputs.
        unsigned long long* src = scratch + bin_base[t];
        for (int i = threadIdx.x; i < n; i += blockDim.x) src[i] = seg[i];
        __syncthreads();
        for (int i0 = 0; i0 < n; i0 += blockDim.x) {
            const int i = i0 + threadIdx.x;
            const unsigned long long key = i < n ? src[i] : 0;
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

void launch_binning(hipStream_t st, const RasterParams& p, const float* faces, const float* textures,
                    float* faces_info, BinWorkspace& ws) {
    const int nfaces = p.B * p.NF;
    const int nbins = p.B * p.bins_x * p.bins_y;
    (void)hipMemsetAsync(ws.bin_count, 0, sizeof(int) * (size_t)nbins, st);
    (void)hipMemsetAsync(ws.counters, 0, sizeof(unsigned long long) * 4, st);
    k_face_setup<<<(nfaces + 255) / 256, 256, 0, st>>>(p, faces, textures, faces_info, ws.geo, ws.face_rect, ws.bin_count);
    k_bin_alloc<<<(nbins + 255) / 256, 256, 0, st>>>(nbins, ws.bin_count, ws.bin_base, ws.bin_cursor, ws.counters);
}

void launch_bin_fill_sort(hipStream_t st, const RasterParams& p, BinWorkspace& ws) {
    const int nfaces = p.B * p.NF;
    const int nbins = p.B * p.bins_x * p.bins_y;
    k_bin_fill<<<(nfaces + 255) / 256, 256, 0, st>>>(p, ws.face_rect, ws.bin_base, ws.bin_cursor, ws.pool);
    k_bin_sort<<<nbins, 256, 0, st>>>(ws.bin_count, ws.bin_base, ws.pool, ws.pool_scratch);
}

}  // namespace jr
