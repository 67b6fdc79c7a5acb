// SoftRas forward rasteriser for gfx950 (MI355X).
//
// Replaces forward_soft_rasterize_cuda_kernel (SRK:243-456).  The reference runs one thread per
// pixel over ALL faces.  Here ONE WAVEFRONT owns an 8x8-pixel tile (lane = pixel) and walks its
// 32x32 bin's ascending face list 64 faces at a time with the roles of the lanes switched:
//
//   cull  (lane = face)   each lane takes one list entry, drops it unless the entry's tile mask
//                         has this tile's bit, loads the face's border box and evaluates the
//                         reference's border test (SRK:28-34) against the tile's 8 column and 8 row
//                         pixel centres.  Every compare is a v_cmp whose 64-bit result IS the
//                         wavefront ballot over the 64 faces: 32 compares cull 64 faces x 64 pixels.
//   stage (lane = face)   surviving faces copy their packed geometry record into LDS.
//   raster(lane = pixel)  each pixel ANDs its column ballot with its row ballot: a private bitmask
//                         of the faces that pass ITS border test.  It then pops its own bits in
//                         ascending face order and runs the exact per-(pixel,face) arithmetic on
//                         the LDS record of ITS face — lanes stay busy although neighbouring pixels
//                         see different face subsets (lane compaction by bitmask).
//
// The per-pixel state machine (alpha, online softmax over depth, K-nearest buffer) lives in VGPRs;
// it is sequential in face order, which is why the lists are sorted.  No MFMA: no dense contraction.
#include "jr_kernels.h"

namespace jr {

template <int KCAP>
struct KBuffer {
    int id[KCAP];
    float z[KCAP];
    int size;
    float max_z;
    int max_slot;

    __device__ inline void init() {
#pragma unroll
        for (int k = 0; k < KCAP; k++) { id[k] = -1; z[k] = 0.f; }
        size = 0; max_z = -1.f; max_slot = -1;
    }
    // K-nearest insert with the reference's slot semantics (SRK:369-385): append while not
    // full (tracking the first largest depth), afterwards overwrite the largest-depth slot
    // when strictly nearer and rescan, first maximum wins.
    __device__ inline void insert(int fn, float zp, int K) {
        const bool filling = size < K;
        if (!filling && !(zp < max_z)) return;
        const int slot = filling ? size : max_slot;
#pragma unroll
        for (int k = 0; k < KCAP; k++) {
            const bool hit = k == slot;
            id[k] = hit ? fn : id[k];
            z[k] = hit ? zp : z[k];
        }
        if (filling) {
            if (zp > max_z) { max_z = zp; max_slot = size; }
            size++;
        } else {
            float m = -1.f;
            int ms = max_slot;
#pragma unroll
            for (int k = 0; k < KCAP; k++) {
                const bool gt = (k < K) && (z[k] > m);
                m = gt ? z[k] : m;
                ms = gt ? k : ms;
            }
            max_z = m; max_slot = ms;
        }
    }
};

template <int DIST, int RGB, int KCAP>
__global__ __launch_bounds__(64) void k_softras_forward(
    RasterParams p, int ntiles_total, const float* __restrict__ textures,
    const FaceGeo* __restrict__ geo, const int* __restrict__ bin_count,
    const int* __restrict__ bin_base, const unsigned long long* __restrict__ pool,
    float* __restrict__ aggrs, float* __restrict__ rgba, int32_t* __restrict__ ids) {
    __shared__ FaceRec s_rec[CHUNK];

    // XCD-aware order: consecutive workgroup ids land on different XCDs (id % 8); give each XCD a
    // contiguous run of tiles so that the 16 tiles of a bin (same list, same records) share an L2.
    const int per_xcd = gridDim.x >> 3;
    const int t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= ntiles_total) return;
    const int bin = t >> 4, sub = t & 15;
    const int bins_per_img = p.bins_x * p.bins_y;
    const int b = bin / bins_per_img;
    const int bb = bin - b * bins_per_img;
    const int by = bb / p.bins_x, bx = bb - by * p.bins_x;
    const int col0 = bx * BIN + (sub & 3) * TILE, row0 = by * BIN + (sub >> 2) * TILE;
    if (col0 >= p.IS || row0 >= p.IS) return;            // tile lies outside the image

    const int lane = threadIdx.x, lx = lane & 7, ly = lane >> 3;
    const int col = col0 + lx, row = row0 + ly;
    const bool valid = col < p.IS && row < p.IS;
    const float xp = pixel_centre(col, p.IS);
    const float yp = pixel_centre(p.IS - 1 - row, p.IS);                      // SRK:280-283
    float xc[8], yc[8];                                                       // wave-uniform centres
#pragma unroll
    for (int c = 0; c < 8; c++) {
        xc[c] = pixel_centre(col0 + c, p.IS);
        yc[c] = pixel_centre(p.IS - 1 - (row0 + c), p.IS);
    }

    // ---- per-pixel state (SRK:291-309) ----
    float c0 = 1.f, c1 = 1.f, c2 = 1.f;
    float alpha = p.alpha == 2 ? 1.f : 0.f;
    float ssum = expf(p.eps / p.gamma), smax = p.eps;
    if (RGB == 0) { c0 = p.bg[0]; c1 = p.bg[1]; c2 = p.bg[2]; }
    else if (RGB == 1) { c0 = p.bg[0] * ssum; c1 = p.bg[1] * ssum; c2 = p.bg[2] * ssum; }
    float depth_min = 10000000.f;
    int face_min = -1;
    KBuffer<KCAP> q;
    q.init();

    const int n = bin_count[bin];
    const unsigned long long* seg = pool + bin_base[bin];
    const FaceGeo* gbase = geo + (size_t)b * p.NF;
    const float* tbase = textures + (size_t)b * p.NF * p.T * 3;

    for (int s0 = 0; s0 < n; s0 += CHUNK) {
        // ---- cull + stage: lane = face ----
        const int idx = s0 + lane;
        const unsigned long long e = idx < n ? seg[idx] : 0ull;
        const int fn_f = (int)(e >> 32);
        const bool need = (e >> sub) & 1ull;
        if (!ballot(need)) continue;            // no face of this chunk touches this tile
        const FaceGeo* gp = gbase + fn_f;
        float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
        if (need) box = *reinterpret_cast<const float4*>(gp);                 // xlo xhi ylo yhi
        unsigned long long cx[8], ry[8];
        bool anyx = false, anyy = false;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            // check_border (SRK:28-34, :316): a pixel is culled when strictly outside the grown box
            const bool px = need && !(xc[c] > box.y) && !(xc[c] < box.x);
            const bool py = need && !(yc[c] > box.w) && !(yc[c] < box.z);
            cx[c] = ballot(px); ry[c] = ballot(py);
            anyx |= px; anyy |= py;
        }
        __syncthreads();                        // previous chunk's readers are done with s_rec
        if (anyx && anyy) {
            const float4* src = reinterpret_cast<const float4*>(gp);
            float4* dst = reinterpret_cast<float4*>(&s_rec[lane]);
#pragma unroll
            for (int k = 0; k < 9; k++) dst[k] = src[k];
            s_rec[lane].id = fn_f;
            const float* tx_ = tbase + (size_t)fn_f * p.T * 3;
            if (p.tex == 1) {
#pragma unroll
                for (int k = 0; k < 9; k++) s_rec[lane].col[k] = tx_[k];
            } else if (p.T == 1) {
                s_rec[lane].col[0] = tx_[0]; s_rec[lane].col[1] = tx_[1]; s_rec[lane].col[2] = tx_[2];
            }
        }
        __syncthreads();

        // ---- raster: lane = pixel; private mask of the faces that pass this pixel's border test ----
        unsigned long long M = valid ? (select8(cx, lx) & select8(ry, ly)) : 0ull;
        while (M) {
            const int j = __builtin_ctzll(M);
            M &= M - 1;
            const FaceRec& fr = s_rec[j];
            const FaceGeo& r = fr.g;
            const Bary w = barycentric(r, xp, yp);
            float D;
            if (DIST == 0) {                                                   // SRK:331-333
                if (!pixel_inside(w)) continue;
                D = 1.f;
            } else if (DIST == 1) {                                            // SRK:335-338
                const float dis = barycentric_dist(w);
                if (-dis >= p.thr) continue;
                D = coverage(-dis / p.sigma);
            } else {                                                           // SRK:340-344
                const Dist dd = euclidean_p2f(r, w, xp, yp);
                const float dis = dd.dx * dd.dx + dd.dy * dd.dy;
                if (dd.sign < 0 && dis >= p.thr) continue;
                D = coverage(-dd.sign * dis / p.sigma);
            }
            // alpha aggregation happens before the depth cull (SRK:350-358)
            if (p.alpha == 0) { if (D > 0.5f) alpha = 1.f; }
            else if (p.alpha == 1) alpha += D;
            else alpha = (float)((double)alpha * (1. - (double)D));

            const Bary wc = barycentric_clip(w);
            const float zp = depth_of(r, wc);
            if (zp < p.near_ || zp > p.far_) continue;                        // SRK:365
            const int fn = fr.id;
            q.insert(fn, zp, p.K);

            if (RGB == 0) {                                                    // SRK:390-397
                if (zp < depth_min && pixel_inside(w) && (p.double_side || r.front)) {
                    depth_min = zp; face_min = fn;
                    if (p.tex == 0) {
                        if (p.T == 1) { c0 = fr.col[0]; c1 = fr.col[1]; c2 = fr.col[2]; }
                        else {
                            const float* tx_ = tbase + ((size_t)fn * p.T + surface_texel(wc, p.R)) * 3;
                            c0 = tx_[0]; c1 = tx_[1]; c2 = tx_[2];
                        }
                    } else {                                                   // SRK:168-171
                        c0 = ((wc.w0 * fr.col[0] / r.z[0] + wc.w1 * fr.col[3] / r.z[1]) + wc.w2 * fr.col[6] / r.z[2]) * zp;
                        c1 = ((wc.w0 * fr.col[1] / r.z[0] + wc.w1 * fr.col[4] / r.z[1]) + wc.w2 * fr.col[7] / r.z[2]) * zp;
                        c2 = ((wc.w0 * fr.col[2] / r.z[0] + wc.w1 * fr.col[5] / r.z[1]) + wc.w2 * fr.col[8] / r.z[2]) * zp;
                    }
                }
            } else if (RGB == 1) {                                             // SRK:399-419
                if (r.front || p.double_side) {
                    const float zn = (p.far_ - zp) / (p.far_ - p.near_);
                    float ed = 1.f;
                    if (zn > smax) { ed = expf((smax - zn) / p.gamma); smax = zn; }
                    const float ez = expf((zn - smax) / p.gamma);
                    ssum = ed * ssum + ez * D;
                    float k0, k1, k2;
                    if (p.tex == 0) {
                        if (p.T == 1) { k0 = fr.col[0]; k1 = fr.col[1]; k2 = fr.col[2]; }
                        else {
                            const float* tx_ = tbase + ((size_t)fn * p.T + surface_texel(wc, p.R)) * 3;
                            k0 = tx_[0]; k1 = tx_[1]; k2 = tx_[2];
                        }
                    } else {
                        k0 = ((wc.w0 * fr.col[0] / r.z[0] + wc.w1 * fr.col[3] / r.z[1]) + wc.w2 * fr.col[6] / r.z[2]) * zp;
                        k1 = ((wc.w0 * fr.col[1] / r.z[0] + wc.w1 * fr.col[4] / r.z[1]) + wc.w2 * fr.col[7] / r.z[2]) * zp;
                        k2 = ((wc.w0 * fr.col[2] / r.z[0] + wc.w1 * fr.col[5] / r.z[1]) + wc.w2 * fr.col[8] / r.z[2]) * zp;
                    }
                    c0 = ed * c0 + ez * D * k0;
                    c1 = ed * c1 + ez * D * k1;
                    c2 = ed * c2 + ez * D * k2;
                }
            }
        }
    }

    if (!valid) return;
    // ---- finalise (SRK:426-455) ----
    const size_t pp = (size_t)p.IS * p.IS;
    const size_t pn = (size_t)row * p.IS + col;
    float a_out;
    if (p.alpha == 0) a_out = alpha;
    else if (p.alpha == 1) a_out = alpha / p.NF;
    else a_out = (float)(1. - (double)alpha);
    float o0 = p.bg[0], o1 = p.bg[1], o2 = p.bg[2], g0 = 0.f, g1 = 0.f;
    if (RGB == 0) {
        if (face_min != -1) { o0 = c0; o1 = c1; o2 = c2; }
        g0 = depth_min; g1 = (float)face_min;
    } else if (RGB == 1) {
        o0 = c0 / ssum; o1 = c1 / ssum; o2 = c2 / ssum;
        g0 = ssum; g1 = smax;
    }
    float* out = rgba + (size_t)b * 4 * pp + pn;
    out[0] = o0; out[pp] = o1; out[2 * pp] = o2; out[3 * pp] = a_out;
    float* ag = aggrs + (size_t)b * 2 * pp + pn;
    ag[0] = g0; ag[pp] = g1;
    int32_t* io = ids + (size_t)b * p.K * pp + pn;
#pragma unroll
    for (int k = 0; k < KCAP; k++)
        if (k < p.K) io[(size_t)k * pp] = q.id[k];
}

template <int DIST, int RGB>
static void launch_k(hipStream_t st, const RasterParams& p, int ntiles, const float* textures,
                     const BinWorkspace& ws, float* aggrs, float* rgba, int32_t* ids) {
    const int grid = ((ntiles + 7) / 8) * 8;
    if (p.K <= 16)
        k_softras_forward<DIST, RGB, 16><<<grid, 64, 0, st>>>(
            p, ntiles, textures, ws.geo, ws.bin_count, ws.bin_base, ws.pool, aggrs, rgba, ids);
    else
        k_softras_forward<DIST, RGB, 64><<<grid, 64, 0, st>>>(
            p, ntiles, textures, ws.geo, ws.bin_count, ws.bin_base, ws.pool, aggrs, rgba, ids);
}

void launch_softras_forward(hipStream_t st, const RasterParams& p, const float* textures,
                            const BinWorkspace& ws, float* aggrs, float* rgba, int32_t* ids) {
    const int ntiles = p.B * p.bins_x * p.bins_y * SUBS * SUBS;
#define JR_FWD(D, R) launch_k<D, R>(st, p, ntiles, textures, ws, aggrs, rgba, ids)
    const int rgb = p.rgb == 0 ? 0 : (p.rgb == 1 ? 1 : 2);
    switch (p.dist * 3 + rgb) {
        case 0: JR_FWD(0, 0); break;
        case 1: JR_FWD(0, 1); break;
        case 2: JR_FWD(0, 2); break;
        case 3: JR_FWD(1, 0); break;
        case 4: JR_FWD(1, 1); break;
        case 5: JR_FWD(1, 2); break;
        case 6: JR_FWD(2, 0); break;
        case 7: JR_FWD(2, 1); break;
        default: JR_FWD(2, 2); break;
    }
#undef JR_FWD
}

}  // namespace jr
