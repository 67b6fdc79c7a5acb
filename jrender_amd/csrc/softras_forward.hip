// SoftRas forward rasteriser for gfx950 (MI355X).
//
// Replaces forward_soft_rasterize_cuda_kernel (SRK:243-456).  The reference runs one thread per
// pixel over ALL faces; here one 256-thread workgroup owns a 16x16-pixel screen tile (four
// wavefronts, each an 8x8 quad) and walks only the tile's ascending face list (binning.hip).
// Faces are staged CHUNK at a time into LDS records that every lane reads by broadcast; a
// wavefront skips a face with one s_cbranch_execz when none of its 64 pixels passes the border
// test.  The per-pixel state machine (alpha, online softmax over depth, K-nearest buffer) is kept
// entirely in VGPRs; it is inherently sequential in face order, which is why the lists are sorted.
// No MFMA: there is no dense contraction on this path.
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
__global__ __launch_bounds__(WG_THREADS) void k_softras_forward(
    RasterParams p, int ntiles_total, const float* __restrict__ faces,
    const float* __restrict__ textures, const float* __restrict__ infos,
    const int* __restrict__ tile_count, const int* __restrict__ tile_base,
    const int* __restrict__ pool, float* __restrict__ aggrs, float* __restrict__ rgba,
    int32_t* __restrict__ ids) {
    __shared__ float4 s_raw[CHUNK * REC_F4];
    FaceRec* s_rec = reinterpret_cast<FaceRec*>(s_raw);

    // XCD-aware tile order: consecutive workgroup ids land on different XCDs (id % 8); give each
    // XCD a contiguous run of tiles so that neighbouring tiles (which share faces) share an L2.
    const int per_xcd = gridDim.x >> 3;
    const int t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= ntiles_total) return;

    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int b = t / tiles_per_img;
    const int tt = t - b * tiles_per_img;
    const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int col = tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int row = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool valid = col < p.IS && row < p.IS;
    const float xp = pixel_centre(col, p.IS);
    const float yp = pixel_centre(p.IS - 1 - row, p.IS);                      // SRK:280-283

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

    const int n = tile_count[t];
    const int* list = pool + tile_base[t];
    const float* fbase = faces + (size_t)b * p.NF * 9;
    const float* ibase = infos + (size_t)b * p.NF * 27;
    const float* tbase = textures + (size_t)b * p.NF * p.T * 3;

    for (int s0 = 0; s0 < n; s0 += CHUNK) {
        const int cn = min(CHUNK, n - s0);
        __syncthreads();
        if (tid < cn) {
            const int fn = list[s0 + tid];
            FaceRec r;
            build_face_rec(r, fbase + (size_t)fn * 9, ibase + (size_t)fn * 27, p.rad, fn);
            const float* tx_ = tbase + (size_t)fn * p.T * 3;
            if (p.tex == 1) {
#pragma unroll
                for (int k = 0; k < 9; k++) r.col[k] = tx_[k];
            } else if (p.T == 1) {
                r.col[0] = tx_[0]; r.col[1] = tx_[1]; r.col[2] = tx_[2];
            }
            s_rec[tid] = r;
        }
        __syncthreads();

        for (int j = 0; j < cn; j++) {
            const float4 bb = s_raw[j * REC_F4];
            // check_border (SRK:28-34, :316): cull when strictly outside the grown box
            if (!valid || xp > bb.y || xp < bb.x || yp > bb.w || yp < bb.z) continue;
            const FaceRec& r = s_rec[j];
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
            const int fn = r.id;
            q.insert(fn, zp, p.K);

            if (RGB == 0) {                                                    // SRK:390-397
                if (zp < depth_min && pixel_inside(w) && (p.double_side || r.front)) {
                    depth_min = zp; face_min = fn;
                    if (p.tex == 0) {
                        if (p.T == 1) { c0 = r.col[0]; c1 = r.col[1]; c2 = r.col[2]; }
                        else {
                            const float* tx_ = tbase + ((size_t)fn * p.T + surface_texel(wc, p.R)) * 3;
                            c0 = tx_[0]; c1 = tx_[1]; c2 = tx_[2];
                        }
                    } else {                                                   // SRK:168-171
                        c0 = ((wc.w0 * r.col[0] / r.z[0] + wc.w1 * r.col[3] / r.z[1]) + wc.w2 * r.col[6] / r.z[2]) * zp;
                        c1 = ((wc.w0 * r.col[1] / r.z[0] + wc.w1 * r.col[4] / r.z[1]) + wc.w2 * r.col[7] / r.z[2]) * zp;
                        c2 = ((wc.w0 * r.col[2] / r.z[0] + wc.w1 * r.col[5] / r.z[1]) + wc.w2 * r.col[8] / r.z[2]) * zp;
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
                        if (p.T == 1) { k0 = r.col[0]; k1 = r.col[1]; k2 = r.col[2]; }
                        else {
                            const float* tx_ = tbase + ((size_t)fn * p.T + surface_texel(wc, p.R)) * 3;
                            k0 = tx_[0]; k1 = tx_[1]; k2 = tx_[2];
                        }
                    } else {
                        k0 = ((wc.w0 * r.col[0] / r.z[0] + wc.w1 * r.col[3] / r.z[1]) + wc.w2 * r.col[6] / r.z[2]) * zp;
                        k1 = ((wc.w0 * r.col[1] / r.z[0] + wc.w1 * r.col[4] / r.z[1]) + wc.w2 * r.col[7] / r.z[2]) * zp;
                        k2 = ((wc.w0 * r.col[2] / r.z[0] + wc.w1 * r.col[5] / r.z[1]) + wc.w2 * r.col[8] / r.z[2]) * zp;
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
static void launch_k(hipStream_t st, const RasterParams& p, int ntiles, const float* faces,
                     const float* textures, const float* infos, const BinWorkspace& ws,
                     float* aggrs, float* rgba, int32_t* ids) {
    const int grid = ((ntiles + 7) / 8) * 8;
    if (p.K <= 16)
        k_softras_forward<DIST, RGB, 16><<<grid, WG_THREADS, 0, st>>>(
            p, ntiles, faces, textures, infos, ws.tile_count, ws.tile_base, ws.pool, aggrs, rgba, ids);
    else
        k_softras_forward<DIST, RGB, 64><<<grid, WG_THREADS, 0, st>>>(
            p, ntiles, faces, textures, infos, ws.tile_count, ws.tile_base, ws.pool, aggrs, rgba, ids);
}

void launch_softras_forward(hipStream_t st, const RasterParams& p, const float* faces,
                            const float* textures, const float* infos, const BinWorkspace& ws,
                            float* aggrs, float* rgba, int32_t* ids) {
    const int ntiles = p.B * p.tiles_x * p.tiles_y;
#define JR_FWD(D, R) launch_k<D, R>(st, p, ntiles, faces, textures, infos, ws, aggrs, rgba, ids)
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
