// SoftRas backward for gfx950 (MI355X).
//
// Replaces backward_soft_rasterize_cuda_kernel (SRK:1177-1360).  The reference runs one thread
// per pixel that gathers its <=K buffered faces from global memory (63 scattered floats per
// pixel-face pair) and issues 9+3T float atomics per pair.  Here the walk is turned around:
// a workgroup owns a 16x16 tile and streams the tile's ascending face list through LDS (same
// records as the forward); every lane keeps its K buffered ids SORTED in registers with a cursor,
// so "is this face in my buffer" is one compare.  Contributions of the lanes that hold the face
// are summed across the wavefront with DPP row shifts/broadcasts and lane 63 issues ONE atomic
// per gradient component per (wave, face) instead of one per pixel.
#include "jr_kernels.h"

namespace jr {

// Wavefront sum (64 lanes, all active): inclusive DPP scan, total lands in lane 63.
__device__ inline float wave_sum_to_lane63(float v) {
    int x;
#define JR_DPP_ADD(ctrl, rmask)                                                            \
    x = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false); \
    v += __builtin_bit_cast(float, x);
    JR_DPP_ADD(0x111, 0xf)  // row_shr:1
    JR_DPP_ADD(0x112, 0xf)  // row_shr:2
    JR_DPP_ADD(0x114, 0xf)  // row_shr:4
    JR_DPP_ADD(0x118, 0xf)  // row_shr:8
    JR_DPP_ADD(0x142, 0xa)  // row_bcast:15 into rows 1,3
    JR_DPP_ADD(0x143, 0xc)  // row_bcast:31 into rows 2,3
#undef JR_DPP_ADD
    return v;
}

template <int N>
__device__ inline void sort_ascending(int (&s)[N]) {
#pragma unroll
    for (int k = 2; k <= N; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const int a = s[i], b = s[l];
                    const int lo = min(a, b), hi = max(a, b);
                    s[i] = up ? lo : hi;
                    s[l] = up ? hi : lo;
                }
            }
}

template <int DIST, int RGB, int KCAP>
__global__ __launch_bounds__(WG_THREADS) void k_softras_backward(
    RasterParams p, int ntiles_total, const float* __restrict__ faces,
    const float* __restrict__ textures, const float* __restrict__ infos,
    const int* __restrict__ tile_count, const int* __restrict__ tile_base,
    const int* __restrict__ pool, const float* __restrict__ rgba, const float* __restrict__ aggrs,
    const int32_t* __restrict__ ids, const float* __restrict__ grad_rgba,
    float* __restrict__ grad_faces, float* __restrict__ grad_textures) {
    __shared__ float4 s_raw[CHUNK * REC_F4];
    FaceRec* s_rec = reinterpret_cast<FaceRec*>(s_raw);

    const int per_xcd = gridDim.x >> 3;
    const int t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= ntiles_total) return;
    const int n = tile_count[t];
    if (n == 0) return;

    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int b = t / tiles_per_img;
    const int tt = t - b * tiles_per_img;
    const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int col = tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int row = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool valid = col < p.IS && row < p.IS;
    const size_t pp = (size_t)p.IS * p.IS;
    const size_t pn = valid ? (size_t)row * p.IS + col : 0;
    const float xp = pixel_centre(col, p.IS);
    const float yp = pixel_centre(p.IS - 1 - row, p.IS);                      // SRK:1218-1221

    // this pixel's buffered face ids; the reference stops at the first -1 (SRK:1236-1238)
    constexpr int BIG = 0x7fffffff;
    int mine[KCAP];
    {
        const int32_t* ip = ids + (size_t)b * p.K * pp + pn;
        bool live = valid;
#pragma unroll
        for (int k = 0; k < KCAP; k++) {
            int v = -1;
            if (live && k < p.K) v = ip[(size_t)k * pp];
            live = live && v != -1;
            mine[k] = live ? v : BIG;
        }
    }
    if (!__syncthreads_or(mine[0] != BIG)) return;   // nothing buffered anywhere in this tile
    sort_ascending(mine);
    int cur = mine[0];

    float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
    float ssum = 1.f, smax = 0.f;
    if (valid) {
        const float* gp = grad_rgba + (size_t)b * 4 * pp + pn;
        const float* op = rgba + (size_t)b * 4 * pp + pn;
        g0 = gp[0]; g1 = gp[pp]; g2 = gp[2 * pp]; g3 = gp[3 * pp];
        o0 = op[0]; o1 = op[pp]; o2 = op[2 * pp]; o3 = op[3 * pp];
        ssum = aggrs[(size_t)b * 2 * pp + pn];                                  // SRK:1230-1231
        smax = aggrs[(size_t)b * 2 * pp + pp + pn];
    }

    const int* list = pool + tile_base[t];
    const float* fbase = faces + (size_t)b * p.NF * 9;
    const float* ibase = infos + (size_t)b * p.NF * 27;
    const float* tbase = textures + (size_t)b * p.NF * p.T * 3;
    float* gfbase = grad_faces + (size_t)b * p.NF * 9;
    float* gtbase = grad_textures + (size_t)b * p.NF * p.T * 3;
    const bool tex_reduce = p.tex == 1 || p.T == 1;   // texture gradient identical target for all lanes

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
            const FaceRec& r = s_rec[j];
            const int fn = r.id;
            const bool has = cur == fn;
            if (!__builtin_amdgcn_ballot_w64(has)) continue;     // wave-uniform skip

            float gv[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // x0 y0 z0 x1 y1 z1 x2 y2 z2
            float gt[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (has) {
                // advance the sorted cursor
#pragma unroll
                for (int k = 0; k + 1 < KCAP; k++) mine[k] = mine[k + 1];
                mine[KCAP - 1] = BIG;
                cur = mine[0];
                // check_border is repeated by the reference's backward (SRK:1244)
                const bool outside = xp > r.xhi || xp < r.xlo || yp > r.yhi || yp < r.ylo;
                if (!outside) {
                    const Bary w = barycentric(r, xp, yp);
                    float D, dis = 0.f;
                    Dist dd;
                    dd.sign = 0.f; dd.dx = 0.f; dd.dy = 0.f; dd.t0 = 0.f; dd.t1 = 0.f; dd.t2 = 0.f;
                    if (DIST == 0) D = 1.f;                                       // SRK:1258-1270
                    else if (DIST == 1) { dis = barycentric_dist(w); D = coverage(-dis / p.sigma); }
                    else {
                        dd = euclidean_p2f(r, w, xp, yp);
                        dis = dd.dx * dd.dx + dd.dy * dd.dy;
                        D = coverage(-dd.sign * dis / p.sigma);
                    }
                    float ca = g3;                                                // SRK:1281-1291
                    if (p.alpha == 1) ca /= p.NF;
                    else if (p.alpha == 2)
                        ca = (float)((double)ca * ((double)(1 - o3) / fmax((double)(1 - D), 1e-6)));
                    float cxy = 0.f;
                    cxy += ca;
                    const Bary wc = barycentric_clip(w);                          // SRK:1294-1296
                    const float zp = depth_of(r, wc);
                    const int texel = p.tex == 0 ? surface_texel(wc, p.R) : 0;
                    float tgs = 0.f;   // scale of the texture gradient for this pair
                    bool tex_on = false;
                    if (RGB == 0) {                                               // SRK:1299-1306
                        if ((float)fn == smax) { tgs = 1.f; tex_on = true; }
                    } else if (RGB == 1) {                                        // SRK:1308-1332
                        const float zn = (p.far_ - zp) / (p.far_ - p.near_);
                        const float zs = D * expf((zn - smax) / p.gamma) / ssum;
                        tgs = zs; tex_on = true;
                        float k0, k1, k2;
                        if (p.tex == 0) {
                            if (p.T == 1) { k0 = r.col[0]; k1 = r.col[1]; k2 = r.col[2]; }
                            else {
                                const float* tx_ = tbase + ((size_t)fn * p.T + texel) * 3;
                                k0 = tx_[0]; k1 = tx_[1]; k2 = tx_[2];
                            }
                        } else {                                                   // SRK:1147-1149 (affine)
                            k0 = (wc.w0 * r.col[0] + wc.w1 * r.col[3]) + wc.w2 * r.col[6];
                            k1 = (wc.w0 * r.col[1] + wc.w1 * r.col[4]) + wc.w2 * r.col[7];
                            k2 = (wc.w0 * r.col[2] + wc.w1 * r.col[5]) + wc.w2 * r.col[8];
                        }
                        float crgb = 0.f;
                        crgb += g0 * (k0 - o0);
                        crgb += g1 * (k1 - o1);
                        crgb += g2 * (k2 - o2);
                        crgb *= zs;
                        cxy += crgb / D;
                        const float cz = crgb / p.gamma / (p.near_ - p.far_) * zp * zp;
                        gv[2] = cz * wc.w0 / r.z[0] / r.z[0];
                        gv[5] = cz * wc.w1 / r.z[1] / r.z[1];
                        gv[8] = cz * wc.w2 / r.z[2] / r.z[2];
                    }
                    if (tex_on) {
                        if (p.tex == 1) {            // backward_sample_texture vertex: w[j]*grad (SRK:1170-1172)
                            const float wj[3] = {wc.w0, wc.w1, wc.w2};
#pragma unroll
                            for (int jv = 0; jv < 3; jv++) {
                                gt[3 * jv + 0] = tgs * (wj[jv] * g0);
                                gt[3 * jv + 1] = tgs * (wj[jv] * g1);
                                gt[3 * jv + 2] = tgs * (wj[jv] * g2);
                            }
                        } else if (p.T == 1) {
                            gt[0] = tgs * g0; gt[1] = tgs * g1; gt[2] = tgs * g2;
                        } else {                     // per-lane texel: direct atomics
                            float* gtx = gtbase + ((size_t)fn * p.T + texel) * 3;
                            atomicAdd(gtx + 0, tgs * g0);
                            atomicAdd(gtx + 1, tgs * g1);
                            atomicAdd(gtx + 2, tgs * g2);
                        }
                    }
                    cxy *= D * (1 - D) / p.sigma;                                 // SRK:1336
                    if (DIST == 1) {                                              // SRK:1118-1132
                        const int q = w.w0 > w.w1 ? (w.w1 > w.w2 ? 2 : 1) : (w.w0 > w.w2 ? 2 : 0);
                        const double mul = dis > 0 ? (2. * (double)sqrtf(dis)) : (2. * (double)sqrtf(-dis));
#pragma unroll
                        for (int l = 0; l < 2; l++) {
                            const float ql = q == 0 ? r.inv[l] : (q == 1 ? r.inv[3 + l] : r.inv[6 + l]);
#pragma unroll
                            for (int k = 0; k < 3; k++) {
                                float s = 0.f;
                                s += -ql * r.inv[3 * k + 0] * xp;
                                s += -ql * r.inv[3 * k + 1] * yp;
                                s += -ql * r.inv[3 * k + 2] * 1.f;
                                float v = s * cxy;
                                v = (float)((double)v * mul);
                                gv[3 * k + l] = v;
                            }
                        }
                    } else if (DIST == 2) {                                       // SRK:1341-1347
                        const float w0s[3] = {w.w0, w.w1, w.w2};
                        const float ts[3] = {dd.t0, dd.t1, dd.t2};
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            gv[3 * k + 0] = 2 * dd.sign * cxy * (ts[k] + w0s[k]) * dd.dx;
                            gv[3 * k + 1] = 2 * dd.sign * cxy * (ts[k] + w0s[k]) * dd.dy;
                        }
                    }
                }
            }
            // wavefront reduction, one atomic per component (SRK:1349-1358 issues one per pixel)
            float* gf = gfbase + (size_t)fn * 9;
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const float s = wave_sum_to_lane63(gv[k]);
                if (lane == 63) atomicAdd(gf + k, s);
            }
            if (tex_reduce) {
                float* gtx = gtbase + (size_t)fn * p.T * 3;
                const int nt = p.tex == 1 ? 9 : 3;
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    if (k < nt) {
                        const float s = wave_sum_to_lane63(gt[k]);
                        if (lane == 63) atomicAdd(gtx + k, s);
                    }
                }
            }
        }
    }
}

template <int DIST, int RGB>
static void launch_k(hipStream_t st, const RasterParams& p, int ntiles, const float* faces,
                     const float* textures, const float* infos, const BinWorkspace& ws,
                     const float* rgba, const float* aggrs, const int32_t* ids, const float* grad_rgba,
                     float* grad_faces, float* grad_textures) {
    const int grid = ((ntiles + 7) / 8) * 8;
    if (p.K <= 16)
        k_softras_backward<DIST, RGB, 16><<<grid, WG_THREADS, 0, st>>>(
            p, ntiles, faces, textures, infos, ws.tile_count, ws.tile_base, ws.pool, rgba, aggrs, ids,
            grad_rgba, grad_faces, grad_textures);
    else
        k_softras_backward<DIST, RGB, 64><<<grid, WG_THREADS, 0, st>>>(
            p, ntiles, faces, textures, infos, ws.tile_count, ws.tile_base, ws.pool, rgba, aggrs, ids,
            grad_rgba, grad_faces, grad_textures);
}

void launch_softras_backward(hipStream_t st, const RasterParams& p, const float* faces,
                             const float* textures, const float* rgba, const float* infos,
                             const float* aggrs, const int32_t* ids, const float* grad_rgba,
                             const BinWorkspace& ws, float* grad_faces, float* grad_textures) {
    const int ntiles = p.B * p.tiles_x * p.tiles_y;
    (void)hipMemsetAsync(grad_faces, 0, sizeof(float) * (size_t)p.B * p.NF * 9, st);          // SRK:1374
    (void)hipMemsetAsync(grad_textures, 0, sizeof(float) * (size_t)p.B * p.NF * p.T * 3, st); // SRK:1375
#define JR_BWD(D, R) \
    launch_k<D, R>(st, p, ntiles, faces, textures, infos, ws, rgba, aggrs, ids, grad_rgba, grad_faces, grad_textures)
    const int rgb = p.rgb == 0 ? 0 : (p.rgb == 1 ? 1 : 2);
    switch (p.dist * 3 + rgb) {
        case 0: JR_BWD(0, 0); break;
        case 1: JR_BWD(0, 1); break;
        case 2: JR_BWD(0, 2); break;
        case 3: JR_BWD(1, 0); break;
        case 4: JR_BWD(1, 1); break;
        case 5: JR_BWD(1, 2); break;
        case 6: JR_BWD(2, 0); break;
        case 7: JR_BWD(2, 1); break;
        default: JR_BWD(2, 2); break;
    }
#undef JR_BWD
}

}  // namespace jr
