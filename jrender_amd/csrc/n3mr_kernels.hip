// NMR ("n3mr", Kato et al. 2018) hard rasteriser + approximate gradients for gfx950.
//
// Replaces the five JIT ops of jrender/renderer/dr/n3mr/cuda/rasterize.py ("N3K"):
//   forward_face_index_map   N3K:35-164   -> k_n3mr_zbuffer + k_n3mr_resolve
//   forward_texture_sampling N3K:228-298  -> fused into k_n3mr_resolve
//   backward_pixel_map       N3K:352-610  -> k_n3mr_backward_pixel_map
//   backward_textures        N3K:660-694  -> k_n3mr_backward_textures
//   backward_depth_map       N3K:739-788  -> k_n3mr_backward_depth
// and the host tensor ops between them (background compositing, alpha = face_index >= 0,
// N3F:135-148), which are fused into the resolve kernel.
//
// What is different from the reference's organisation:
//  * the z-buffer is a 64-bit atomicMin on (depth bits << 32 | face id) per pixel instead of a
//    per-pixel spin lock (N3K:140-161).  Depths are > near >= 0, so the float bit pattern orders
//    like the value; equal depths resolve to the LOWEST face index — deterministic, and the same
//    answer a serial run of the reference gives (its GPU run is racy on ties).
//  * one WAVEFRONT per face (lanes over the bounding-box pixels / over the scan lines of an edge)
//    instead of one thread per face: a face covering thousands of pixels no longer serialises.
//  * the winner's weights, face_inv, texture sample, colour and alpha are written by one per-pixel
//    resolve pass after the depth test (no read-modify-write under a lock).
#include "jr_kernels.h"

namespace jr {

struct N3Params {
    int B, NF, TS, IS;
    float near_, far_, eps;
    int return_rgb, return_alpha, return_depth;
    float bg[3];
};

// pixel-space inverse of the face (N3K:66-86); p[k] = 0.5 * (ndc * is + is - 1)
__device__ inline void n3_face_inv(const float* __restrict__ face, int is, float (&px)[3], float (&py)[3],
                                   float (&inv)[9]) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        px[k] = 0.5f * (face[3 * k] * is + is - 1);
        py[k] = 0.5f * (face[3 * k + 1] * is + is - 1);
    }
    inv[0] = py[1] - py[2]; inv[1] = px[2] - px[1]; inv[2] = px[1] * py[2] - px[2] * py[1];
    inv[3] = py[2] - py[0]; inv[4] = px[0] - px[2]; inv[5] = px[2] * py[0] - px[0] * py[2];
    inv[6] = py[0] - py[1]; inv[7] = px[1] - px[0]; inv[8] = px[0] * py[1] - px[1] * py[0];
    const float den = (px[2] * (py[0] - py[1]) + px[0] * (py[1] - py[2])) + px[1] * (py[2] - py[0]);
#pragma unroll
    for (int k = 0; k < 9; k++) inv[k] = inv[k] / den;
}

// barycentric weights of the integer pixel (xi, yi), clamped and renormalised, and the depth
// (N3K:120-134).  Returns false when the pixel centre fails one of the three edge tests (N3K:113-116).
__device__ inline bool n3_pixel(const float* __restrict__ f, const float (&inv)[9], int xi, int yi, int is,
                                float (&w)[3], float& zp) {
    const float yp = (float)(2 * yi + 1 - is) / (float)is;        // == (float)((2.*yi + 1 - is) / is)
    const float xp = (float)(2 * xi + 1 - is) / (float)is;
    if (((yp - f[1]) * (f[3] - f[0]) < (xp - f[0]) * (f[4] - f[1])) ||
        ((yp - f[4]) * (f[6] - f[3]) < (xp - f[3]) * (f[7] - f[4])) ||
        ((yp - f[7]) * (f[0] - f[6]) < (xp - f[6]) * (f[1] - f[7])))
        return false;
    float ws = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float v = (inv[3 * k] * xi + inv[3 * k + 1] * yi) + inv[3 * k + 2];
        w[k] = fminf(fmaxf(v, 0.f), 1.f);                          // min(max(w, 0.), 1.): exact selection
        ws += w[k];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) w[k] = w[k] / ws;
    zp = 1.0f / ((w[0] / f[2] + w[1] / f[5]) + w[2] / f[8]);     // double reciprocal == IEEE float quotient
    return true;
}

__global__ __launch_bounds__(256) void k_n3mr_zbuffer(N3Params p, const float* __restrict__ faces,
                                                      float* __restrict__ faces_inv,
                                                      unsigned long long* __restrict__ zkey) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= p.B * p.NF) return;
    const int bn = wave / p.NF, fn = wave - bn * p.NF;
    const float* f = faces + (size_t)wave * 9;
    if ((f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0])) {               // back side, N3K:63
        if (lane < 9) faces_inv[(size_t)wave * 9 + lane] = 0.f;                        // (the reference's output is pre-zeroed; no memset launch here)
        return;
    }
    float px[3], py[3], inv[9];
    n3_face_inv(f, p.IS, px, py, inv);
    if (lane < 9) faces_inv[(size_t)wave * 9 + lane] = inv[lane];
    float x_min = p.IS, y_min = p.IS, x_max = 0, y_max = 0;                           // N3K:89-99
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (px[k] < x_min) x_min = px[k];
        if (px[k] > x_max) x_max = px[k];
        if (py[k] < y_min) y_min = py[k];
        if (py[k] > y_max) y_max = py[k];
    }
    const int ix0 = max(0, (int)x_min), ix1 = min(p.IS - 1, (int)x_max);
    const int iy0 = max(0, (int)y_min), iy1 = min(p.IS - 1, (int)y_max);
    if (ix1 < ix0 || iy1 < iy0) return;
    const int hgt = iy1 - iy0 + 1;
    const long npix = (long)(ix1 - ix0 + 1) * hgt;
    for (long idx = lane; idx < npix; idx += 64) {
        const int xi = ix0 + (int)(idx / hgt), yi = iy0 + (int)(idx % hgt);
        float w[3], zp;
        if (!n3_pixel(f, inv, xi, yi, p.IS, w, zp)) continue;
        // N3K:136 (zp <= near || far <= zp) AND N3K:147 (zp < depth_map, false for NaN): a zero-area face
        // gives w_sum == 0 -> zp = NaN, which the reference never draws; reject unordered depths here
        if (!(zp > p.near_ && zp < p.far_)) continue;
        const unsigned long long key = ((unsigned long long)__builtin_bit_cast(unsigned, zp) << 32) | (unsigned)fn;
        atomicMin(&zkey[(size_t)bn * p.IS * p.IS + (size_t)yi * p.IS + xi], key);
    }
}

__device__ inline float n3_bcast(float v, int s) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), s));
}

// Round 5: the same pass with the per-FACE work in the lanes.  Above, every one of a face's 64 lanes computes the nine IEEE quotients of
// its inverse, the bounding box and the back-face test - ~250 wave-uniform VALU instructions per front face before the first pixel
// (SQ_INSTS_VALU 2.1e7 per launch of 78 000 faces: the kernel is VALU-bound at 34 of its 41 us).  Here a wavefront takes G consecutive
// faces: lane l < G sets face l up (G = 16: a quarter of the lanes busy for ~1/16 of the former work; 4 900 wavefronts keep the GPU
// filled, with G = 64 there would be 1.2 per SIMD), then the wavefront walks its front faces one after the other with the lanes over the
// pixels of the bounding box as before - the face's 24 values arrive by v_readlane.  Bounding boxes up to 64 pixels high (all but
// screen-filling faces) map lane -> (column, row) with one reciprocal instead of a 64-bit division per pixel.  Same arithmetic per face
// and per pixel, same keys: bit-identical maps.
template <int G>
__global__ __launch_bounds__(256) void k_n3mr_zbuffer_grouped(N3Params p, const float* __restrict__ faces,
                                                              float* __restrict__ faces_inv,
                                                              unsigned long long* __restrict__ zkey) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int total = p.B * p.NF, face0 = wave * G;
    if (face0 >= total) return;
    const bool have = lane < G && face0 + lane < total;
    const int fi = have ? face0 + lane : face0;
    float f[9];
#pragma unroll
    for (int k = 0; k < 9; k++) f[k] = faces[(size_t)fi * 9 + k];
    const bool back = (f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0]);                  // back side, N3K:63
    float px[3], py[3], inv[9];
    n3_face_inv(f, p.IS, px, py, inv);
    if (have) {
#pragma unroll
        for (int k = 0; k < 9; k++) faces_inv[(size_t)fi * 9 + k] = back ? 0.f : inv[k];             // (the reference's output is pre-zeroed; no memset launch here)
    }
    float x_min = p.IS, y_min = p.IS, x_max = 0, y_max = 0;                                          // N3K:89-99
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (px[k] < x_min) x_min = px[k];
        if (px[k] > x_max) x_max = px[k];
        if (py[k] < y_min) y_min = py[k];
        if (py[k] > y_max) y_max = py[k];
    }
    const int ix0 = max(0, (int)x_min), ix1 = min(p.IS - 1, (int)x_max);
    const int iy0 = max(0, (int)y_min), iy1 = min(p.IS - 1, (int)y_max);
    const int bn_l = fi / p.NF, fn_l = fi - bn_l * p.NF;
    unsigned long long todo = ballot(have && !back && ix0 <= ix1 && iy0 <= iy1);
    while (todo) {
        const int s = __builtin_ctzll(todo);
        todo &= todo - 1;
        float bf[9], binv[9];
#pragma unroll
        for (int k = 0; k < 9; k++) { bf[k] = n3_bcast(f[k], s); binv[k] = n3_bcast(inv[k], s); }
        const int bx0 = __builtin_amdgcn_readlane(ix0, s), bx1 = __builtin_amdgcn_readlane(ix1, s);
        const int by0 = __builtin_amdgcn_readlane(iy0, s), by1 = __builtin_amdgcn_readlane(iy1, s);
        const int bn = __builtin_amdgcn_readlane(bn_l, s), fn = __builtin_amdgcn_readlane(fn_l, s);
        unsigned long long* zk = zkey + (size_t)bn * p.IS * p.IS;
        const int hgt = by1 - by0 + 1;
        auto pixel = [&](int xi, int yi) {
            float w[3], zp;
            if (!n3_pixel(bf, binv, xi, yi, p.IS, w, zp)) return;
            // N3K:136 (zp <= near || far <= zp) AND N3K:147 (zp < depth_map, false for NaN): a zero-area face
            // gives w_sum == 0 -> zp = NaN, which the reference never draws; reject unordered depths here
            if (!(zp > p.near_ && zp < p.far_)) return;
            const unsigned long long key = ((unsigned long long)__builtin_bit_cast(unsigned, zp) << 32) | (unsigned)fn;
            atomicMin(&zk[(size_t)yi * p.IS + xi], key);
        };
        if (hgt <= 64) {
            // lane -> (column q, row r) of a block of 64 / hgt columns: q = floor(lane / hgt) through one reciprocal ((lane + 0.5) / hgt is
            // at least 0.5 / 64 away from an integer: exact), then whole blocks of columns per trip
            const float rh = __builtin_amdgcn_rcpf((float)hgt);
            const int q = (int)(((float)lane + 0.5f) * rh), r = lane - q * hgt;
            const int cpi = __builtin_amdgcn_readfirstlane((int)(64.5f * rh));            // columns per trip, >= 1
            for (int c0 = bx0; c0 <= bx1; c0 += cpi) {
                const int xi = c0 + q;
                if (q < cpi && xi <= bx1) pixel(xi, by0 + r);
            }
        } else {
            const long npix = (long)(bx1 - bx0 + 1) * hgt;
            for (long idx = lane; idx < npix; idx += 64) pixel(bx0 + (int)(idx / hgt), by0 + (int)(idx % hgt));
        }
    }
}

// per pixel: winner -> face_index / depth / weights / face_inv; texture sample, background, alpha
__global__ __launch_bounds__(256) void k_n3mr_resolve(
    N3Params p, const float* __restrict__ faces, const float* __restrict__ textures,
    const float* __restrict__ faces_inv, unsigned long long* __restrict__ zkey,
    int32_t* __restrict__ face_index_map, float* __restrict__ weight_map, float* __restrict__ depth_map,
    float* __restrict__ face_inv_map, float* __restrict__ rgb_map, float* __restrict__ alpha_map,
    int32_t* __restrict__ sampling_index_map, float* __restrict__ sampling_weight_map) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long pp = (long)p.IS * p.IS;
    if (i >= p.B * pp) return;
    const unsigned long long key = zkey[i];
    zkey[i] = ~0ull;                       // the keys leave this kernel cleared: the next forward of the context needs no memset launch in front of its z-buffer pass
    const bool hit = key != ~0ull;
    const int fn = hit ? (int)(unsigned)key : -1;
    face_index_map[i] = fn;
    if (p.return_alpha) alpha_map[i] = hit ? 1.f : 0.f;                               // N3F:145-148
    float w[3] = {0.f, 0.f, 0.f}, depth = p.far_;
    float inv[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int bn = (int)(i / pp);
    const float* f = faces + ((size_t)bn * p.NF + (hit ? fn : 0)) * 9;
    if (hit) {
        const long r = i - bn * pp;
        const int yi = (int)(r / p.IS), xi = (int)(r - (long)yi * p.IS);
        const float* fi = faces_inv + ((size_t)bn * p.NF + fn) * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) inv[k] = fi[k];
        n3_pixel(f, inv, xi, yi, p.IS, w, depth);                 // the same arithmetic that won the test
    }
    depth_map[i] = depth;
#pragma unroll
    for (int k = 0; k < 3; k++) weight_map[3 * i + k] = w[k];
    if (p.return_depth) {
#pragma unroll
        for (int k = 0; k < 9; k++) face_inv_map[9 * i + k] = inv[k];
    }
    if (!p.return_rgb) return;
    float pix[3] = {p.bg[0], p.bg[1], p.bg[2]};                                        // N3F:135-143
    int sidx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float swt[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (hit) {                                                                         // N3K:264-296
        const int ts = p.TS;
        const float* tex = textures + ((size_t)bn * p.NF + fn) * ts * ts * ts * 3;
        float tif[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float t = w[k] * (ts - 1) * (depth / f[3 * k + 2]);
            t = fmaxf(t, 0.f);
            t = fminf(t, ts - 1 - p.eps);
            tif[k] = t;
        }
        float np_[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int pn = 0; pn < 8; pn++) {
            float ww = 1;
            int ti[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (((pn >> k) % 2) == 0) { ww *= 1 - (tif[k] - (int)tif[k]); ti[k] = (int)tif[k]; }
                else { ww *= tif[k] - (int)tif[k]; ti[k] = (int)tif[k] + 1; }
            }
            const int isc = ti[0] * ts * ts + ti[1] * ts + ti[2];
#pragma unroll
            for (int k = 0; k < 3; k++) np_[k] += ww * tex[isc * 3 + k];
            sidx[pn] = isc; swt[pn] = ww;
        }
        pix[0] = np_[0]; pix[1] = np_[1]; pix[2] = np_[2];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) rgb_map[3 * i + k] = pix[k];
#pragma unroll
    for (int k = 0; k < 8; k++) { sampling_index_map[8 * i + k] = sidx[k]; sampling_weight_map[8 * i + k] = swt[k]; }
}

// ---- NMR's approximate image gradient (N3K:352-610) ---------------------------------------------------
// For every edge of a face and both image axes the reference walks the scan positions d0 the edge crosses;
// at each one it compares the pixel just inside the edge with (a) every pixel from the outside neighbour to
// the image border ("out" walk, only when the inside pixel shows this face) and (b) every pixel of this face
// from the inside pixel to the opposite edge ("in" walk), and accumulates -max(diff, 0)/(dist +- eps) into the
// two vertices of the edge, where diff = sum_k (map_k[m] - ref_k) * grad_k[m] over alpha, r, g, b.
//
// Organisation here (one wavefront per face):
//   * a per-pixel PACK pass writes S[m] = sum_k map_k[m]*grad_k[m] and the four gradients as one float4 + one
//     float, in row-major AND column-major order, so that diff = S[m] - sum_k ref_k*grad_k[m] costs two loads
//     per visited pixel and every walk is contiguous in memory whichever axis it runs along (the column walks
//     of the reference touch one cache line per pixel);
//   * lanes = scan positions for the set-up (cross points, inside/outside pixels, divisions) and for the short
//     "in" walks (each lane walks its own scan line);
//   * the long "out" walks take the scan lines one after the other and spread the PIXEL walk over the 64 lanes;
//   * gradient-only arithmetic in float with v_rcp_f32 (the reference promotes dist to double and divides);
//     per-lane partial sums are reduced once per face and stored without atomics, like the reference.
struct N3Planes {
    const float4* sg;        // (g_alpha, g_r, g_g, g_b)   (round 5: the four gradients travel together - the products with the reference pixel pair up
    const float* gb;         // S = sum_k map_k * grad_k       without register moves; the scalar plane holds S)
    const int32_t* fidx;
};

// One 32x32-pixel tile per workgroup: the row-major copies are written as the pixels are read; the column-major copies go
// through an LDS tile so that they, too, leave as contiguous runs (round 4: a thread writing its own transposed element touched
// one cache line per lane - 0.034 of the backward's 0.41 ms).
// sum_k map_k * grad_k over (alpha, r, g, b) in the association both the pack pass and the walks use
__device__ inline float n3_dot4(float a, float c0, float c1, float c2, const float4 g) {
    return __builtin_fmaf(c1, g.z, a * g.x) + __builtin_fmaf(c2, g.w, c0 * g.y);
}
constexpr int N3_PACK_TILE = 32;
__global__ __launch_bounds__(256) void k_n3mr_pack(
    N3Params p, const int32_t* __restrict__ face_index_map, const float* __restrict__ rgb_map,
    const float* __restrict__ alpha_map, const float* __restrict__ grad_rgb_map,
    const float* __restrict__ grad_alpha_map, float4* __restrict__ sg, float* __restrict__ gb,
    float4* __restrict__ sg_t, float* __restrict__ gb_t, int32_t* __restrict__ fidx_r, int32_t* __restrict__ fidx_t,
    int* __restrict__ line_count, int nsub) {
    __shared__ float4 s_v[N3_PACK_TILE][N3_PACK_TILE + 1];
    __shared__ float s_g[N3_PACK_TILE][N3_PACK_TILE + 1];
    __shared__ int32_t s_f[N3_PACK_TILE][N3_PACK_TILE + 1];
    // (the crossing lists of the line-walk kernel start empty: cleared here instead of by a memset launch of their own)
    for (int c = blockIdx.x * 256 + threadIdx.x; c < nsub; c += gridDim.x * 256) line_count[c] = 0;
    const int is = p.IS, tiles = (is + N3_PACK_TILE - 1) / N3_PACK_TILE;
    const int bn = blockIdx.x / (tiles * tiles), tt = blockIdx.x - bn * tiles * tiles;
    const int y0 = (tt / tiles) * N3_PACK_TILE, x0 = (tt % tiles) * N3_PACK_TILE;
    const size_t pbase = (size_t)bn * is * is;
    const int tx = threadIdx.x & 31, ty0 = threadIdx.x >> 5;          // 8 rows of 32 threads
#pragma unroll
    for (int r = 0; r < N3_PACK_TILE; r += 8) {
        const int ty = ty0 + r, x = x0 + tx, y = y0 + ty;
        if (x >= is || y >= is) continue;
        const size_t i = pbase + (size_t)y * is + x;
        float a = 0.f, ga = 0.f, c[3] = {0.f, 0.f, 0.f}, g[3] = {0.f, 0.f, 0.f};
        if (p.return_alpha) { ga = grad_alpha_map[i]; a = alpha_map[i]; }
        if (p.return_rgb)
            for (int k = 0; k < 3; k++) { g[k] = grad_rgb_map[3 * i + k]; c[k] = rgb_map[3 * i + k]; }
        const float S = n3_dot4(a, c[0], c[1], c[2], make_float4(ga, g[0], g[1], g[2]));
        const float4 v = make_float4(ga, g[0], g[1], g[2]);
        const int32_t fi = face_index_map[i];
        sg[i] = v; gb[i] = S; fidx_r[i] = fi;
        s_v[ty][tx] = v; s_g[ty][tx] = S; s_f[ty][tx] = fi;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < N3_PACK_TILE; r += 8) {
        const int cx = ty0 + r, cy = tx, x = x0 + cx, y = y0 + cy;      // column x of the tile, consecutive threads along y
        if (x >= is || y >= is) continue;
        const size_t it = pbase + (size_t)x * is + y;
        sg_t[it] = s_v[cy][cx]; gb_t[it] = s_g[cy][cx]; fidx_t[it] = s_f[cy][cx];
    }
}

struct N3Ref { float a, c0, c1, c2; };       // the reference pixel of a walk (alpha, r, g, b)

// diff = S - sum_k ref_k * grad_k.  Both sums go through n3_dot4 - (alpha, g) and (r, b) paired: one v_pk_mul_f32 + one v_pk_fma_f32 + one
// add where four products and three adds cost 20 issue cycles per visited pixel group - so a pixel with the reference's colours gives exactly 0.
__device__ inline float n3_diff(const float4 g, float S, const N3Ref& r) {
    return S - n3_dot4(r.a, r.c0, r.c1, r.c2, g);
}

// -= diff / (dist +- eps) for the two vertices of the edge (N3K:496-505, :583-592)
__device__ inline void n3_push(float diff, int d1, float cross, float ta, float tb, bool ha, bool hb,
                               float two_over_is, float eps, float& acc_a, float& acc_b) {
    const float dd = (float)d1 - cross;
    if (ha) {
        float dist = ta * dd * two_over_is;
        dist = (0 < dist) ? dist + eps : dist - eps;
        acc_a -= diff * __builtin_amdgcn_rcpf(dist);
    }
    if (hb) {
        float dist = tb * dd * two_over_is;
        dist = (0 < dist) ? dist + eps : dist - eps;
        acc_b -= diff * __builtin_amdgcn_rcpf(dist);
    }
}


// One (edge, axis) of one face.  q = the edge's two vertices and the opposite one in pixel coordinates with
// the walking axis second; `in` = planes in which consecutive scan positions are contiguous (lane = scan
// line), `out` = planes in which consecutive walk positions are contiguous (lane = walk position).
__device__ inline void n3_edge_axis(const N3Params& p, int axis, const float (&q)[3][2], int fn, size_t mbase,
                                    const int32_t* __restrict__ face_index_map,
                                    const float* __restrict__ rgb_map, const float* __restrict__ alpha_map,
                                    const N3Planes& in, const N3Planes& out, int lane, float& acc_a, float& acc_b) {
    const int is = p.IS;
    const float q00 = q[0][0], q01 = q[0][1], q10 = q[1][0], q11 = q[1][1], q20 = q[2][0], q21 = q[2][1];
    const int direction = axis == 0 ? (q00 < q10 ? -1 : 1) : (q00 < q10 ? 1 : -1);         // N3K:407-411
    const int d0_from = (int)fmax((double)ceilf(fminf(q00, q10)), 0.);
    const int d0_to = (int)fmin((double)fmaxf(q00, q10), is - 1.);
    const float e10 = q10 - q00;
    const float slope = (q11 - q01) / e10;
    const float s20 = (q21 - q01) / (q20 - q00), s12 = (q11 - q21) / (q10 - q20);
    const float two_over_is = 2.f / is;
    const bool use_rgb = p.return_rgb, use_a = p.return_alpha;
    for (int c0 = d0_from; c0 <= d0_to; c0 += 64) {
        // ---- lane = scan position ----
        const int d0 = c0 + lane;
        const float d1_cross = slope * (d0 - q00) + q01;
        const int d1_in = 0 < direction ? (int)floorf(d1_cross) : (int)ceilf(d1_cross);
        const int d1_out = d1_in + direction;
        const bool ok = d0 <= d0_to && !(d1_in < 0 || is <= d1_in) && !(d1_out < 0 || is <= d1_out);
        const bool ha = q10 != d0, hb = q00 != d0;
        const float ta = e10 / (q10 - d0), tb = e10 / (d0 - q00);
        N3Ref rin = {0.f, 0.f, 0.f, 0.f}, rout = {0.f, 0.f, 0.f, 0.f};
        int fin = -1, from = 1, to = 0;
        if (ok) {
            const size_t idx_in = mbase + (axis == 0 ? (size_t)d1_in * is + d0 : (size_t)d0 * is + d1_in);
            const size_t idx_out = mbase + (axis == 0 ? (size_t)d1_out * is + d0 : (size_t)d0 * is + d1_out);
            fin = face_index_map[idx_in];
            if (use_a) { rin.a = alpha_map[idx_in]; rout.a = alpha_map[idx_out]; }
            if (use_rgb) {
                rin.c0 = rgb_map[idx_in * 3]; rin.c1 = rgb_map[idx_in * 3 + 1]; rin.c2 = rgb_map[idx_in * 3 + 2];
                rout.c0 = rgb_map[idx_out * 3]; rout.c1 = rgb_map[idx_out * 3 + 1]; rout.c2 = rgb_map[idx_out * 3 + 2];
            }
            const float cross2 = ((d0 - q00) * (d0 - q20) < 0) ? s20 * (d0 - q00) + q01 : s12 * (d0 - q20) + q21;
            const int d1_limit = 0 < direction ? (int)ceilf(cross2) : (int)floorf(cross2);    // N3K:520-528
            from = max(min(d1_in, d1_limit), 0);
            to = min(max(d1_in, d1_limit), is - 1);
        }
        // ---- "in" walk: every lane walks its own scan line (contiguous across lanes in the `in` planes) ----
        for (int k = 0; ballot(from + k <= to) != 0ull; k++) {
            const int d1 = from + k;
            if (d1 > to) continue;
            const size_t m = mbase + (size_t)d1 * is + d0;
            if (in.fidx[m] != fn) continue;
            const float diff = n3_diff(in.sg[m], in.gb[m], rout);
            if (diff <= 0) continue;
            n3_push(diff, d1, d1_cross, ta, tb, ha, hb, two_over_is, p.eps, acc_a, acc_b);
        }
        // ---- "out" walks: scan lines one after the other, the pixel walk spread over the lanes ----
        unsigned long long vis = ballot(ok && fin == fn);
        while (vis) {
            const int s = __builtin_ctzll(vis);
            vis &= vis - 1;
            const int bd0 = c0 + s;
            const int b_out = __builtin_amdgcn_readlane(d1_out, s);
            const float b_cross = n3_bcast(d1_cross, s), b_ta = n3_bcast(ta, s), b_tb = n3_bcast(tb, s);
            const bool b_ha = q10 != bd0, b_hb = q00 != bd0;
            const N3Ref r = {n3_bcast(rin.a, s), n3_bcast(rin.c0, s), n3_bcast(rin.c1, s), n3_bcast(rin.c2, s)};
            const int d1_limit = 0 < direction ? is - 1 : 0;
            const int wf = max(min(b_out, d1_limit), 0), wt = min(max(b_out, d1_limit), is - 1);
            for (int d1 = wf + lane; d1 <= wt; d1 += 64) {
                const size_t m = mbase + (size_t)bd0 * is + d1;
                const float diff = n3_diff(out.sg[m], out.gb[m], r);
                if (diff <= 0) continue;
                n3_push(diff, d1, b_cross, b_ta, b_tb, b_ha, b_hb, two_over_is, p.eps, acc_a, acc_b);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_n3mr_backward_pixel_map(
    N3Params p, const float* __restrict__ faces, const int32_t* __restrict__ face_index_map,
    const float* __restrict__ rgb_map, const float* __restrict__ alpha_map,
    N3Planes rowmajor, N3Planes colmajor, float* __restrict__ grad_faces) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= p.B * p.NF) return;
    const int bn = wave / p.NF, fn = wave - bn * p.NF;
    const int is = p.IS;
    const float* face = faces + (size_t)wave * 9;
    if ((face[7] - face[1]) * (face[3] - face[0]) < (face[4] - face[1]) * (face[6] - face[0])) return;
    const size_t mbase = (size_t)bn * is * is;
    float g[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int edge = 0; edge < 3; edge++) {
        const int pi0 = edge % 3, pi1 = (edge + 1) % 3, pi2 = (edge + 2) % 3;
        const int pis[3] = {pi0, pi1, pi2};
        float pp_[3][2];
#pragma unroll
        for (int n = 0; n < 3; n++)
#pragma unroll
            for (int d = 0; d < 2; d++) pp_[n][d] = 0.5f * (face[3 * pis[n] + d] * is + is - 1);
#pragma unroll
        for (int axis = 0; axis < 2; axis++) {
            float q[3][2];
#pragma unroll
            for (int n = 0; n < 3; n++)
#pragma unroll
                for (int d = 0; d < 2; d++) q[n][d] = pp_[n][(d + axis) % 2];
            float acc_a = 0.f, acc_b = 0.f;
            // axis 0: scan positions are columns, walks run along rows  -> in = row-major, out = column-major
            // axis 1: scan positions are rows, walks run along columns  -> in = column-major, out = row-major
            n3_edge_axis(p, axis, q, fn, mbase, face_index_map, rgb_map, alpha_map, axis == 0 ? rowmajor : colmajor,
                         axis == 0 ? colmajor : rowmajor, lane, acc_a, acc_b);
            g[pi0 * 3 + (1 - axis)] += acc_a;
            g[pi1 * 3 + (1 - axis)] += acc_b;
        }
    }
#pragma unroll
    for (int k = 0; k < 9; k++) {
        float v = g[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) grad_faces[(size_t)wave * 9 + k] = v;
    }
}

__device__ inline float n3_wave_sum(float v) {            // all lanes active; result uniform
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));   // row_mirror
    return (n3_bcast(v, 0) + n3_bcast(v, 16)) + (n3_bcast(v, 32) + n3_bcast(v, 48));
}

// ---- out-walks per SCAN LINE (tune::n3_line_walks, round 4) ------------------------------------------------------------------
// The out-walks of the pixel-map gradient are 0.375 ms of the NMR backward and move 2.45 GB through the L2s (6.5 TB/s: every one
// of the ~120 M pixel visits fetches its 20 bytes from behind them; profiles/r04_experiments.md calls 7, 18) although the two
// packed planes are 40 MB: a walk runs along ONE image column or row, and the ~140 walks that run along the same line are
// spread over faces all around the mesh.  So the walks are regrouped by line: the per-face kernel appends a 36-byte crossing
// record to its line's list (one atomic), and here one workgroup per line copies the line's 20 B x image_size into LDS once
// and runs all its crossings from there - one crossing per wavefront at a time, lanes along the walk - adding the two sums
// of a crossing to the face's gradient with float atomics (the per-face kernel has stored its part before, on the same stream).
// A line's list is N3_LINE_PARTS sub-lists, chosen by the low bits of the face index: a line through the middle of the object
// takes several hundred appends, and same-address atomics retire one at a time; a workgroup of the walk kernel takes ONE sub-list,
// so the lines with the most crossings (and the longest walks) are also spread over N3_LINE_PARTS workgroups.
constexpr int N3_LINE_PARTS = tune::n3_line_parts;      // power of two
constexpr int N3_LINE_CAP = 1024 / N3_LINE_PARTS;       // crossings a sub-list holds; what it turns away is walked by the per-face kernel
static_assert((N3_LINE_PARTS & (N3_LINE_PARTS - 1)) == 0 && N3_LINE_PARTS <= 16, "JR_TUNE_N3_LINE_PARTS");
struct N3Crossing {
    int packed;                                  // d1_out (13 bits) | direction > 0 (bit 13) | ha, hb (14, 15) | output component of the edge's two vertices (16-19, 20-23)
    int face;                                    // bn * NF + fn
    float cross, ta, tb, ra, r0, r1, r2;         // crossing point, N3K:496-505's slopes, the reference pixel (alpha, r, g, b)
};
static_assert(sizeof(N3Crossing) == 36, "crossing record");

// Round 5: the walk loop was VALU-issue bound (22 VGPRs, 8 wavefronts per SIMD, ~117 issue cycles per 64 visited pixels at the
// measured opcode prices against ~6 LDS cycles).  What it sheds:
//   * the sign of dist = t * (d1 - cross) * 2 / is is the SAME for every pixel of an out-walk (the walk starts beyond the crossing
//     and runs away from it): the +-eps is chosen once per crossing and dist is one FMA (was mul, compare, select, add); a walk
//     that does straddle its crossing point - or touches it exactly - keeps the per-pixel compare (UNI = false);
//   * which of the edge's two vertices take a gradient (ha, hb) is wave-uniform: three instantiations instead of two selects per pixel;
//   * the trip count is wave-uniform (scalar loop, no per-lane compare except in the last partial group of 64), four groups per trip
//     with their LDS reads issued together and one address update per trip;
//   * the two sums of a crossing leave through row_bcast steps and ONE v_readlane each (was four + three adds).
template <bool HA, bool HB, bool UNI>
__device__ inline void n3_line_push(float diff, float dd, float fa, float ea, float fb, float eb, float eps, float& pa, float& pb) {
    // (a real branch around this body: if-converted, the four pixels of a trip were packed ACROSS pixels - 14 register moves per trip -
    // and a group of 64 pixels without a positive diff, e.g. zero image gradients, paid both v_rcp all the same)
    asm volatile("");
    if (HA) {
        float dist;
        if (UNI) dist = __builtin_fmaf(fa, dd, ea);
        else { dist = fa * dd; dist += 0 < dist ? eps : -eps; }
        pa = __builtin_fmaf(-diff, __builtin_amdgcn_rcpf(dist), pa);
    }
    if (HB) {
        float dist;
        if (UNI) dist = __builtin_fmaf(fb, dd, eb);
        else { dist = fb * dd; dist += 0 < dist ? eps : -eps; }
        pb = __builtin_fmaf(-diff, __builtin_amdgcn_rcpf(dist), pb);
    }
}
template <bool HA, bool HB, bool UNI>
__device__ inline void n3_line_walk(const float4* s_g, const float* s_S, int wf, int wt, int lane, const N3Ref& r, float cross,
                                    float fa, float ea, float fb, float eb, float eps, float& pa, float& pb) {
    const int n = wt + 1 - wf, groups = n >> 6, rem = n & 63;          // wave-uniform: scalar loop control
    int d1 = wf + lane, gi = 0;
    for (; gi + 4 <= groups; gi += 4, d1 += 256) {
        float4 g[4];
        float S[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { g[k] = s_g[d1 + 64 * k]; S[k] = s_S[d1 + 64 * k]; }
        const float dd = (float)d1 - cross;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float diff = n3_diff(g[k], S[k], r);
            if (diff <= 0) continue;
            n3_line_push<HA, HB, UNI>(diff, dd + 64.f * k, fa, ea, fb, eb, eps, pa, pb);      // ((float)d1 - cross, one rounding further)
        }
    }
    for (; gi < groups; gi++, d1 += 64) {
        const float diff = n3_diff(s_g[d1], s_S[d1], r);
        if (diff <= 0) continue;
        n3_line_push<HA, HB, UNI>(diff, (float)d1 - cross, fa, ea, fb, eb, eps, pa, pb);
    }
    if (lane < rem) {
        const float diff = n3_diff(s_g[d1], s_S[d1], r);
        if (diff > 0) n3_line_push<HA, HB, UNI>(diff, (float)d1 - cross, fa, ea, fb, eb, eps, pa, pb);
    }
}
__device__ inline float n3_wave_sum_last(float v) {       // all lanes active; -> the sum of the 64 lanes in LANE 63 (other lanes: partial sums)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));   // row_mirror: every lane holds its row's sum
    // rows 1, 3 += lane 15 of the row before; rows 2, 3 += lane 31: lane 63 holds the sum of all 64.  As ONE instruction each - rows outside
    // the row mask keep their value, which the builtin can only express as mov + dpp-mov + add; the two wait states a DPP read needs
    // after the VALU write of its source are spelled out, the compiler's hazard pass does not look into asm.
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
    return v;
}

__global__ __launch_bounds__(256) void k_n3mr_backward_line_walks(
    N3Params p, int nsub, const float4* __restrict__ sg2, const float* __restrict__ gb2, const int* __restrict__ line_count,
    const N3Crossing* __restrict__ line_rec, float* __restrict__ grad_faces) {
    extern __shared__ float4 s_line[];           // [is] (g_alpha, g_r, g_g, g_b), then [is] S
    // the sub-lists of ONE line go to one XCD (workgroup ids are dealt round-robin to the eight): its L2 serves the line once
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3, is = p.IS;
    const int rank = (k / N3_LINE_PARTS) * 8 + xcd;           // of the line in launch order
    if (rank >= nsub / N3_LINE_PARTS) return;
    int d0, axis, bn;
    if (tune::n3_line_fast) {
        // launch order = expected load, heaviest first: the lines through the middle of the image carry the most crossings (an object is
        // usually centred) and the longest lists, and a workgroup that starts late with 128 crossings is the kernel's tail.  Both
        // orientations alternate, positions run from the centre outwards: is/2, is/2 - 1, is/2 + 1, ...
        const int q = rank % (2 * is), j = q >> 1;
        bn = rank / (2 * is); axis = q & 1;
        d0 = (is >> 1) + ((j & 1) ? -((j + 1) >> 1) : (j >> 1));
    } else {
        d0 = rank % is; axis = (rank / is) & 1; bn = rank / (2 * is);
    }
    const int S = ((bn * 2 + axis) * is + d0) * N3_LINE_PARTS + (k % N3_LINE_PARTS);      // sub-list S of that scan line (the producer's index)
    const int n = min(line_count[S], N3_LINE_CAP);
    if (n <= 0) return;
    float* s_gbl = reinterpret_cast<float*>(s_line + is);
    const size_t P = (size_t)p.B * is * is;
    const size_t base = (size_t)(1 - axis) * P + (size_t)bn * is * is + (size_t)d0 * is;   // the orientation in which the walks of this line are contiguous
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float two_over_is = 2.f / is;
    const int* words = reinterpret_cast<const int*>(line_rec + (size_t)S * N3_LINE_CAP);
    constexpr int cstep = 4;
    int w = !tune::n3_line_fast && wid < n && lane < 9 ? words[wid * 9 + lane] : 0;   // (round-4 loop) the record of this wavefront's first crossing, one word per lane
    if (tune::n3_line_fast) {
        // the line's copy with ALL its loads in flight at once, the first record's behind them: a workgroup starts walking one memory
        // round trip after its list length arrived (it was one per 256 pixels plus the record's; a CU holds 8 of these workgroups and
        // runs ~40 of them, their start-up latencies were the gap between the kernel's issue time and its duration)
        float4 v[4];
        float sv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = min((int)threadIdx.x + 256 * k, is - 1);
            v[k] = sg2[base + i]; sv[k] = gb2[base + i];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = (int)threadIdx.x + 256 * k;
            if (i < is) { s_line[i] = v[k]; s_gbl[i] = sv[k]; }
        }
        for (int i = threadIdx.x + 1024; i < is; i += 256) { s_line[i] = sg2[base + i]; s_gbl[i] = gb2[base + i]; }
    } else {
        for (int i = threadIdx.x; i < is; i += 256) { s_line[i] = sg2[base + i]; s_gbl[i] = gb2[base + i]; }
    }
    __syncthreads();
    if (tune::n3_line_fast) {
        // Per CROSSING the round-4 loop spent ~100 VALU instructions around ~85 of walking (SQ_INSTS_VALU 7.3e7 for 4.0e5 crossings of six
        // 64-pixel groups each): nine v_readlane for the record, float compares for the side of the crossing point, two reductions with four
        // v_readlane each, two atomics behind the compiler's uniform-address sequence (mbcnt, bcnt, cvt, mul).  Here
        //   * the record is read with SCALAR loads (its address is wave-uniform; the list was written by the previous kernel), the next
        //     record's before this crossing's walk;
        //   * an out-walk starts at d1_out = floor(cross) + 1 (up) or ceil(cross) - 1 (down) and runs AWAY from the crossing point
        //     (k_n3mr_backward_pixel_map_all, N3K:413-447): d1 - cross has the sign of the direction for every pixel, so the +-eps of a
        //     vertex is the sign test of its slope's bit pattern - scalar;
        //   * the two sums of a crossing end in lane 63 of their reduction (row_bcast steps) and are added to grad_faces FROM that lane:
        //     no v_readlane, and a per-lane offset (+ lane - 63) keeps the compiler's uniform-address atomic sequence away.
        const int wid_u = __builtin_amdgcn_readfirstlane(wid);
        const N3Crossing* recs = line_rec + (size_t)S * N3_LINE_CAP;
        const float eps = p.eps;
        const int lane_m63 = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) - 63;     // (= lane - 63, from a source the compiler does not fold against `lane == 63`)
        N3Crossing rc = recs[wid_u];
        for (int c = wid_u; c < n; c += cstep) {
            const N3Crossing cur = rc;
            rc = recs[min(c + cstep, n - 1)];
            const int packed = cur.packed, d1_out = packed & 0x1fff;
            const bool up = packed & (1 << 13), ha = packed & (1 << 14), hb = packed & (1 << 15);
            const int wf = up ? d1_out : 0, wt = up ? is - 1 : d1_out;        // (0 <= d1_out < is: the producer's `ok`)
            const N3Ref r = {cur.ra, cur.r0, cur.r1, cur.r2};
            // 0 < f * dd  <=>  f and dd have the same sign (f == +-0: dist = 0, the reference subtracts eps)
            const int tab = __builtin_bit_cast(int, cur.ta), tbb = __builtin_bit_cast(int, cur.tb);
            const float ea = (up ? tab > 0 : (tab < 0 && tab != (int)0x80000000)) ? eps : -eps;
            const float eb = (up ? tbb > 0 : (tbb < 0 && tbb != (int)0x80000000)) ? eps : -eps;
            const float fa = cur.ta * two_over_is, fb = cur.tb * two_over_is;
            float pa = 0.f, pb = 0.f;
            if (ha && hb) n3_line_walk<true, true, true>(s_line, s_gbl, wf, wt, lane, r, cur.cross, fa, ea, fb, eb, eps, pa, pb);
            else if (ha) n3_line_walk<true, false, true>(s_line, s_gbl, wf, wt, lane, r, cur.cross, fa, ea, fb, eb, eps, pa, pb);
            else if (hb) n3_line_walk<false, true, true>(s_line, s_gbl, wf, wt, lane, r, cur.cross, fa, ea, fb, eb, eps, pa, pb);
            const float sa = ha ? n3_wave_sum_last(pa) : 0.f, sb = hb ? n3_wave_sum_last(pb) : 0.f;
            if (lane == 63) {
                if (sa != 0.f) atomicAdd(grad_faces + (unsigned)(cur.face * 9 + ((packed >> 16) & 15) + lane_m63), sa);
                if (sb != 0.f) atomicAdd(grad_faces + (unsigned)(cur.face * 9 + ((packed >> 20) & 15) + lane_m63), sb);
            }
        }
        return;
    }
    for (int c = wid; c < n; c += cstep) {
        const int packed = __builtin_amdgcn_readlane(w, 0), face = __builtin_amdgcn_readlane(w, 1);
        const float cross = n3_bcast(__builtin_bit_cast(float, w), 2), ta = n3_bcast(__builtin_bit_cast(float, w), 3),
                    tb = n3_bcast(__builtin_bit_cast(float, w), 4);
        const N3Ref r = {n3_bcast(__builtin_bit_cast(float, w), 5), n3_bcast(__builtin_bit_cast(float, w), 6),
                         n3_bcast(__builtin_bit_cast(float, w), 7), n3_bcast(__builtin_bit_cast(float, w), 8)};
        w = c + cstep < n && lane < 9 ? words[(c + cstep) * 9 + lane] : 0;  // the next record is in flight during the walk
        const int d1_out = packed & 0x1fff;
        const bool up = packed & (1 << 13), ha = packed & (1 << 14), hb = packed & (1 << 15);
        const int ia = (packed >> 16) & 15, ib = (packed >> 20) & 15;
        const int d1_limit = up ? is - 1 : 0;
        const int wf = max(min(d1_out, d1_limit), 0), wt = min(max(d1_out, d1_limit), is - 1);
        // -= diff / (dist +- eps), dist = t * (d1 - cross) * 2 / is (N3K:496-505; the constant factors folded: one rounding apart)
        const float fa = ta * two_over_is, fb = tb * two_over_is, eps = p.eps;
        float pa = 0.f, pb = 0.f;
        for (int d1 = wf + lane; d1 <= wt; d1 += 64) {
            const float diff = n3_diff(s_line[d1], s_gbl[d1], r);
            if (diff <= 0) continue;
            const float dd = (float)d1 - cross;
            if (ha) { float dist = fa * dd; dist += 0 < dist ? eps : -eps; pa -= diff * __builtin_amdgcn_rcpf(dist); }
            if (hb) { float dist = fb * dd; dist += 0 < dist ? eps : -eps; pb -= diff * __builtin_amdgcn_rcpf(dist); }
        }
        const float sa = n3_wave_sum(pa), sb = n3_wave_sum(pb);
        if (lane == 0) {
            if (sa != 0.f) atomicAdd(grad_faces + (size_t)face * 9 + ia, sa);
            if (sb != 0.f) atomicAdd(grad_faces + (size_t)face * 9 + ib, sb);
        }
    }
}

// ---- round 4: the same gradient with ALL SIX (edge, axis) passes of a face in the lanes at once --------------------------
// k_n3mr_backward_pixel_map above runs the six (edge, axis) passes of a face one after the other: each sets up with
// 5 of 64 lanes busy (a face of the 39k-face sphere spans ~5 scan positions per edge), does its short "in" walks, and
// then takes its out-walks ONE scan line at a time - a chain of dependent load round trips (2 - 8 per walk) that only
// other wavefronts can hide: ~7 000 wavefront-instructions per face, VALU 57 % busy, the rest waiting.  Here
//   * lane = (edge, axis, scan position) over all six passes: one set-up, one round of "in" walks (loads issued
//     unconditionally at clamped addresses so that they overlap), ~30 lanes busy instead of 5;
//   * the out-walks of the whole face form ONE list and are taken FOUR at a time: the four lines' loads of a walk step
//     are independent and issued together (4x the memory-level parallelism per wavefront), their parameters are
//     wave-uniform (v_readlane from the scan lane that owns the line);
//   * planes: [0] row-major, [1] column-major copies of (g_alpha, g_r, g_g, g_b | S | face index) in ONE buffer each, so
//     that "the orientation in which this walk is contiguous" is an index offset, not a pointer select.
// Same arithmetic per visited pixel (n3_diff / n3_push); only the order of the float sums differs.

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(JR_TUNE_N3_PIXMAP_WAVES))) void k_n3mr_backward_pixel_map_all(
    N3Params p, const float* __restrict__ faces, const int32_t* __restrict__ face_index_map,
    const float* __restrict__ rgb_map, const float* __restrict__ alpha_map,
    const float4* __restrict__ sg2, const float* __restrict__ gb2, const int32_t* __restrict__ fidx2,
    int* __restrict__ line_count, N3Crossing* __restrict__ line_rec, float* __restrict__ grad_faces) {
    // XCD-aware order (tune::n3_xcd_group = G): workgroup ids go round-robin to the 8 XCDs, so with faces in launch order
    // every XCD walks rows and columns all over the image and its 4 MB L2 keeps missing.  Neighbouring faces of a mesh walk
    // the same rows / columns: runs of G consecutive workgroups (4 G faces) go to ONE XCD.  (One contiguous eighth of the
    // faces per XCD is 1.8x SLOWER: with fill_back half of the face array is back-facing and four XCDs get nothing to do.)
    int wg = (int)blockIdx.x;
    if (tune::n3_xcd_group > 0) {
        constexpr int G = tune::n3_xcd_group > 0 ? tune::n3_xcd_group : 1;
        const int xcd = wg & 7, k = wg >> 3;                 // k-th workgroup of its XCD
        wg = ((k / G) * 8 + xcd) * G + (k % G);              // (beyond the last face: those wavefronts exit)
    }
    const int wave = (wg * (int)blockDim.x + (int)threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= p.B * p.NF) return;
    const int bn = wave / p.NF, fn = wave - bn * p.NF;
    const int is = p.IS;
    const float* face = faces + (size_t)wave * 9;
    if ((face[7] - face[1]) * (face[3] - face[0]) < (face[4] - face[1]) * (face[6] - face[0])) {
        if (lane < 9) grad_faces[(size_t)wave * 9 + lane] = 0.f;      // (back face: no gradient; written here, there is no memset launch in front of this kernel)
        return;
    }
    const size_t P = (size_t)p.B * is * is, mbase = (size_t)bn * is * is;
    const float two_over_is = 2.f / is;
    const bool use_rgb = p.return_rgb, use_a = p.return_alpha;
    float pp_[3][2];                                        // vertices in pixel coordinates (N3K:381-386)
#pragma unroll
    for (int n = 0; n < 3; n++)
#pragma unroll
        for (int d = 0; d < 2; d++) pp_[n][d] = 0.5f * (face[3 * n + d] * is + is - 1);
    // scan ranges of the six passes, c = 2 * edge + axis (N3K:413-414)
    int first[6], off[7];
    off[0] = 0;
#pragma unroll
    for (int c = 0; c < 6; c++) {
        const int e = c >> 1, ax = c & 1, e1 = (e + 1) % 3;
        const float a = pp_[e][ax], b = pp_[e1][ax];
        const int d0_from = (int)fmax((double)ceilf(fminf(a, b)), 0.);
        const int d0_to = (int)fmin((double)fmaxf(a, b), is - 1.);
        first[c] = d0_from;
        off[c + 1] = off[c] + max(d0_to - d0_from + 1, 0);
    }
    const int total = off[6];
    float g[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto add_uniform = [&](int k, float v) {                // k is wave-uniform
        switch (k) {
            case 0: g[0] += v; break; case 1: g[1] += v; break; case 2: g[2] += v; break;
            case 3: g[3] += v; break; case 4: g[4] += v; break; case 5: g[5] += v; break;
            case 6: g[6] += v; break; case 7: g[7] += v; break; default: g[8] += v; break;
        }
    };
    for (int t0 = 0; t0 < total; t0 += 64) {
        // ---- lane = (pass, scan position) ----
        const int t = t0 + lane;
        const bool have = t < total;
        int c = 0;
#pragma unroll
        for (int k = 1; k < 6; k++) c += t >= off[k] ? 1 : 0;
        if (!have) c = 0;
        int cfirst = first[0], coff = off[0];
#pragma unroll
        for (int k = 1; k < 6; k++) { cfirst = c == k ? first[k] : cfirst; coff = c == k ? off[k] : coff; }
        const int d0 = have ? cfirst + (t - coff) : 0;
        const int edge = c >> 1, axis = c & 1;
        const int pi0 = edge, pi1 = edge == 2 ? 0 : edge + 1, pi2 = edge == 0 ? 2 : edge - 1;
        auto coord = [&](int v, int d) {                    // pp_[v][d] for per-lane v, d
            const float x = v == 0 ? pp_[0][0] : (v == 1 ? pp_[1][0] : pp_[2][0]);
            const float y = v == 0 ? pp_[0][1] : (v == 1 ? pp_[1][1] : pp_[2][1]);
            return d == 0 ? x : y;
        };
        // the edge's two vertices and the opposite one, walking axis second (N3K:398-405)
        const float q00 = coord(pi0, axis), q01 = coord(pi0, 1 - axis), q10 = coord(pi1, axis), q11 = coord(pi1, 1 - axis),
                    q20 = coord(pi2, axis), q21 = coord(pi2, 1 - axis);
        const int direction = axis == 0 ? (q00 < q10 ? -1 : 1) : (q00 < q10 ? 1 : -1);         // N3K:407-411
        const float e10 = q10 - q00;
        const float slope = (q11 - q01) / e10;
        const float s20 = (q21 - q01) / (q20 - q00), s12 = (q11 - q21) / (q10 - q20);
        const float d1_cross = slope * (d0 - q00) + q01;
        const int d1_in = 0 < direction ? (int)floorf(d1_cross) : (int)ceilf(d1_cross);
        const int d1_out = d1_in + direction;
        const bool ok = have && !(d1_in < 0 || is <= d1_in) && !(d1_out < 0 || is <= d1_out);
        const bool ha = q10 != d0, hb = q00 != d0;
        const float ta = e10 / (q10 - d0), tb = e10 / (d0 - q00);
        N3Ref rin = {0.f, 0.f, 0.f, 0.f}, rout = {0.f, 0.f, 0.f, 0.f};
        int fin = -1, from = 1, to = 0;
        if (ok) {
            const size_t idx_in = mbase + (axis == 0 ? (size_t)d1_in * is + d0 : (size_t)d0 * is + d1_in);
            const size_t idx_out = mbase + (axis == 0 ? (size_t)d1_out * is + d0 : (size_t)d0 * is + d1_out);
            fin = face_index_map[idx_in];
            if (use_a) { rin.a = alpha_map[idx_in]; rout.a = alpha_map[idx_out]; }
            if (use_rgb) {
                rin.c0 = rgb_map[idx_in * 3]; rin.c1 = rgb_map[idx_in * 3 + 1]; rin.c2 = rgb_map[idx_in * 3 + 2];
                rout.c0 = rgb_map[idx_out * 3]; rout.c1 = rgb_map[idx_out * 3 + 1]; rout.c2 = rgb_map[idx_out * 3 + 2];
            }
            const float cross2 = ((d0 - q00) * (d0 - q20) < 0) ? s20 * (d0 - q00) + q01 : s12 * (d0 - q20) + q21;
            const int d1_limit = 0 < direction ? (int)ceilf(cross2) : (int)floorf(cross2);    // N3K:520-528
            from = max(min(d1_in, d1_limit), 0);
            to = min(max(d1_in, d1_limit), is - 1);
        }
        float acc_a = 0.f, acc_b = 0.f;
        const int ia = pi0 * 3 + (1 - axis), ib = pi1 * 3 + (1 - axis);     // the two vertices of this lane's edge, coordinate 1 - axis (N3K:496-505)
        // ---- "in" walks: every lane walks its own scan line, in the orientation in which neighbouring scan positions
        //      are contiguous (plane `axis`) ----
        const size_t in_base = (size_t)axis * P + mbase + (size_t)d0;
        for (int k = 0; !(JR_TUNE_DIAG & 4096) && ballot(from + k <= to) != 0ull; k++) {      // (diagnostic bit 12: no "in" walks)
            const int d1 = from + k;
            const bool act = d1 <= to;
            const size_t m = in_base + (size_t)(act ? d1 : from) * is;      // (from is 1 for idle lanes: a valid element)
            const int fm = fidx2[m];
            const float4 v = sg2[m];
            const float gv = gb2[m];
            if (!act || fm != fn) continue;
            const float diff = n3_diff(v, gv, rout);
            if (diff <= 0) continue;
            n3_push(diff, d1, d1_cross, ta, tb, ha, hb, two_over_is, p.eps, acc_a, acc_b);
        }
        // ---- the walks that stay here: the lines of all six passes, four at a time, the pixel walk spread over the lanes ----
        // ---- "out" walks (N3K:470-507).  tune::n3_line_walks: they are NOT run here.  Every walk is handed to the scan line it runs
        //      along (a crossing record appended to the line's list); k_n3mr_backward_line_walks stages each line of the packed
        //      planes in LDS ONCE and runs all its crossings from there.  A line whose list is full keeps its walk here.
        bool walk_here = ok && fin == fn;
        if (line_rec && walk_here) {
            const int L = ((bn * 2 + axis) * is + d0) * N3_LINE_PARTS + (wave & (N3_LINE_PARTS - 1));
            const int pos = atomicAdd(&line_count[L], 1);
            if (pos < N3_LINE_CAP) {
                N3Crossing c;
                c.packed = d1_out | (0 < direction ? 1 << 13 : 0) | (ha ? 1 << 14 : 0) | (hb ? 1 << 15 : 0) | (ia << 16) | (ib << 20);
                c.face = wave; c.cross = d1_cross; c.ta = ta; c.tb = tb;
                c.ra = rin.a; c.r0 = rin.c0; c.r1 = rin.c1; c.r2 = rin.c2;
                line_rec[(size_t)L * N3_LINE_CAP + pos] = c;
                walk_here = false;
            }
        }
        unsigned long long vis = (JR_TUNE_DIAG & 2048) ? 0ull : ballot(walk_here);                   // (diagnostic bit 11: no "out" walks)
        const unsigned long long ha_m = ballot(ha), hb_m = ballot(hb);
        while (vis) {
            // out-walks in flight: with the per-line regrouping this loop only takes what a full line list turned away, and one
            // at a time keeps the registers (68 instead of 97: 7 wavefronts per SIMD instead of 4) for the common path
            constexpr int NL = tune::n3_line_walks ? 1 : tune::n3_walks;
            int s[NL], len[NL], wf[NL], ia_u[NL], ib_u[NL];
            size_t base[NL];
            float cross[NL], bta[NL], btb[NL];
            N3Ref r[NL];
            bool bha[NL], bhb[NL];
            int maxlen = -1;
#pragma unroll
            for (int u = 0; u < NL; u++) {
                const bool valid = vis != 0ull;
                s[u] = valid ? __builtin_ctzll(vis) : s[0];
                vis &= vis - 1;                              // (0 stays 0)
                const int bd0 = __builtin_amdgcn_readlane(d0, s[u]), bax = __builtin_amdgcn_readlane(axis, s[u]);
                const int bdir = __builtin_amdgcn_readlane(direction, s[u]), b_out = __builtin_amdgcn_readlane(d1_out, s[u]);
                ia_u[u] = __builtin_amdgcn_readlane(ia, s[u]); ib_u[u] = __builtin_amdgcn_readlane(ib, s[u]);
                cross[u] = n3_bcast(d1_cross, s[u]); bta[u] = n3_bcast(ta, s[u]); btb[u] = n3_bcast(tb, s[u]);
                bha[u] = (ha_m >> s[u]) & 1ull; bhb[u] = (hb_m >> s[u]) & 1ull;
                r[u] = {n3_bcast(rin.a, s[u]), n3_bcast(rin.c0, s[u]), n3_bcast(rin.c1, s[u]), n3_bcast(rin.c2, s[u])};
                const int d1_limit = 0 < bdir ? is - 1 : 0;
                wf[u] = max(min(b_out, d1_limit), 0);
                const int wt = min(max(b_out, d1_limit), is - 1);
                len[u] = valid ? wt - wf[u] : -1;
                maxlen = max(maxlen, len[u]);
                base[u] = (size_t)(1 - bax) * P + mbase + (size_t)bd0 * is + wf[u];   // the orientation in which this walk is contiguous
            }
            float pa[NL], pb[NL];
#pragma unroll
            for (int u = 0; u < NL; u++) { pa[u] = 0.f; pb[u] = 0.f; }
            for (int o = lane; o <= maxlen; o += 64) {
                float4 v[NL];
                float gv[NL];
#pragma unroll
                for (int u = 0; u < NL; u++) {                // eight independent loads in flight
                    const size_t m = base[u] + (size_t)min(o, max(len[u], 0));
                    v[u] = sg2[m];
                    gv[u] = gb2[m];
                }
#pragma unroll
                for (int u = 0; u < NL; u++) {
                    if (o > len[u]) continue;
                    const float diff = n3_diff(v[u], gv[u], r[u]);
                    if (diff <= 0) continue;
                    n3_push(diff, wf[u] + o, cross[u], bta[u], btb[u], bha[u], bhb[u], two_over_is, p.eps, pa[u], pb[u]);
                }
            }
            // the walk's per-lane partial sums go straight into the face's nine per-lane accumulators: which two of them is
            // wave-uniform (the pass of the scan lane that owns the line), so this is a scalar branch, not a reduction
#pragma unroll
            for (int u = 0; u < NL; u++) {
                if (len[u] < 0) continue;                    // uniform
                add_uniform(ia_u[u], pa[u]);
                add_uniform(ib_u[u], pb[u]);
            }
        }
        // ---- this lane's own ("in" walk) sums ----
#pragma unroll
        for (int k = 0; k < 9; k++) g[k] += (have && ia == k ? acc_a : 0.f) + (have && ib == k ? acc_b : 0.f);
    }
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const float v = n3_wave_sum(g[k]);
        if (lane == 0) grad_faces[(size_t)wave * 9 + k] = v;
    }
}

__global__ __launch_bounds__(256) void k_n3mr_backward_textures(
    N3Params p, const int32_t* __restrict__ face_index_map, const float* __restrict__ sampling_weight_map,
    const int32_t* __restrict__ sampling_index_map, const float* __restrict__ grad_rgb_map,
    float* __restrict__ grad_textures) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long pp = (long)p.IS * p.IS;
    if (i >= p.B * pp) return;
    const int fn = face_index_map[i];
    if (fn < 0) return;
    const int bn = (int)(i / pp), ts = p.TS;
    float* gt = grad_textures + ((size_t)bn * p.NF + fn) * ts * ts * ts * 3;
    const float g0 = grad_rgb_map[3 * i], g1 = grad_rgb_map[3 * i + 1], g2 = grad_rgb_map[3 * i + 2];
#pragma unroll
    for (int pn = 0; pn < 8; pn++) {                                                   // N3K:685-692
        const float w = sampling_weight_map[8 * i + pn];
        float* t = gt + (size_t)sampling_index_map[8 * i + pn] * 3;
        atomicAdd(t, w * g0); atomicAdd(t + 1, w * g1); atomicAdd(t + 2, w * g2);
    }
}

__global__ __launch_bounds__(256) void k_n3mr_backward_depth(
    N3Params p, const float* __restrict__ faces, const float* __restrict__ depth_map,
    const int32_t* __restrict__ face_index_map, const float* __restrict__ face_inv_map,
    const float* __restrict__ weight_map, const float* __restrict__ grad_depth_map,
    float* __restrict__ grad_faces) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long pp = (long)p.IS * p.IS;
    if (i >= p.B * pp) return;
    const int fn = face_index_map[i];
    if (fn < 0) return;
    const int bn = (int)(i / pp), is = p.IS;
    const float* face = faces + ((size_t)bn * p.NF + fn) * 9;
    const float depth = depth_map[i], depth2 = depth * depth, gd = grad_depth_map[i];
    const float* finv = face_inv_map + 9 * i;
    const float* w = weight_map + 3 * i;
    float* gf = grad_faces + ((size_t)bn * p.NF + fn) * 9;
    for (int k = 0; k < 3; k++) {                                                      // N3K:768-771
        const float zk = face[3 * k + 2];
        atomicAdd(&gf[3 * k + 2], gd * w[k] * depth2 / (zk * zk));
    }
    float tmp[3] = {0.f, 0.f, 0.f};                                                    // N3K:773-779
    for (int k = 0; k < 3; k++)
        for (int l = 0; l < 3; l++) tmp[k] += -finv[3 * l + k] / face[3 * l + 2];
    for (int k = 0; k < 3; k++)
        for (int l = 0; l < 2; l++) atomicAdd(&gf[3 * k + l], -gd * tmp[l] * w[k] * depth2 * is / 2);
}

// backward_textures + backward_depth_map with ONE WAVEFRONT PER FACE instead of one thread per pixel with
// float atomics on the face's 9 + 3*TS^3 accumulators (every pixel of a face hits the same addresses: the
// per-pixel kernels above spend their time serialising in L2).  The wavefront walks the face's bounding box
// (the same box the z-buffer pass rasterised, so it contains every pixel the face owns), keeps the depth
// gradient in registers and the texel gradients in LDS, and writes each face's results once, without
// atomics and without a prior memset of grad_textures.  grad_faces already holds the pixel-map part
// (k_n3mr_backward_pixel_map ran before on the same stream) and is updated in place by its only owner.
constexpr int N3_TEX_LDS = 1536;            // floats of texel gradient per wavefront (TS <= 8)
// Sum of 16 per-lane values over the 16 lanes of a DPP row, "transposed": lane i of the row ends up with the row total of v[i].
// Butterfly with halving payload (8 + 4 + 2 + 1 exchanges): partners row_mirror, row_half_mirror, quad_perm [3,2,1,0], [1,0,3,2]
// (the scheme of softras_backward.hip: row_transpose_reduce).
template <int CTRL>
__device__ inline float n3_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ inline float n3_row_transpose_reduce(const float (&v)[16], int li) {
    float a[8], b[4], c[2];
    const bool h8 = li & 8, h4 = li & 4, h2 = li & 2, h1 = li & 1;
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = (h8 ? v[j + 8] : v[j]) + n3_dpp<0x140>(h8 ? v[j] : v[j + 8]);
#pragma unroll
    for (int j = 0; j < 4; j++) b[j] = (h4 ? a[j + 4] : a[j]) + n3_dpp<0x141>(h4 ? a[j] : a[j + 4]);
#pragma unroll
    for (int j = 0; j < 2; j++) c[j] = (h2 ? b[j + 2] : b[j]) + n3_dpp<0x1B>(h2 ? b[j] : b[j + 2]);
    return (h1 ? c[1] : c[0]) + n3_dpp<0xB1>(h1 ? c[0] : c[1]);
}

__global__ __launch_bounds__(256) void k_n3mr_backward_face(
    N3Params p, const float* __restrict__ faces, const int32_t* __restrict__ face_index_map,
    const float* __restrict__ depth_map, const float* __restrict__ face_inv_map,
    const float* __restrict__ weight_map, const float* __restrict__ sampling_weight_map,
    const int32_t* __restrict__ sampling_index_map, const float* __restrict__ grad_rgb_map,
    const float* __restrict__ grad_depth_map, float* __restrict__ grad_faces, float* __restrict__ grad_textures) {
    __shared__ float s_tex[4][N3_TEX_LDS];
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63, wl = threadIdx.x >> 6;
    if (wave >= p.B * p.NF) return;
    const int bn = wave / p.NF, fn = wave - bn * p.NF;
    const int is = p.IS, ts = p.TS, ntex = p.return_rgb ? ts * ts * ts * 3 : 0;
    float* acc = s_tex[wl];
    for (int k = lane; k < ntex; k += 64) acc[k] = 0.f;
    float treg[24];
#pragma unroll
    for (int k = 0; k < 24; k++) treg[k] = 0.f;
    const float* f = faces + (size_t)wave * 9;
    float g[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool mine = false;                                   // this lane found a pixel the face owns
    const bool front = !((f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0]));   // N3K:63
    if (front) {
        float x_min = is, y_min = is, x_max = 0, y_max = 0;                               // N3K:89-99
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float px = 0.5f * (f[3 * k] * is + is - 1), py = 0.5f * (f[3 * k + 1] * is + is - 1);
            if (px < x_min) x_min = px;
            if (px > x_max) x_max = px;
            if (py < y_min) y_min = py;
            if (py > y_max) y_max = py;
        }
        const int ix0 = max(0, (int)x_min), ix1 = min(is - 1, (int)x_max);
        const int iy0 = max(0, (int)y_min), iy1 = min(is - 1, (int)y_max);
        const int wid = ix1 - ix0 + 1;
        const long npix = (ix1 < ix0 || iy1 < iy0) ? 0 : (long)wid * (iy1 - iy0 + 1);
        const size_t mbase = (size_t)bn * is * is;
        // lanes along rows.  Boxes up to 64 pixels wide (all but screen-filling faces): lane -> (row q, column r) of a block of 64 / wid
        // rows through one reciprocal ((lane + 0.5) / wid is at least 0.5 / 64 away from an integer: exact) instead of a 64-bit division
        // and remainder per pixel (round 5: ~80 of this kernel's ~280 VALU instructions per face)
        const bool small = tune::n3_face_fast && wid >= 1 && wid <= 64;
        const float rw = __builtin_amdgcn_rcpf((float)max(wid, 1));
        const int q = (int)(((float)lane + 0.5f) * rw), r = lane - q * wid;
        const int rpi = small ? __builtin_amdgcn_readfirstlane((int)(64.5f * rw)) : 1;       // rows per trip
        const long trips = npix <= 0 ? 0 : (small ? (long)(iy1 - iy0 + rpi) / rpi : (npix + 63) / 64);
        for (long t = 0; t < trips; t++) {
            int yi, xi;
            if (small) {
                yi = iy0 + (int)t * rpi + q; xi = ix0 + r;
                if (q >= rpi || yi > iy1) continue;
            } else {
                const long idx = t * 64 + lane;
                if (idx >= npix) continue;
                yi = iy0 + (int)(idx / wid); xi = ix0 + (int)(idx % wid);
            }
            const size_t i = mbase + (size_t)yi * is + xi;
            if (face_index_map[i] != fn) continue;
            mine = true;
            if (p.return_depth) {                                                          // N3K:768-779
                const float depth = depth_map[i], depth2 = depth * depth, gd = grad_depth_map[i];
                const float* finv = face_inv_map + 9 * i;
                const float* w = weight_map + 3 * i;
                for (int k = 0; k < 3; k++) {
                    const float zk = f[3 * k + 2];
                    g[3 * k + 2] += gd * w[k] * depth2 / (zk * zk);
                }
                float tmp[3] = {0.f, 0.f, 0.f};
                for (int k = 0; k < 3; k++)
                    for (int l = 0; l < 3; l++) tmp[k] += -finv[3 * l + k] / f[3 * l + 2];
                for (int k = 0; k < 3; k++)
                    for (int l = 0; l < 2; l++) g[3 * k + l] += -gd * tmp[l] * w[k] * depth2 * is / 2;
            }
            if (p.return_rgb) {                                                            // N3K:685-692
                const float g0 = grad_rgb_map[3 * i], g1 = grad_rgb_map[3 * i + 1], g2 = grad_rgb_map[3 * i + 2];
                const float4* wq = reinterpret_cast<const float4*>(sampling_weight_map + 8 * i);
                const int4* iq = reinterpret_cast<const int4*>(sampling_index_map + 8 * i);
                const float4 w0 = wq[0], w1 = wq[1];
                const int4 i0 = iq[0], i1 = iq[1];
                const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                const int ix[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
                // texture_size 2: every pixel samples the cube's eight corners, corner pn -> texel bitreverse3(pn) (N3K:275-284
                // with (int)t = 0).  ds_add_f32 retires about one lane per clock, so 24 LDS atomics per pixel were this
                // kernel's time; the corners are summed in 24 registers per lane instead and reduced once per face.  Any other
                // index (larger cubes; a coordinate clamped to exactly ts - 1 when eps = 0) takes the LDS atomics.
                bool corner = ts == 2;
#pragma unroll
                for (int pn = 0; pn < 8; pn++) corner = corner && ix[pn] == (((pn & 1) << 2) | (pn & 2) | ((pn >> 2) & 1));
                if (corner) {
#pragma unroll
                    for (int pn = 0; pn < 8; pn++) {
                        const int tx = (((pn & 1) << 2) | (pn & 2) | ((pn >> 2) & 1)) * 3;
                        treg[tx] += w[pn] * g0; treg[tx + 1] += w[pn] * g1; treg[tx + 2] += w[pn] * g2;
                    }
                } else {
#pragma unroll
                    for (int pn = 0; pn < 8; pn++) {
                        float* t = acc + ix[pn] * 3;
                        atomicAdd(t, w[pn] * g0); atomicAdd(t + 1, w[pn] * g1); atomicAdd(t + 2, w[pn] * g2);
                    }
                }
            }
        }
        if (!tune::n3_face_fast && p.return_rgb && ts == 2 && ballot(mine) != 0ull) {   // (uniform) the register sums join the LDS accumulators
            __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
            for (int k = 0; k < 24; k++) {
                const float v = n3_wave_sum(treg[k]);
                if (lane == 0) acc[k] += v;
            }
        }
    }
    // three faces out of four own no pixel here (back faces, hidden front faces): they skip the reductions (round 4)
    const bool some = ballot(mine) != 0ull;
    if (tune::n3_face_fast) {
        // Round 5: the 24 texel sums and the 9 vertex components were 33 separate wave reductions (~13 VALU instructions each: 430 per face
        // that owns a pixel).  Two "transposing" row reductions instead - lane i of every 16-lane row ends up with the row total of value i
        // (15 exchanges for 16 values) - then two cross-row steps: ~140 instructions for 32 values; the 33rd keeps its own reduction.
        const bool tex2 = p.return_rgb && ts == 2;
        if (some && (tex2 || p.return_depth)) {
            float va[16], vb[16];
#pragma unroll
            for (int k = 0; k < 16; k++) va[k] = treg[k];
#pragma unroll
            for (int k = 0; k < 8; k++) { vb[k] = treg[16 + k]; vb[8 + k] = g[k]; }
            const int li = lane & 15;
            float ra = n3_row_transpose_reduce(va, li), rb = n3_row_transpose_reduce(vb, li);
            ra += __shfl_xor(ra, 16); rb += __shfl_xor(rb, 16);
            ra += __shfl_xor(ra, 32); rb += __shfl_xor(rb, 32);
            if (tex2) {
                __builtin_amdgcn_s_waitcnt(0);
                if (lane < 16) acc[lane] += ra;
                if (lane < 8) acc[16 + lane] += rb;
            }
            if (p.return_depth) {
                if (lane >= 8 && lane < 16 && rb != 0.f) grad_faces[(size_t)wave * 9 + (lane - 8)] += rb;
                const float v8 = n3_wave_sum(g[8]);
                if (lane == 0 && v8 != 0.f) grad_faces[(size_t)wave * 9 + 8] += v8;
            }
        }
    } else if (p.return_depth && some) {
#pragma unroll
        for (int k = 0; k < 9; k++) {
            float v = g[k];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0 && v != 0.f) grad_faces[(size_t)wave * 9 + k] += v;
        }
    }
    if (ntex) {
        __builtin_amdgcn_s_waitcnt(0);      // this wavefront's LDS atomics have landed (single-wave ownership)
        float* gt = grad_textures + (size_t)wave * ntex;
        for (int k = lane; k < ntex; k += 64) gt[k] = acc[k];
    }
}

static N3Params make_n3(int B, int NF, int TS, int IS, float near_, float far_, float eps, const float* bg,
                        int rrgb, int ralpha, int rdepth) {
    N3Params p;
    p.B = B; p.NF = NF; p.TS = TS; p.IS = IS; p.near_ = near_; p.far_ = far_; p.eps = eps;
    p.return_rgb = rrgb; p.return_alpha = ralpha; p.return_depth = rdepth;
    for (int k = 0; k < 3; k++) p.bg[k] = bg ? bg[k] : 0.f;
    return p;
}

void launch_n3mr_forward(hipStream_t st, const float* faces, const float* textures, float* faces_inv,
                         unsigned long long* zkey, int32_t* face_index_map, float* weight_map, float* depth_map,
                         float* face_inv_map, float* rgb_map, float* alpha_map, int32_t* sampling_index_map,
                         float* sampling_weight_map, int B, int NF, int TS, int IS, float near_, float far_,
                         float eps, const float* bg, int rrgb, int ralpha, int rdepth, bool zkey_clean) {
    const N3Params p = make_n3(B, NF, TS, IS, near_, far_, eps, bg, rrgb, ralpha, rdepth);
    const long P = (long)B * IS * IS;
    if (!zkey_clean) (void)hipMemsetAsync(zkey, 0xff, sizeof(unsigned long long) * P, st);     // (first use / after a regrow: k_n3mr_resolve clears what it read)
    const long waves = (long)B * NF;
    if (tune::n3_zbuf_group > 0) {
        constexpr int G = tune::n3_zbuf_group > 0 ? tune::n3_zbuf_group : 1;
        const long groups = (waves + G - 1) / G;
        k_n3mr_zbuffer_grouped<G><<<(unsigned)((groups * 64 + 255) / 256), 256, 0, st>>>(p, faces, faces_inv, zkey);
    } else
        k_n3mr_zbuffer<<<(unsigned)((waves * 64 + 255) / 256), 256, 0, st>>>(p, faces, faces_inv, zkey);
    k_n3mr_resolve<<<(unsigned)((P + 255) / 256), 256, 0, st>>>(p, faces, textures, faces_inv, zkey, face_index_map,
                                                               weight_map, depth_map, face_inv_map, rgb_map, alpha_map,
                                                               sampling_index_map, sampling_weight_map);
}

// scan lines per launch of the line-walk kernel (two orientations per image); the LDS copy of a line is 20 B per pixel
// The crossing lists are sized for the worst case - 1 024 records of 36 B per scan line and orientation: 72 KB per image row,
// i.e. 19 MB for one 256^2 image, 75 MB for one at 1024^2, 600 MB for a batch of eight of those - and stay in the context's
// scratch until jr_ctx_trim (the per-face walks needed 44 B per pixel).  Above N3_LINE_LIST_BUDGET the backward keeps the
// per-face walks instead of growing the arena silently: the regrouping is a latency / L2 optimisation of launches
// that size, a batch of 64 images has enough faces in flight without it.
constexpr size_t N3_LINE_LIST_BUDGET = (size_t)1 << 30;
static size_t n3_line_list_bytes(int B, int IS) {
    return (size_t)B * 2 * IS * N3_LINE_PARTS * (sizeof(int) + sizeof(N3Crossing) * N3_LINE_CAP) + 64;
}
static bool n3_use_line_walks(int B, int IS) {
    return tune::n3_line_walks && (size_t)IS * 20 <= 65536 && IS < (1 << 13) && n3_line_list_bytes(B, IS) <= N3_LINE_LIST_BUDGET;
}
size_t n3mr_backward_scratch_bytes(int B, int IS) {
    size_t bytes = (size_t)B * IS * IS * (2 * 16 + 2 * 4 + 2 * 4);
    if (n3_use_line_walks(B, IS)) bytes += n3_line_list_bytes(B, IS);
    return bytes;
}

void launch_n3mr_backward(hipStream_t st, const float* faces, const int32_t* face_index_map,
                          const float* weight_map, const float* depth_map, const float* face_inv_map,
                          const float* rgb_map, const float* alpha_map, const float* sampling_weight_map,
                          const int32_t* sampling_index_map, const float* grad_rgb_map, const float* grad_alpha_map,
                          const float* grad_depth_map, float* grad_faces, float* grad_textures, void* scratch,
                          int B, int NF, int TS, int IS, float eps, int rrgb, int ralpha, int rdepth) {
    const N3Params p = make_n3(B, NF, TS, IS, 0.f, 0.f, eps, nullptr, rrgb, ralpha, rdepth);
    const long P = (long)B * IS * IS, waves = (long)B * NF;
    // k_n3mr_backward_pixel_map_all writes every face's nine components itself (zeros for back faces)
    if (!((rrgb || ralpha) && tune::n3_pixmap_all)) (void)hipMemsetAsync(grad_faces, 0, sizeof(float) * (size_t)B * NF * 9, st);
    if (rrgb || ralpha) {
        // scratch (n3mr_backward_scratch_bytes): [sg | sg_t] float4, [gb | gb_t] float, [fidx | fidx_t] int32 - row-major | column-major
        float4* sg = static_cast<float4*>(scratch);
        float4* sg_t = sg + P;
        float* gb = reinterpret_cast<float*>(sg_t + P);
        float* gb_t = gb + P;
        int32_t* fidx_r = reinterpret_cast<int32_t*>(gb_t + P);
        int32_t* fidx_t = fidx_r + P;
        const int ptiles = (IS + N3_PACK_TILE - 1) / N3_PACK_TILE;
        const bool line_walks = tune::n3_pixmap_all && n3_use_line_walks(B, IS);
        const int nlines = B * 2 * IS * N3_LINE_PARTS;            // sub-lists of the scan lines
        int* line_count = reinterpret_cast<int*>(fidx_t + P);
        N3Crossing* line_rec = reinterpret_cast<N3Crossing*>(line_count + ((nlines + 15) & ~15));     // (N3Crossing is 4-byte aligned)
        k_n3mr_pack<<<(unsigned)(B * ptiles * ptiles), 256, 0, st>>>(p, face_index_map, rgb_map, alpha_map, grad_rgb_map,
                                                                    grad_alpha_map, sg, gb, sg_t, gb_t, fidx_r, fidx_t,
                                                                    line_count, line_walks ? nlines : 0);
        if (tune::n3_pixmap_all) {
            constexpr long GG = 8 * (tune::n3_xcd_group > 0 ? tune::n3_xcd_group : 1);        // whole runs for every XCD
            k_n3mr_backward_pixel_map_all<<<(unsigned)(((waves * 64 + 255) / 256 + GG - 1) / GG * GG), 256, 0, st>>>(
                p, faces, face_index_map, rgb_map, alpha_map, sg, gb, fidx_r, line_walks ? line_count : nullptr, line_walks ? line_rec : nullptr, grad_faces);
            if (line_walks)
                k_n3mr_backward_line_walks<<<(unsigned)((nlines + 8 * N3_LINE_PARTS - 1) / (8 * N3_LINE_PARTS) * (8 * N3_LINE_PARTS)), 256, (size_t)IS * 20, st>>>(
                    p, nlines, sg, gb, line_count, line_rec, grad_faces);
        } else {
            const N3Planes rowmajor = {sg, gb, face_index_map}, colmajor = {sg_t, gb_t, fidx_t};
            k_n3mr_backward_pixel_map<<<(unsigned)((waves * 64 + 255) / 256), 256, 0, st>>>(
                p, faces, face_index_map, rgb_map, alpha_map, rowmajor, colmajor, grad_faces);
        }
    }
    if (!rrgb && !rdepth) return;
    if (!rrgb || (size_t)TS * TS * TS * 3 <= (size_t)N3_TEX_LDS) {
        k_n3mr_backward_face<<<(unsigned)((waves * 64 + 255) / 256), 256, 0, st>>>(
            p, faces, face_index_map, depth_map, face_inv_map, weight_map, sampling_weight_map, sampling_index_map,
            grad_rgb_map, grad_depth_map, grad_faces, grad_textures);
        return;
    }
    // texture cubes too large for the per-wavefront LDS accumulators: per-pixel kernels with global atomics
    (void)hipMemsetAsync(grad_textures, 0, sizeof(float) * (size_t)B * NF * TS * TS * TS * 3, st);
    k_n3mr_backward_textures<<<(unsigned)((P + 255) / 256), 256, 0, st>>>(
        p, face_index_map, sampling_weight_map, sampling_index_map, grad_rgb_map, grad_textures);
    if (rdepth)
        k_n3mr_backward_depth<<<(unsigned)((P + 255) / 256), 256, 0, st>>>(
            p, faces, depth_map, face_index_map, face_inv_map, weight_map, grad_depth_map, grad_faces);
}

}  // namespace jr
