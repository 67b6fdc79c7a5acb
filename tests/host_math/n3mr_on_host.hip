// TEST INFRASTRUCTURE ONLY - tests/test_device_math_on_host.py.
//
// jrender_amd/csrc/n3mr_kernels.hip compiled with every __device__ function ALSO built for the host (the macro below; nothing of the
// file is restated): n3_face_inv and n3_pixel - the face set-up and the per-(face, pixel) coverage test, clamped weights and depth of the
// NMR forward (N3K:63-134) - run here on the CPU.  The loop around them is this file's own and follows k_n3mr_zbuffer line by line in
// serial form: back-side test, pixel-space box, depth range, and the depth test as the minimum of the packed key (depth bits << 32 |
// face index) - lowest face index on equal depths, which is what a serial run of the reference gives.  Compared with the reference's own
// kernels compiled for the host (oracle/_ref: N3mrOracle): faces_inv, the face-index map, the depth map and the winner's weights, bit for bit.
#include <hip/hip_runtime.h>
#include <math.h>
#undef __device__
#define __device__ __attribute__((host)) __attribute__((device))
#include "../../jrender_amd/csrc/n3mr_kernels.hip"

#include <stdint.h>
#include <string.h>

extern "C" {

// faces [NF,9] -> faces_inv [NF,9], face_index_map [IS,IS] (-1 = none), depth_map [IS,IS] (far where none, N3K:57), weight_map [IS,IS,3] (0 where none)
int hm_n3mr_zbuffer(const float* faces, int NF, int IS, float near_, float far_, float* faces_inv, int32_t* face_index_map,
                    float* depth_map, float* weight_map) {
    using namespace jr;
    const size_t pp = (size_t)IS * IS;
    unsigned long long* zkey = new unsigned long long[pp];
    for (size_t i = 0; i < pp; i++) zkey[i] = ~0ull;
    memset(weight_map, 0, sizeof(float) * pp * 3);
    for (int fn = 0; fn < NF; fn++) {
        const float* f = faces + (size_t)fn * 9;
        float* finv = faces_inv + (size_t)fn * 9;
        if ((f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0])) { for (int k = 0; k < 9; k++) finv[k] = 0.f; continue; }   // back side, N3K:63
        float px[3], py[3], inv[9];
        n3_face_inv(f, IS, px, py, inv);
        for (int k = 0; k < 9; k++) finv[k] = inv[k];
        float x_min = IS, y_min = IS, x_max = 0, y_max = 0;                                // N3K:89-99
        for (int k = 0; k < 3; k++) {
            if (px[k] < x_min) x_min = px[k];
            if (px[k] > x_max) x_max = px[k];
            if (py[k] < y_min) y_min = py[k];
            if (py[k] > y_max) y_max = py[k];
        }
        const int ix0 = (int)x_min > 0 ? (int)x_min : 0, ix1 = (int)x_max < IS - 1 ? (int)x_max : IS - 1;
        const int iy0 = (int)y_min > 0 ? (int)y_min : 0, iy1 = (int)y_max < IS - 1 ? (int)y_max : IS - 1;
        for (int xi = ix0; xi <= ix1; xi++)
            for (int yi = iy0; yi <= iy1; yi++) {
                float w[3], zp;
                if (!n3_pixel(f, inv, xi, yi, IS, w, zp)) continue;
                if (!(zp > near_ && zp < far_)) continue;
                uint32_t bits; memcpy(&bits, &zp, 4);
                const unsigned long long key = ((unsigned long long)bits << 32) | (unsigned)fn;
                unsigned long long& cell = zkey[(size_t)yi * IS + xi];
                if (key < cell) { cell = key; for (int k = 0; k < 3; k++) weight_map[((size_t)yi * IS + xi) * 3 + k] = w[k]; }
            }
    }
    for (size_t i = 0; i < pp; i++) {
        if (zkey[i] == ~0ull) { face_index_map[i] = -1; depth_map[i] = far_; }
        else { face_index_map[i] = (int32_t)(zkey[i] & 0xffffffffull); const uint32_t bits = (uint32_t)(zkey[i] >> 32); memcpy(&depth_map[i], &bits, 4); }
    }
    delete[] zkey;
    return 0;
}

}  // extern "C"
