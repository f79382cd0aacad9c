// Shared accumulator epilogue of the conv / DCN MFMA kernels.
// Tile = (NT*8) rows x 16 columns of pixels per block quarter; wave `wv` owns rows
// [wv*2*NT, (wv+1)*2*NT); N-tile n covers rows +2n, +2n+1.  C/D map of the 32x32 MFMA:
// col = lane&31 (pixel: row (lane>>4)&1, column lane&15), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
// (output channel) -> each lane owns 4 runs of 4 consecutive channels of ONE pixel.
#pragma once
#include "common.h"

struct EpiArgs {
    const float *bias;
    const char *res;
    char *out;
    int Ho, Wo, Cout, out_cs, res_cs, relu, out_mode;
};

template <typename T, int MT, int NT>
__device__ __forceinline__ void tile_epilogue(f32x16 (&acc)[MT][NT], const EpiArgs &a, int b, int oy0, int ox0,
                                              int cout0, int wv, int r, int h)
{
    using E = ET<T>;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int py = wv * (2 * NT) + n * 2 + (r >> 4), px = r & 15;
        const int oy = oy0 + py, ox = ox0 + px;
        if (oy >= a.Ho || ox >= a.Wo) continue;
        const size_t opix = ((size_t)b * a.Ho + oy) * a.Wo + ox;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = cout0 + m * 32 + 8 * g + 4 * h;
                if (c >= a.Cout) continue;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[m][n][4 * g + i] + a.bias[c + i];
                const bool full = (c + 4 <= a.Cout);
                if (a.res) {
                    const T *rp = reinterpret_cast<const T *>(a.res) + opix * a.res_cs + c;
                    if (full) {
                        float rv[4];
                        load4<T>(rp, rv);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] += rv[i];
                    } else {
                        for (int i = 0; i < 4 && c + i < a.Cout; ++i) v[i] += E::to_f32(rp[i]);
                    }
                }
                if (a.relu) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
                }
                if (a.out_mode == H3D_OUT_NHWC) {
                    T *op = reinterpret_cast<T *>(a.out) + opix * a.out_cs + c;
                    if (full) store4<T>(op, v[0], v[1], v[2], v[3]);
                    else for (int i = 0; i < 4 && c + i < a.Cout; ++i) op[i] = E::from_f32(v[i]);
                } else if (a.out_mode == H3D_OUT_NHWC_F32) {
                    float *op = reinterpret_cast<float *>(a.out) + opix * a.out_cs + c;
                    if (full) store4<float>(op, v[0], v[1], v[2], v[3]);
                    else for (int i = 0; i < 4 && c + i < a.Cout; ++i) op[i] = v[i];
                } else {  // NCHW fp32: consecutive lanes = consecutive x -> coalesced rows
                    float *op = reinterpret_cast<float *>(a.out);
                    for (int i = 0; i < 4 && c + i < a.Cout; ++i)
                        op[(((size_t)b * a.Cout + c + i) * a.Ho + oy) * a.Wo + ox] = v[i];
                }
            }
        }
    }
}
