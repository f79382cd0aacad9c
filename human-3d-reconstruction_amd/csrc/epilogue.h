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

// FAST = the hot configuration only (NHWC output of type T, Cout % 4 == 0): a fraction of the general
// epilogue's code, which matters because the conv/DCN kernels otherwise approach the 64 KB instruction
// cache that two CUs share.  Launchers pick FAST whenever the op qualifies.
template <typename T, int MT, int NT, bool FAST = false>
__device__ __forceinline__ void tile_epilogue(f32x16 (&acc)[MT][NT], const EpiArgs &a, int b, int oy0, int ox0,
                                              int cout0, int wv, int r, int h)
{
    using E = ET<T>;
    if constexpr (FAST) {
        const bool has_res = a.res != nullptr;
        const float lo = a.relu ? 0.f : -__builtin_inff();   // fmaxf(v, -inf) == v
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int py = wv * (2 * NT) + n * 2 + (r >> 4), px = r & 15;
            const int oy = oy0 + py, ox = ox0 + px;
            if (oy >= a.Ho || ox >= a.Wo) continue;
            const size_t opix = ((size_t)b * a.Ho + oy) * a.Wo + ox;
            T *op = reinterpret_cast<T *>(a.out) + opix * a.out_cs + cout0 + 4 * h;
            const T *rp = reinterpret_cast<const T *>(a.res) + opix * a.res_cs + cout0 + 4 * h;
            const float *bp = a.bias + cout0 + 4 * h;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float4 bias4[4];       // the packed bias covers every row of the tile (rows are padded): no guards,
#pragma unroll                         // the four loads of an M-tile in flight together
                for (int g = 0; g < 4; ++g) bias4[g] = *reinterpret_cast<const float4 *>(bp + m * 32 + 8 * g);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = m * 32 + 8 * g;
                    if (cout0 + co + 4 * h >= a.Cout) continue;
                    const float4 bv = bias4[g];
                    float v0 = acc[m][n][4 * g + 0] + bv.x, v1 = acc[m][n][4 * g + 1] + bv.y;
                    float v2 = acc[m][n][4 * g + 2] + bv.z, v3 = acc[m][n][4 * g + 3] + bv.w;
                    if (has_res) {
                        float rv[4];
                        load4<T>(rp + co, rv);
                        v0 += rv[0]; v1 += rv[1]; v2 += rv[2]; v3 += rv[3];
                    }
                    store4<T>(op + co, fmaxf(v0, lo), fmaxf(v1, lo), fmaxf(v2, lo), fmaxf(v3, lo));   // (store4<f16_t> clamps to the fp16 range)
                }
            }
        }
        return;
    }
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

// LDS-transposed epilogue (bf16 / fp16 NHWC outputs, Cout % 8 == 0, out_cs % 8 == 0, 16-byte aligned bases).
// The MFMA C layout gives each lane 4 consecutive channels of one pixel, so a direct store writes 8-byte
// pieces 2*out_cs bytes apart: 16 partial writes per 128-byte line.  Here every wave first drops its
// 32 pixels x 32*MT channels (bias, residual, ReLU applied in fp32, then rounded) into its own LDS region
// [32 px][64*MT + 16 B], then streams them out 16 bytes per lane with the lanes of a pixel adjacent:
// whole lines per store instruction.  No workgroup barrier: a wave only touches its own region
// (`lw`, epi_lds_stride<MT, NT>() bytes = NT * 32 * (64*MT + 16) rounded up to whole KiB for the residual DMA; it must
// no longer be read by anyone else).  Residual images must be smaller than 2 GiB (buffer offsets).
template <int MT, int NT = 1> constexpr int epi_lds_stride() { return (NT * 32 * (64 * MT + 16) + 1023) / 1024 * 1024; }

template <typename T, int MT, int NT, bool HASRES>
__device__ __forceinline__ void tile_epilogue_lds_impl(f32x16 (&acc)[MT][NT], const EpiArgs &a, int b, int oy0, int ox0,
                                                       int cout0, int wv, int l, char *lw)
{
    constexpr int ROWB = 64 * MT + 16;
    constexpr int SPP = 4 * MT;                  // 16-byte slots per pixel
    constexpr int NPX = 32 * NT;                 // pixels of this wave: rows wv*2NT .. +2NT-1 of the tile, 16 wide
    const int r = l & 31, h = l >> 5;
    if constexpr (HASRES) {
        // residual tile: LDS-DMA straight into this wave's region in the layout the result will have
        // ([NPX px][SPP data slots + 1 pad slot]); pad slots, pixels outside the image and channels past Cout
        // carry an out-of-range offset (zeros, no traffic).  Whole lines per request, no registers.
        const size_t img_bytes = (size_t)a.Ho * a.Wo * a.res_cs * 2;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)(a.res + (size_t)b * img_bytes), 0, (int)img_bytes, 0x00020000);
        constexpr int PIECES = (NPX * ROWB + 1023) / 1024;
#pragma unroll
        for (int pc = 0; pc < PIECES; ++pc) {
            const int q = pc * 64 + l;
            const int p = q / (SPP + 1), sl = q - p * (SPP + 1);
            const int oy = oy0 + 2 * NT * wv + (p >> 4), ox = ox0 + (p & 15);
            const bool ok = p < NPX && sl < SPP && oy < a.Ho && ox < a.Wo && cout0 + sl * 8 < a.Cout;
            const int voff = ok ? ((oy * a.Wo + ox) * a.res_cs + cout0 + sl * 8) * 2 : 0x7ffffff0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(lw + pc * 1024), 16, voff, 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0)
        __builtin_amdgcn_wave_barrier();
    }
    {
        const float *bp = a.bias + cout0 + 4 * h;
        const float lo = a.relu ? 0.f : -__builtin_inff();
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float4 bias4[4];           // padded bias rows: unguarded, the four loads of an M-tile in flight together
#pragma unroll
            for (int g = 0; g < 4; ++g) bias4[g] = *reinterpret_cast<const float4 *>(bp + m * 32 + 8 * g);
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = m * 32 + 8 * g;
                    const float4 bv = bias4[g];
                    float v0 = acc[m][n][4 * g + 0] + bv.x, v1 = acc[m][n][4 * g + 1] + bv.y;
                    float v2 = acc[m][n][4 * g + 2] + bv.z, v3 = acc[m][n][4 * g + 3] + bv.w;
                    char *slot = lw + (n * 32 + r) * ROWB + (co + 4 * h) * 2;      // this lane's 4 channels of its pixel
                    if constexpr (HASRES) {
                        const u32x2 rr = *reinterpret_cast<const u32x2 *>(slot);
                        float r0, r1, r2, r3;
                        EP<T>::unpack2(rr[0], r0, r1);
                        EP<T>::unpack2(rr[1], r2, r3);
                        v0 += r0; v1 += r1; v2 += r2; v3 += r3;
                    }
                    const u32x2 pk = {EP<T>::pack2(EP<T>::clamp(v0, lo), EP<T>::clamp(v1, lo)), EP<T>::pack2(EP<T>::clamp(v2, lo), EP<T>::clamp(v3, lo))};
                    *reinterpret_cast<u32x2 *>(slot) = pk;
                }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): my wave's LDS writes are done
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < NT * SPP / 2; ++it) {
        const int q = it * 64 + l;
        const int p = q / SPP, sl = q - p * SPP;
        const int oy = oy0 + 2 * NT * wv + (p >> 4), ox = ox0 + (p & 15);
        if (oy < a.Ho && ox < a.Wo && cout0 + sl * 8 < a.Cout) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(lw + p * ROWB + sl * 16);
            const size_t opix = ((size_t)b * a.Ho + oy) * a.Wo + ox;
            *reinterpret_cast<u32x4 *>(a.out + (opix * a.out_cs + cout0 + sl * 8) * 2) = v;
        }
    }
}

template <typename T, int MT, int NT = 1>       // T: bf16_t | f16_t (the rounding of the stored result and the residual's format)
__device__ __forceinline__ void tile_epilogue_lds(f32x16 (&acc)[MT][NT], const EpiArgs &a, int b, int oy0, int ox0,
                                                  int cout0, int wv, int l, char *lw)
{
    static_assert(sizeof(T) == 2, "2-byte NHWC outputs");
    if (a.res) tile_epilogue_lds_impl<T, MT, NT, true>(acc, a, b, oy0, ox0, cout0, wv, l, lw);     // wave-uniform
    else tile_epilogue_lds_impl<T, MT, NT, false>(acc, a, b, oy0, ox0, cout0, wv, l, lw);
}
