// Dense convolution stack of DLA-34 on gfx950: implicit-GEMM on MFMA with the input halo tile
// staged once in LDS and re-read by all kh*kw taps (no im2col buffer, no per-tap re-fetch).
//
// Replaces the cuDNN calls behind nn.Conv2d / BatchNorm2d / ReLU / residual add of
// /root/reference/src/lib/models/model.py:32-61 (BasicBlock), 148-166 (Root), 200-207 (project),
// 274-284 (conv levels), 451-460 (heads).  BatchNorm (eval) is folded into weights/bias by the
// host packer; bias, residual add and ReLU are the epilogue.
//
// GEMM orientation: M = output channels (A operand = weights), N = pixels (B operand = halo
// tile), so an accumulator lane owns ONE pixel and 4-channel runs -> NHWC stores of 8/16 B per
// lane and fully coalesced NCHW rows for the head outputs.
//
// Block = 256 threads = 4 waves; output tile = TH x 16 pixels of one image x (32*MT) channels.
// wave w owns tile rows [w*TH/4, (w+1)*TH/4): NT = TH/8 N-tiles of 32 pixels (2 rows x 16).
// LDS image of the halo: pixel stride SB = CK*sizeof(T)+16 bytes (odd multiple of 16 B), row
// stride RB = multiple of 256 B, which makes every ds_read_b128 fragment read conflict-free for
// stride-1 convs (bank = (addr/4)%64, 16-lane groups {0-3,12-15,20-27},{4-11,16-19,28-31}).
#include "common.h"
#include "epilogue.h"

struct ConvArgs {
    const char *in;
    const char *w;
    const float *bias;
    const char *res;
    char *out;
    int B, H, W, Cin, in_cs;
    int Ho, Wo, Cout, out_cs, res_cs;
    int relu, out_mode;
    int tiles_x, tiles_y;
    float wscale;      // f16x3 plans: 2^-wexp, the packed filters are 2^wexp times the layer's (h3d_op.wexp); 1 otherwise
};

template <typename T, int KS, int STRIDE, int MT, int CK, int TH, int WAVES = 4>
struct ConvCfg {
    static constexpr int TW = 16;
    static constexpr int ES = sizeof(T);
    static constexpr int PAD = KS / 2;
    static constexpr int IN_H = (TH - 1) * STRIDE + KS;
    static constexpr int IN_W = (TW - 1) * STRIDE + KS;
    static constexpr int SB = CK * ES + 16;
    static constexpr int RB = ((IN_W * SB + 255) / 256) * 256;
    static constexpr int WB = KS * KS * CK * ES + 16;
    static constexpr int BN = 32 * MT;
    static constexpr int NT = TH / (2 * WAVES);   // N-tiles (2 rows x 16 px) per wave
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int VPP = CK * ES / 16;  // 16-byte vectors per pixel per chunk
    static constexpr int LDS_IN = IN_H * RB;
    static constexpr int LDS_W = BN * WB;
    static constexpr int LDS_MAIN = LDS_IN + LDS_W;
    static constexpr int LDS_EPI = sizeof(T) == 2 ? WAVES * ((NT * 32 * (64 * MT + 16) + 1023) / 1024 * 1024) : 0;   // tile_epilogue_lds
    static constexpr int LDS = (LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI) <= 160 * 1024 ? (LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI) : LDS_MAIN;
    static constexpr bool EPI_LDS_OK = sizeof(T) == 2 && MT >= 2 && LDS_EPI <= LDS;
};

template <typename T, int KS, int STRIDE, int MT, int CK, int TH, int WAVES = 4, int EPI = 0>   // EPI: 0 general, 1 lean NHWC, 2 LDS-transposed
__global__ __launch_bounds__(64 * WAVES) void conv_kernel(ConvArgs a)
{
    using C = ConvCfg<T, KS, STRIDE, MT, CK, TH, WAVES>;
    using E = ET<T>;
    constexpr int ES = C::ES;
    __shared__ __attribute__((aligned(16))) char smem[C::LDS];
    char *s_in = smem;
    char *s_w = smem + C::LDS_IN;

    const int tid = threadIdx.x;
    const int wv = tid >> 6, l = tid & 63, r = l & 31, h = l >> 5;
    const int tiles = a.tiles_x * a.tiles_y;
    const int b = blockIdx.x / tiles;
    const int t = blockIdx.x - b * tiles;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * C::TW;
    const int cout0 = blockIdx.y * C::BN;

    f32x16 acc[MT][C::NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < C::NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    // per-lane LDS byte offsets of the B (pixel) fragments at tap (0,0), k = 0
    int boff[C::NT];
#pragma unroll
    for (int n = 0; n < C::NT; ++n) {
        const int py = wv * (2 * C::NT) + n * 2 + (r >> 4), px = r & 15;
        boff[n] = (py * STRIDE) * C::RB + (px * STRIDE) * C::SB + 8 * h * ES;
    }
    const int aoff = r * C::WB + 8 * h * ES;

    const size_t in_img = (size_t)b * a.H * a.W;
    // ---- software pipeline over channel chunks: the global loads of chunk c+1 (input halo
    //      [IN_H][IN_W][CK], zero outside the image, and weight slab [BN][KS*KS][CK]) are issued into
    //      registers BEFORE the MFMAs of chunk c and written to LDS after them, so only the first
    //      chunk's memory latency is exposed.
    constexpr int WV = KS * KS * C::VPP;
    constexpr int NH = C::IN_H * C::IN_W * C::VPP, NW = C::BN * WV;
    constexpr int NV = (NH + NW + C::THREADS - 1) / C::THREADS;
    u32x4 stg[NV];
    auto load_stage = [&](int c0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + j * C::THREADS;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (i < NH) {
                const int v = i % C::VPP, pix = i / C::VPP;
                const int iy = pix / C::IN_W, ix = pix - iy * C::IN_W;
                const int gy = oy0 * STRIDE - C::PAD + iy, gx = ox0 * STRIDE - C::PAD + ix;
                if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                    val = *reinterpret_cast<const u32x4 *>(
                        a.in + ((in_img + (size_t)gy * a.W + gx) * a.in_cs + c0) * ES + v * 16);
            } else if (i < NH + NW) {
                const int q0 = i - NH;
                const int row = q0 / WV, q = q0 - row * WV;
                const int tap = q / C::VPP, v = q - tap * C::VPP;
                val = *reinterpret_cast<const u32x4 *>(
                    a.w + (((size_t)(cout0 + row) * (KS * KS) + tap) * a.Cin + c0) * ES + v * 16);
            }
            stg[j] = val;
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + j * C::THREADS;
            if (i < NH) {
                const int v = i % C::VPP, pix = i / C::VPP;
                const int iy = pix / C::IN_W, ix = pix - iy * C::IN_W;
                if constexpr (std::is_same_v<T, x3_t>)      // f16x3 plans: fp32 in memory, (hi | lo) fp16 terms per 8-channel group in LDS
                    x3_store4(s_in + iy * C::RB + ix * C::SB + (v >> 1) * 32, v & 1, stg[j]);
                else
                    *reinterpret_cast<u32x4 *>(s_in + iy * C::RB + ix * C::SB + v * 16) = stg[j];
            } else if (i < NH + NW) {
                const int q0 = i - NH;
                const int row = q0 / WV, q = q0 - row * WV;
                const int tap = q / C::VPP, v = q - tap * C::VPP;
                *reinterpret_cast<u32x4 *>(s_w + row * C::WB + tap * CK * ES + v * 16) = stg[j];
            }
        }
    };

    load_stage(0);
    for (int c0 = 0; c0 < a.Cin; c0 += CK) {
        if (c0) __syncthreads();            // every wave is done reading the previous chunk
        store_stage();
        __syncthreads();
        if (c0 + CK < a.Cin) load_stage(c0 + CK);
        // ---- contraction over this chunk: taps x CK/16 k-steps -------------------------------------
        if constexpr (std::is_same_v<T, x3_t> && KS == 3 && CK == 16 && (MT + C::NT) * 16 + MT * C::NT * 16 <= 176) {
            // f16x3: left to itself hipcc reads ONE 16-byte fragment half, waits for it (lgkmcnt(0)) and issues one or two MFMAs -- an LDS
            // round trip exposed per 48-96 matrix cycles, 2 waves a SIMD to cover it (SQ counters: pipe 0.43 busy, waves parked 0.53 of
            // the time).  Two fragment sets: tap t+1 is read while tap t multiplies, the order pinned by sched_group_barrier.
            typename E::frag fa[2][MT], fb[2][C::NT];
            auto rd = [&](int tap, int q) {
                const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
                for (int n = 0; n < C::NT; ++n) fb[q][n] = E::lds_frag(s_in + boff[n] + dy * C::RB + dx * C::SB);
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[q][m] = E::lds_frag(s_w + aoff + m * 32 * C::WB + tap * CK * ES);
            };
            constexpr int NRD = 2 * (MT + C::NT), NMM = 3 * MT * C::NT;
            rd(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap + 1 < 9) rd(tap + 1, (tap + 1) & 1);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < C::NT; ++n) E::mma(acc[m][n], fa[tap & 1][m], fb[tap & 1][n]);
                if (tap + 1 < 9) {
#pragma unroll
                    for (int i = 0; i < NRD; ++i) {      // one fragment read behind each of the first MFMAs
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, NMM - NRD, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, NMM, 0);
                }
            }
        } else
#pragma unroll
        for (int tap = 0; tap < KS * KS; ++tap) {
            const int dy = tap / KS, dx = tap - dy * KS;
#pragma unroll
            for (int kk = 0; kk < CK / 16; ++kk) {
                typename E::frag fa[MT], fb[C::NT];
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    fa[m] = E::lds_frag(s_w + aoff + m * 32 * C::WB + (tap * CK + kk * 16) * ES);
#pragma unroll
                for (int n = 0; n < C::NT; ++n)
                    fb[n] = E::lds_frag(s_in + boff[n] + dy * C::RB + dx * C::SB + kk * 16 * ES);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < C::NT; ++n) E::mma(acc[m][n], fa[m], fb[n]);
            }
        }
    }

    if constexpr (std::is_same_v<T, x3_t>) {      // the filters were packed times 2^wexp (normal fp16 lo terms): an exact power-of-two unscale
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < C::NT; ++n)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][n][i] *= a.wscale;
    }
    EpiArgs e;
    e.bias = a.bias; e.res = a.res; e.out = a.out; e.Ho = a.Ho; e.Wo = a.Wo; e.Cout = a.Cout;
    e.out_cs = a.out_cs; e.res_cs = a.res_cs; e.relu = a.relu; e.out_mode = a.out_mode;
    if constexpr (EPI == 2 && C::EPI_LDS_OK) {
        __syncthreads();                          // halo and filters are no longer read
        tile_epilogue_lds<T, MT, C::NT>(acc, e, b, oy0, ox0, cout0, __builtin_amdgcn_readfirstlane(wv), l, smem + wv * epi_lds_stride<MT, C::NT>());
    } else {
        tile_epilogue<typename StoreT<T>::type, MT, C::NT, EPI == 1>(acc, e, b, oy0, ox0, cout0, wv, r, h);
    }
}

template <typename T, int KS, int STRIDE, int MT, int CK, int TH, int WAVES = 4>
static int launch_conv_cfg(const ConvArgs &a0, hipStream_t st)
{
    using C = ConvCfg<T, KS, STRIDE, MT, CK, TH, WAVES>;
    static_assert(C::LDS <= 160 * 1024, "LDS budget");
    ConvArgs a = a0;
    a.tiles_x = cdiv(a.Wo, C::TW);
    a.tiles_y = cdiv(a.Ho, TH);
    dim3 grid(a.B * a.tiles_x * a.tiles_y, cdiv(a.Cout, C::BN));
    const bool lean = a.out_mode == H3D_OUT_NHWC && a.Cout % 4 == 0 && ((uintptr_t)a.bias & 15) == 0;
    const int epi = (C::EPI_LDS_OK && lean && a.Cout % 8 == 0 && a.out_cs % 8 == 0 && ((uintptr_t)a.out & 15) == 0 &&
                     (!a.res || (a.res_cs % 8 == 0 && ((uintptr_t)a.res & 15) == 0)))
                        ? 2 : lean ? 1 : 0;
    if (h3d_note_kernel("conv_kernel<%s, %d, %d, %d, %d, %d, %d, %d>", h3d_tname<T>(), KS, STRIDE, MT,
                        CK, TH, WAVES, epi))
        return H3D_OK;
    if constexpr (C::EPI_LDS_OK) {
        if (epi == 2) {
            hipLaunchKernelGGL((conv_kernel<T, KS, STRIDE, MT, CK, TH, WAVES, 2>), grid, dim3(C::THREADS), 0, st, a);
            H3D_CHECK_LAUNCH("conv_kernel");
            return H3D_OK;
        }
    }
    if (epi == 1)
        hipLaunchKernelGGL((conv_kernel<T, KS, STRIDE, MT, CK, TH, WAVES, 1>), grid, dim3(C::THREADS), 0, st, a);
    else
        hipLaunchKernelGGL((conv_kernel<T, KS, STRIDE, MT, CK, TH, WAVES, 0>), grid, dim3(C::THREADS), 0, st, a);
    H3D_CHECK_LAUNCH("conv_kernel");
    return H3D_OK;
}

// 2-byte element types (bf16_t, f16_t): one tiling table
template <typename T> static int launch_conv_t(const h3d_op &op, const ConvArgs &a, hipStream_t st)
{
    static_assert(sizeof(T) == 2, "float has its own table below");
    const int cin = op.Cin, co = op.Cout;
    // workgroups a (TH rows x 16 px) x (bn channels) tiling produces; small-resolution layers
    // (level3..5, B*H*W <= 256k px) are re-tiled finer so every CU gets >= 4 workgroups
    auto nblk = [&](int th, int bn) { return (long)op.B * cdiv(op.Wo, 16) * cdiv(op.Ho, th) * cdiv(co, bn); };
    constexpr long WANT = 1024;
    if (op.ksize == 3 && op.stride == 1) {
        if (cin % 32 == 0) {
            if (co <= 32) return launch_conv_cfg<T, 3, 1, 1, 32, 16>(a, st);
            if (co <= 64) {
                if (nblk(16, 64) >= WANT) return launch_conv_cfg<T, 3, 1, 2, 32, 16, 8>(a, st);   // 8 waves x (32 px x 64 ch)
                return launch_conv_cfg<T, 3, 1, 2, 32, 8>(a, st);
            }
            // 8 waves x (32 px x 128 ch), CK = 32: a chunk's MFMA time now exceeds the prefetch latency
            if (nblk(16, 128) >= WANT) return launch_conv_cfg<T, 3, 1, 4, 32, 16, 8>(a, st);
            if (nblk(8, 128) >= WANT) return launch_conv_cfg<T, 3, 1, 4, 16, 8>(a, st);
            return launch_conv_cfg<T, 3, 1, 2, 32, 8>(a, st);
        }
        if (co <= 32) return launch_conv_cfg<T, 3, 1, 1, 16, 16>(a, st);
        if (co <= 64) return launch_conv_cfg<T, 3, 1, 2, 16, 16>(a, st);
        return launch_conv_cfg<T, 3, 1, 4, 16, 16>(a, st);
    }
    if (op.ksize == 3 && op.stride == 2) {
        if (co <= 32) return launch_conv_cfg<T, 3, 2, 1, 16, 8>(a, st);
        if (co <= 64 || nblk(8, 128) < WANT) return launch_conv_cfg<T, 3, 2, 2, 16, 8>(a, st);
        return launch_conv_cfg<T, 3, 2, 4, 16, 8>(a, st);
    }
    if (op.ksize == 1 && op.stride == 1) {
        if (cin % 64 == 0) {
            if (co > 32 && (op.reserved & 0x1000)) {           // tuning override (tools/ab_conv1x1.py): 0x1000 | MT << 4 | TH >> 3
                const int mt = (op.reserved >> 4) & 15, th = (op.reserved & 15) * 8;
                if (mt == 4 && th == 16) return launch_conv_cfg<T, 1, 1, 4, 64, 16>(a, st);
                if (mt == 4 && th == 8) return launch_conv_cfg<T, 1, 1, 4, 64, 8>(a, st);
                if (mt == 2 && th == 16) return launch_conv_cfg<T, 1, 1, 2, 64, 16>(a, st);
                if (mt == 2 && th == 8) return launch_conv_cfg<T, 1, 1, 2, 64, 8>(a, st);
            }
            if (co <= 32) return launch_conv_cfg<T, 1, 1, 1, 64, 16>(a, st);
            // 8-row tiles throughout: these layers are HBM / latency bound and twice the workgroups hide more of it
            // (tools/ab_conv1x1.py, batch 64: 448->128 @64x64 0.085 -> 0.073 ms, 1280->512 @16x16 0.042 -> 0.036, ...)
            if (co <= 64) return launch_conv_cfg<T, 1, 1, 2, 64, 8>(a, st);
            return launch_conv_cfg<T, 1, 1, 4, 64, 8>(a, st);
        }
        if (co <= 32) return launch_conv_cfg<T, 1, 1, 1, 16, 16>(a, st);
        if (co <= 64) return launch_conv_cfg<T, 1, 1, 2, 16, 16>(a, st);
        return launch_conv_cfg<T, 1, 1, 4, 16, 16>(a, st);
    }
    if (op.ksize == 1 && op.stride == 2) {        // the 1x1 stride-2 skip convs of the Hourglass / ResNet residual blocks
        if (co <= 32) return launch_conv_cfg<T, 1, 2, 1, 16, 8>(a, st);
        if (co <= 64) return launch_conv_cfg<T, 1, 2, 2, 16, 8>(a, st);
        return launch_conv_cfg<T, 1, 2, 4, 16, 8>(a, st);
    }
    H3D_FAIL(H3D_ERR_UNSUPPORTED, "conv: ksize=%d stride=%d not covered (k in {1,3}, stride in {1,2})",
             op.ksize, op.stride);
}

template <> int launch_conv_t<float>(const h3d_op &op, const ConvArgs &a, hipStream_t st)
{
    const int co = op.Cout;
    if (op.ksize == 3 && op.stride == 1) {
        if (co <= 32) return launch_conv_cfg<float, 3, 1, 1, 16, 16>(a, st);
        return launch_conv_cfg<float, 3, 1, 2, 16, 16>(a, st);
    }
    if (op.ksize == 3 && op.stride == 2) {
        if (co <= 32) return launch_conv_cfg<float, 3, 2, 1, 16, 8>(a, st);
        return launch_conv_cfg<float, 3, 2, 2, 16, 8>(a, st);
    }
    if (op.ksize == 1 && op.stride == 1) {
        if (co <= 32) return launch_conv_cfg<float, 1, 1, 1, 16, 16>(a, st);
        return launch_conv_cfg<float, 1, 1, 2, 16, 16>(a, st);
    }
    if (op.ksize == 1 && op.stride == 2) {
        if (co <= 32) return launch_conv_cfg<float, 1, 2, 1, 16, 8>(a, st);
        return launch_conv_cfg<float, 1, 2, 2, 16, 8>(a, st);
    }
    H3D_FAIL(H3D_ERR_UNSUPPORTED, "conv: ksize=%d stride=%d not covered (k in {1,3}, stride in {1,2})",
             op.ksize, op.stride);
}

// f16x3 plans: the f32 plan's layouts (4-byte elements, 16-channel chunks) on 3 fp16 MFMAs per step instead of 8 fp32 ones.
// (tuning override, tools/ab_conv.py --dtype f16x3: reserved = 0x1000 | MT << 8 | WAVES << 4 | TH >> 3)
template <> int launch_conv_t<x3_t>(const h3d_op &op, const ConvArgs &a, hipStream_t st)
{
    const int co = op.Cout;
    if (op.ksize == 3 && op.stride == 1) {
        if (op.reserved & 0x1000) {
            switch (op.reserved & 0xfff) {
            case 0x142: return launch_conv_cfg<x3_t, 3, 1, 1, 16, 16>(a, st);
            case 0x182: return launch_conv_cfg<x3_t, 3, 1, 1, 16, 16, 8>(a, st);
            case 0x242: return launch_conv_cfg<x3_t, 3, 1, 2, 16, 16>(a, st);
            case 0x282: return launch_conv_cfg<x3_t, 3, 1, 2, 16, 16, 8>(a, st);
            case 0x482: return launch_conv_cfg<x3_t, 3, 1, 4, 16, 16, 8>(a, st);
            case 0x441: return launch_conv_cfg<x3_t, 3, 1, 4, 16, 8>(a, st);
            case 0x241: return launch_conv_cfg<x3_t, 3, 1, 2, 16, 8>(a, st);
            case 0x484: return launch_conv_cfg<x3_t, 3, 1, 4, 16, 32, 8>(a, st);      // two N-tiles per wave: a filter fragment feeds 6 MFMAs
            case 0x284: return launch_conv_cfg<x3_t, 3, 1, 2, 16, 32, 8>(a, st);
            default: H3D_FAIL(H3D_ERR_ARG, "conv (f16x3): unknown tuning override %#x", op.reserved);
            }
        }
        // (tools/ab_conv_x3.py, batch 64, same process: 64 -> 64 @128x128 1.170 ms on 4-wave tiles vs 0.985 on 8-wave ones; 512 -> 512
        //  @16x16 0.639 on 64-channel tiles vs 0.566 on 128-channel ones although that grid is half a round of workgroups)
        if (co <= 32) return launch_conv_cfg<x3_t, 3, 1, 1, 16, 16, 8>(a, st);      // (16 -> 16 @512x512: 0.912 ms on 4-wave tiles, 0.773 on 8-wave ones)
        if (co <= 64) return launch_conv_cfg<x3_t, 3, 1, 2, 16, 16, 8>(a, st);
        // (two N-tiles per wave on 64-channel tiles against one on 128-channel tiles, both with the pipelined fragment reads: 128 -> 128
        //  @64x64 1.656 -> 1.609 ms for the seven launches, 256 -> 256 @32x32 1.383 -> 1.331; 32-row tiles on a 16-row map: 0.54 -> 0.87)
        if (a.Ho % 32 == 0) return launch_conv_cfg<x3_t, 3, 1, 2, 16, 32, 8>(a, st);
        return launch_conv_cfg<x3_t, 3, 1, 4, 16, 16, 8>(a, st);
    }
    if (op.ksize == 3 && op.stride == 2) {
        if (op.reserved & 0x1000) {         // tuning override (tools/ab_conv_x3.py --stride 2)
            switch (op.reserved & 0xfff) {
            case 0x282: return launch_conv_cfg<x3_t, 3, 2, 2, 16, 16, 8>(a, st);
            case 0x182: return launch_conv_cfg<x3_t, 3, 2, 1, 16, 16, 8>(a, st);
            case 0x141: return launch_conv_cfg<x3_t, 3, 2, 1, 16, 8>(a, st);
            case 0x241: return launch_conv_cfg<x3_t, 3, 2, 2, 16, 8>(a, st);
            default: H3D_FAIL(H3D_ERR_ARG, "conv (f16x3, stride 2): unknown tuning override %#x", op.reserved);
            }
        }
        // (tools/ab_conv_x3.py --stride 2, batch 64, same process: 8-wave 16-row tiles against the f32 plan's 4-wave 8-row ones, which
        //  leave ONE 4-wave workgroup on a CU: 32 -> 64 @256 0.357 -> 0.285 ms, 64 -> 128 0.303 -> 0.228, 128 -> 256 0.251 -> 0.191, 256 -> 512 0.219 -> 0.145)
        if (co <= 32) return launch_conv_cfg<x3_t, 3, 2, 1, 16, 8>(a, st);
        return launch_conv_cfg<x3_t, 3, 2, 2, 16, 16, 8>(a, st);
    }
    if (op.ksize == 1 && op.stride == 1) {
        // 64-channel chunks where the layer allows (Root convs over a concat: 128 ... 1280 input channels): a 16-channel chunk of a
        // 1x1 conv is two barriers and a global round trip for 2 * MT * NT * 3 MFMAs per wave (0x2000: tuning override, 16-channel chunks)
        // ... and 128-channel tiles above 64 output channels: a 64-channel tile walks the whole input once per channel block (tools/ab_conv_x3.py
        // --ksize 1, batch 64: 448 -> 128 @64x64 0.218 -> 0.156 ms, 256 -> 128 0.144 -> 0.108, 896 -> 256 0.149 -> 0.126, 1280 -> 512 0.102 -> 0.080)
        if (op.Cin % 64 == 0 && co > 64 && !(op.reserved & 0x6000)) return launch_conv_cfg<x3_t, 1, 1, 4, 64, 8>(a, st);
        if (op.Cin % 64 == 0 && co > 32 && !(op.reserved & 0x2000)) return launch_conv_cfg<x3_t, 1, 1, 2, 64, 8>(a, st);      // (0x4000: force these for > 64 channels)      // (8-row tiles: 52 KB of LDS, three workgroups per CU)
        if (co <= 32) return launch_conv_cfg<x3_t, 1, 1, 1, 16, 16>(a, st);
        return launch_conv_cfg<x3_t, 1, 1, 2, 16, 16>(a, st);
    }
    if (op.ksize == 1 && op.stride == 2) {
        if (co <= 32) return launch_conv_cfg<x3_t, 1, 2, 1, 16, 8>(a, st);
        return launch_conv_cfg<x3_t, 1, 2, 2, 16, 8>(a, st);
    }
    H3D_FAIL(H3D_ERR_UNSUPPORTED, "conv: ksize=%d stride=%d not covered (k in {1,3}, stride in {1,2})",
             op.ksize, op.stride);
}

bool h3d_gemm1_takes(const h3d_op &op);
int h3d_launch_gemm1(const h3d_op &op, hipStream_t st);

int h3d_launch_conv(const h3d_op &op, hipStream_t st)
{
    if (!op.in || !op.w || !op.bias || !op.out) H3D_FAIL(H3D_ERR_ARG, "conv: null pointer");
    const int es = h3d_dtype_bytes(op.dtype);
    if (!es) H3D_FAIL(H3D_ERR_DTYPE, "conv: dtype %d", op.dtype);
    if (op.Cin % 16 || op.in_cs % (16 / es) || op.Cin > op.in_cs)
        H3D_FAIL(H3D_ERR_SHAPE, "conv: Cin=%d (stride %d) must be a multiple of 16", op.Cin, op.in_cs);
    const int pad = op.ksize / 2;
    const int ho = (op.H + 2 * pad - op.ksize) / op.stride + 1;
    const int wo = (op.W + 2 * pad - op.ksize) / op.stride + 1;
    if (ho != op.Ho || wo != op.Wo)
        H3D_FAIL(H3D_ERR_SHAPE, "conv: output %dx%d does not match (H+2p-k)/s+1 = %dx%d", op.Ho, op.Wo, ho, wo);
    if (op.wrows < ((op.Cout + 127) / 128) * 128)
        H3D_FAIL(H3D_ERR_SHAPE, "conv: packed weight rows %d < Cout %d padded to 128", op.wrows, op.Cout);
    if (op.out_mode != H3D_OUT_NCHW_F32 && (op.out_cs % 4 || op.Cout > op.out_cs))
        H3D_FAIL(H3D_ERR_SHAPE, "conv: out channel stride %d must be a multiple of 4 and >= Cout %d", op.out_cs, op.Cout);
    if (op.in2 && (op.in2_cs % 4)) H3D_FAIL(H3D_ERR_SHAPE, "conv: residual channel stride %d", op.in2_cs);
    ConvArgs a;
    a.in = (const char *)op.in; a.w = (const char *)op.w; a.bias = op.bias;
    a.res = (const char *)op.in2; a.out = (char *)op.out;
    a.B = op.B; a.H = op.H; a.W = op.W; a.Cin = op.Cin; a.in_cs = op.in_cs;
    a.Ho = op.Ho; a.Wo = op.Wo; a.Cout = op.Cout; a.out_cs = op.out_cs; a.res_cs = op.in2_cs;
    a.relu = op.relu; a.out_mode = op.out_mode; a.tiles_x = a.tiles_y = 0;
    if (op.wexp < -60 || op.wexp > 60 || (op.wexp && op.dtype != H3D_F16X3)) H3D_FAIL(H3D_ERR_ARG, "conv: wexp %d (an H3D_F16X3 filter exponent)", op.wexp);
    a.wscale = ldexpf(1.f, -op.wexp);
    if (h3d_gemm1_takes(op)) return h3d_launch_gemm1(op, st);      // 1x1 stride 1, bf16: the GEMM kernel (csrc/gemm1.hip)
    if (op.dtype == H3D_BF16) return launch_conv_t<bf16_t>(op, a, st);
    if (op.dtype == H3D_F16) return launch_conv_t<f16_t>(op, a, st);
    if (op.dtype == H3D_F32) return launch_conv_t<float>(op, a, st);
    if (op.dtype == H3D_F16X3) return launch_conv_t<x3_t>(op, a, st);
    H3D_FAIL(H3D_ERR_DTYPE, "conv: dtype %d", op.dtype);
}

// =================================================================================================
// Stem: base_layer = Conv2d(3, C0=16, 7x7, s1, p3, bias=False) + BN + ReLU (model.py:231-235),
// reading the NCHW fp32 image batch and writing NHWC T.  v1: direct FMA, weights via scalar loads.
// w: fp32 [16][3][7][7] (BN folded), bias fp32 [16].
template <typename T>
__global__ __launch_bounds__(256) void stem_kernel(const float *__restrict__ img, const float *__restrict__ w,
                                                   const float *__restrict__ bias, T *__restrict__ out,
                                                   int B, int H, int W, int out_cs, int tiles_x, int tiles_y)
{
    constexpr int TS = 16, HS = TS + 6;
    __shared__ float s[3][HS][HS + 1];
    const int tiles = tiles_x * tiles_y;
    const int b = blockIdx.x / tiles;
    const int t = blockIdx.x - b * tiles;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int oy0 = ty * TS, ox0 = tx * TS;
    for (int i = threadIdx.x; i < 3 * HS * HS; i += 256) {
        const int c = i / (HS * HS), rem = i - c * HS * HS;
        const int iy = rem / HS, ix = rem - iy * HS;
        const int gy = oy0 - 3 + iy, gx = ox0 - 3 + ix;
        float v = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = img[(((size_t)b * 3 + c) * H + gy) * W + gx];
        s[c][iy][ix] = v;
    }
    __syncthreads();
    const int py = threadIdx.x >> 4, px = threadIdx.x & 15;
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = bias[o];
    for (int c = 0; c < 3; ++c)
        for (int dy = 0; dy < 7; ++dy)
#pragma unroll
            for (int dx = 0; dx < 7; ++dx) {
                const float x = s[c][py + dy][px + dx];
#pragma unroll
                for (int o = 0; o < 16; ++o) acc[o] = fmaf(w[((o * 3 + c) * 7 + dy) * 7 + dx], x, acc[o]);
            }
    const int oy = oy0 + py, ox = ox0 + px;
    if (oy < H && ox < W) {
        T *op = out + (((size_t)b * H + oy) * W + ox) * out_cs;
#pragma unroll
        for (int o = 0; o < 16; o += 4)
            store4<T>(op + o, fmaxf(acc[o], 0.f), fmaxf(acc[o + 1], 0.f), fmaxf(acc[o + 2], 0.f),
                      fmaxf(acc[o + 3], 0.f));
    }
}

// bf16 stem on MFMA (v_mfma_f32_16x16x32_bf16): A = weights [16 out ch][K], B = pixels [K][16 px].
// The image tile is staged in LDS as interleaved pixels of 4 bf16 (c0,c1,c2,0), so for tap row dy the
// K run (dx=0..6, c=0..3) of a pixel is 28 CONTIGUOUS elements: one K=32 step per dy (4 zero-weight
// pad elements), 7 MFMAs per 16 pixels.  C/D map: col = lane&15 (pixel), row = 4*(lane>>4)+reg
// (output channel) -> each lane owns 4 consecutive channels of one pixel: 8-byte NHWC stores,
// 512 contiguous bytes per wave.  w: [16][7][32] bf16 (k = dx*4 + c), prepared by the host.
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
template <typename T>      // bf16_t | f16_t
__global__ __launch_bounds__(256) void stem_mfma_kernel(const float *__restrict__ img, const T *__restrict__ w,
                                                        const float *__restrict__ bias, T *__restrict__ out, int B,
                                                        int H, int W, int out_cs, int tiles_x, int tiles_y)
{
    constexpr int TH = 16, TW = 64, IH = TH + 6, IW = TW + 6 + 2;   // +2: the K=32 run of the last pixel reads 8 px
    __shared__ __attribute__((aligned(16))) uint2 s[IH][IW];        // 4 bf16 per pixel
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, p = l & 15, q = l >> 4;
    const int tiles = tiles_x * tiles_y;
    const int b = blockIdx.x / tiles;
    const int t = blockIdx.x - b * tiles;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    // weights: this lane's A fragments for the 7 tap rows (row = out channel p, k = 8q..8q+7)
    u32x4 fa[7];
#pragma unroll
    for (int dy = 0; dy < 7; ++dy) fa[dy] = *reinterpret_cast<const u32x4 *>(w + (p * 7 + dy) * 32 + 8 * q);
    const size_t plane = (size_t)H * W;
    const float *im = img + (size_t)b * 3 * plane;
    for (int i = tid; i < IH * IW; i += 256) {
        const int iy = i / IW, ix = i - iy * IW;
        const int gy = oy0 - 3 + iy, gx = ox0 - 3 + ix;
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const size_t o = (size_t)gy * W + gx;
            c0 = im[o]; c1 = im[plane + o]; c2 = im[2 * plane + o];
        }
        s[iy][ix] = uint2{EP<T>::pack2(c0, c1), EP<T>::pack2(c2, 0.f)};
    }
    __syncthreads();
    float bs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bs[i] = bias[4 * q + i];
    // 64 groups of 16 consecutive pixels; wave wv takes rows 4wv..4wv+3, 4 groups per row
#pragma unroll 1
    for (int g = 0; g < 16; ++g) {
        const int py = wv * 4 + (g >> 2), px0 = (g & 3) * 16;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 7; ++dy) {
            // K elements 8q..8q+7 of the run starting at pixel (py+dy, px0+p): pixels +2q, +2q+1
            const uint2 lo = s[py + dy][px0 + p + 2 * q], hi = s[py + dy][px0 + p + 2 * q + 1];
            const u32x4 fb = {lo.x, lo.y, hi.x, hi.y};
            if constexpr (std::is_same_v<T, f16_t>)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fa[dy]), __builtin_bit_cast(f16x8_t, fb), acc, 0, 0, 0);
            else
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[dy]), __builtin_bit_cast(bf16x8_t, fb), acc, 0, 0, 0);
        }
        const int oy = oy0 + py, ox = ox0 + px0 + p;
        if (oy < H && ox < W)
            store4<T>(out + ((size_t)(b * H + oy) * W + ox) * out_cs + 4 * q, fmaxf(acc[0] + bs[0], 0.f),
                           fmaxf(acc[1] + bs[1], 0.f), fmaxf(acc[2] + bs[2], 0.f), fmaxf(acc[3] + bs[3], 0.f));
    }
}

// f16x3 stem (round 5): the same interleaved-pixel K = 28 (+4) scheme with the fp32 image split into (hi | lo) fp16 terms while it
// is staged (two LDS tiles) and the filters pre-split by the host ([16][7][4 groups of (8 hi | 8 lo)], times 2^wexp): three
// v_mfma_f32_16x16x32_f16 per tap row -- lo.hi + hi.lo + hi.hi -- instead of 147 scalar FMAs per output value; fp32 NHWC output.
__global__ __launch_bounds__(256) void stem_x3_kernel(const float *__restrict__ img, const char *__restrict__ w, const float *__restrict__ bias,
                                                      float *__restrict__ out, int B, int H, int W, int out_cs, int tiles_x, int tiles_y, float wscale)
{
    constexpr int TH = 16, TW = 64, IH = TH + 6, IW = TW + 6 + 2;
    __shared__ __attribute__((aligned(16))) uint2 s_hi[IH][IW], s_lo[IH][IW];      // 4 fp16 per pixel: (c0, c1, c2, 0)
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, p = l & 15, q = l >> 4;
    const int tiles = tiles_x * tiles_y;
    const int b = blockIdx.x / tiles;
    const int t = blockIdx.x - b * tiles;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    u32x4 fah[7], fal[7];      // A fragments of the 7 tap rows: row = out channel p, K group q -> 8 hi terms, 8 lo terms
#pragma unroll
    for (int dy = 0; dy < 7; ++dy) {
        const char *g = w + ((size_t)(p * 7 + dy) * 32 + 8 * q) * 4;
        fah[dy] = *reinterpret_cast<const u32x4 *>(g);
        fal[dy] = *reinterpret_cast<const u32x4 *>(g + 16);
    }
    const size_t plane = (size_t)H * W;
    const float *im = img + (size_t)b * 3 * plane;
    for (int i = tid; i < IH * IW; i += 256) {
        const int iy = i / IW, ix = i - iy * IW;
        const int gy = oy0 - 3 + iy, gx = ox0 - 3 + ix;
        u32x4 raw = {0u, 0u, 0u, 0u};
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const size_t o = (size_t)gy * W + gx;
            raw = u32x4{__float_as_uint(im[o]), __float_as_uint(im[plane + o]), __float_as_uint(im[2 * plane + o]), 0u};
        }
        u32x2 hi, lo;
        x3_split4(raw, hi, lo);
        s_hi[iy][ix] = uint2{hi[0], hi[1]};
        s_lo[iy][ix] = uint2{lo[0], lo[1]};
    }
    __syncthreads();
    float bs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bs[i] = bias[4 * q + i];
#pragma unroll 1
    for (int g = 0; g < 16; ++g) {
        const int py = wv * 4 + (g >> 2), px0 = (g & 3) * 16;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 7; ++dy) {
            const uint2 h0 = s_hi[py + dy][px0 + p + 2 * q], h1 = s_hi[py + dy][px0 + p + 2 * q + 1];
            const uint2 l0 = s_lo[py + dy][px0 + p + 2 * q], l1 = s_lo[py + dy][px0 + p + 2 * q + 1];
            const u32x4 fbh = {h0.x, h0.y, h1.x, h1.y}, fbl = {l0.x, l0.y, l1.x, l1.y};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fal[dy]), __builtin_bit_cast(f16x8_t, fbh), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fah[dy]), __builtin_bit_cast(f16x8_t, fbl), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fah[dy]), __builtin_bit_cast(f16x8_t, fbh), acc, 0, 0, 0);
        }
        const int oy = oy0 + py, ox = ox0 + px0 + p;
        if (oy < H && ox < W)
            store4<float>(out + ((size_t)(b * H + oy) * W + ox) * out_cs + 4 * q, fmaxf(fmaf(acc[0], wscale, bs[0]), 0.f), fmaxf(fmaf(acc[1], wscale, bs[1]), 0.f),
                          fmaxf(fmaf(acc[2], wscale, bs[2]), 0.f), fmaxf(fmaf(acc[3], wscale, bs[3]), 0.f));
    }
}

int h3d_launch_stem_s2(const h3d_op &op, hipStream_t st);      // csrc/extra.hip: the 7x7 stride-2 stems of ResNet / Hourglass

int h3d_launch_stem(const h3d_op &op, hipStream_t st)
{
    if (!op.in || !op.w || !op.bias || !op.out) H3D_FAIL(H3D_ERR_ARG, "stem: null pointer");
    if (op.stride == 2) return h3d_launch_stem_s2(op, st);
    if (op.Cin != 3 || op.Cout != 16 || op.ksize != 7 || op.Ho != op.H || op.Wo != op.W || op.out_cs % 4)
        H3D_FAIL(H3D_ERR_SHAPE, "stem: expects 7x7 3->16 stride 1 (got k=%d %d->%d)", op.ksize, op.Cin, op.Cout);
    const int tx = cdiv(op.W, 16), ty = cdiv(op.H, 16);
    dim3 grid(op.B * tx * ty);
    if (op.dtype == H3D_BF16 || op.dtype == H3D_F16) {
        // op.w: bf16 / fp16 [16][7][32] (k = dx*4 + c, zero padded) -- see engine.PackedWeights.stem
        const int mx = cdiv(op.W, 64), my = cdiv(op.H, 16);
        if (h3d_note_kernel("stem_mfma_kernel<%s>", op.dtype == H3D_F16 ? "f16_t" : "unsigned short")) return H3D_OK;
        if (op.dtype == H3D_F16)
            hipLaunchKernelGGL(stem_mfma_kernel<f16_t>, dim3(op.B * mx * my), dim3(256), 0, st, (const float *)op.in, (const f16_t *)op.w,
                               op.bias, (f16_t *)op.out, op.B, op.H, op.W, op.out_cs, mx, my);
        else
            hipLaunchKernelGGL(stem_mfma_kernel<bf16_t>, dim3(op.B * mx * my), dim3(256), 0, st, (const float *)op.in, (const bf16_t *)op.w,
                               op.bias, (bf16_t *)op.out, op.B, op.H, op.W, op.out_cs, mx, my);
        H3D_CHECK_LAUNCH("stem_mfma_kernel");
        return H3D_OK;
    }
    if (op.dtype == H3D_F16X3) {
        // op.w: the [16][7][32] (k = dx*4 + c, zero padded) filter bank times 2^wexp as (hi | lo) fp16 terms per 8 k -- engine.PackedWeights.stem
        if (op.wexp < -60 || op.wexp > 60) H3D_FAIL(H3D_ERR_ARG, "stem: wexp %d", op.wexp);
        const int mx = cdiv(op.W, 64), my = cdiv(op.H, 16);
        if (h3d_note_kernel("stem_x3_kernel")) return H3D_OK;
        hipLaunchKernelGGL(stem_x3_kernel, dim3(op.B * mx * my), dim3(256), 0, st, (const float *)op.in, (const char *)op.w, op.bias, (float *)op.out, op.B,
                           op.H, op.W, op.out_cs, mx, my, ldexpf(1.f, -op.wexp));
        H3D_CHECK_LAUNCH("stem_x3_kernel");
        return H3D_OK;
    }
    if (h3d_note_kernel("stem_kernel<%s>", op.dtype == H3D_BF16 ? "unsigned short" : "float")) return H3D_OK;
    if (op.dtype == H3D_BF16)
        hipLaunchKernelGGL(stem_kernel<bf16_t>, grid, dim3(256), 0, st, (const float *)op.in, (const float *)op.w,
                           op.bias, (bf16_t *)op.out, op.B, op.H, op.W, op.out_cs, tx, ty);
    else if (op.dtype == H3D_F32)
        hipLaunchKernelGGL(stem_kernel<float>, grid, dim3(256), 0, st, (const float *)op.in, (const float *)op.w,
                           op.bias, (float *)op.out, op.B, op.H, op.W, op.out_cs, tx, ty);
    else
        H3D_FAIL(H3D_ERR_DTYPE, "stem: dtype %d", op.dtype);
    H3D_CHECK_LAUNCH("stem_kernel");
    return H3D_OK;
}

// =================================================================================================
// Elementwise NHWC kernels (HBM-bound; 16 B per lane)
template <typename T> struct Vec16 { static constexpr int N = 16 / sizeof(T); };

// 2x2/2 max pool (floor semantics of nn.MaxPool2d(2, stride=2), model.py:201)
template <typename T>
__global__ void maxpool_kernel(const T *__restrict__ in, T *__restrict__ out, int B, int H, int W, int C,
                               int in_cs, int Ho, int Wo, int out_cs)
{
    constexpr int N = Vec16<T>::N;
    const int vpc = C / N;
    const size_t total = (size_t)B * Ho * Wo * vpc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int v = i % vpc;
        const size_t p = i / vpc;
        const int ox = p % Wo;
        const size_t q = p / Wo;
        const int oy = q % Ho, b = q / Ho;
        float m[N], x[N];
        const T *base = in + (((size_t)b * H + oy * 2) * W + ox * 2) * in_cs + v * N;
        unpack16<T>(*reinterpret_cast<const u32x4 *>(base), m);
        unpack16<T>(*reinterpret_cast<const u32x4 *>(base + in_cs), x);
#pragma unroll
        for (int k = 0; k < N; ++k) m[k] = fmaxf(m[k], x[k]);
        unpack16<T>(*reinterpret_cast<const u32x4 *>(base + (size_t)W * in_cs), x);
#pragma unroll
        for (int k = 0; k < N; ++k) m[k] = fmaxf(m[k], x[k]);
        unpack16<T>(*reinterpret_cast<const u32x4 *>(base + (size_t)W * in_cs + in_cs), x);
#pragma unroll
        for (int k = 0; k < N; ++k) m[k] = fmaxf(m[k], x[k]);
        *reinterpret_cast<u32x4 *>(out + p * out_cs + v * N) = pack16<T>(m);
    }
}

// Depthwise ConvTranspose2d(C, C, k=2f, stride=f, padding=f/2, groups=C, bias=False) + skip add
// (IDAUp.forward: layers[i] = up(proj(layers[i])); node(layers[i] + layers[i-1]), model.py:384-390).
// out[oy][ox][c] = skip + sum over (i,j) with (oy + p - i) % f == 0: w[c][i][j] * in[(oy+p-i)/f][(ox+p-j)/f][c]
// -> exactly two i and two j per output pixel.  w: fp32 [k*k][C] (tap-major so channel vectors load).
// Workgroup = UP_ROWS output rows x a segment of 256/vpc pixels (vpc = 16-byte channel vectors per pixel); the
// k*k x C tap weights are staged in LDS once per workgroup, so the inner loop has no global weight loads,
// no integer division, and 5 independent 16-byte loads per thread in flight (4 taps + skip).
constexpr int UP_ROWS = 8;
template <typename T, bool F16OUT = false, bool WLDS = true>   // F16OUT: bf16 inputs, fp16 output (H3D_OUT_NHWC_F16)
__global__ __launch_bounds__(256) void upadd_kernel(const T *__restrict__ in, const T *__restrict__ skip,
                                                    const float *__restrict__ w, T *__restrict__ out, int B, int H, int W,
                                                    int C, int in_cs, int skip_cs, int Ho, int Wo, int out_cs, int f)
{
    constexpr int N = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) float s_w[];     // [k*k][C + 4]
    const int vpc = C / N, k = 2 * f, p = f / 2;
    // rows of C + 4 floats: neighbouring pixels of a wave read neighbouring taps (kj = (ox + p) % f + ...), and with rows
    // of C floats (a multiple of 256 B) the same channel vector of two taps sits on the same banks -- the SQ counters
    // showed 75 % of this kernel's LDS cycles as bank conflicts and the waves LDS-issue-stalled 41 % of the time; the 16-byte
    // skew puts tap t + 1 on the slots tap t leaves free
    const int wrow = WLDS ? C + 4 : C;
    if constexpr (WLDS) {   // a compile-time choice: a run-time pointer select would turn the ds_reads into flat loads
        const int c4 = C / 4;
        for (int i = threadIdx.x; i < k * k * c4; i += 256) {
            const int t = i / c4, j = i - t * c4;
            *reinterpret_cast<f32x4 *>(s_w + t * wrow + 4 * j) = *reinterpret_cast<const f32x4 *>(w + 4 * i);
        }
        __syncthreads();
    }
    const int ppb = 256 / vpc;                                       // pixels per workgroup row segment
    const int v = threadIdx.x % vpc, pxl = threadIdx.x / vpc;
    const int ox = blockIdx.x * ppb + pxl;
    const int rows_per_img = (Ho + UP_ROWS - 1) / UP_ROWS;
    const int b = blockIdx.y / rows_per_img, oy0 = (blockIdx.y - b * rows_per_img) * UP_ROWS;
    if (pxl >= ppb || ox >= Wo) return;
    // the two contributing input columns and their kernel columns are fixed for this thread
    const int rx = (ox + p) % f;
    int ixs[2], kjs[2];
    bool xok[2];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
        kjs[c2] = rx + c2 * f;
        const int num = ox + p - kjs[c2];
        ixs[c2] = num / f;                       // exact when num >= 0 (multiple of f)
        xok[c2] = num >= 0 && ixs[c2] < W;
    }
    for (int oy = oy0; oy < min(oy0 + UP_ROWS, Ho); ++oy) {
        const size_t pix = ((size_t)b * Ho + oy) * Wo + ox;
        const int ry = (oy + p) % f;
        u32x4 xv[4], sv;
        bool ok[4];
        int tap[4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int ki = ry + a * f;
            const int num = oy + p - ki;
            const int iy = num / f;
            const bool yok = num >= 0 && iy < H;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const int q = a * 2 + c2;
                ok[q] = yok && xok[c2];
                tap[q] = ki * k + kjs[c2];
                xv[q] = u32x4{0u, 0u, 0u, 0u};
                if (ok[q]) xv[q] = *reinterpret_cast<const u32x4 *>(in + (((size_t)b * H + iy) * W + ixs[c2]) * in_cs + v * N);
            }
        }
        sv = *reinterpret_cast<const u32x4 *>(skip + pix * skip_cs + v * N);
        float acc[N], x[N];
#pragma unroll
        for (int n = 0; n < N; ++n) acc[n] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!ok[q]) continue;                // same association order as before: taps (a, c2) ascending
            unpack16<T>(xv[q], x);
            const float *wp;
            if constexpr (WLDS) wp = s_w + tap[q] * wrow + v * N;
            else wp = w + tap[q] * C + v * N;
#pragma unroll
            for (int n = 0; n < N; ++n) acc[n] = fmaf(wp[n], x[n], acc[n]);
        }
        unpack16<T>(sv, x);
#pragma unroll
        for (int n = 0; n < N; ++n) acc[n] += x[n];
        if constexpr (F16OUT) *reinterpret_cast<u32x4 *>(out + pix * out_cs + v * N) = pack16_f16(acc);
        else *reinterpret_cast<u32x4 *>(out + pix * out_cs + v * N) = pack16<T>(acc);
    }
}

template <typename T>
__global__ void copy_kernel(const T *__restrict__ in, T *__restrict__ out, size_t npix, int C, int in_cs, int out_cs)
{
    constexpr int N = Vec16<T>::N;
    const int vpc = C / N;
    const size_t total = npix * vpc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int v = i % vpc;
        const size_t p = i / vpc;
        *reinterpret_cast<u32x4 *>(out + p * out_cs + v * N) = *reinterpret_cast<const u32x4 *>(in + p * in_cs + v * N);
    }
}

static inline int ew_grid(size_t total) { size_t g = (total + 255) / 256; return (int)(g > 2048 * 8 ? 2048 * 8 : (g ? g : 1)); }

int h3d_launch_elementwise(const h3d_op &op, hipStream_t st)
{
    if (!op.in || !op.out) H3D_FAIL(H3D_ERR_ARG, "elementwise: null pointer");
    const int es = h3d_dtype_bytes(op.dtype);
    if (!es) H3D_FAIL(H3D_ERR_DTYPE, "elementwise: dtype %d", op.dtype);
    const bool f16 = op.dtype == H3D_F16;
    const int n = 16 / es;
    if (op.Cin % n || op.in_cs % n || op.out_cs % n || op.Cin != op.Cout)
        H3D_FAIL(H3D_ERR_SHAPE, "elementwise: channels %d/%d strides %d/%d must be multiples of %d", op.Cin, op.Cout,
                 op.in_cs, op.out_cs, n);
    const size_t total = (size_t)op.B * op.Ho * op.Wo * (op.Cin / n);
    dim3 grid(ew_grid(total)), blk(256);
    const bool f16out = op.kind == H3D_OP_UPADD && op.out_mode == H3D_OUT_NHWC_F16;
    if (f16out && op.dtype != H3D_BF16) H3D_FAIL(H3D_ERR_DTYPE, "upadd: fp16 output is an option of bf16 plans (fp16 plans write fp16 anyway)");
    // tap table in LDS only while staging it is cheap next to the workgroup's 8 rows x (256 / vectors per pixel) pixels.
    // With the skewed rows (no bank conflicts, see upadd_kernel), same box, batch 64: 4 KiB (64 ch, f = 2) 0.070 -> 0.063 ms;
    // 8 KiB (128 ch) 0.057 -> 0.043; 16 KiB with 64 channels (f = 4) 0.102 -> 0.075; 16 KiB with 256 channels (8-pixel
    // segments) 0.029 -> 0.041: stays in global memory.  Tuning override: reserved 1 = never, 2 = whenever it fits 64 KiB
    const size_t up_wbytes = (size_t)op.ksize * op.ksize * op.Cin * sizeof(float);
    const bool up_wlds = op.reserved == 1 ? false : op.reserved == 2 ? up_wbytes <= 64 * 1024 : (up_wbytes <= 8192 || (up_wbytes <= 16384 && op.Cin <= 64));
    // (measured and dropped in round 3: f = 2 tap weights in registers, rows in pairs with ten loads in flight -- 0.341 vs 0.325 ms
    //  over the six f = 2 launches of the batch-64 plan: the kernel sits at the 4.4-4.9 TB/s these mixed read / write streams reach)
    if (h3d_note_kernel("%s<%s%s%s>", op.kind == H3D_OP_MAXPOOL ? "maxpool_kernel" : op.kind == H3D_OP_UPADD ? "upadd_kernel" : "copy_kernel",
                        f16 ? "f16_t" : es == 2 ? "unsigned short" : "float", op.kind == H3D_OP_UPADD ? (f16out ? ", true" : ", false") : "",
                        op.kind == H3D_OP_UPADD ? (up_wlds ? ", true" : ", false") : ""))
        return H3D_OK;
    if (op.kind == H3D_OP_MAXPOOL) {
        if (op.Ho != op.H / 2 || op.Wo != op.W / 2) H3D_FAIL(H3D_ERR_SHAPE, "maxpool: output must be floor(H/2) x floor(W/2)");
        if (f16)
            hipLaunchKernelGGL(maxpool_kernel<f16_t>, grid, blk, 0, st, (const f16_t *)op.in, (f16_t *)op.out, op.B, op.H,
                               op.W, op.Cin, op.in_cs, op.Ho, op.Wo, op.out_cs);
        else if (es == 2)
            hipLaunchKernelGGL(maxpool_kernel<bf16_t>, grid, blk, 0, st, (const bf16_t *)op.in, (bf16_t *)op.out, op.B, op.H,
                               op.W, op.Cin, op.in_cs, op.Ho, op.Wo, op.out_cs);
        else
            hipLaunchKernelGGL(maxpool_kernel<float>, grid, blk, 0, st, (const float *)op.in, (float *)op.out, op.B, op.H,
                               op.W, op.Cin, op.in_cs, op.Ho, op.Wo, op.out_cs);
        H3D_CHECK_LAUNCH("maxpool_kernel");
    } else if (op.kind == H3D_OP_UPADD) {
        const int f = op.stride;
        if (!op.in2 || !op.w) H3D_FAIL(H3D_ERR_ARG, "upadd: null pointer");
        if (op.ksize != 2 * f || op.Ho != op.H * f || op.Wo != op.W * f || op.in2_cs % n)
            H3D_FAIL(H3D_ERR_SHAPE, "upadd: expects k=2f, out = f*in (k=%d f=%d)", op.ksize, f);
        const int vpc = op.Cin / n;
        if (vpc > 256) H3D_FAIL(H3D_ERR_UNSUPPORTED, "upadd: C=%d", op.Cin);
        const size_t wfl = (size_t)op.ksize * op.ksize * op.Cin;
        // tap table in LDS when it fits the default 64 KiB (tuning override: reserved 1 = never, 2 = always)
        const bool wlds = up_wlds;
        const size_t lds = wlds ? (wfl + 4 * (size_t)op.ksize * op.ksize) * sizeof(float) : 0;     // rows of C + 4 floats
        const dim3 ugrid(cdiv(op.Wo, 256 / vpc), op.B * cdiv(op.Ho, UP_ROWS));
#define H3D_UPADD_LAUNCH(K, TT)                                                                                              \
    do {                                                                                                                      \
        hipLaunchKernelGGL(K, ugrid, blk, lds, st, (const TT *)op.in, (const TT *)op.in2, (const float *)op.w, (TT *)op.out, \
                           op.B, op.H, op.W, op.Cin, op.in_cs, op.in2_cs, op.Ho, op.Wo, op.out_cs, f);                        \
    } while (0)
        if (wlds) {
            if (f16out) H3D_UPADD_LAUNCH((upadd_kernel<bf16_t, true, true>), bf16_t);
            else if (f16) H3D_UPADD_LAUNCH((upadd_kernel<f16_t, false, true>), f16_t);
            else if (es == 2) H3D_UPADD_LAUNCH((upadd_kernel<bf16_t, false, true>), bf16_t);
            else H3D_UPADD_LAUNCH((upadd_kernel<float, false, true>), float);
        } else {
            if (f16out) H3D_UPADD_LAUNCH((upadd_kernel<bf16_t, true, false>), bf16_t);
            else if (f16) H3D_UPADD_LAUNCH((upadd_kernel<f16_t, false, false>), f16_t);
            else if (es == 2) H3D_UPADD_LAUNCH((upadd_kernel<bf16_t, false, false>), bf16_t);
            else H3D_UPADD_LAUNCH((upadd_kernel<float, false, false>), float);
        }
#undef H3D_UPADD_LAUNCH
        H3D_CHECK_LAUNCH("upadd_kernel");
    } else {
        if (op.Ho != op.H || op.Wo != op.W) H3D_FAIL(H3D_ERR_SHAPE, "copy: shape");
        const size_t npix = (size_t)op.B * op.H * op.W;
        if (f16)
            hipLaunchKernelGGL(copy_kernel<f16_t>, grid, blk, 0, st, (const f16_t *)op.in, (f16_t *)op.out, npix, op.Cin,
                               op.in_cs, op.out_cs);
        else if (es == 2)
            hipLaunchKernelGGL(copy_kernel<bf16_t>, grid, blk, 0, st, (const bf16_t *)op.in, (bf16_t *)op.out, npix, op.Cin,
                               op.in_cs, op.out_cs);
        else
            hipLaunchKernelGGL(copy_kernel<float>, grid, blk, 0, st, (const float *)op.in, (float *)op.out, npix, op.Cin,
                               op.in_cs, op.out_cs);
        H3D_CHECK_LAUNCH("copy_kernel");
    }
    return H3D_OK;
}

// =================================================================================================
// Layout converters at the host boundary (reference tensors are NCHW fp32)
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float *__restrict__ src, T *__restrict__ dst, int B, int C, int HW, int dst_cs)
{
    // tile 32 pixels x 32 channels through LDS so both sides are coalesced
    __shared__ float tile[32][33];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 8 rows per pass
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, p = p0 + tx;
        tile[k][tx] = (c < C && p < HW) ? src[((size_t)b * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int p = p0 + k, c = c0 + tx;
        if (p < HW && c < C) dst[((size_t)b * HW + p) * dst_cs + c] = ET<T>::from_f32(tile[tx][k]);
    }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T *__restrict__ src, float *__restrict__ dst, int B, int C, int HW, int src_cs)
{
    __shared__ float tile[32][33];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int p = p0 + k, c = c0 + tx;
        tile[k][tx] = (p < HW && c < C) ? ET<T>::to_f32(src[((size_t)b * HW + p) * src_cs + c]) : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, p = p0 + tx;
        if (c < C && p < HW) dst[((size_t)b * C + c) * HW + p] = tile[tx][k];
    }
}

extern "C" int h3d_nchw_f32_to_nhwc(const float *src, void *dst, int dtype, int B, int C, int H, int W, int dst_cs,
                                    void *stream)
{
    if (!src || !dst) H3D_FAIL(H3D_ERR_ARG, "nchw_to_nhwc: null pointer");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || dst_cs < C) H3D_FAIL(H3D_ERR_SHAPE, "nchw_to_nhwc: bad shape");
    dim3 grid(cdiv(H * W, 32), cdiv(C, 32), B);
    if (dtype == H3D_BF16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, src, (bf16_t *)dst, B, C,
                           H * W, dst_cs);
    else if (dtype == H3D_F16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)stream, src, (f16_t *)dst, B, C,
                           H * W, dst_cs);
    else if (dtype == H3D_F32)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, src, (float *)dst, B, C,
                           H * W, dst_cs);
    else
        H3D_FAIL(H3D_ERR_DTYPE, "nchw_to_nhwc: dtype %d", dtype);
    H3D_CHECK_LAUNCH("nchw_to_nhwc_kernel");
    return H3D_OK;
}

extern "C" int h3d_nhwc_to_nchw_f32(const void *src, int dtype, float *dst, int B, int C, int H, int W, int src_cs,
                                    void *stream)
{
    if (!src || !dst) H3D_FAIL(H3D_ERR_ARG, "nhwc_to_nchw: null pointer");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || src_cs < C) H3D_FAIL(H3D_ERR_SHAPE, "nhwc_to_nchw: bad shape");
    dim3 grid(cdiv(H * W, 32), cdiv(C, 32), B);
    if (dtype == H3D_BF16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t *)src, dst,
                           B, C, H * W, src_cs);
    else if (dtype == H3D_F16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const f16_t *)src, dst,
                           B, C, H * W, src_cs);
    else if (dtype == H3D_F32)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)src, dst, B,
                           C, H * W, src_cs);
    else
        H3D_FAIL(H3D_ERR_DTYPE, "nhwc_to_nchw: dtype %d", dtype);
    H3D_CHECK_LAUNCH("nhwc_to_nchw_kernel");
    return H3D_OK;
}
