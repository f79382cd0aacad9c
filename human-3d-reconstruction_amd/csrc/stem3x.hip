// base_layer + level0 + level1 of DLA-34 in one kernel for the f16x3 plans (round 5):
//   7x7 3->16 (BN, ReLU) -> 3x3 16->16 (BN, ReLU) -> 3x3 stride 2 16->32 (BN, ReLU)       (model.py:231-249)
// The f16x3 twin of csrc/stem3.hip: fp32 images in, fp32 level1 map out, every fp32 product as three fp16 MFMAs on split operands
// (csrc/common.h ET<x3_t>).  The two full-resolution 16-channel maps -- 1.07 GB each in fp32 at batch 64 and 512x512, written and read
// back by the three separate launches (stem_x3_kernel 0.39 ms + 16 -> 16 0.77 + 16 -> 32 stride 2 0.43) -- stay in LDS as finished
// (hi | lo) operand fragments: an intermediate is split ONCE, by the lane that produced it, exactly as the consumer's staging would
// split the stored fp32 value (same numbers: the fused and the unfused plan differ by fp32 accumulation order only).
//
// One workgroup (8 waves) = 8 x 16 level1 pixels <- 17 x 33 level0 pixels <- 19 x 35 stem pixels <- 25 x 41 image pixels.
//   P0  image patch -> LDS, two tiles of interleaved pixels (c0, c1, c2, 0) in fp16: the hi terms and the lo terms
//   P1  stem on v_mfma_f32_16x16x32_f16: per tap row one K = 32 step (7 taps x 4 interleaved channels + 4 zero weights), three MFMAs
//       (lo.hi, hi.lo, hi.hi); 2^-e0 scale + bias + ReLU, ZERO outside the image (level0's padding) -> split -> tile S (80 B per pixel:
//       [hi c0-7 | lo c0-7 | hi c8-15 | lo c8-15] + 16 B pad)
//   P2  level0 on the same instruction: K = 32 is a PAIR of taps x 16 channels, 5 K steps (the 10th tap has zero weights) -> tile L0
//   P3  level1 (stride 2, 32 channels) on v_mfma_f32_32x32x16_f16: one K step per tap; waves 0-3 take 32 pixels each
// All filters live in registers (7 + 5 + 9 fragment pairs per lane: 168 VGPRs); 115 KB of LDS: one workgroup per CU.
#include "common.h"

struct Stem3xArgs {
    const float *img;      // [B,3,H,W] fp32
    const char *w;         // float32-TYPED split filters: [16][7][32] stem (k = dx*4 + c) | [5][16][32] level0 (k = tapsel*16 + c) | [32][9][16] level1
    const float *bias;     // [16 | 16 | 32 | 2^-e0, 2^-e1, 2^-e2, 0]: the three banks were packed times 2^e (engine.PackedWeights.stem3_x3)
    float *out;            // [B,Ho,Wo,out_cs] level1, fp32
    int B, H, W, Ho, Wo, out_cs;
    int tiles_x, tiles_y, tpb;
};

constexpr int S3X_IW = 44, S3X_IH = 25;                  // image patch: 41 columns + the 8-pixel K run of the last stem pixel
constexpr int S3X_SH = 19, S3X_SW = 35, S3X_LH = 17, S3X_LW = 33;
constexpr int S3X_PX = 80;                               // bytes per pixel of S and L0
constexpr int S3X_LDS_I = S3X_IH * S3X_IW * 8;           // one of the two image tiles
constexpr int S3X_LDS_S = (S3X_SH * S3X_SW + 1) * S3X_PX, S3X_LDS_L = (S3X_LH * S3X_LW + 1) * S3X_PX;
constexpr int S3X_LDS = 2 * S3X_LDS_I + S3X_LDS_S + S3X_LDS_L + 128;      // + level1's 32 biases (read in P3 only: 16 registers less to hold)
constexpr int S3X_NPF = (S3X_IH * S3X_IW + 511) / 512;                   // image-patch items per thread (3)

typedef __attribute__((ext_vector_type(4))) float f32x4_s3x;

__device__ __forceinline__ f32x4_s3x s3x_mma16(const u32x4 &ah, const u32x4 &al, const u32x4 &bh, const u32x4 &bl, f32x4_s3x acc)
{
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, al), __builtin_bit_cast(f16x8_t, bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, ah), __builtin_bit_cast(f16x8_t, bl), acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, ah), __builtin_bit_cast(f16x8_t, bh), acc, 0, 0, 0);
}

__global__ __launch_bounds__(512, 2) void stem3x_kernel(Stem3xArgs a)
{
    __shared__ __attribute__((aligned(16))) char smem[S3X_LDS];
    uint2 *s_ih = reinterpret_cast<uint2 *>(smem), *s_il = reinterpret_cast<uint2 *>(smem + S3X_LDS_I);
    char *s_s = smem + 2 * S3X_LDS_I, *s_l = s_s + S3X_LDS_S;
    float *s_b2 = reinterpret_cast<float *>(s_l + S3X_LDS_L);

    const int tid = threadIdx.x, l = tid & 63, p = l & 15, q = l >> 4, r = l & 31, h = l >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = a.tiles_x * a.tiles_y, ntile = a.B * tiles;
    const size_t plane = (size_t)a.H * a.W;

    // ---- filters (registers) ---------------------------------------------------------------------------------------------
    u32x4 f0h[7], f0l[7], f1h[5], f1l[5], f2h[9], f2l[9];
#pragma unroll
    for (int dy = 0; dy < 7; ++dy) {       // 16x16x32 A operand: row = out channel p, K group q (8 of the 32 k)
        const char *g = a.w + ((size_t)(p * 7 + dy) * 32 + 8 * q) * 4;
        f0h[dy] = *reinterpret_cast<const u32x4 *>(g);
        f0l[dy] = *reinterpret_cast<const u32x4 *>(g + 16);
    }
    const char *w1 = a.w + 16 * 7 * 32 * 4;
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const char *g = w1 + ((size_t)(ks * 16 + p) * 32 + 8 * q) * 4;
        f1h[ks] = *reinterpret_cast<const u32x4 *>(g);
        f1l[ks] = *reinterpret_cast<const u32x4 *>(g + 16);
    }
    const char *w2 = w1 + 5 * 16 * 32 * 4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {    // 32x32x16 A operand: row = out channel r, K = channels 8h .. 8h+7
        const char *g = w2 + ((size_t)(r * 9 + tap) * 16 + 8 * h) * 4;
        f2h[tap] = *reinterpret_cast<const u32x4 *>(g);
        f2l[tap] = *reinterpret_cast<const u32x4 *>(g + 16);
    }
    float b0[4], b1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { b0[i] = a.bias[4 * q + i]; b1[i] = a.bias[16 + 4 * q + i]; }
    if (tid < 32) s_b2[tid] = a.bias[32 + tid];
    const float s0 = a.bias[64], s1 = a.bias[65], s2 = a.bias[66];

    // image patch of tile `ti` -> registers (three fp32 planes of up to S3X_NPF pixels per thread): issued while the previous tile
    // computes, so that a tile does not start with an exposed global round trip (the first version did: 1.15 ms for the launch)
    float pf[S3X_NPF][3];
    auto load_patch = [&](int ti) {
        const int pb = ti / tiles, pt = ti - pb * tiles;
        const int pty = pt / a.tiles_x, ptx = pt - pty * a.tiles_x;
        const int piy0 = 2 * (pty * 8) - 5, pix0 = 2 * (ptx * 16) - 5;
        const float *im = a.img + (size_t)pb * 3 * plane;
#pragma unroll
        for (int j = 0; j < S3X_NPF; ++j) {
            const int i = tid + j * 512;
            const int iy = i / S3X_IW, ix = i - iy * S3X_IW;
            const int gy = piy0 + iy, gx = pix0 + ix;
            pf[j][0] = pf[j][1] = pf[j][2] = 0.f;
            if (ti < ntile && i < S3X_IH * S3X_IW && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && ix < 41) {
                const size_t o = (size_t)gy * a.W + gx;
                pf[j][0] = im[o]; pf[j][1] = im[plane + o]; pf[j][2] = im[2 * plane + o];
            }
        }
    };
    const int t_first = blockIdx.x * a.tpb, t_end = min((int)(blockIdx.x + 1) * a.tpb, ntile);
    load_patch(t_first);
    for (int ti = t_first; ti < t_end; ++ti) {
        const int b = ti / tiles, t = ti - b * tiles;
        const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
        const int oy0 = ty * 8, ox0 = tx * 16;                    // level1 tile origin
        const int ly0 = 2 * oy0 - 1, lx0 = 2 * ox0 - 1;           // level0 region origin (full resolution)
        const int sy0 = ly0 - 1, sx0 = lx0 - 1;                   // stem region origin (the image patch starts 3 pixels further out)
        // ---- P0: image patch (prefetched), split -> LDS -------------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < S3X_NPF; ++j) {
            const int i = tid + j * 512;
            if (i < S3X_IH * S3X_IW) {
                u32x2 hi, lo;
                x3_split4(u32x4{__float_as_uint(pf[j][0]), __float_as_uint(pf[j][1]), __float_as_uint(pf[j][2]), 0u}, hi, lo);
                s_ih[i] = uint2{hi[0], hi[1]};
                s_il[i] = uint2{lo[0], lo[1]};
            }
        }
        __syncthreads();
        load_patch(ti + 1);                                       // (ti + 1 == ntile or the next workgroup's tile: loads nothing / is not used)
        // ---- P1: stem -> S ---------------------------------------------------------------------------------------------------
        for (int g = wv; g < (S3X_SH * S3X_SW + 15) / 16; g += 8) {
            const int f = min(16 * g + p, S3X_SH * S3X_SW - 1);
            const int sy = f / S3X_SW, sx = f - sy * S3X_SW;
            f32x4_s3x acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 7; ++dy) {
                const int o = (sy + dy) * S3X_IW + sx + 2 * q;      // K elements 8q .. 8q+7 of the run that starts at pixel (sy + dy, sx): pixels +2q, +2q+1
                const uint2 h0 = s_ih[o], h1 = s_ih[o + 1], l0 = s_il[o], l1 = s_il[o + 1];
                acc = s3x_mma16(f0h[dy], f0l[dy], u32x4{h0.x, h0.y, h1.x, h1.y}, u32x4{l0.x, l0.y, l1.x, l1.y}, acc);
            }
            const int gy = sy0 + sy, gx = sx0 + sx;
            const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            u32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = __float_as_uint(in ? fmaxf(fmaf(acc[i], s0, b0[i]), 0.f) : 0.f);
            if (16 * g + p < S3X_SH * S3X_SW) x3_store4(s_s + f * S3X_PX + (q >> 1) * 32, q & 1, v);      // channels 4q .. 4q+3 of stem pixel f
        }
        __syncthreads();
        // ---- P2: level0 -> L0 ------------------------------------------------------------------------------------------------
        for (int g = wv; g < (S3X_LH * S3X_LW + 15) / 16; g += 8) {
            const int f = min(16 * g + p, S3X_LH * S3X_LW - 1);
            const int ly = f / S3X_LW, lx = f - ly * S3X_LW;
            f32x4_s3x acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) {
                const int tap = min(2 * ks + (q >> 1), 8);           // (the 10th tap of the last pair has zero filters: any finite operand will do)
                const int dy = tap / 3, dx = tap - 3 * dy;
                const char *src = s_s + ((ly + dy) * S3X_SW + lx + dx) * S3X_PX + (q & 1) * 32;
                acc = s3x_mma16(f1h[ks], f1l[ks], *reinterpret_cast<const u32x4 *>(src), *reinterpret_cast<const u32x4 *>(src + 16), acc);
            }
            const int gy = ly0 + ly, gx = lx0 + lx;
            const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            u32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = __float_as_uint(in ? fmaxf(fmaf(acc[i], s1, b1[i]), 0.f) : 0.f);
            if (16 * g + p < S3X_LH * S3X_LW) x3_store4(s_l + f * S3X_PX + (q >> 1) * 32, q & 1, v);
        }
        __syncthreads();
        // ---- P3: level1 (stride 2) -> global ---------------------------------------------------------------------------------
        if (wv < 4) {
            const int py = 2 * wv + (r >> 4), px = r & 15;
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap - 3 * dy;
                const char *src = s_l + ((2 * py + dy) * S3X_LW + 2 * px + dx) * S3X_PX + h * 32;
                const u32x4 bh = *reinterpret_cast<const u32x4 *>(src), bl = *reinterpret_cast<const u32x4 *>(src + 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, f2l[tap]), __builtin_bit_cast(f16x8_t, bh), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, f2h[tap]), __builtin_bit_cast(f16x8_t, bl), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, f2h[tap]), __builtin_bit_cast(f16x8_t, bh), acc, 0, 0, 0);
            }
            const int oy = oy0 + py, ox = ox0 + px;
            if (oy < a.Ho && ox < a.Wo) {
                float *op = a.out + ((size_t)(b * a.Ho + oy) * a.Wo + ox) * a.out_cs + 4 * h;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {      // accumulator registers 4g .. 4g+3 = channels 8g + 4h .. + 3
                    const f32x4 bb = *reinterpret_cast<const f32x4 *>(s_b2 + 8 * g4 + 4 * h);
                    store4<float>(op + 8 * g4, fmaxf(fmaf(acc[4 * g4 + 0], s2, bb[0]), 0.f), fmaxf(fmaf(acc[4 * g4 + 1], s2, bb[1]), 0.f),
                                  fmaxf(fmaf(acc[4 * g4 + 2], s2, bb[2]), 0.f), fmaxf(fmaf(acc[4 * g4 + 3], s2, bb[3]), 0.f));
                }
            }
        }
        // (no barrier here: the next tile's P0 writes the image tiles, last read in P1; its P1 writes S, last read in P2 -- both behind
        //  barriers every wave has passed; L0 is rewritten in the next P2, behind the next tile's first two barriers)
    }
}

int h3d_launch_stem3x(const h3d_op &op, hipStream_t st)
{
    if (op.in2) H3D_FAIL(H3D_ERR_UNSUPPORTED, "stem3 (f16x3): the fused residual branch is an option of the 2-byte plans");
    Stem3xArgs a;
    a.img = (const float *)op.in; a.w = (const char *)op.w; a.bias = op.bias; a.out = (float *)op.out;
    a.B = op.B; a.H = op.H; a.W = op.W; a.Ho = op.Ho; a.Wo = op.Wo; a.out_cs = op.out_cs;
    a.tiles_x = cdiv(op.Wo, 16); a.tiles_y = cdiv(op.Ho, 8);
    const int ntiles = op.B * a.tiles_x * a.tiles_y;
    int tpb = 2;                                   // consecutive tiles per workgroup (the 21 filter fragment pairs are fetched once per workgroup)
    while (tpb < 64 && cdiv(ntiles, 2 * tpb) >= 256) tpb *= 2;
    a.tpb = tpb;
    if (h3d_note_kernel("stem3x_kernel")) return H3D_OK;
    hipLaunchKernelGGL(stem3x_kernel, dim3(cdiv(ntiles, tpb)), dim3(512), 0, st, a);
    H3D_CHECK_LAUNCH("stem3x_kernel");
    return H3D_OK;
}
