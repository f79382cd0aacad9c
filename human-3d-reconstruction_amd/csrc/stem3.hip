// base_layer + level0 + level1 of DLA-34 in one kernel (bf16 plans):
//   7x7 3->16 (BN, ReLU) -> 3x3 16->16 (BN, ReLU) -> 3x3 stride 2 16->32 (BN, ReLU)       (model.py:231-249)
// The two 16-channel full-resolution maps are the fattest tensors of the network (537 MB each at batch 64 and
// 512x512) and nothing else reads them (DLAUp starts at level 2), so they never leave the chip here: the kernel reads
// the fp32 NCHW images and writes the level1 map.
//
// One workgroup (8 waves) = 8 x 16 level1 pixels.  It needs the 17 x 33 level0 pixels under them, those need the
// 19 x 35 stem pixels around them, those the 25 x 41 image pixels: every stage recomputes its halo (1.3x / 1.1x).
//   P0  image patch -> LDS as interleaved pixels of 4 bf16 (c0, c1, c2, 0): a 7-tap row of the stem is a contiguous
//       K = 28 (+4 zero-weight) run
//   P1  stem on v_mfma_f32_16x16x32_bf16: one K step per tap row, 7 MFMAs per 16 pixels (as csrc/conv.hip
//       stem_mfma_kernel); bias + ReLU -> bf16 -> LDS tile S (pixel stride 48 B); pixels outside the image are
//       written as ZERO: they are level0's zero padding, not the stem of a padded image
//   P2  level0 on the same instruction: K = 32 is a PAIR of taps x 16 channels, 5 K steps (the 10th tap has zero
//       weights); bias + ReLU -> bf16 -> LDS tile L0, zero outside the image
//   P3  level1 (stride 2, 32 channels) on v_mfma_f32_32x32x16_bf16: one K step per tap, waves 0-3 take 32 pixels each
// All filters live in registers (7 + 5 + 9 fragments per lane).  76 KB of LDS: two workgroups per CU.
// Intermediates are rounded to bf16 exactly where the unfused kernels round them.
#include "common.h"
#include "epilogue.h"

typedef __attribute__((ext_vector_type(4))) float f32x4_s3;

struct Stem3Args {
    const float *img;      // [B,3,H,W] fp32
    const uint16_t *w;     // (bf16 | fp16) [16][7][32] stem (k = dx*4 + c) | [5][16][32] level0 (k = tapsel*16 + c) | [32][9][16] level1
    const float *bias;     // [16 | 16 | 32]
    void *out;             // [B,Ho,Wo,out_cs] level1
    int B, H, W, Ho, Wo, out_cs;
    int tiles_x, tiles_y;
    int dbg;               // ABLATE builds: stop after phase dbg (1..3)
    int tpb;               // consecutive tiles per workgroup
    // PROJ (round 4): level2's residual branch in the same launch -- project(max_pool2x2(level1)) (Tree.downsample + Tree.project,
    // model.py:200-207, 211-212: level2 is not a level_root, so nothing else reads the pooled map)
    const uint16_t *wproj;   // [64][32] T: 1x1 filters (BatchNorm folded)
    const float *bproj;      // [64]
    void *res_out;           // [B,Ho/2,Wo/2,res_cs] T
    int res_cs;
};

// consecutive tiles per workgroup (image patch prefetched one tile ahead; the 12 filter fragments are fetched once per workgroup): a
// launch-time choice (Stem3Args::tpb).  Round 4, same process (tools/ab_lib.py), batch 64 (32768 tiles on 512 workgroup slots): 4 / 8 /
// 16 / 32 / 64 tiles per workgroup -> 0.496 / 0.467 / 0.447 / 0.443 / 0.431 ms; batch 8 (4096 tiles): 4 / 8 / 16 -> 0.071 / 0.068 / 0.092
#ifndef S3K_TPB_FORCE
#define S3K_TPB_FORCE 0
#endif
constexpr int S3K_IW = 56, S3K_IH = 25;              // image patch (uint2 per pixel)
constexpr int S3K_ROWB = 1792, S3K_PXB = 48;         // S and L0 tiles: 48 B per pixel, rows 0 mod 256 B
constexpr int S3K_SH = 19, S3K_LH = 17, S3K_LW = 33;    // (the stem region is 19 x 35)
constexpr int S3K_LDS_I = S3K_IH * S3K_IW * 8, S3K_LDS_S = S3K_SH * S3K_ROWB, S3K_LDS_L = S3K_LH * S3K_ROWB;
constexpr int S3K_LDS = S3K_LDS_I + S3K_LDS_S + S3K_LDS_L + 4096;     // + slack: masked lanes of the last groups read past a row
// PROJ: the 4 KB slack above doubles as two buffers (tile parity) of 4 pooled level1 rows x 8 pixels x 32 channels (512 B per row):
// nothing is ever WRITTEN there by P1 / P2, and what their masked lanes read from it is unused
constexpr int S3K_POOL_OFF = S3K_LDS - 4096;

// bias + ReLU + bf16 rounding of one accumulator quad, zero outside the image -- branch free: the ReLU runs on the packed
// pairs (v_pk_max_i16 against 0: a negative bf16 is a negative int16, rounding never changes the sign, so
// round(relu(x)) == relu(round(x)) bit for bit) and the image test is a select.  (-2 % on the kernel: its waves are parked
// on the four barriers of a tile 55 % of the time, not short of vector issue slots.)
template <typename T>      // (the same holds for fp16: sign-magnitude, rounding keeps the sign)
__device__ __forceinline__ u32x2 stem3_epi(const f32x4_s3 &acc, const float (&b)[4], bool in)
{
    typedef short s16x2_s3 __attribute__((ext_vector_type(2)));
    uint32_t lo = EP<T>::pack2(acc[0] + b[0], acc[1] + b[1]), hi = EP<T>::pack2(acc[2] + b[2], acc[3] + b[3]);
    lo = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_s3, lo), s16x2_s3{0, 0}));
    hi = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_s3, hi), s16x2_s3{0, 0}));
    return u32x2{in ? lo : 0u, in ? hi : 0u};
}

template <typename T>
__device__ __forceinline__ f32x4_s3 stem3_mfma16(const u32x4 &fa, const u32x4 &fb, const f32x4_s3 &acc)
{
    if constexpr (std::is_same_v<T, f16_t>)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fa), __builtin_bit_cast(f16x8_t, fb), acc, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa), __builtin_bit_cast(bf16x8_t, fb), acc, 0, 0, 0);
}

template <typename T, bool PROJ = false>      // bf16_t | f16_t
__global__ __launch_bounds__(512, 4) void stem3_kernel(Stem3Args a)   // 4 waves per SIMD = two workgroups per CU: at most 128 VGPRs
{
    __shared__ __attribute__((aligned(16))) char smem[S3K_LDS];
    uint2 *s_i = reinterpret_cast<uint2 *>(smem);
    char *s_s = smem + S3K_LDS_I;
    char *s_l = s_s + S3K_LDS_S;

    const int tid = threadIdx.x, l = tid & 63, p = l & 15, q = l >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform for the compiler too: the group index arithmetic of P1 / P2 runs on the scalar unit
    const int tiles = a.tiles_x * a.tiles_y;
    const int ntile = a.B * tiles;
    const int t_first = blockIdx.x * a.tpb;
    const size_t plane = (size_t)a.H * a.W;

    // ---- P0 (software pipelined over the S3K_TPB tiles of this workgroup): the image patch of tile i+1 is fetched into
    //      registers while tile i computes -- with one patch per workgroup every workgroup paid a full memory round
    //      trip before its first MFMA, 40 % of the kernel (tools/ablate_stem3.py).
    //      The tile origin is a multiple of 32 pixels, so the patch starts 5 pixels left of a 16-byte boundary: rows are
    //      fetched as aligned float4 (columns ix0 - 3 .. ix0 + 56, 15 vectors per row and plane; W % 4 == 0 makes every
    //      vector entirely inside or outside the image), one (row, 4-pixel group) item per thread.
    constexpr int VPRW = 15, NITEM = S3K_IH * VPRW;       // 375 items
    f32x4 pc0, pc1, pc2;
    auto load_patch = [&](int ti) {
        pc0 = f32x4{0.f, 0.f, 0.f, 0.f}; pc1 = pc0; pc2 = pc0;
        if (tid < NITEM && ti < ntile) {
            const int pb = ti / tiles, pt = ti - pb * tiles;
            const int pty = pt / a.tiles_x, ptx = pt - pty * a.tiles_x;
            const int iy = tid / VPRW, j = tid - iy * VPRW;
            const int gy = 16 * pty - 5 + iy, gx = 32 * ptx - 5 - 3 + 4 * j;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                const float *im = a.img + (size_t)pb * 3 * plane + (size_t)gy * a.W + gx;
                pc0 = *reinterpret_cast<const f32x4 *>(im);
                pc1 = *reinterpret_cast<const f32x4 *>(im + plane);
                pc2 = *reinterpret_cast<const f32x4 *>(im + 2 * plane);
            }
        }
    };
    auto store_patch = [&]() {
        if (tid < NITEM) {
            const int iy = tid / VPRW, j = tid - iy * VPRW;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ix = 4 * j - 3 + e;
                if (ix >= 0 && ix < S3K_IW) s_i[iy * S3K_IW + ix] = uint2{EP<T>::pack2(pc0[e], pc1[e]), EP<T>::pack2(pc2[e], 0.f)};
            }
        }
    };
    load_patch(t_first);
    store_patch();
    // filters of P1 / P2 (16x16x32 A operand: row = p, K = 8q..8q+7)
    u32x4 fa0[7], fa1[5];
#pragma unroll
    for (int dy = 0; dy < 7; ++dy) fa0[dy] = *reinterpret_cast<const u32x4 *>(a.w + (p * 7 + dy) * 32 + 8 * q);
    const uint16_t *w1 = a.w + 16 * 7 * 32;
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) fa1[ks] = *reinterpret_cast<const u32x4 *>(w1 + (ks * 16 + p) * 32 + 8 * q);
    float b0[4], b1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { b0[i] = a.bias[4 * q + i]; b1[i] = a.bias[16 + 4 * q + i]; }

    // PROJ: the 1x1 `project` conv on the 4 x 8 pooled pixels of tile `tj` (written by waves 0-3 in that tile's P3, at least one
    // workgroup barrier ago) by waves 4 and 5 (idle in P3): wave 4 + m computes output channels 32 m .. 32 m + 31 of all 32 pixels
    // (one full N tile), K = 32 as two accumulating 32x32x16 MFMAs -- the instruction and K order of csrc/conv.hip's 1x1 kernel:
    // bit-identical to it.  Its two filter fragments stay in registers for the whole launch (fetched per tile they put two
    // dependent memory round trips into every tile: 0.71 ms instead of 0.48)
    [[maybe_unused]] u32x4 fpw[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    if constexpr (PROJ) {
        if (wv == 4 || wv == 5) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fpw[ks] = *reinterpret_cast<const u32x4 *>(a.wproj + ((wv - 4) * 32 + (l & 31)) * 32 + 16 * ks + 8 * (l >> 5));
        }
    }
    [[maybe_unused]] auto proj_tile = [&](int tj) {
        if constexpr (PROJ) {
            const int r = l & 31, h = l >> 5, m = wv - 4;
            const int pb = tj / tiles, pt = tj - pb * tiles;
            const int pty = pt / a.tiles_x, ptx = pt - pty * a.tiles_x;
            const char *lp = smem + S3K_POOL_OFF + ((tj & 1) * 4 + (r >> 3)) * 512 + (r & 7) * 64 + 16 * h;      // pooled pixel (row r / 8, column r % 8)
            f32x16 pacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) pacc[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                typename ET<T>::frag fa_, fb_;
                fb_.v = *reinterpret_cast<const u32x4 *>(lp + 32 * ks);
                fa_.v = fpw[ks];
                ET<T>::mma(pacc, fa_, fb_);
            }
            const int qy = pty * 4 + (r >> 3), qx = ptx * 8 + (r & 7);
            const int Hp = a.Ho >> 1, Wp = a.Wo >> 1;
            if (qy < Hp && qx < Wp) {
                T *rp = reinterpret_cast<T *>(a.res_out) + (((size_t)pb * Hp + qy) * Wp + qx) * a.res_cs + m * 32 + 4 * h;
                const float ninf = -__builtin_inff();
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *reinterpret_cast<const float4 *>(a.bproj + m * 32 + 8 * g + 4 * h);
                    const u32x2 pk = {EP<T>::pack2(EP<T>::clamp(pacc[4 * g + 0] + bv.x, ninf), EP<T>::clamp(pacc[4 * g + 1] + bv.y, ninf)),
                                      EP<T>::pack2(EP<T>::clamp(pacc[4 * g + 2] + bv.z, ninf), EP<T>::clamp(pacc[4 * g + 3] + bv.w, ninf))};
                    *reinterpret_cast<u32x2 *>(rp + 8 * g) = pk;
                }
            }
        }
    };

    for (int ti = t_first; ti < min(t_first + a.tpb, ntile); ++ti) {
    const int b = ti / tiles;
    const int t = ti - b * tiles;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int oy1 = ty * 8, ox1 = tx * 16;                   // level1 tile origin
    const int ly0 = 2 * oy1 - 1, lx0 = 2 * ox1 - 1;          // level0 region origin (full-resolution coordinates)
    const int sy0 = ly0 - 1, sx0 = lx0 - 1;                  // stem region origin
    __syncthreads();                                         // this tile's patch is in LDS; the previous tile is done with S / L0
    if (H3D_DBG(a) == 1) return;
    if (ti + 1 < t_first + a.tpb) load_patch(ti + 1);        // (zeros past the last tile)

    // ---- P1: stem -> S.  The kernel is bound by the LDS array (SQ_LDS_IDX_ACTIVE 0.75 per CU cycle, profiles/r03_pmc_sq_summary):
    //      a flat pixel list (rounds 2-3: 42 groups of 16 pixels, one 1-KB patch read per MFMA) re-reads every patch row for each
    //      of the 7 tap rows.  Now a wave owns a 16-column STRIP of up to 5 output rows: each of the 11 patch rows under it is
    //      read ONCE and feeds the (up to) 5 accumulators whose tap row it is -- 11 reads for 35 MFMAs.  Waves 0-3 / 4-7 take
    //      columns 0-15 / 16-31, rows 5 (wv & 3) ..; the last three columns (57 pixels) stay a flat list on waves 4-7.
    //      Same MFMAs in the same order per output pixel (tap rows ascending): bit-identical. ------------------------------
    {
        const int r0 = 5 * (wv & 3), c0 = 16 * (wv >> 2);
        f32x4_s3 acc[5];
#pragma unroll
        for (int o = 0; o < 5; ++o) acc[o] = f32x4_s3{0.f, 0.f, 0.f, 0.f};
        const uint2 *src = s_i + r0 * S3K_IW + c0 + p + 2 * q;          // (rows past the patch: LDS of the S tile, results unused)
#pragma unroll
        for (int pr = 0; pr < 11; ++pr) {
            const uint2 lo = src[pr * S3K_IW], hi = src[pr * S3K_IW + 1];
            const u32x4 fb = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
            for (int o = 0; o < 5; ++o)
                if (pr - o >= 0 && pr - o < 7) acc[o] = stem3_mfma16<T>(fa0[pr - o], fb, acc[o]);
        }
#pragma unroll
        for (int o = 0; o < 5; ++o) {
            const int sy = r0 + o, sx = c0 + p;
            if (sy < S3K_SH) {
                const int gy = sy0 + sy, gx = sx0 + sx;
                const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                *reinterpret_cast<u32x2 *>(s_s + sy * S3K_ROWB + sx * S3K_PXB + q * 8) = stem3_epi<T>(acc[o], b0, in);
            }
        }
        if (wv >= 4) {                                                   // columns 32..34: flat pixel 16 (wv - 4) + p of 19 x 3
            const int fi = 16 * (wv - 4) + p;
            const int sy = fi / 3, sx = 32 + fi - 3 * sy;
            f32x4_s3 accr = {0.f, 0.f, 0.f, 0.f};
            const uint2 *srcr = s_i + sy * S3K_IW + sx + 2 * q;           // (fi >= 57: rows below the patch, unused)
#pragma unroll
            for (int dy = 0; dy < 7; ++dy) {
                const uint2 lo = srcr[dy * S3K_IW], hi = srcr[dy * S3K_IW + 1];
                const u32x4 fb = {lo.x, lo.y, hi.x, hi.y};
                accr = stem3_mfma16<T>(fa0[dy], fb, accr);
            }
            if (sy < S3K_SH) {
                const int gy = sy0 + sy, gx = sx0 + sx;
                const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                *reinterpret_cast<u32x2 *>(s_s + sy * S3K_ROWB + sx * S3K_PXB + q * 8) = stem3_epi<T>(accr, b0, in);
            }
        }
    }
    __syncthreads();
    if (H3D_DBG(a) == 2) return;

    // ---- P2: level0 -> L0 (K step = taps 2ks, 2ks+1 x 16 channels; lane group q: tap 2ks + (q >> 1), channels 8(q & 1)..) ----
    int toff[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const int tap = min(2 * ks + (q >> 1), 8);          // the 10th "tap" has zero weights: read tap 8 again
        toff[ks] = (tap / 3) * S3K_ROWB + (tap % 3) * S3K_PXB + (q & 1) * 16;
    }
    // two groups per iteration: two independent MFMA chains and their loads in flight.  Flat pixel list as in P1: 17 x 33 = 561
    // pixels = 36 groups (round 2: 51); group g and group g + 8 per iteration, a wave advances by 256 pixels
    {
        constexpr int NG = (S3K_LH * S3K_LW + 15) / 16;
        int ly0v = (wv * 16 + p) / S3K_LW, lx0v = (wv * 16 + p) - ly0v * S3K_LW;
#pragma unroll 1
        for (int g0i = wv; g0i < NG; g0i += 16) {
            f32x4_s3 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            int lyv[2], lxv[2];                                          // group g0i and group g0i + 8 (128 pixels on)
            lyv[0] = ly0v; lxv[0] = lx0v;
            lxv[1] = lx0v + 128 % S3K_LW; lyv[1] = ly0v + 128 / S3K_LW;
            if (lxv[1] >= S3K_LW) { lxv[1] -= S3K_LW; lyv[1] += 1; }
            const char *src[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) src[u] = s_s + lyv[u] * S3K_ROWB + lxv[u] * S3K_PXB;      // (pixels past the end: reads inside the LDS slack)
#pragma unroll
            for (int ks = 0; ks < 5; ++ks)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const u32x4 fb = *reinterpret_cast<const u32x4 *>(src[u] + toff[ks]);
                    acc[u] = stem3_mfma16<T>(fa1[ks], fb, acc[u]);
                }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (lyv[u] < S3K_LH) {                                   // (flat index < 561)
                    const int ly = lyv[u], lx = lxv[u];
                    const int gy = ly0 + ly, gx = lx0 + lx;
                    const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                    *reinterpret_cast<u32x2 *>(s_l + ly * S3K_ROWB + lx * S3K_PXB + q * 8) = stem3_epi<T>(acc[u], b1, in);
                }
            }
            lx0v += 256 % S3K_LW;
            ly0v += 256 / S3K_LW;
            if (lx0v >= S3K_LW) { lx0v -= S3K_LW; ly0v += 1; }
        }
    }
    __syncthreads();
    if (H3D_DBG(a) == 3) return;
    store_patch();                                           // the patch area was last read in P1

    // ---- P3: level1 (3x3 stride 2, 32 channels): waves 0..3, one 32-pixel N tile each -------------------------
    if (wv < 4) {
        const int r = l & 31, h = l >> 5;
        const uint16_t *w2 = a.w + 16 * 7 * 32 + 5 * 16 * 32;
        const int py = 2 * wv + (r >> 4), px = r & 15;
        const char *base = s_l + (2 * py) * S3K_ROWB + (2 * px) * S3K_PXB + h * 16;
        f32x16 acc[1][1];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[0][0][i] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const u32x4 fa = *reinterpret_cast<const u32x4 *>(w2 + (r * 9 + tap) * 16 + 8 * h);
            const u32x4 fb = *reinterpret_cast<const u32x4 *>(base + (tap / 3) * S3K_ROWB + (tap % 3) * S3K_PXB);
            typename ET<T>::frag fa_, fb_;
            fa_.v = fa; fb_.v = fb;
            ET<T>::mma(acc[0][0], fa_, fb_);
        }
        EpiArgs e;
        e.bias = a.bias + 32; e.res = nullptr; e.out = (char *)a.out; e.Ho = a.Ho; e.Wo = a.Wo; e.Cout = 32;
        e.out_cs = a.out_cs; e.res_cs = 0; e.relu = 1; e.out_mode = H3D_OUT_NHWC;
        if constexpr (!PROJ) {
            tile_epilogue<T, 1, 1, true>(acc, e, b, oy1, ox1, 0, wv, r, h);
        } else {
            // level1 output as tile_epilogue's FAST path writes it, then: 2x2 max-pool of the wave's 2 x 16 pixels by two cross-lane
            // exchanges on the PACKED values (after the ReLU every value is >= +0, so the integer maximum of the bit patterns is the
            // floating-point maximum, bf16 and fp16 alike) -> 8 pooled pixels x 32 channels into this tile's buffer of pooled rows;
            // waves 4-7 (idle in P3) turn the PREVIOUS tile's buffer into the residual map meanwhile (proj_tile below)
            typedef short s16x2_p __attribute__((ext_vector_type(2)));
            char *lp = smem + S3K_POOL_OFF + ((ti & 1) * 4 + wv) * 512;
            const int oy = oy1 + py, ox = ox1 + px;
            const bool in = oy < a.Ho && ox < a.Wo;
            T *op = reinterpret_cast<T *>(a.out) + (((size_t)b * a.Ho + oy) * a.Wo + ox) * a.out_cs + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4 *>(a.bias + 32 + 8 * g + 4 * h);
                const float v0 = fmaxf(acc[0][0][4 * g + 0] + bv.x, 0.f), v1 = fmaxf(acc[0][0][4 * g + 1] + bv.y, 0.f);
                const float v2 = fmaxf(acc[0][0][4 * g + 2] + bv.z, 0.f), v3 = fmaxf(acc[0][0][4 * g + 3] + bv.w, 0.f);
                if (in) store4<T>(op + 8 * g, v0, v1, v2, v3);
                u32x2 pk;
                if constexpr (std::is_same_v<T, f16_t>) pk = u32x2{pack_f16x2(__builtin_amdgcn_fmed3f(v0, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v1, -65504.f, 65504.f)),
                                                                  pack_f16x2(__builtin_amdgcn_fmed3f(v2, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v3, -65504.f, 65504.f))};
                else pk = u32x2{EP<T>::pack2(v0, v1), EP<T>::pack2(v2, v3)};
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    uint32_t m0 = pk[e2];
                    m0 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_p, m0), __builtin_bit_cast(s16x2_p, (uint32_t)__shfl_xor((int)m0, 1))));
                    m0 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_p, m0), __builtin_bit_cast(s16x2_p, (uint32_t)__shfl_xor((int)m0, 16))));
                    pk[e2] = m0;
                }
                if ((r & 17) == 0) *reinterpret_cast<u32x2 *>(lp + (r >> 1) * 64 + (8 * g + 4 * h) * 2) = pk;      // pooled pixel r / 2 of this wave's row
            }
        }
    } else if constexpr (PROJ) {
        if (wv < 6 && ti > t_first) proj_tile(ti - 1);
    }
    }   // tiles of this workgroup
    if constexpr (PROJ) {                                    // the last tile's pooled rows
        const int t_last = min(t_first + a.tpb, ntile) - 1;
        __syncthreads();
        if ((wv == 4 || wv == 5) && t_last >= t_first) proj_tile(t_last);
    }
}

int h3d_launch_stem3x(const h3d_op &op, hipStream_t st);      // csrc/stem3x.hip: the f16x3 twin (fp32 storage, split-operand fp16 MFMAs)

int h3d_launch_stem3(const h3d_op &op, hipStream_t st)
{
    if (!op.in || !op.w || !op.bias || !op.out) H3D_FAIL(H3D_ERR_ARG, "stem3: null pointer");
    if (op.dtype != H3D_BF16 && op.dtype != H3D_F16 && op.dtype != H3D_F16X3) H3D_FAIL(H3D_ERR_DTYPE, "stem3: bf16 / fp16 / f16x3 plans only (dtype %d)", op.dtype);
    if (op.Cin != 3 || op.Cout != 32 || op.Ho != (op.H - 1) / 2 + 1 || op.Wo != (op.W - 1) / 2 + 1 || op.out_cs % 4 || op.out_cs < 32)
        H3D_FAIL(H3D_ERR_SHAPE, "stem3: expects 3 -> 16 -> 16 -> 32 channels, output %dx%d (got %d -> %d, %dx%d)", (op.H - 1) / 2 + 1,
                 (op.W - 1) / 2 + 1, op.Cin, op.Cout, op.Ho, op.Wo);
    if (((uintptr_t)op.bias & 15) || ((uintptr_t)op.w & 15)) H3D_FAIL(H3D_ERR_ARG, "stem3: weights / bias must be 16-byte aligned");
    if (op.dtype == H3D_F16X3) return h3d_launch_stem3x(op, st);
    // the image patch is fetched as aligned float4 (load_patch): a row must be whole vectors, or a vector straddles the right edge (and
    // the last one of the batch the end of the buffer).  csrc/stem3x.hip loads single pixels and takes any width.
    if (op.W % 4 || ((uintptr_t)op.in & 15))
        H3D_FAIL(H3D_ERR_SHAPE, "stem3: the 2-byte kernel needs an image width that is a multiple of 4 and a 16-byte aligned image (W = %d)", op.W);
    Stem3Args a;
    a.wproj = nullptr; a.bproj = nullptr; a.res_out = nullptr; a.res_cs = 0;
    const bool proj = op.in2 != nullptr;       // in2 = level2's residual map [B,Ho/2,Wo/2,in2_cs] (output); its filters follow level1's in w / bias
    if (proj) {
        if (op.in2_cs % 4 || op.in2_cs < 64 || ((uintptr_t)op.in2 & 7) || (op.Ho | op.Wo) & 1)
            H3D_FAIL(H3D_ERR_SHAPE, "stem3: the fused residual branch needs an even level1 map and a 64-channel output (stride %d)", op.in2_cs);
        a.wproj = (const uint16_t *)op.w + 16 * 7 * 32 + 5 * 16 * 32 + 32 * 9 * 16;
        a.bproj = op.bias + 64;
        a.res_out = (void *)op.in2; a.res_cs = op.in2_cs;
    }
    a.img = (const float *)op.in; a.w = (const uint16_t *)op.w; a.bias = op.bias; a.out = op.out;
    a.B = op.B; a.H = op.H; a.W = op.W; a.Ho = op.Ho; a.Wo = op.Wo; a.out_cs = op.out_cs;
    a.tiles_x = cdiv(op.Wo, 16); a.tiles_y = cdiv(op.Ho, 8);
    a.dbg = op.reserved;
    if (h3d_note_kernel(proj ? "stem3_kernel<%s, true>" : "stem3_kernel<%s>", op.dtype == H3D_F16 ? "f16_t" : "unsigned short")) return H3D_OK;
    // as many tiles per workgroup as still give every one of the 512 workgroup slots (two per CU) a workgroup: batch 64 at 512 x 512 =
    // 32768 tiles -> 64 per workgroup (one round of equal workgroups: no tail, one prologue per slot); an 8-image shard (4096) -> 8
    const int ntiles = op.B * a.tiles_x * a.tiles_y;
    int tpb = 4;
    while (tpb < 64 && cdiv(ntiles, 2 * tpb) >= 512) tpb *= 2;
    if (S3K_TPB_FORCE) tpb = S3K_TPB_FORCE;
    a.tpb = tpb;
    const dim3 grid(cdiv(ntiles, tpb));
    if (proj) {
        if (op.dtype == H3D_F16) hipLaunchKernelGGL((stem3_kernel<f16_t, true>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((stem3_kernel<bf16_t, true>), grid, dim3(512), 0, st, a);
    } else if (op.dtype == H3D_F16) hipLaunchKernelGGL(stem3_kernel<f16_t>, grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL(stem3_kernel<bf16_t>, grid, dim3(512), 0, st, a);
    H3D_CHECK_LAUNCH("stem3_kernel");
    return H3D_OK;
}
