// 3x3 stride-1 convolution, generation 2 (bf16): implicit GEMM fed entirely by LDS-DMA.
// (reference: the nn.Conv2d(k=3, s=1, p=1) + BN (+ residual) + ReLU of DLA-34's BasicBlock, model.py:42-71;
//  same math as csrc/conv.hip, which keeps stride 2, 1x1, fp32 and non-NHWC outputs.)
//
// One workgroup = (2*WAVES) x 16 output pixels x (32*MT) output channels; wave w owns pixel rows
// 2w, 2w+1 (one 32-pixel MFMA N-tile) and all MT M-tiles.  The contraction walks "stages" of 16 input
// channels.  A stage's operands are two LDS images that `buffer_load ... lds` writes directly (no
// staging registers, no ds_write, no address arithmetic in the loop):
//   halo    [2*WAVES+2 rows][64 slots of 16 B]: pixel ix of a row owns slots 3ix, 3ix+1 (its 16
//           channels) and a pad slot 3ix+2, so the pixel stride is 48 B (odd multiple of 16 B) and the
//           row stride 1 KiB = ONE DMA wave-instruction; lanes whose pixel lies outside the image
//           (or that own pad slots) carry an out-of-range buffer offset and the DMA writes zeros.
//   weights [32*MT rows][19 slots]: slot 2*tap+h = input channels 8h..8h+7 of tap `tap`, slot 18 pad
//           (row stride 304 B).  The host packs the filter bank stage-major in exactly this order
//           (engine.PackedWeights.conv_stream), so the DMA is a linear copy of MT*9728 bytes.
// Both strides make every 16-lane ds_read_b128 group hit 16 distinct 16-byte bank slots.
// Two ring slots: stage s+1 is in flight while stage s feeds 9*MT MFMAs per wave; one barrier a stage.
#include "common.h"
#include "epilogue.h"

struct Conv2Args {
    const char *in;
    const char *wimg;   // [Cin/16][G][32][19][16 B], G = row groups of 32 output channels (zero padded)
    const float *bias;
    const char *res;
    char *out;
    int B, H, W, Cin, in_cs;
    int Ho, Wo, Cout, out_cs, res_cs, relu, out_mode;
    int G, tiles_x, tiles_y;
    int dbg;   // ABLATE builds only (op.reserved >> 16): 1 = no DMA after stage 0, 2 = no fragment reads / MFMA
    int xcd;   // h3d_tile_id mode
    int f16;   // host side only: fp16 plan (the launcher picks the f16_t instantiation)
    unsigned long long *stamps;   // profiling builds: per workgroup and wave (0-7) the cycles spent at the stage barrier; slot 7: wave 0's {DMA issue} (tools/stamp_conv2.py)
};

template <int MT, int WAVES, int S = 1, int SLOTS = 2, int NT = 1>
struct Conv2Cfg {
    static constexpr int TH = 2 * WAVES * NT;           // a wave owns NT N-tiles (2 rows x 16 px each): filter fragments are reused NT times
    static constexpr int HROWS = (TH - 1) * S + 3;      // halo rows
    static constexpr int HCOLS = 15 * S + 3;            // halo pixels per row: 18 (stride 1) / 33 (stride 2)
    static constexpr int PPR = S;                       // 1 KiB DMA pieces per halo row (3 slots per pixel)
    static constexpr int ROWB = PPR * 1024;
    static constexpr int ROWS = HROWS * PPR;            // DMA pieces of the halo image
    static constexpr int HALO = ROWS * 1024;
    static constexpr int WROW = 19 * 16;                // 304 B
    static constexpr int WGRP = 32 * WROW;              // one 32-row group: 9728 B
    static constexpr int WPIECES = (MT * WGRP + 1023) / 1024;
    static constexpr int WALLOC = WPIECES * 1024;
    static constexpr int SLOT = HALO + WALLOC;
    static constexpr int LDS_EPI = WAVES * ((NT * 32 * (64 * MT + 16) + 1023) / 1024 * 1024);   // tile_epilogue_lds regions
    static constexpr int LDS_RING = SLOTS * SLOT;   // SLOTS = 1: single-stage layers (Cin = 16) keep more workgroups per CU
    static constexpr int LDS = LDS_RING > LDS_EPI ? LDS_RING : LDS_EPI;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int HP = (ROWS + WAVES - 1) / WAVES;       // halo pieces per wave
    static constexpr int WP = (WPIECES + WAVES - 1) / WAVES;    // weight pieces per wave
};

typedef __attribute__((address_space(3))) void lds_void;

// LDS-DMA of one stage into the ring slot at `base` (a plain function of plain arguments: the buffer-descriptor
// type does not exist in the host pass, and a lambda capturing one silently drops the kernel's stub)
template <int MT, int WAVES, int S, int NT>
__device__ __forceinline__ void conv2_issue(const char *in_b, int in_bytes, const char *wimg, int w_bytes, char *base,
                                            const int *hoff, int woff, int wv, int s, int wsrc, int dbg = 0)
{
    using C = Conv2Cfg<MT, WAVES, S, 2, NT>;
    // descriptors are rebuilt from wave-uniform scalars at every call (4 SGPRs each, no memory traffic)
    const auto r_in = __builtin_amdgcn_make_buffer_rsrc((void *)in_b, 0, in_bytes, 0x00020000);
    const auto r_w = __builtin_amdgcn_make_buffer_rsrc((void *)wimg, 0, w_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < C::HP; ++j) {
        const int p = wv + j * WAVES;
        if (p < C::ROWS && !(dbg & 8)) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_in, (lds_void *)(base + p * 1024), 16, hoff[j], s * 32, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < C::WP; ++j) {
        const int p = wv + j * WAVES;
        if (p < C::WPIECES && !(dbg & 16))
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lds_void *)(base + C::HALO + p * 1024), 16, woff, wsrc + p * 1024, 0, 0);
    }
}

template <typename T, int MT, int WAVES, int EPI, int S = 1, int SLOTS = 2, int NT = 1, bool PIPE = false>   // T: bf16_t | f16_t; EPI: 0 general, 1 lean NHWC, 2 LDS-transposed NHWC (epilogue.h); S: stride
__global__ __launch_bounds__(64 * WAVES) void conv2_kernel(Conv2Args a)
{
    using C = Conv2Cfg<MT, WAVES, S, SLOTS, NT>;
    using E = ET<T>;
    static_assert(sizeof(T) == 2, "2-byte element types");
    __shared__ __attribute__((aligned(1024))) char smem[C::LDS];

    const int tid = threadIdx.x;
    const int l = tid & 63, r = l & 31, h = l >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform for the compiler too
    const int tiles = a.tiles_x * a.tiles_y;
    const int bid = h3d_tile_id(blockIdx.x, gridDim.x, a.xcd);
    const int b = bid / tiles;
    const int t = bid - b * tiles;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int oy0 = ty * C::TH, ox0 = tx * 16;
    const int g0 = blockIdx.y * MT;                               // first 32-row group of this workgroup
    const int cout0 = g0 * 32;
    const int nst = a.Cin >> 4;

    // ---- DMA descriptors: the image of batch b, and the packed filter bank ----------------------------
    const size_t img_bytes = (size_t)a.H * a.W * a.in_cs * 2;
    const char *in_b = a.in + (size_t)b * img_bytes;
    const int w_bytes = nst * a.G * C::WGRP;
    // per-lane source offsets of my halo pieces (stage 0); the stage adds 32 B through soffset
    int hoff[C::HP];
#pragma unroll
    for (int j = 0; j < C::HP; ++j) {
        const int pc = wv + j * WAVES;                            // DMA piece: halo row pc / PPR, slots (pc % PPR) * 64 ...
        const int iy = pc / C::PPR, slot = (pc - iy * C::PPR) * 64 + l;
        const int ix = slot / 3, sub = slot - 3 * ix;
        const int gy = oy0 * S - 1 + iy, gx = ox0 * S - 1 + ix;
        const bool ok = pc < C::ROWS && ix < C::HCOLS && sub < 2 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        hoff[j] = ok ? ((gy * a.W + gx) * a.in_cs + sub * 8) * 2 : 0x7ffffff0;
    }
    const int woff = l * 16;

    // fragment offsets inside a slot
    const int py = 2 * NT * wv + (r >> 4), px = r & 15;      // N-tile n adds 2 rows
    const int boff = py * S * C::ROWB + px * S * 48 + h * 16;
    const int aoff = C::HALO + r * C::WROW + h * 16;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    // ring of SLOTS stages: stage s + SLOTS - 1 is issued when stage s starts, so a stage has SLOTS - 1 stages of
    // MFMA time to land (with 2 slots the SQ counters showed the waves parked ~48 % of the time)
    constexpr int AHEAD = SLOTS > 1 ? SLOTS - 1 : 1;
    constexpr int PMIN = C::ROWS / WAVES + C::WPIECES / WAVES;     // fewest DMA instructions any wave issues per stage
    static_assert(AHEAD <= 2 && PMIN <= 63, "counted vmcnt wait");
    constexpr int WAIT_PMIN = 0x0f70 | (PMIN & 15) | ((PMIN >> 4) << 14);     // s_waitcnt vmcnt(PMIN): 6-bit field, split
#pragma unroll
    for (int j = 0; j < AHEAD; ++j)
        if (j < nst && (SLOTS > 1 || j == 0))
            conv2_issue<MT, WAVES, S, NT>(in_b, (int)img_bytes, a.wimg, w_bytes, smem + j * C::SLOT, hoff, woff, wv, j, (j * a.G + g0) * C::WGRP, H3D_DBG(a));
    int cslot = 0, pslot = AHEAD % SLOTS;          // slot consumed by stage s / filled with stage s + AHEAD
#ifdef H3D_ABLATE
    unsigned long long t_bar = 0, t_iss = 0;
    const unsigned long long t_start = __builtin_readcyclecounter();
#endif
    for (int s = 0; s < nst; ++s) {
        if constexpr (SLOTS == 1) {
            // one slot, several stages: the stage is fetched once nobody reads its predecessor any more; the wait is
            // covered by the CU's other workgroups (the slot is small enough for two or three of them)
            if (s > 0) {
                __syncthreads();
                conv2_issue<MT, WAVES, S, NT>(in_b, (int)img_bytes, a.wimg, w_bytes, smem, hoff, woff, wv, s, (s * a.G + g0) * C::WGRP, H3D_DBG(a));
            }
        }
        // my pieces of stage s have landed once only stage s+1's may be outstanding (vmcnt retires in order)
        if (AHEAD == 2 && s + 1 < nst) __builtin_amdgcn_s_waitcnt(WAIT_PMIN);
        else __builtin_amdgcn_s_waitcnt(0x0f70);
#ifdef H3D_ABLATE
        const unsigned long long tb0 = __builtin_readcyclecounter();
#endif
        h3d_barrier_keep_vmcnt();                 // ... everyone's have; the slot of stage s-1 is no longer being read (not
                                                  // __syncthreads: its release fence waits for the younger stage's DMA too)
#ifdef H3D_ABLATE
        const unsigned long long tb1 = __builtin_readcyclecounter();
        t_bar += tb1 - tb0;
#endif
        if (SLOTS > 1 && s + AHEAD < nst && !(H3D_DBG(a) & 1))
            conv2_issue<MT, WAVES, S, NT>(in_b, (int)img_bytes, a.wimg, w_bytes, smem + pslot * C::SLOT, hoff, woff, wv, s + AHEAD, ((s + AHEAD) * a.G + g0) * C::WGRP, H3D_DBG(a));
#ifdef H3D_ABLATE
        t_iss += __builtin_readcyclecounter() - tb1;
#endif
        const char *sl = smem + cslot * C::SLOT;
        cslot = cslot + 1 == SLOTS ? 0 : cslot + 1;
        pslot = pslot + 1 == SLOTS ? 0 : pslot + 1;
        if (H3D_DBG(a) & 2) continue;
        if constexpr (PIPE) {
            // fragments of tap t+1 are read while tap t is multiplied: left to itself hipcc keeps 24 fragment registers and
            // waits (lgkmcnt(1)) in front of every MFMA pair for a read issued two instructions earlier; with two fragment
            // sets (40 registers) and the issue order pinned by sched_group_barrier the wait becomes lgkmcnt(MT + NT)
            typename E::frag fb[2][NT], fa[2][MT];
            auto rd = [&](int tap, int q) {
                const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
                for (int n = 0; n < NT; ++n) fb[q][n] = E::lds_frag(sl + boff + (2 * n * S + dy) * C::ROWB + dx * 48);
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[q][m] = E::lds_frag(sl + aoff + m * C::WGRP + tap * 32);
            };
            rd(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap + 1 < 9) rd(tap + 1, (tap + 1) & 1);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) E::mma(acc[m][n], fa[tap & 1][m], fb[tap & 1][n]);
                if (tap + 1 < 9) __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
            }
        } else {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - 3 * dy;
            typename E::frag fb[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) fb[n] = E::lds_frag(sl + boff + (2 * n * S + dy) * C::ROWB + dx * 48);
            typename E::frag fa[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) fa[m] = E::lds_frag(sl + aoff + m * C::WGRP + tap * 32);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) E::mma(acc[m][n], fa[m], fb[n]);
        }
        }
    }

#ifdef H3D_ABLATE
    if (a.stamps && blockIdx.x < 65536 && blockIdx.y == 0 && l == 0) {
        const unsigned long long t_loop = __builtin_readcyclecounter() - t_start;
        if (wv < 4) a.stamps[blockIdx.x * H3D_NSTAMP + wv] = t_bar;                 // waves 0-3: the oldest wave of each SIMD
        if (wv >= WAVES - 3) a.stamps[blockIdx.x * H3D_NSTAMP + 4 + (wv - (WAVES - 3))] = t_bar;   // the three youngest
        if (wv == 0) a.stamps[blockIdx.x * H3D_NSTAMP + 7] = (t_iss << 32) | (t_loop & 0xffffffffull);
    }
#endif
    EpiArgs e;
    e.bias = a.bias; e.res = a.res; e.out = a.out; e.Ho = a.Ho; e.Wo = a.Wo; e.Cout = a.Cout;
    e.out_cs = a.out_cs; e.res_cs = a.res_cs; e.relu = a.relu; e.out_mode = a.out_mode;
    if constexpr (EPI == 2) {
        __syncthreads();                          // nobody reads the ring any more
        tile_epilogue_lds<T, MT, NT>(acc, e, b, oy0, ox0, cout0, wv, l, smem + wv * epi_lds_stride<MT, NT>());
    } else {
        tile_epilogue<T, MT, NT, EPI == 1>(acc, e, b, oy0, ox0, cout0, wv, r, h);
    }
}

template <typename T, int MT, int WAVES, int S, int SLOTS, int NT, bool PIPE>
static int launch_conv2_t(const Conv2Args &a, dim3 grid, int epi, hipStream_t st)
{
    using C = Conv2Cfg<MT, WAVES, S, SLOTS, NT>;
    if constexpr (PIPE) {
        if (epi == 2) {
            if (h3d_note_kernel("conv2_kernel<%s, %d, %d, %d, %d, %d, %d, true>", h3d_tname<T>(), MT, WAVES, epi, S, SLOTS, NT)) return H3D_OK;
            hipLaunchKernelGGL((conv2_kernel<T, MT, WAVES, 2, S, SLOTS, NT, true>), grid, dim3(C::THREADS), 0, st, a);
            H3D_CHECK_LAUNCH("conv2_kernel");
            return H3D_OK;
        }
    }
    if (h3d_note_kernel("conv2_kernel<%s, %d, %d, %d, %d, %d, %d>", h3d_tname<T>(), MT, WAVES, epi, S, SLOTS, NT)) return H3D_OK;
    if (epi == 2)
        hipLaunchKernelGGL((conv2_kernel<T, MT, WAVES, 2, S, SLOTS, NT>), grid, dim3(C::THREADS), 0, st, a);
    else if (epi == 1)
        hipLaunchKernelGGL((conv2_kernel<T, MT, WAVES, 1, S, SLOTS, NT>), grid, dim3(C::THREADS), 0, st, a);
    else
        hipLaunchKernelGGL((conv2_kernel<T, MT, WAVES, 0, S, SLOTS, NT>), grid, dim3(C::THREADS), 0, st, a);
    H3D_CHECK_LAUNCH("conv2_kernel");
    return H3D_OK;
}

template <int MT, int WAVES, int S = 1, int SLOTS = 2, int NT = 1, bool PIPE = false>
static int launch_conv2_cfg(const Conv2Args &a0, hipStream_t st)
{
    using C = Conv2Cfg<MT, WAVES, S, SLOTS, NT>;
    static_assert(C::LDS <= 160 * 1024, "LDS budget");
    Conv2Args a = a0;
    a.tiles_x = cdiv(a.Wo, 16);
    a.tiles_y = cdiv(a.Ho, C::TH);
    a.xcd = h3d_xcd_mode();
#ifdef H3D_ABLATE
    a.stamps = h3d_stamp_buffer();
#else
    a.stamps = nullptr;
#endif
    dim3 grid(a.B * a.tiles_x * a.tiles_y, cdiv(cdiv(a.Cout, 32), MT));
    const bool lean = a.out_mode == H3D_OUT_NHWC && a.Cout % 4 == 0 && ((uintptr_t)a.bias & 15) == 0;
    const int epi = (MT >= 2 && lean && a.Cout % 8 == 0 && a.out_cs % 8 == 0 && ((uintptr_t)a.out & 15) == 0 && !(a.dbg & 4)) ? 2 : lean ? 1 : 0;
    if (a.f16) return launch_conv2_t<f16_t, MT, WAVES, S, SLOTS, NT, PIPE>(a, grid, epi, st);
    return launch_conv2_t<bf16_t, MT, WAVES, S, SLOTS, NT, PIPE>(a, grid, epi, st);
}

int h3d_launch_conv_stream(const h3d_op &op, hipStream_t st)
{
    if (!op.in || !op.w || !op.bias || !op.out) H3D_FAIL(H3D_ERR_ARG, "conv_stream: null pointer");
    if (op.dtype != H3D_BF16 && op.dtype != H3D_F16) H3D_FAIL(H3D_ERR_DTYPE, "conv_stream: bf16 / fp16 plans only (dtype %d)", op.dtype);
    if (op.ksize != 3 || (op.stride != 1 && op.stride != 2) || op.Ho != (op.H - 1) / op.stride + 1 || op.Wo != (op.W - 1) / op.stride + 1)
        H3D_FAIL(H3D_ERR_UNSUPPORTED, "conv_stream: covers 3x3 p1 with stride 1 or 2 (k=%d s=%d, %dx%d -> %dx%d)", op.ksize, op.stride,
                 op.H, op.W, op.Ho, op.Wo);
    if (op.Cin % 16 || op.in_cs % 8 || op.Cin > op.in_cs)
        H3D_FAIL(H3D_ERR_SHAPE, "conv_stream: Cin=%d (stride %d) must be a multiple of 16", op.Cin, op.in_cs);
    if ((size_t)op.H * op.W * op.in_cs * 2 >= 0x7ffffff0ull) H3D_FAIL(H3D_ERR_SHAPE, "conv_stream: image of 2 GiB or more");
    if (op.wrows % 128 || op.wrows < op.Cout)
        H3D_FAIL(H3D_ERR_SHAPE, "conv_stream: packed weight/bias rows %d for Cout %d (Cout padded to 128 expected)", op.wrows, op.Cout);
    if (op.out_mode != H3D_OUT_NCHW_F32 && (op.out_cs % 4 || op.Cout > op.out_cs))
        H3D_FAIL(H3D_ERR_SHAPE, "conv_stream: out channel stride %d", op.out_cs);
    if (op.in2 && (op.in2_cs % 4 || op.Cout > op.in2_cs)) H3D_FAIL(H3D_ERR_SHAPE, "conv_stream: residual stride %d", op.in2_cs);
    Conv2Args a;
    a.in = (const char *)op.in; a.wimg = (const char *)op.w; a.bias = op.bias; a.res = (const char *)op.in2;
    a.out = (char *)op.out; a.B = op.B; a.H = op.H; a.W = op.W; a.Cin = op.Cin; a.in_cs = op.in_cs;
    a.Ho = op.Ho; a.Wo = op.Wo; a.Cout = op.Cout; a.out_cs = op.out_cs; a.res_cs = op.in2_cs; a.relu = op.relu; a.out_mode = op.out_mode;
    a.G = op.wrows / 32; a.tiles_x = a.tiles_y = 0;
    a.f16 = op.dtype == H3D_F16;
    a.dbg = op.reserved >> 16;
    // workgroups a (th rows x 16 px) x (32*mt channels) tiling produces
    const int gq = cdiv(op.Cout, 32);
    auto nblk = [&](int th, int mt) { return (long)op.B * cdiv(op.Wo, 16) * cdiv(op.Ho, th) * cdiv(gq, mt); };
    if (op.stride == 2) {          // the stride-2 halo is (2 TH + 1) x 33 pixels: 4-wave tiles are what fits twice in the LDS
        switch (op.reserved & 0xffff) {           // tuning override (tools/ab_conv.py)
        case 0x4404: return launch_conv2_cfg<4, 4, 2, 1>(a, st);    // 0x4...: ONE ring slot (more workgroups per CU)
        case 0x4408: return launch_conv2_cfg<4, 8, 2, 1>(a, st);
        case 0x4204: return launch_conv2_cfg<2, 4, 2, 1>(a, st);
        default: break;
        }
        // one ring slot (73 KB): two workgroups per CU instead of one 4-wave workgroup with a two-slot ring
        // (tools/ab_conv_s2.py, batch 64: 64->128 0.093 -> 0.073 ms, 128->256 0.081 -> 0.082, 256->512 0.064 -> 0.051)
        if (gq >= 4) return (op.reserved & 0xffff) == 0x2404 ? launch_conv2_cfg<4, 4, 2>(a, st) : launch_conv2_cfg<4, 4, 2, 1>(a, st);
        if (gq >= 2) return launch_conv2_cfg<2, 4, 2>(a, st);
        return launch_conv2_cfg<1, 4, 2>(a, st);
    }
    if (op.reserved & 0xffff) {   // tuning override (profiling): reserved = MT << 8 | WAVES
        switch (op.reserved & 0xffff) {
        case 0x410: return launch_conv2_cfg<4, 16>(a, st);
        case 0x408: return launch_conv2_cfg<4, 8>(a, st);
        case 0x404: return launch_conv2_cfg<4, 4>(a, st);
        case 0x208: return launch_conv2_cfg<2, 8>(a, st);
        case 0x204: return launch_conv2_cfg<2, 4>(a, st);
        case 0x108: return launch_conv2_cfg<1, 8>(a, st);
        case 0x104: return launch_conv2_cfg<1, 4>(a, st);
        case 0x2408: return launch_conv2_cfg<4, 8, 1, 2, 2>(a, st);   // 0x2...: two N-tiles per wave (32 x 16 px tile, 8 waves)
        case 0x2208: return launch_conv2_cfg<2, 8, 1, 2, 2>(a, st);
        case 0x3208: return launch_conv2_cfg<2, 8, 1, 3>(a, st);      // 0x3...: three ring slots
        case 0x3108: return launch_conv2_cfg<1, 8, 1, 3>(a, st);
        case 0x3404: return launch_conv2_cfg<4, 4, 1, 3>(a, st);
        case 0x5408: return launch_conv2_cfg<4, 8, 1, 1>(a, st);      // 0x5...: ONE ring slot
        case 0x5208: return launch_conv2_cfg<2, 8, 1, 1>(a, st);
        case 0x5108: return launch_conv2_cfg<1, 8, 1, 1>(a, st);
        case 0x6410: return launch_conv2_cfg<4, 16, 1, 2, 1, true>(a, st);   // 0x6...: fragment reads one tap ahead (PIPE)
        case 0x6408: return launch_conv2_cfg<4, 8, 1, 2, 1, true>(a, st);
        case 0x6208: return launch_conv2_cfg<2, 8, 1, 2, 1, true>(a, st);
        case 0x6404: return launch_conv2_cfg<4, 4, 1, 2, 1, true>(a, st);
        case 0x1: break;           // 1 = auto configuration (used with the ablation bits)
        default: H3D_FAIL(H3D_ERR_ARG, "conv_stream: unknown tuning override %#x", op.reserved);
        }
    }
    // the three configurations a batch-64 plan selects read their fragments one tap ahead (PIPE): -2 % each (tools/ab_conv.py
    // 6410 6408 6208, same process: 0.548 -> 0.538 ms on the seven 128 -> 128 @64x64 launches, 0.296 -> 0.289 on 64 -> 64 @128x128)
    if (gq >= 4) {
        if (op.Ho % 32 == 0 && nblk(32, 4) >= 256) return launch_conv2_cfg<4, 16, 1, 2, 1, true>(a, st);   // 16 waves: 32 x 16 px share one weight stream
        if (nblk(16, 4) >= 256) return launch_conv2_cfg<4, 8, 1, 2, 1, true>(a, st);
        // grids that leave CUs idle at 128 channels per workgroup (the deep levels of Hourglass-104 at 16 images per GPU, DLA's level 5
        // in an 8-image shard): narrower channel blocks.  Such a launch is a chain of Cin / 16 DMA round trips per workgroup whatever its
        // width, so more, smaller workgroups per CU is what hides them (tools/arch_kernels.py hourglass --codes ..., batch 16, same process:
        // 384 -> 384 @32x32 1.166 -> 0.846 ms for the 18 launches on <2,8>; @16x16 0.708 -> 0.421, @8x8 0.685 -> 0.337 on <1,4>;
        // 512 -> 512 @4x4 1.041 -> 0.559 for the 26 launches)
        if (!(op.reserved & 0x10000000)) {          // (0x10000000: round 4's rule, 128-channel 8-row tiles, for A/B runs)
            if (nblk(16, 2) >= 256) return launch_conv2_cfg<2, 8, 1, 2, 1, true>(a, st);
            if (nblk(16, 1) >= 256) return launch_conv2_cfg<1, 8>(a, st);
            return launch_conv2_cfg<1, 4>(a, st);
        }
        return launch_conv2_cfg<4, 4>(a, st);
    }
    if (gq >= 2) {
        if (nblk(16, 2) >= 256) return launch_conv2_cfg<2, 8, 1, 2, 1, true>(a, st);
        return launch_conv2_cfg<2, 4>(a, st);
    }
    if (op.Cin == 16 && nblk(16, 1) >= 256) return launch_conv2_cfg<1, 8, 1, 1>(a, st);   // one stage: no ring
    if (nblk(16, 1) >= 256) return launch_conv2_cfg<1, 8>(a, st);
    return launch_conv2_cfg<1, 4>(a, st);
}
