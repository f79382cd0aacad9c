// 1x1 convolution (stride 1, or 2: a strided row gather) as a plain GEMM (bf16, fp32 accumulate), fed entirely by LDS-DMA.
// (reference: the Root / project convs of DLA-34, model.py:148-166, 200-207; the bottleneck 1x1 convs of the published
//  ResNet-101-DCN and the 1x1 convs of Hourglass-104's residual blocks; same math and the same epilogue as csrc/conv.hip,
//  which keeps fp32, other kernel sizes / strides and the shapes this kernel does not take.)
//
// A 1x1 convolution has no spatial structure: out[p][co] = sum_ci w[co][ci] * in[p][ci] over the FLAT pixel index p of
// the NHWC batch.  csrc/conv.hip ran it through the halo-tile machinery of the 3x3 kernels (128 px x 128 channels per
// 4-wave workgroup, register staging, two barriers per 64-channel chunk, 1.25 KB of LDS reads per MFMA): 0.14 of the MFMA
// peak on ResNet-101's 1024 <-> 256 layers and 3-3.5 TB/s on DLA-34's HBM-bound roots.  Here:
//   M = output channels (A operand = rows of the packed [rows][Cin] filter bank, exactly H3D_OP_CONV's layout),
//   N = pixels (B operand), so a lane owns one pixel and 4-channel runs and the shared epilogues apply unchanged: the flat
//       pixel array is presented to them as an image of width 16 (tile rows of 16 pixels);
//   one workgroup (8 waves as WM x WN) = (32 MT WM) channels x (32 NT WN) pixels; a wave keeps MT x NT accumulator tiles
//       (LDS reads per MFMA: (MT + NT) / (MT NT) KB = 0.75 / 1 KB);
//   K walks stages of 64 input channels: rows of 128 B, both operands in the same LDS image [rows][8 slots of 16 B] with
//       slot' = slot ^ ((row >> 1) & 7) -- an LDS-DMA wave-instruction writes 1 KiB = 8 whole rows and the swizzle goes
//       into the per-lane SOURCE address, so every 16-lane ds_read_b128 group of a fragment read hits 16 distinct slots;
//   ring of SLOTS stages, AHEAD = SLOTS - 1 in flight, counted vmcnt (every wave issues the same number of pieces per
//       stage), one barrier per stage.  Rows past the last pixel / the last packed filter row lie beyond the buffer
//       descriptors' limits and arrive as zeros.
#include "common.h"
#include "epilogue.h"

struct Gemm1Args {
    const char *in;      // bf16 NHWC, N pixels x in_cs
    const char *w;       // bf16 [wrows][Cin]
    const float *bias;
    const char *res;     // bf16 NHWC or null
    char *out;           // bf16 NHWC
    long long N;         // B * Ho * Wo output pixels
    int H, W, Ho, Wo, stride;   // stride 2 (the 1x1 down-sampling convs of ResNet / Hourglass): input pixel (2 oy, 2 ox)
    int Cin, in_cs, Cout, out_cs, res_cs, relu, wrows;
    int tiles_n, blocks_m;
    int xcd;
    int f16;             // host side only: fp16 plan
};

template <int MT, int NT, int WM, int WN, int SLOTS>
struct Gemm1Cfg {
    static constexpr int WAVES = WM * WN;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    static constexpr int KS = 64;                        // input channels per stage: rows of 128 B
    static constexpr int SLOT = (BM + BN) * 128;
    static constexpr int PA = BM / 8 / WAVES, PB = BN / 8 / WAVES;   // 1 KiB pieces per wave and stage
    static constexpr int PPW = PA + PB;
    static constexpr int AHEAD = SLOTS - 1;
    static constexpr int LDS_RING = SLOTS * SLOT;
    static constexpr int LDS_EPI = WAVES * epi_lds_stride<MT, NT>();
    static constexpr int LDS = LDS_RING > LDS_EPI ? LDS_RING : LDS_EPI;
    static_assert(BM % (8 * WAVES) == 0 && BN % (8 * WAVES) == 0, "every wave issues the same number of pieces");
    static_assert(AHEAD >= 1 && AHEAD * PPW <= 63, "counted vmcnt wait");
};

typedef __attribute__((address_space(3))) void lds_void_g1;

// LDS-DMA of stage s into the ring slot at `slot` (a plain function of plain arguments: the buffer-descriptor type does
// not exist in the host pass, and a lambda capturing one silently drops the kernel's host stub).  Descriptors are
// rebuilt from wave-uniform scalars at every call (4 SGPRs each, no memory traffic).
template <int PA, int PB, int WAVES, int BM>
__device__ __forceinline__ void gemm1_issue(const char *w0, int w_bytes, const char *in0, int in_bytes, char *slot,
                                            const int *offa, const int *offb, int wv, int s)
{
    const auto r_w = __builtin_amdgcn_make_buffer_rsrc((void *)w0, 0, w_bytes, 0x00020000);
    const auto r_in = __builtin_amdgcn_make_buffer_rsrc((void *)in0, 0, in_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < PA; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lds_void_g1 *)(slot + (wv + j * WAVES) * 1024), 16, offa[j], s * 128, 0, 0);
#pragma unroll
    for (int j = 0; j < PB; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_in, (lds_void_g1 *)(slot + BM * 128 + (wv + j * WAVES) * 1024), 16, offb[j], s * 128, 0, 0);
}

template <typename T, int MT, int NT, int WM, int WN, int SLOTS, int OCC>   // T: bf16_t | f16_t; OCC: waves per SIMD the register allocation must allow (2 workgroups per CU: 4 for 8 waves)
__global__ __launch_bounds__(64 * WM * WN, OCC) void gemm1_kernel(Gemm1Args a)
{
    using C = Gemm1Cfg<MT, NT, WM, WN, SLOTS>;
    using E = ET<T>;
    static_assert(sizeof(T) == 2, "2-byte element types");
    __shared__ __attribute__((aligned(1024))) char smem[C::LDS];

    const int tid = threadIdx.x;
    const int l = tid & 63, r = l & 31, h = l >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv / WN, wn = wv - wm * WN;
    // neighbouring ids = the channel blocks of ONE pixel tile: they share its rows in the die's L2
    const int bid = h3d_tile_id(blockIdx.x, gridDim.x, a.xcd);
    const int tn = bid / a.blocks_m, bm = bid - tn * a.blocks_m;
    const long long p0 = (long long)tn * C::BN;
    const int cout0 = bm * C::BM;
    const int nst = a.Cin / C::KS;

    // ---- DMA sources, rebased to this workgroup's first pixel / filter row (per-lane offsets stay small) --------------
    const long long in_left = (a.N - p0) * a.in_cs * 2;
    const char *in0 = a.in + (size_t)p0 * a.in_cs * 2;
    int in_bytes = (int)(in_left < 0x7ffffff0ll ? in_left : 0x7ffffff0ll);
    const char *w0 = a.w + (size_t)cout0 * a.Cin * 2;
    const int w_bytes = (a.wrows - cout0) * a.Cin * 2;
    // lane l of a piece: row 8 * piece + (l >> 3), LDS slot l & 7 <- source slot (l & 7) ^ ((row >> 1) & 7)
    int offa[C::PA], offb[C::PB];
#pragma unroll
    for (int j = 0; j < C::PA; ++j) {
        const int row = (wv + j * C::WAVES) * 8 + (l >> 3);
        offa[j] = row * a.Cin * 2 + (((l & 7) ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < C::PB; ++j) {
        const int row = (wv + j * C::WAVES) * 8 + (l >> 3);
        offb[j] = row * a.in_cs * 2 + (((l & 7) ^ ((row >> 1) & 7)) << 4);
    }
    if (a.stride != 1) {
        // strided input: the pixel rows of the tile are output pixels (b, oy, ox) reading input pixel (s oy, s ox); offsets
        // are absolute (the launcher has checked that the whole input lies below 2 GiB), rows past the last pixel out of range
        in0 = a.in;
        in_bytes = (int)((long long)a.N / (a.Ho * a.Wo) * a.H * a.W * a.in_cs * 2);
        const int hw = a.Ho * a.Wo;
#pragma unroll
        for (int j = 0; j < C::PB; ++j) {
            const int row = (wv + j * C::WAVES) * 8 + (l >> 3);
            const long long pp = p0 + row;
            int off = 0x7ffffff0;
            if (pp < a.N) {
                const int b = (int)(pp / hw), rr = (int)(pp - (long long)b * hw);
                const int oy = rr / a.Wo, ox = rr - oy * a.Wo;
                off = ((b * a.H + oy * a.stride) * a.W + ox * a.stride) * a.in_cs * 2 + (((l & 7) ^ ((row >> 1) & 7)) << 4);
            }
            offb[j] = off;
        }
    }
    auto issue = [&](int s, char *slot) { gemm1_issue<C::PA, C::PB, C::WAVES, C::BM>(w0, w_bytes, in0, in_bytes, slot, offa, offb, wv, s); };

    // fragment offsets inside a slot: row r of my first tile, K step kk -> slot (2 kk + h) ^ ((r >> 1) & 7)
    const int sw = (r >> 1) & 7;
    int xk[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) xk[kk] = ((2 * kk + h) ^ sw) << 4;
    const int arow = (wm * MT * 32 + r) * 128;
    const int brow = C::BM * 128 + (wn * NT * 32 + r) * 128;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

#pragma unroll
    for (int j = 0; j < C::AHEAD; ++j)
        if (j < nst) issue(j, smem + j * C::SLOT);
    int cslot = 0, pslot = C::AHEAD % SLOTS;
    for (int s = 0; s < nst; ++s) {
        // my pieces of stage s have landed once only those of the (at most AHEAD - 1) younger stages are outstanding
        const int younger = min(C::AHEAD - 1, nst - 1 - s);
        if (younger <= 0) __builtin_amdgcn_s_waitcnt(0x0f70);
        else if (younger == 1) __builtin_amdgcn_s_waitcnt(0x0f70 | (C::PPW & 15) | ((C::PPW >> 4) << 14));
        else __builtin_amdgcn_s_waitcnt(0x0f70 | ((2 * C::PPW) & 15) | (((2 * C::PPW) >> 4) << 14));
        h3d_barrier_keep_vmcnt();                 // ... everyone's have; the slot of stage s-1 is no longer being read (NOT
                                                  // __syncthreads: its fence would wait for the younger stages' DMA as well)
        if (s + C::AHEAD < nst) issue(s + C::AHEAD, smem + pslot * C::SLOT);
        const char *sl = smem + cslot * C::SLOT;
        cslot = cslot + 1 == SLOTS ? 0 : cslot + 1;
        pslot = pslot + 1 == SLOTS ? 0 : pslot + 1;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            typename E::frag fa[MT], fb[NT];
#pragma unroll
            for (int m = 0; m < MT; ++m) fa[m] = E::lds_frag(sl + arow + m * 32 * 128 + xk[kk]);
#pragma unroll
            for (int n = 0; n < NT; ++n) fb[n] = E::lds_frag(sl + brow + n * 32 * 128 + xk[kk]);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) E::mma(acc[m][n], fa[m], fb[n]);
        }
    }

    // ---- epilogue: the flat pixel array as an image of width 16; wave (wm, wn) owns rows 2 NT wn .. of the tile -------
    EpiArgs e;
    e.bias = a.bias; e.res = a.res; e.out = a.out; e.Ho = (int)(a.N >> 4); e.Wo = 16; e.Cout = a.Cout;
    e.out_cs = a.out_cs; e.res_cs = a.res_cs; e.relu = a.relu; e.out_mode = H3D_OUT_NHWC;
    __syncthreads();                              // nobody reads the ring any more
    tile_epilogue_lds<T, MT, NT>(acc, e, 0, (int)(p0 >> 4), 0, cout0 + wm * MT * 32, wn, l, smem + wv * epi_lds_stride<MT, NT>());
}

template <int MT, int NT, int WM, int WN, int SLOTS, int WGS = 1>   // WGS: workgroups per CU the LDS and register budgets are cut for
static int launch_gemm1_cfg(const Gemm1Args &a0, hipStream_t st)
{
    using C = Gemm1Cfg<MT, NT, WM, WN, SLOTS>;
    static_assert(C::LDS * WGS <= 160 * 1024, "LDS budget");
    constexpr int OCC = WGS * WM * WN / 4;
    Gemm1Args a = a0;
    a.tiles_n = (int)((a.N + C::BN - 1) / C::BN);
    a.blocks_m = cdiv(a.Cout, C::BM);
    a.xcd = h3d_xcd_mode();
    if (h3d_note_kernel("gemm1_kernel<%s, %d, %d, %d, %d, %d, %d>", a.f16 ? "f16_t" : "unsigned short", MT, NT, WM, WN, SLOTS, OCC)) return H3D_OK;
    if (a.f16) hipLaunchKernelGGL((gemm1_kernel<f16_t, MT, NT, WM, WN, SLOTS, OCC>), dim3(a.tiles_n * a.blocks_m), dim3(C::THREADS), 0, st, a);
    else hipLaunchKernelGGL((gemm1_kernel<bf16_t, MT, NT, WM, WN, SLOTS, OCC>), dim3(a.tiles_n * a.blocks_m), dim3(C::THREADS), 0, st, a);
    H3D_CHECK_LAUNCH("gemm1_kernel");
    return H3D_OK;
}

// Does the GEMM kernel take this H3D_OP_CONV?  (bf16, 1x1 stride 1, whole 64-channel stages, more than 64 output channels,
// the lean NHWC epilogue, a pixel count the width-16 view covers exactly, a grid worth the tiles)
bool h3d_gemm1_takes(const h3d_op &op)
{
    if ((op.dtype != H3D_BF16 && op.dtype != H3D_F16) || op.ksize != 1 || (op.stride != 1 && op.stride != 2) || op.out_mode != H3D_OUT_NHWC) return false;
    if (op.stride == 2 && (long long)op.B * op.H * op.W * op.in_cs * 2 >= 0x7ffffff0ll) return false;   // absolute 32-bit offsets
    if (op.reserved & 0x3000) return false;                  // tuning overrides 0x1000 (tile shape), 0x2000: the halo-tile kernel of csrc/conv.hip
    const long long N = (long long)op.B * op.Ho * op.Wo;
    if (op.Cin % 64 || op.Cin < 128 || op.in_cs % 8 || op.Cout % 8 || op.out_cs % 8 || (N & 15)) return false;
    if (((uintptr_t)op.in & 15) || ((uintptr_t)op.out & 15) || ((uintptr_t)op.bias & 15) || ((uintptr_t)op.w & 15)) return false;
    if (op.in2 && (op.in2_cs % 8 || ((uintptr_t)op.in2 & 15) || N * op.in2_cs * 2 >= 0x7ffffff0ll)) return false;
    if ((long long)op.wrows * op.Cin * 2 >= 0x7ffffff0ll || (N >> 4) > 0x7fffffff || 256ll * op.in_cs * 2 >= 0x7ffffff0ll) return false;
    if (cdiv(op.Cout, 128) * 128 > op.wrows) return false;   // (h3d_launch_conv has checked this already)
    if (op.reserved & 0x4000) return true;                   // tuning override: this kernel whatever the shape
    // <= 64 output channels (DLA-34's level-2 root, ResNet's 256 -> 64): the layer streams its input once, the halo-tile
    // kernel's 4-wave tiles do that as fast or faster (tools/ab_gemm1.py: 0.075 vs 0.077-0.093 ms, 0.147 vs 0.149-0.163)
    if (op.Cout <= 64) return false;
    return ((N + 127) / 128) * cdiv(op.Cout, 128) >= 128;
}

int h3d_launch_gemm1(const h3d_op &op, hipStream_t st)
{
    Gemm1Args a;
    a.in = (const char *)op.in; a.w = (const char *)op.w; a.bias = op.bias; a.res = (const char *)op.in2; a.out = (char *)op.out;
    a.N = (long long)op.B * op.Ho * op.Wo; a.Cin = op.Cin; a.in_cs = op.in_cs; a.Cout = op.Cout; a.out_cs = op.out_cs;
    a.res_cs = op.in2_cs; a.relu = op.relu; a.wrows = op.wrows; a.tiles_n = a.blocks_m = 0; a.xcd = 0;
    a.H = op.H; a.W = op.W; a.Ho = op.Ho; a.Wo = op.Wo; a.stride = op.stride;
    a.f16 = op.dtype == H3D_F16;
    // a tile's epilogue reads the bias of ALL its channel rows unguarded, and its filter rows must exist or lie past the end
    // of the bank: a channel block of BM rows needs cdiv(Cout, BM) * BM packed rows (128 is guaranteed, see above)
    const bool ok256 = cdiv(op.Cout, 256) * 256 <= op.wrows;
    switch (op.reserved & 0xf00) {                           // tuning override (tests, tools/ab_gemm1.py)
    case 0x100:
        if (!ok256) H3D_FAIL(H3D_ERR_SHAPE, "conv 1x1 (gemm): the 256-channel tile needs %d packed rows, got %d", cdiv(op.Cout, 256) * 256, op.wrows);
        return launch_gemm1_cfg<4, 2, 2, 4, 2>(a, st);
    case 0x200: return launch_gemm1_cfg<2, 2, 2, 4, 3>(a, st);
    case 0x300: return launch_gemm1_cfg<2, 2, 2, 2, 2, 2>(a, st);
    default: break;
    }
    // 256 x 256 (8 waves of 128 x 64, one workgroup per CU) when its grid fills the chip's 256 CUs in whole rounds;
    // otherwise 128 x 128 (4 waves of 64 x 64, 64 KB: two workgroups per CU, four times the workgroups).
    // tools/ab_gemm1.py on the three networks' layers, ms: 1024 -> 256 @48x48 x32 (288 tiles = 1.1 rounds) 0.064 vs 0.060,
    // 2048 -> 512 @24x24 x32 (144) 0.067 vs 0.060, 1280 -> 512 @16x16 x64 (128) 0.041 vs 0.030; 256 -> 1024 (1152) 0.094 vs
    // 0.098, 512 -> 256 @96x96 (1152) 0.120 vs 0.130, 896 -> 256 @32x32 x64 0.037 vs 0.042; the halo-tile kernel: 0.081,
    // 0.086, 0.037; 0.120, 0.167, 0.058.
    const long long nA = ((a.N + 255) / 256) * cdiv(op.Cout, 256);
    const long long rounds = (nA + 255) / 256;
    if (ok256 && op.Cout >= 256 && nA >= 200 && 10 * nA >= 7 * rounds * 256) return launch_gemm1_cfg<4, 2, 2, 4, 2>(a, st);
    return launch_gemm1_cfg<2, 2, 2, 2, 2, 2>(a, st);
}
