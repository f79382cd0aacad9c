// Fused output heads of DLASeg (reference models/model.py:451-460, 485-489):
//     for head in heads:  z[head] = Conv1x1(head_conv -> C_head)( ReLU( Conv3x3(64 -> head_conv)(y) ) )
// The reference runs 2 cuDNN convs per head, re-reading y for every head and round-tripping the
// head_conv(=256)-channel intermediate (16.8 MB/image in fp32) through memory for each of them:
// 37 % of the network's multiply-accumulates and the largest single source of HBM traffic.
//
// Here ONE workgroup (8 waves) keeps the 64-channel input halo tile of a TH x 32 pixel tile in LDS
// for ALL heads of the launch and never writes the intermediate: each 64-channel slab of it leaves
// the MFMA accumulators as bf16 (bias + ReLU applied) and is fed straight back as the B operand of
// the 1x1 contraction (accumulator tile -> next MFMA's operand: the 32x32 C/D map puts the pixel on
// the lane and channels in the registers, exactly the B-operand shape; the K order is the
// accumulator row order, which the host bakes into the packed 1x1 weights).
//
// Weight pipeline: the 3x3 filters stream through a 2-slot LDS ring one tap ROW (3 taps, 24 KB) per
// stage by LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, no ds_write pass), issued one
// stage (48 MFMAs per wave) ahead; one barrier per stage.  The ring rows are unpadded (an LDS-DMA
// wave-instruction writes 1 KB contiguously) and XOR-swizzled through the per-lane SOURCE address
// so the A-fragment ds_read_b128 stay conflict-free.  The 1x1 slice of a slab arrives the same way.
// Outputs are written as contiguous NCHW fp32 rows (lane = pixel), the reference's head layout.
#include "common.h"
#include <type_traits>
#include <utility>

constexpr int HEADS_MAX = 16;
constexpr int HC_IN = 64;     // channels of y (DLA-34 first_level = 2)
constexpr int HC_SLAB = 64;   // intermediate channels per slab (two 32-row MFMA tiles)
constexpr int HC_MT2 = 3;     // up to 96 output channels per head
#ifndef HEADS_PF_WIDE
#define HEADS_PF_WIDE 0 // the same pipeline in the 2- and 3-tile bodies (wide heads)
#endif
#ifndef HEADS_PF
#define HEADS_PF 2      // measured 0: 1.363, 1: 1.348, 2: 1.335, 3: 1.343 ms (batch 64, the narrow-heads launch)
#endif

struct HeadsArgs {
    const char *in;      // NHWC T feature map
    const char *w1;      // [nheads*head_conv][9][64] T
    const float *b1;     // [nheads*head_conv]
    const char *w2[HEADS_MAX];   // per head: [96 rows][head_conv] T, K in accumulator-row order
    const float *b2[HEADS_MAX];  // per head: fp32 [96]
    float *out[HEADS_MAX];       // per head: NCHW fp32 [B,C,H,W]
    int C[HEADS_MAX];
    int nheads, head_conv;
    int B, H, W, in_cs;
    int tiles_x, tiles_y;
    int dbg;   // profiling ablation (h3d_op.reserved): 1 = skip the weight loads after the prologue
    int xcd;   // h3d_tile_id mode
    float s1;                // f16x3 plans: 2^-wexp of the 3x3 filters AND of b1 (h3d_heads_desc.wexp); 1 otherwise
    float s2[HEADS_MAX];     // ... 2^-wexp2 of a head's 1x1 filters
    unsigned long long *stamps;   // profiling builds: per workgroup {kernel start, end, cycles wave 0 spent in the stage barriers, in gemm2, waiting for its own DMA pieces}
};

template <typename T, int TH>
struct HeadsCfg {
    static constexpr int TW = 32;
    static constexpr int ES = sizeof(T);
    static constexpr int NT = TH / 8;                 // N-tiles (rows of 32 pixels) per wave, 8 waves
    static constexpr int IN_H = TH + 2, IN_W = TW + 2;
    static constexpr int SB = HC_IN * ES + 16;        // halo pixel stride (padded: register-staged once)
    static constexpr int RB = IN_W * SB;              // halo row stride
    static constexpr int RBW = HC_IN * ES;            // weight row bytes (64 K elements), UNPADDED: LDS-DMA image
    static constexpr int CPR = RBW / 16;              // 16-byte chunks per weight row (8 bf16 / 16 f32)
    static constexpr int SWZ_SHIFT = ES == 2 ? 1 : 0; // chunk' = chunk ^ ((row >> SWZ_SHIFT) & (CPR-1))
    static constexpr int TAPS = ES == 2 ? 3 : 1;      // taps per stage (f32: LDS only fits one)
    static constexpr int TRS = 9 / TAPS;              // stages per slab
    static constexpr int SLOT = TAPS * HC_SLAB * RBW; // ring slot bytes
    static constexpr int LDS_IN = IN_H * RB;
    static constexpr int LDS_W2 = 32 * HC_MT2 * RBW;  // 96 rows
    static constexpr int LDS_BIAS = 1024 + 1024;      // per head: b1 (<= 256 floats) + b2 (96 floats; its DMA piece writes a whole KiB), two heads in flight
    static constexpr int LDS = LDS_IN + 2 * SLOT + LDS_W2 + 2 * LDS_BIAS;
    static constexpr int RPI = 1024 / RBW;            // weight rows per LDS-DMA wave-instruction
    static constexpr int VPR = HC_IN * ES / 16;       // 16-byte vectors per halo pixel
};

// 16-byte LDS read the compiler's wait-count pass does not see (see heads_kernel's stage): the caller waits.
template <int OFF>
__device__ __forceinline__ void lds_read16_async(u32x4 &dst, uint32_t addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// Biases of one head -> LDS by LDS-DMA (wave 0: b1[head_conv] <= 1 KiB, wave 1: b2[96]); lanes past the end of either
// array carry offsets outside the buffer and write zeros.  Reading them from global memory where they are used put
// a full memory round trip at the end of every slab and before every group of output stores.
__device__ __forceinline__ void heads_issue_bias(const float *b1, int hc, const float *b2, char *dst, int wv, int l)
{
    if (wv == 0) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)b1, 0, hc * 4, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)dst, 16, l * 16, 0, 0, 0);
    } else if (wv == 1) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)b2, 0, 96 * 4, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(dst + 1024), 16, l * 16 < 512 ? l * 16 : 0x7ffffff0, 0, 0, 0);
    }
}

// Second contraction of a head: acc2[m2] += W2[rows of tile m2][slab K] . ReLU(acc + b1), then acc = 0.
// The accumulator registers 8s..8s+7 of a 32x32 tile are K-step s of the B operand (lane = pixel);
// the matching A fragment (K in accumulator-row order) sits at K offset m*32 + h*16 + s*8 of the
// host-permuted W2 row (swizzled LDS image, see HeadsCfg).
// BIASC: the accumulators were STARTED at b1 (C operand of the slab's first MFMA, see heads_kernel), so the slab leaves
// them as conv + bias and nothing is added or zeroed here; on bf16 the ReLU runs on the packed pairs (v_pk_max_i16
// against 0: a negative bf16 is a negative int16, and rounding never changes the sign).  That is 64 vector
// instructions per slab and wave instead of 224 -- all eight waves do this at the same time, right after a barrier,
// with the MFMA pipe idle.
template <typename T, int TH, int NT, int M2, bool BIASC>
__device__ __forceinline__ void gemm2(f32x16 (&acc)[2][NT], f32x16 (&acc2)[M2][NT], const char *s_b1 /* LDS: this slab's 64 b1 */,
                                      const char *s_w2, int r, int h, int sw, float s1 = 1.f)
{
    using C = HeadsCfg<T, TH>;
    using E = ET<T>;
    constexpr int ES = sizeof(T);
    typedef short s16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            typename E::frag fb[NT];
            // accumulator registers 8sb..8sb+7 hold channels m*32 + 16sb + 4h + {0..3} and + 8 + {0..3}
            float bias[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if constexpr (!BIASC) {
                const f32x4 b_lo = *reinterpret_cast<const f32x4 *>(s_b1 + (m * 32 + 16 * sb + 4 * h) * 4);
                const f32x4 b_hi = *reinterpret_cast<const f32x4 *>(s_b1 + (m * 32 + 16 * sb + 4 * h + 8) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { bias[j] = b_lo[j]; bias[4 + j] = b_hi[j]; }
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if constexpr (BIASC) {
                        x[j] = acc[m][n][8 * sb + j];
                    } else {
                        x[j] = fmaxf(acc[m][n][8 * sb + j] + bias[j], 0.f);
                        acc[m][n][8 * sb + j] = 0.f;
                    }
                }
                if constexpr (ES == 4) {
                    if constexpr (BIASC) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) x[j] = fmaxf(x[j], 0.f);
                    }
                    if constexpr (std::is_same_v<T, x3_t>) {
                        // f16x3: the fp32 slab values, unscaled (the 3x3 filters and b1 were packed times 2^wexp), split into (hi | lo) fp16 terms
#pragma unroll
                        for (int j = 0; j < 8; ++j) x[j] *= s1;
                        fb[n] = E::split8(x);
                    } else {
                        fb[n].lo = f32x4{x[0], x[1], x[2], x[3]};
                        fb[n].hi = f32x4{x[4], x[5], x[6], x[7]};
                    }
                } else {
                    if constexpr (std::is_same_v<T, f16_t>) {       // fp16: ReLU and the saturation at 65504 in one v_med3_f32 each
#pragma unroll
                        for (int j = 0; j < 8; ++j) x[j] = __builtin_amdgcn_fmed3f(x[j], 0.f, 65504.f);
                    }
                    uint32_t pk[4] = {EP<T>::pack2(x[0], x[1]), EP<T>::pack2(x[2], x[3]), EP<T>::pack2(x[4], x[5]), EP<T>::pack2(x[6], x[7])};
                    if constexpr (BIASC && !std::is_same_v<T, f16_t>) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            pk[j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk[j]), s16x2{0, 0}));
                    }
                    fb[n].v = u32x4{pk[0], pk[1], pk[2], pk[3]};
                }
            }
            const int c = (m * 32 + h * 16 + sb * 8) * ES / 16;        // first 16-byte chunk of this K run
#pragma unroll
            for (int m2 = 0; m2 < M2; ++m2) {
                const char *row = s_w2 + (m2 * 32 + r) * C::RBW;
                const typename E::frag fa = E::lds_frag2(row + ((c ^ sw) << 4), row + (((c + 1) ^ sw) << 4));
#pragma unroll
                for (int n = 0; n < NT; ++n) E::mma(acc2[m2][n], fa, fb[n]);
            }
        }
    }
}

// LDS-DMA of one weight stage / one 1x1 slice in the BUFFER form: per-lane 32-bit offsets that do not depend on the stage
// (row, tap and swizzled chunk of the lane's 16 bytes) + a scalar stage offset, instead of a 64-bit per-lane address rebuilt
// for every piece (global_load_lds): an LDS-DMA instruction costs its wave 100-185 issue cycles as it is, and the address
// arithmetic in front of each of the 3 + 2 pieces per wave and stage came on top (the weight stream was 10 % of the kernel:
// ablation in DESIGN.md 2.3).  Plain functions of plain arguments: a lambda capturing a buffer descriptor drops the host stub.
template <typename T, int TH>
__device__ __forceinline__ void heads_dma_w1(const char *w1, int w1_bytes, char *slot, int wv, int l, int soff)
{
    using C = HeadsCfg<T, TH>;
    constexpr int ES = sizeof(T);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)w1, 0, w1_bytes, 0x00020000);
    constexpr int NI = C::TAPS * HC_SLAB / C::RPI;          // wave-instructions per stage (24 / 16)
#pragma unroll
    for (int j = 0; j < NI / 8; ++j) {
        const int g = j * 8 + wv;                            // 1 KB run index
        const int row_all = g * C::RPI + l / C::CPR;         // = tap_in_stage * 64 + row
        const int tp = row_all / HC_SLAB, row = row_all - tp * HC_SLAB;
        const int c = (l % C::CPR) ^ ((row >> C::SWZ_SHIFT) & (C::CPR - 1));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(slot + g * 1024), 16,
                                                 ((row * 9 + tp) * HC_IN) * ES + c * 16, soff, 0, 0);
    }
}
template <typename T, int TH>
__device__ __forceinline__ void heads_dma_w2(const char *w2, int head_conv, char *s_w2, int wv, int l, int soff)
{
    using C = HeadsCfg<T, TH>;
    constexpr int ES = sizeof(T);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)w2, 0, 32 * HC_MT2 * head_conv * ES, 0x00020000);
    constexpr int NI = 32 * HC_MT2 / C::RPI;                 // 12 / 24 wave-instructions
#pragma unroll
    for (int j = 0; j < (NI + 7) / 8; ++j) {
        const int g = j * 8 + wv;
        if (g < NI) {
            const int row = g * C::RPI + l / C::CPR;
            const int c = (l % C::CPR) ^ ((row >> C::SWZ_SHIFT) & (C::CPR - 1));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(s_w2 + g * 1024), 16,
                                                     row * head_conv * ES + c * 16, soff, 0, 0);
        }
    }
}

template <typename T, int TH, int M2, bool BIASC, bool MIXED = false>
__global__ __launch_bounds__(512) void heads_kernel(HeadsArgs a)
{
    using C = HeadsCfg<T, TH>;
    using E = ET<T>;
    constexpr int ES = C::ES;
    constexpr int NT = C::NT;
    __shared__ __attribute__((aligned(16))) char smem[C::LDS];
    char *s_in = smem;
    char *s_ring = smem + C::LDS_IN;
    char *s_w2 = s_ring + 2 * C::SLOT;
    char *s_bias = s_w2 + C::LDS_W2;

    const int tid = threadIdx.x;
    const int wv = tid >> 6, l = tid & 63, r = l & 31, h = l >> 5;
    const int tiles = a.tiles_x * a.tiles_y;
    const int bid = h3d_tile_id(blockIdx.x, gridDim.x, a.xcd);
    const int b = bid / tiles;
    const int t = bid - b * tiles;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * C::TW;
    const int slabs = a.head_conv / HC_SLAB;
    const int nstages = a.nheads * slabs * C::TRS;       // tap-row stages of the whole launch
    const int sw = (r >> C::SWZ_SHIFT) & (C::CPR - 1);   // XOR swizzle of this lane's A-fragment rows

    // ---- LDS-DMA of stage s: TAPS taps x 64 rows of W1 for (head, slab) -> ring slot s & 1 -----------
    // lane i of a wave-instruction lands at byte i*16 of a 1 KB run = RPI rows; it fetches source
    // chunk (i % CPR) ^ swz(row) so that LDS position p of a row holds chunk p ^ swz(row).
    const int wvs = __builtin_amdgcn_readfirstlane(wv);
    const int w1_bytes = a.nheads * a.head_conv * 9 * HC_IN * ES;
    auto issue_w1 = [&](int s) {
        if ((H3D_DBG(a) & 1) && s > 0) return;
        const int hs = s / C::TRS, tr = s - hs * C::TRS;        // hs = head * slabs + slab
        heads_dma_w1<T, TH>(a.w1, w1_bytes, s_ring + (s & 1) * C::SLOT, wvs, l, ((hs * HC_SLAB * 9 + tr * C::TAPS) * HC_IN) * ES);
    };
    // ---- LDS-DMA of the 1x1 slice [96 rows][64 K of this slab] -> s_w2 --------------------------------
    auto issue_w2 = [&](int head, int slab) {
        if (H3D_DBG(a) & 1) return;
        heads_dma_w2<T, TH>(a.w2[head], a.head_conv, s_w2, wvs, l, slab * HC_SLAB * ES);
    };

    // ---- prologue: stage 0 weights in flight, halo tile (all 64 channels, zero outside the image) -----
    issue_w1(0);
    heads_issue_bias(a.b1, a.head_conv, a.b2[0], s_bias, __builtin_amdgcn_readfirstlane(wv), l);
    {
        const size_t in_img = (size_t)b * a.H * a.W;
        constexpr int NHV = C::IN_H * C::IN_W * C::VPR;
        constexpr int HALF = (NHV + 1) / 2;                 // two batches bound the staging registers
        for (int part = 0; part < 2; ++part) {
            const int base = part * HALF;
            stage_vectors<HALF, 512>(
                tid,
                [&](int j) -> u32x4 {
                    const int i = base + j;
                    if (i >= NHV) return u32x4{0u, 0u, 0u, 0u};
                    const int v = i % C::VPR, pix = i / C::VPR;
                    const int iy = pix / C::IN_W, ix = pix - iy * C::IN_W;
                    const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
                    if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                        return *reinterpret_cast<const u32x4 *>(a.in + ((in_img + (size_t)gy * a.W + gx) * a.in_cs) * ES + v * 16);
                    return u32x4{0u, 0u, 0u, 0u};
                },
                [&](int j, u32x4 val) {
                    const int i = base + j;
                    if (i < NHV) {
                        const int v = i % C::VPR, pix = i / C::VPR;
                        const int iy = pix / C::IN_W, ix = pix - iy * C::IN_W;
                        if constexpr (std::is_same_v<T, x3_t>) x3_store4(s_in + iy * C::RB + ix * C::SB + (v >> 1) * 32, v & 1, val);
                        else *reinterpret_cast<u32x4 *>(s_in + iy * C::RB + ix * C::SB + v * 16) = val;
                    }
                });
        }
    }
    __syncthreads();   // (emits s_waitcnt vmcnt(0): the LDS-DMA of stage 0 has landed)

    int boff[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) boff[n] = (wv * NT + n) * C::RB + r * C::SB + 8 * h * ES;

#ifdef H3D_ABLATE
    unsigned long long t_bar = 0, t_g2 = 0, t_dma = 0, t_stage = 0;
    const unsigned long long t_start = __builtin_readcyclecounter();
#endif
    int s = 0;
    // one head: M2H = 32-row tiles of ITS 1x1 output.  MIXED launches (all heads of the model in one launch, so that the halo
    // tile is staged once per workgroup for all of them) instantiate the body for every width up to M2 and pick per head
    auto head_body = [&](auto m2_tag, int head) {
        constexpr int M2H = decltype(m2_tag)::value;
        const char *s_b = s_bias + (head & 1) * C::LDS_BIAS;
        f32x16 acc[2][NT], acc2[M2H][NT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
#pragma unroll
        for (int m = 0; m < M2H; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc2[m][n][i] = 0.f;
        for (int slab = 0; slab < slabs; ++slab) {
            // one tap-row stage; FIRST: the slab's first MFMA of every accumulator tile takes its C operand from the
            // bias tile (accumulator row = channel), so neither a zeroing pass nor a bias pass exists
            auto stage = [&](auto first_tag, const char *slot, int tr) {
                constexpr bool FIRST = decltype(first_tag)::value;
                // the bias tile goes from LDS straight INTO the accumulators (accumulator row = channel).  Round 3 built it in 32
                // registers first and copied it into each N-tile: those 32 registers were live together with the fragment
                // pipeline and pushed the 3-tile body into scratch (256 VGPRs + 24-36 bytes spilled)
                auto load_bias = [&](f32x16 &dst, int m) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 bv = *reinterpret_cast<const f32x4 *>(s_b + (slab * HC_SLAB + m * 32 + 8 * g + 4 * h) * 4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) dst[4 * g + i] = bv[i];
                    }
                };
                constexpr int NSTEP = C::TAPS * (HC_IN / 16);
                auto plain = [&]() {
#pragma unroll
                    for (int tp = 0; tp < C::TAPS; ++tp) {
                        const int tap = tr * C::TAPS + tp;
                        const int dy = tap / 3, dx = tap - dy * 3;
                        const char *br = s_in + dy * C::RB + dx * C::SB;
#pragma unroll
                        for (int kk = 0; kk < HC_IN / 16; ++kk) {
                            const int c = (kk * 16 + 8 * h) * ES / 16;
                            typename E::frag fa[2], fb[NT];
#pragma unroll
                            for (int m = 0; m < 2; ++m) {
                                const char *row = slot + (tp * HC_SLAB + m * 32 + r) * C::RBW;
                                fa[m] = E::lds_frag2(row + ((c ^ sw) << 4), row + (((c + 1) ^ sw) << 4));
                            }
#pragma unroll
                            for (int n = 0; n < NT; ++n) fb[n] = E::lds_frag(br + boff[n] + kk * 16 * ES);
#pragma unroll
                            for (int m = 0; m < 2; ++m)
#pragma unroll
                                for (int n = 0; n < NT; ++n) {
                                    if (FIRST && tp == 0 && kk == 0) { load_bias(acc[m][n], m); }
                                    E::mma(acc[m][n], fa[m], fb[n]);
                                }
                        }
                    }
                                };
                // (two levels: inside this generic lambda only a condition that does not depend on M2H keeps the fp32
                // instantiation from type-checking the 2-byte fragment code)
                if constexpr (ES == 2) {
                if constexpr ((M2H == 1 && HEADS_PF > 0) || (M2H > 1 && HEADS_PF_WIDE > 0)) {
                    // bf16, narrow heads (the 3-tile instantiation spills with it, in the 2-tile one it measures the same as the
                    // compiler's schedule: 0.279 / 0.287 vs 0.280 ms at depth 2 / 1): explicit software pipeline.  The fragments of step i+1 (a step = one tap x 16 channels: 2 filter
                    // + NT pixel fragments feeding 2*NT MFMAs) are requested BEFORE the MFMAs of step i and waited for with a
                    // counted s_waitcnt.  Left to itself hipcc issues every ds_read right in front of its MFMA behind
                    // lgkmcnt(0) -- and does so even when the loads are hoisted in the source -- so the reads are inline asm
                    // (invisible to its wait-count pass) and the waits are ours: LDS returns in order, NT + 2 newer reads
                    // may stay in flight.  sched_barrier pins the MFMAs behind their wait.
                    const uint32_t ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)slot + r * C::RBW;
                    uint32_t fa_base[HC_IN / 16];
#pragma unroll
                    for (int kk = 0; kk < HC_IN / 16; ++kk) fa_base[kk] = ring + (((2 * kk + h) ^ sw) << 4);
                    const uint32_t fb_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)s_in + boff[0] + tr * C::RB;
                    constexpr int PF = M2H == 1 ? HEADS_PF : HEADS_PF_WIDE, NBUF = PF + 1;          // steps in flight ahead of the MFMAs
                    static_assert(PF * (NT + 2) <= 15, "lgkmcnt field");
                    u32x4 fa[NBUF][2], fb[NBUF][NT];
                    auto fetch = [&](auto step_tag) {
                        constexpr int STEP = decltype(step_tag)::value, BUF = STEP % NBUF;
                        constexpr int tp = STEP / (HC_IN / 16), kk = STEP - tp * (HC_IN / 16);
                        lds_read16_async<(tp * HC_SLAB) * C::RBW>(fa[BUF][0], fa_base[kk]);
                        lds_read16_async<(tp * HC_SLAB + 32) * C::RBW>(fa[BUF][1], fa_base[kk]);
                        static_for<NT>([&](auto n_tag) {
                            constexpr int n = decltype(n_tag)::value;
                            lds_read16_async<n * C::RB + tp * C::SB + kk * 32>(fb[BUF][n], fb_base);
                        });
                    };
                    static_for<PF>([&](auto step_tag) { fetch(step_tag); });
                    static_for<NSTEP>([&](auto step_tag) {
                        constexpr int STEP = decltype(step_tag)::value, BUF = STEP % NBUF;
                        if constexpr (STEP + PF < NSTEP) fetch(std::integral_constant<int, STEP + PF>{});
                        constexpr int NEWER = (STEP + PF < NSTEP ? PF : NSTEP - 1 - STEP) * (NT + 2);   // reads issued after step STEP's
                        __builtin_amdgcn_s_waitcnt(0xc07f | (NEWER << 8));                              // lgkmcnt(NEWER)
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int m = 0; m < 2; ++m)
#pragma unroll
                            for (int n = 0; n < NT; ++n) {
                                if (FIRST && STEP == 0) { load_bias(acc[m][n], m); }
                                typename E::frag a_, b_;
                                a_.v = fa[BUF][m]; b_.v = fb[BUF][n];
                                E::mma(acc[m][n], a_, b_);
                            }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                } else {
                    plain();
                }
                } else if constexpr (std::is_same_v<T, x3_t>) {
                    // (a third ring slot for the one-tile heads -- the 64 rows of the 1x1 slice they never read pay for it --, filters two
                    //  stages ahead behind counted vmcnt waits: 4.364 -> 4.381 ms, tools/ab_lib.py; the one-tap stages do not wait for their DMA)
                    // f16x3: one tap x 64 channels per stage = 4 steps of (2 filter + NT pixel fragments, 32 bytes each) feeding 6 NT
                    // MFMAs.  Two fragment sets, step k+1 read while step k multiplies (the compiler's own schedule: two 16-byte reads,
                    // lgkmcnt(0), three MFMAs); the fragments are plain loads here, the order is pinned by sched_group_barrier
                    const int dy = tr / 3, dx = tr - dy * 3;
                    const char *br = s_in + dy * C::RB + dx * C::SB;
                    typename E::frag fa[2][2], fb[2][NT];
                    auto rd = [&](int kk, int q) {
                        const int c = (kk * 16 + 8 * h) * ES / 16;
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const char *row = slot + (m * 32 + r) * C::RBW;
                            fa[q][m] = E::lds_frag2(row + ((c ^ sw) << 4), row + (((c + 1) ^ sw) << 4));
                        }
#pragma unroll
                        for (int n = 0; n < NT; ++n) fb[q][n] = E::lds_frag(br + boff[n] + kk * 16 * ES);
                    };
                    constexpr int NRD = 2 * (2 + NT), NMM = 6 * NT;
                    rd(0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
#pragma unroll
                    for (int kk = 0; kk < HC_IN / 16; ++kk) {
                        if (kk + 1 < HC_IN / 16) rd(kk + 1, (kk + 1) & 1);
#pragma unroll
                        for (int m = 0; m < 2; ++m)
#pragma unroll
                            for (int n = 0; n < NT; ++n) {
                                if (FIRST && kk == 0) { load_bias(acc[m][n], m); }
                                E::mma(acc[m][n], fa[kk & 1][m], fb[kk & 1][n]);
                            }
                        if (kk + 1 < HC_IN / 16) {
#pragma unroll
                            for (int i = 0; i < (NRD < NMM ? NRD : NMM); ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            }
                            if (NMM > NRD) __builtin_amdgcn_sched_group_barrier(0x008, NMM - NRD, 0);
                        } else {
                            __builtin_amdgcn_sched_group_barrier(0x008, NMM, 0);
                        }
                    }
                } else {
                    plain();
                }
            };
#pragma unroll 1
            for (int tr = 0; tr < C::TRS; ++tr, ++s) {
                if (s + 1 < nstages) issue_w1(s + 1);          // slot (s+1)&1 was last read in stage s-1
                if (tr == (C::TRS > 1 ? 1 : 0)) issue_w2(head, slab);   // after the barrier that follows the previous gemm2
                if (slab == 0 && tr == C::TRS - 1 && head + 1 < a.nheads)   // next head's biases: their slot was last read in the
                    heads_issue_bias(a.b1 + (head + 1) * a.head_conv, a.head_conv, a.b2[head + 1],   // previous head's epilogue, at least one barrier ago
                                     s_bias + ((head + 1) & 1) * C::LDS_BIAS, __builtin_amdgcn_readfirstlane(wv), l);
                const char *slot = s_ring + (s & 1) * C::SLOT;
#ifdef H3D_ABLATE
                const unsigned long long ts0 = __builtin_readcyclecounter();
#endif
                if ((H3D_DBG(a) & 4) && wv >= 4) { /* profiling: only the older wave of each SIMD computes (timing only) */ }
                else if ((H3D_DBG(a) & 8) && wv < 4) { /* profiling: only the younger wave computes */ }
                else if (BIASC && tr == 0) stage(std::true_type{}, slot, tr);
                else stage(std::false_type{}, slot, tr);
#ifdef H3D_ABLATE
                t_stage += __builtin_readcyclecounter() - ts0;
#endif
#ifdef H3D_ABLATE
                const unsigned long long tb0 = __builtin_readcyclecounter();
                __builtin_amdgcn_s_waitcnt(0x0f70);          // my own DMA pieces of the next stage
                const unsigned long long tb1 = __builtin_readcyclecounter();
                t_dma += tb1 - tb0;
#endif
                if (!(H3D_DBG(a) & 2))   // (profiling: 2 = no stage barrier -- wrong results, timing only)
                __syncthreads();   // vmcnt(0) + barrier: next stage's weights (and the 1x1 slice) have landed
#ifdef H3D_ABLATE
                t_bar += __builtin_readcyclecounter() - tb1;
#endif
            }
            // ---- slab done: X = ReLU(acc [+ b1]) -> B operand; acc2 += W2[:, slab] . X ----------------------
#ifdef H3D_ABLATE
            const unsigned long long tg0 = __builtin_readcyclecounter();
#endif
            gemm2<T, TH, NT, M2H, BIASC>(acc, acc2, s_b + slab * HC_SLAB * 4, s_w2, r, h, sw, a.s1);
#ifdef H3D_ABLATE
            t_g2 += __builtin_readcyclecounter() - tg0;
#endif
            if (C::TRS == 1) __syncthreads();   // (f32 path) s_w2 is rewritten in the very next stage
        }
        // ---- head done: z = acc2 + b2 -> NCHW fp32 (lane = pixel: coalesced rows) ------------------------
        const int C_head = a.C[head];
        float *out = a.out[head];
        const size_t cstride = (size_t)a.H * a.W;
        [[maybe_unused]] const float s2 = a.s2[head];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int oy = oy0 + wv * NT + n, ox = ox0 + r;
            if (oy < a.H && ox < a.W) {
                float *op = out + (((size_t)b * C_head + 4 * h) * a.H + oy) * a.W + ox;
#pragma unroll
                for (int m2 = 0; m2 < M2H; ++m2) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c = m2 * 32 + 8 * g + 4 * h;
                        float *p = op + (size_t)(m2 * 32 + 8 * g) * cstride;
                        const f32x4 b2v = *reinterpret_cast<const f32x4 *>(s_b + 1024 + c * 4);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (c + i < C_head) {
                                if constexpr (std::is_same_v<T, x3_t>) p[i * cstride] = fmaf(acc2[m2][n][4 * g + i], s2, b2v[i]);
                                else p[i * cstride] = acc2[m2][n][4 * g + i] + b2v[i];
                            }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        };
    for (int head = 0; head < a.nheads; ++head) {
        if constexpr (!MIXED) {
            head_body(std::integral_constant<int, M2>{}, head);
        } else {
            const int m2h = __builtin_amdgcn_readfirstlane((a.C[head] + 31) >> 5);
            if (m2h <= 1) head_body(std::integral_constant<int, 1>{}, head);
            else if (M2 < 3 || m2h == 2) head_body(std::integral_constant<int, (M2 < 2 ? 1 : 2)>{}, head);
            else head_body(std::integral_constant<int, M2>{}, head);
        }
    }
#ifdef H3D_ABLATE
    if (a.stamps && blockIdx.x < 65536) {
        if (a.dbg & 64) {                       // per-wave barrier time instead (tools/stamp_heads.py --waves)
            if (l == 0) a.stamps[blockIdx.x * H3D_NSTAMP + wv] = t_bar;
        } else if (tid == 0) {
            a.stamps[blockIdx.x * H3D_NSTAMP + 0] = t_start;
            a.stamps[blockIdx.x * H3D_NSTAMP + 1] = __builtin_readcyclecounter();
            a.stamps[blockIdx.x * H3D_NSTAMP + 2] = t_bar;
            a.stamps[blockIdx.x * H3D_NSTAMP + 3] = t_g2;
            a.stamps[blockIdx.x * H3D_NSTAMP + 4] = t_dma;
            a.stamps[blockIdx.x * H3D_NSTAMP + 5] = t_stage;
        }
    }
#endif
}

static_assert(HEADS_MAX == H3D_HEADS_MAX, "header/kernel mismatch");

int h3d_launch_heads(const h3d_op &op, hipStream_t st)
{
    if (!op.in || !op.w || !op.bias || !op.in2) H3D_FAIL(H3D_ERR_ARG, "heads: null pointer");
    const h3d_heads_desc *d = (const h3d_heads_desc *)op.in2;
    const int es = h3d_dtype_bytes(op.dtype);
    if (!es) H3D_FAIL(H3D_ERR_DTYPE, "heads: dtype %d", op.dtype);
    if (op.Cin != HC_IN || op.in_cs % (16 / es) || op.in_cs < HC_IN)
        H3D_FAIL(H3D_ERR_SHAPE, "heads: input must have %d channels (got %d, stride %d)", HC_IN, op.Cin, op.in_cs);
    if (op.Cout <= 0 || op.Cout % HC_SLAB || op.Cout > 256)
        H3D_FAIL(H3D_ERR_SHAPE, "heads: head_conv %d must be a multiple of %d, at most 256", op.Cout, HC_SLAB);
    if (d->nheads <= 0 || d->nheads > HEADS_MAX) H3D_FAIL(H3D_ERR_SHAPE, "heads: %d heads (max %d)", d->nheads, HEADS_MAX);
    HeadsArgs a;
    a.in = (const char *)op.in; a.w1 = (const char *)op.w; a.b1 = op.bias;
    a.dbg = op.reserved & 0xff;
    a.xcd = h3d_xcd_mode();
#ifdef H3D_ABLATE
    a.stamps = h3d_stamp_buffer();
#else
    a.stamps = nullptr;
#endif
    a.nheads = d->nheads; a.head_conv = op.Cout; a.B = op.B; a.H = op.H; a.W = op.W; a.in_cs = op.in_cs;
    for (int i = 0; i < d->nheads; ++i) {
        if (!d->head[i].w2 || !d->head[i].b2 || !d->head[i].out) H3D_FAIL(H3D_ERR_ARG, "heads: head %d null pointer", i);
        if (d->head[i].C <= 0 || d->head[i].C > 32 * HC_MT2)
            H3D_FAIL(H3D_ERR_UNSUPPORTED, "heads: head %d has %d channels (max %d)", i, d->head[i].C, 32 * HC_MT2);
        a.w2[i] = (const char *)d->head[i].w2; a.b2[i] = d->head[i].b2; a.out[i] = d->head[i].out; a.C[i] = d->head[i].C;
        if (d->head[i].wexp2 < -60 || d->head[i].wexp2 > 60 || (d->head[i].wexp2 && op.dtype != H3D_F16X3))
            H3D_FAIL(H3D_ERR_ARG, "heads: head %d wexp2 %d (an H3D_F16X3 filter exponent)", i, d->head[i].wexp2);
        a.s2[i] = ldexpf(1.f, -d->head[i].wexp2);
    }
    if (d->wexp < -60 || d->wexp > 60 || (d->wexp && op.dtype != H3D_F16X3)) H3D_FAIL(H3D_ERR_ARG, "heads: wexp %d (an H3D_F16X3 filter exponent)", d->wexp);
    a.s1 = ldexpf(1.f, -d->wexp);
    int m2 = 1, m2min = HC_MT2;
    for (int i = 0; i < d->nheads; ++i) {
        m2 = max(m2, (d->head[i].C + 31) / 32);   // row tiles of the widest head
        m2min = min(m2min, (d->head[i].C + 31) / 32);
    }
    // heads of different widths in one launch: the kernel picks the 1 / 2 / 3-tile body per head (2-byte plans)
    const bool mixed = m2min != m2 && es == 2;
    const int th = es == 2 ? 16 : 8;
    a.tiles_x = cdiv(op.W, 32); a.tiles_y = cdiv(op.H, th);
    const dim3 grid(op.B * a.tiles_x * a.tiles_y), blk(512);
    const bool biasc = !(op.reserved & 0x200);    // tuning override (tools/ab_heads.py): 0x200 = separate bias / zeroing pass
    if (h3d_note_kernel(mixed ? "heads_kernel<%s, %d, %d, %s, true>" : "heads_kernel<%s, %d, %d, %s>",
                        op.dtype == H3D_BF16 ? "unsigned short" : op.dtype == H3D_F16 ? "f16_t" : op.dtype == H3D_F16X3 ? "x3_t" : "float", th, m2, biasc ? "true" : "false"))
        return H3D_OK;
    if (mixed) {
        if (!biasc) H3D_FAIL(H3D_ERR_UNSUPPORTED, "heads: the separate-bias tuning override applies to launches of equal-width heads");
        if (op.dtype == H3D_BF16) {
            if (m2 == 2) hipLaunchKernelGGL((heads_kernel<bf16_t, 16, 2, true, true>), grid, blk, 0, st, a);
            else hipLaunchKernelGGL((heads_kernel<bf16_t, 16, 3, true, true>), grid, blk, 0, st, a);
        } else {
            if (m2 == 2) hipLaunchKernelGGL((heads_kernel<f16_t, 16, 2, true, true>), grid, blk, 0, st, a);
            else hipLaunchKernelGGL((heads_kernel<f16_t, 16, 3, true, true>), grid, blk, 0, st, a);
        }
        H3D_CHECK_LAUNCH("heads_kernel");
        return H3D_OK;
    }
#define H3D_HEADS_LAUNCH(T, TH, M2)                                                                 \
    do {                                                                                             \
        if (biasc) hipLaunchKernelGGL((heads_kernel<T, TH, M2, true>), grid, blk, 0, st, a);         \
        else hipLaunchKernelGGL((heads_kernel<T, TH, M2, false>), grid, blk, 0, st, a);              \
    } while (0)
    if (op.dtype == H3D_BF16) {
        if (m2 == 1) H3D_HEADS_LAUNCH(bf16_t, 16, 1);
        else if (m2 == 2) H3D_HEADS_LAUNCH(bf16_t, 16, 2);
        else H3D_HEADS_LAUNCH(bf16_t, 16, 3);
    } else if (op.dtype == H3D_F16) {
        if (m2 == 1) H3D_HEADS_LAUNCH(f16_t, 16, 1);
        else if (m2 == 2) H3D_HEADS_LAUNCH(f16_t, 16, 2);
        else H3D_HEADS_LAUNCH(f16_t, 16, 3);
    } else if (op.dtype == H3D_F16X3) {
        if (m2 == 1) H3D_HEADS_LAUNCH(x3_t, 8, 1);
        else if (m2 == 2) H3D_HEADS_LAUNCH(x3_t, 8, 2);
        else H3D_HEADS_LAUNCH(x3_t, 8, 3);
    } else {
        if (m2 == 1) H3D_HEADS_LAUNCH(float, 8, 1);
        else if (m2 == 2) H3D_HEADS_LAUNCH(float, 8, 2);
        else H3D_HEADS_LAUNCH(float, 8, 3);
    }
#undef H3D_HEADS_LAUNCH
    H3D_CHECK_LAUNCH("heads_kernel");
    return H3D_OK;
}
