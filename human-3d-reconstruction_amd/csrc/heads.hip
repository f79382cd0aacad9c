// Fused output heads of DLASeg (reference models/model.py:451-460, 485-489):
//     for head in heads:  z[head] = Conv1x1(head_conv -> C_head)( ReLU( Conv3x3(64 -> head_conv)(y) ) )
// The reference runs 2 cuDNN convs per head, re-reading y for every head and round-tripping the
// head_conv(=256)-channel intermediate (16.8 MB/image in fp32) through memory for each of them:
// 37 % of the network's multiply-accumulates and the largest single source of HBM traffic.
//
// Here ONE workgroup keeps the 64-channel input halo tile of a TH x 32 pixel tile in LDS for ALL
// heads, streams the 3x3 weights tap by tap through a 2-slot LDS ring (prefetched into registers
// one stage ahead), and never writes the intermediate: each 64-channel slab of it leaves the
// MFMA accumulators as bf16 (bias + ReLU applied) and is fed straight back as the B operand of
// the 1x1 contraction (accumulator tile -> next MFMA's operand: the 32x32 C/D map puts the
// pixel on the lane and channels in the registers, exactly the B-operand shape; the K order is
// the accumulator row order, which the host bakes into the packed 1x1 weights).
// Outputs are written as contiguous NCHW fp32 rows (lane = pixel), the reference's head layout.
#include "common.h"

constexpr int HEADS_MAX = 16;
constexpr int HC_IN = 64;     // channels of y (DLA-34 first_level = 2)
constexpr int HC_SLAB = 64;   // intermediate channels per slab (two 32-row MFMA tiles)
constexpr int HC_MT2 = 3;     // up to 96 output channels per head

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
};

template <typename T, int TH>
struct HeadsCfg {
    static constexpr int TW = 32;
    static constexpr int ES = sizeof(T);
    static constexpr int NT = TH / 8;                 // N-tiles (rows of 32 pixels) per wave, 8 waves
    static constexpr int IN_H = TH + 2, IN_W = TW + 2;
    static constexpr int SB = HC_IN * ES + 16;        // halo pixel stride
    static constexpr int RB = IN_W * SB;              // halo row stride
    static constexpr int WB = HC_IN * ES + 16;        // ring row stride (64 K-elements of one tap)
    static constexpr int W2B = HC_SLAB * ES + 16;     // W2 slab row stride
    static constexpr int LDS_IN = IN_H * RB;
    static constexpr int LDS_RING = HC_SLAB * WB;     // one slot: 64 rows
    static constexpr int LDS_W2 = 32 * HC_MT2 * W2B;  // 96 rows
    static constexpr int LDS = LDS_IN + 2 * LDS_RING + LDS_W2;
    static constexpr int VPR = HC_IN * ES / 16;       // 16-byte vectors per 64-element row
    static constexpr int NSTG = (32 * HC_MT2 * VPR + 511) / 512;   // staging vectors per thread (W2 stage is the largest)
};

// Second contraction of a head: acc2[m2] += W2[rows of tile m2][slab K] . ReLU(acc + b1), then acc = 0.
// The accumulator registers 8s..8s+7 of a 32x32 tile are K-step s of the B operand (lane = pixel);
// the matching A fragment (K in accumulator-row order) sits at column m*32 + h*16 + s*8 of the
// host-permuted W2 row.
template <typename T, int NT, int M2>
__device__ __forceinline__ void gemm2(f32x16 (&acc)[2][NT], f32x16 (&acc2)[M2][NT], const float *__restrict__ b1,
                                      const char *s_w2, int w2b, int r, int h)
{
    using E = ET<T>;
    constexpr int ES = sizeof(T);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            typename E::frag fb[NT];
            float bias[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = 8 * sb + j;
                bias[j] = b1[m * 32 + (i & 3) + 8 * (i >> 2) + 4 * h];
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    x[j] = fmaxf(acc[m][n][8 * sb + j] + bias[j], 0.f);
                    acc[m][n][8 * sb + j] = 0.f;
                }
                if constexpr (ES == 4) {
                    fb[n].lo = f32x4{x[0], x[1], x[2], x[3]};
                    fb[n].hi = f32x4{x[4], x[5], x[6], x[7]};
                } else {
                    fb[n].v = u32x4{pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]), pack_bf16x2(x[4], x[5]),
                                    pack_bf16x2(x[6], x[7])};
                }
            }
#pragma unroll
            for (int m2 = 0; m2 < M2; ++m2) {
                const typename E::frag fa = E::lds_frag(s_w2 + (m2 * 32 + r) * w2b + (m * 32 + h * 16 + sb * 8) * ES);
#pragma unroll
                for (int n = 0; n < NT; ++n) E::mma(acc2[m2][n], fa, fb[n]);
            }
        }
    }
}

template <typename T, int TH, int M2>
__global__ __launch_bounds__(512) void heads_kernel(HeadsArgs a)
{
    using C = HeadsCfg<T, TH>;
    using E = ET<T>;
    constexpr int ES = C::ES;
    constexpr int NT = C::NT;
    __shared__ __attribute__((aligned(16))) char smem[C::LDS];
    char *s_in = smem;
    char *s_ring = smem + C::LDS_IN;
    char *s_w2 = s_ring + 2 * C::LDS_RING;

    const int tid = threadIdx.x;
    const int wv = tid >> 6, l = tid & 63, r = l & 31, h = l >> 5;
    const int tiles = a.tiles_x * a.tiles_y;
    const int b = blockIdx.x / tiles;
    const int t = blockIdx.x - b * tiles;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * C::TW;
    const int slabs = a.head_conv / HC_SLAB;
    const int stages_per_head = slabs * 10;            // 9 tap stages + 1 "W2" stage per slab
    const int nstages = a.nheads * stages_per_head;

    // ---- stage descriptor: what global data stage `s` needs in LDS -------------------------------
    // tap stage : 64 rows x 64 K of W1 (rows = intermediate channels of the slab) -> ring slot (k & 1)
    // W2 stage  : 96 rows x 64 K of the head's packed 1x1 weights                 -> s_w2
    // two staging register sets: the loads of stage s+2 are issued while stage s computes and are written
    // to LDS one stage later, so a weight fetch has two stages (~1 us) to land instead of one
    u32x4 stgA[C::NSTG], stgB[C::NSTG];
    auto prefetch_into = [&](int s, u32x4 (&stg)[C::NSTG]) {
        if ((a.dbg & 1) && s > 1) return;
        const int head = s / stages_per_head, q = s - head * stages_per_head;
        const int slab = q / 10, k = q - slab * 10;
#pragma unroll
        for (int j = 0; j < C::NSTG; ++j) {
            const int i = tid + j * 512;
            const int row = i / C::VPR, v = i - row * C::VPR;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (k < 9) {
                if (row < HC_SLAB)
                    val = *reinterpret_cast<const u32x4 *>(
                        a.w1 + (((size_t)(head * a.head_conv + slab * HC_SLAB + row) * 9 + k) * HC_IN) * ES + v * 16);
            } else if (row < 32 * HC_MT2) {
                val = *reinterpret_cast<const u32x4 *>(
                    a.w2[head] + ((size_t)row * a.head_conv + slab * HC_SLAB) * ES + v * 16);
            }
            stg[j] = val;
        }
    };
    auto commit_from = [&](int s, u32x4 (&stg)[C::NSTG]) {
        const int q = s % stages_per_head;
        const int k = q % 10;
#pragma unroll
        for (int j = 0; j < C::NSTG; ++j) {
            const int i = tid + j * 512;
            const int row = i / C::VPR, v = i - row * C::VPR;
            if (k < 9) {
                if (row < HC_SLAB)
                    *reinterpret_cast<u32x4 *>(s_ring + (k & 1) * C::LDS_RING + row * C::WB + v * 16) = stg[j];
            } else if (row < 32 * HC_MT2) {
                *reinterpret_cast<u32x4 *>(s_w2 + row * C::W2B + v * 16) = stg[j];
            }
        }
    };

    // ---- prologue: halo tile (all 64 channels, zero outside the image) + stages 0 and 1 ------------
    prefetch_into(0, stgA);
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
                        *reinterpret_cast<u32x4 *>(s_in + iy * C::RB + ix * C::SB + v * 16) = val;
                    }
                });
        }
    }
    commit_from(0, stgA);
    prefetch_into(min(1, nstages - 1), stgB);
    __syncthreads();

    int boff[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) boff[n] = (wv * NT + n) * C::RB + r * C::SB + 8 * h * ES;
    const int aoff = r * C::WB + 8 * h * ES;

    // One head with M2 row tiles of 1x1 output.  Accumulators are local to this instantiation so
    // no control-flow merge ever joins differently-updated accumulator sets (that costs copies
    // and spills); `s` is the running stage index of the weight pipeline.
    auto run_head = [&](int head, int &s) {
        f32x16 acc[2][NT], acc2[M2][NT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
#pragma unroll
        for (int m = 0; m < M2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc2[m][n][i] = 0.f;
        for (int slab = 0; slab < slabs; ++slab) {
            // 10 stages per slab (9 taps + the 1x1 stage), fully unrolled so the staging-set parity
            // (stage s uses set s & 1; s is even at every slab start) is a compile-time choice
#pragma unroll
            for (int k = 0; k < 10; ++k, ++s) {
                {   // unconditional (stage index clamped at the tail): a path-independent number of
                    // outstanding loads lets hipcc emit a counted s_waitcnt vmcnt(N) at the commit below
                    const int sp = min(s + 2, nstages - 1);
                    if (k & 1) prefetch_into(sp, stgB); else prefetch_into(sp, stgA);
                }
                if (k < 9) {
                    // ---- 3x3 tap k of this slab: acc[2][NT] += W1[64 x 64] . halo(tap)[64 x pixels] ----
                    const int dy = k / 3, dx = k - dy * 3;
                    const char *wr = s_ring + (k & 1) * C::LDS_RING + aoff;
                    const char *br = s_in + dy * C::RB + dx * C::SB;
                    // fragments of k-step kk+1 are read from LDS before the MFMAs of k-step kk issue
                    // (explicit double buffer: hipcc otherwise re-uses one register set and every
                    // k-step waits out its own ds_read latency)
                    typename E::frag fa[2][2], fb[2][NT];
#pragma unroll
                    for (int m = 0; m < 2; ++m) fa[0][m] = E::lds_frag(wr + m * 32 * C::WB);
#pragma unroll
                    for (int n = 0; n < NT; ++n) fb[0][n] = E::lds_frag(br + boff[n]);
#pragma unroll
                    for (int kk = 0; kk < HC_IN / 16; ++kk) {
                        constexpr int NK = HC_IN / 16;
                        const int cur = kk & 1, nxt = cur ^ 1;
                        if (kk + 1 < NK) {
#pragma unroll
                            for (int m = 0; m < 2; ++m) fa[nxt][m] = E::lds_frag(wr + m * 32 * C::WB + (kk + 1) * 16 * ES);
#pragma unroll
                            for (int n = 0; n < NT; ++n) fb[nxt][n] = E::lds_frag(br + boff[n] + (kk + 1) * 16 * ES);
                        }
#pragma unroll
                        for (int m = 0; m < 2; ++m)
#pragma unroll
                            for (int n = 0; n < NT; ++n) E::mma(acc[m][n], fa[cur][m], fb[cur][n]);
                    }
                } else {
                    // ---- slab done: X = ReLU(acc + b1) -> B operand; acc2 += W2[:, slab] . X; acc = 0 ----
                    gemm2<T, NT, M2>(acc, acc2, a.b1 + head * a.head_conv + slab * HC_SLAB, s_w2, C::W2B, r, h);
                }
                {
                    const int sc = min(s + 1, nstages - 1);
                    if ((k + 1) & 1) commit_from(sc, stgB); else commit_from(sc, stgA);
                }
                __syncthreads();
            }
        }
        // ---- head done: z = acc2 + b2 -> NCHW fp32 (lane = pixel: coalesced rows) --------------------
        const int C_head = a.C[head];
        const float *b2 = a.b2[head];
        float *out = a.out[head];
        const size_t cstride = (size_t)a.H * a.W;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int oy = oy0 + wv * NT + n, ox = ox0 + r;
            if (oy < a.H && ox < a.W) {
                float *op = out + (((size_t)b * C_head + 4 * h) * a.H + oy) * a.W + ox;
#pragma unroll
                for (int m2 = 0; m2 < M2; ++m2) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c = m2 * 32 + 8 * g + 4 * h;
                        float *p = op + (size_t)(m2 * 32 + 8 * g) * cstride;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (c + i < C_head) p[i * cstride] = acc2[m2][n][4 * g + i] + b2[c + i];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    };

    int s = 0;
    for (int head = 0; head < a.nheads; ++head) run_head(head, s);
}

static_assert(HEADS_MAX == H3D_HEADS_MAX, "header/kernel mismatch");

int h3d_launch_heads(const h3d_op &op, hipStream_t st)
{
    if (!op.in || !op.w || !op.bias || !op.in2) H3D_FAIL(H3D_ERR_ARG, "heads: null pointer");
    const h3d_heads_desc *d = (const h3d_heads_desc *)op.in2;
    const int es = op.dtype == H3D_BF16 ? 2 : 4;
    if (op.Cin != HC_IN || op.in_cs % (16 / es) || op.in_cs < HC_IN)
        H3D_FAIL(H3D_ERR_SHAPE, "heads: input must have %d channels (got %d, stride %d)", HC_IN, op.Cin, op.in_cs);
    if (op.Cout <= 0 || op.Cout % HC_SLAB) H3D_FAIL(H3D_ERR_SHAPE, "heads: head_conv %d must be a multiple of %d", op.Cout, HC_SLAB);
    if (d->nheads <= 0 || d->nheads > HEADS_MAX) H3D_FAIL(H3D_ERR_SHAPE, "heads: %d heads (max %d)", d->nheads, HEADS_MAX);
    HeadsArgs a;
    a.in = (const char *)op.in; a.w1 = (const char *)op.w; a.b1 = op.bias;
    a.dbg = op.reserved;
    a.nheads = d->nheads; a.head_conv = op.Cout; a.B = op.B; a.H = op.H; a.W = op.W; a.in_cs = op.in_cs;
    for (int i = 0; i < d->nheads; ++i) {
        if (!d->head[i].w2 || !d->head[i].b2 || !d->head[i].out) H3D_FAIL(H3D_ERR_ARG, "heads: head %d null pointer", i);
        if (d->head[i].C <= 0 || d->head[i].C > 32 * HC_MT2)
            H3D_FAIL(H3D_ERR_UNSUPPORTED, "heads: head %d has %d channels (max %d)", i, d->head[i].C, 32 * HC_MT2);
        a.w2[i] = (const char *)d->head[i].w2; a.b2[i] = d->head[i].b2; a.out[i] = d->head[i].out; a.C[i] = d->head[i].C;
    }
    int m2 = 1;
    for (int i = 0; i < d->nheads; ++i) m2 = max(m2, (d->head[i].C + 31) / 32);   // row tiles of the widest head
    const int th = op.dtype == H3D_BF16 ? 16 : 8;
    a.tiles_x = cdiv(op.W, 32); a.tiles_y = cdiv(op.H, th);
    const dim3 grid(op.B * a.tiles_x * a.tiles_y), blk(512);
    if (op.dtype != H3D_BF16 && op.dtype != H3D_F32) H3D_FAIL(H3D_ERR_DTYPE, "heads: dtype %d", op.dtype);
    if (h3d_note_kernel("heads_kernel<%s, %d, %d>", op.dtype == H3D_BF16 ? "unsigned short" : "float", th, m2)) return H3D_OK;
    if (op.dtype == H3D_BF16) {
        if (m2 == 1) hipLaunchKernelGGL((heads_kernel<bf16_t, 16, 1>), grid, blk, 0, st, a);
        else if (m2 == 2) hipLaunchKernelGGL((heads_kernel<bf16_t, 16, 2>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((heads_kernel<bf16_t, 16, 3>), grid, blk, 0, st, a);
    } else {
        if (m2 == 1) hipLaunchKernelGGL((heads_kernel<float, 8, 1>), grid, blk, 0, st, a);
        else if (m2 == 2) hipLaunchKernelGGL((heads_kernel<float, 8, 2>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((heads_kernel<float, 8, 3>), grid, blk, 0, st, a);
    }
    H3D_CHECK_LAUNCH("heads_kernel");
    return H3D_OK;
}
