// First-generation DeformConv network kernel (round 1): global gather of the four corners per (pixel, tap, channel vector), sampled tile
// through LDS into the MFMA contraction, offsets + mask logits read as NHWC fp32 [.., >= 27] straight from a separate conv_offset_mask
// launch (dcn_v2.py:119-122).  Superseded by csrc/dcn2.hip (LDS apron) and csrc/dcn3.hip (offset conv fused in, patch slots): no plan
// dispatches it.  Kept as an A/B reference behind H3D_OP_DCN_V1 in `make EXTRA=1` builds (tests/test_gpu_dcn.py, marker `extra`).
#include "common.h"
#include "epilogue.h"
#include "dcn_sample.h"

struct DcnArgs {
    const char *in;
    const char *w;
    const float *bias;
    const float *om;  // [B,H,W,om_cs] fp32: 0..17 offsets (2t = dh, 2t+1 = dw), 18..26 mask logits
    char *out;
    int B, H, W, Cin, in_cs, om_cs;
    int Cout, out_cs, relu, out_mode;
    int tiles_x, tiles_y;
};

template <typename T, int MT, int CK>
struct DcnCfg {
    static constexpr int ES = sizeof(T);
    static constexpr int SB = CK * ES + 16;       // sampled-tile pixel stride (odd multiple of 16 B)
    static constexpr int WB = 9 * CK * ES + 16;   // weight row stride
    static constexpr int BN = 32 * MT;
    static constexpr int NT = 2;
    static constexpr int VPP = CK * ES / 16;
    static constexpr int LDS_S = 256 * SB;
    static constexpr int LDS = LDS_S + BN * WB;
};

template <typename T, int MT, int CK>
__global__ __launch_bounds__(256) void dcn_kernel(DcnArgs a)
{
    using C = DcnCfg<T, MT, CK>;
    using E = ET<T>;
    constexpr int ES = C::ES;
    constexpr int NV = 16 / ES;  // elements per 16-byte vector
    __shared__ __attribute__((aligned(16))) char smem[C::LDS];
    char *s_s = smem;
    char *s_w = smem + C::LDS_S;

    const int tid = threadIdx.x;
    const int wv = tid >> 6, l = tid & 63, r = l & 31, h = l >> 5;
    const int tiles = a.tiles_x * a.tiles_y;
    const int b = blockIdx.x / tiles;
    const int t = blockIdx.x - b * tiles;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int oy0 = ty * 16, ox0 = tx * 16;
    const int cout0 = blockIdx.y * C::BN;

    f32x16 acc[MT][C::NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < C::NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    const int aoff = r * C::WB + 8 * h * ES;
    int boff[C::NT];
#pragma unroll
    for (int n = 0; n < C::NT; ++n) boff[n] = ((wv * 4 + n * 2) * 16 + r) * C::SB + 8 * h * ES;

    const char *img = a.in + (size_t)b * a.H * a.W * a.in_cs * ES;
    const int bv = tid % C::VPP;       // which 16-byte channel vector of the chunk this thread builds
    const int bp = tid / C::VPP;       // pixel within a build pass
    constexpr int PPP = 256 / C::VPP;  // pixels per build pass

    for (int c0 = 0; c0 < a.Cin; c0 += CK) {
        __syncthreads();  // previous chunk's MFMAs have consumed s_w / s_s
        constexpr int WV = 9 * C::VPP;
        for (int i = tid; i < C::BN * WV; i += 256) {
            const int row = i / WV, q = i - row * WV;
            const int tap = q / C::VPP, v = q - tap * C::VPP;
            const u32x4 val = *reinterpret_cast<const u32x4 *>(
                a.w + (((size_t)(cout0 + row) * 9 + tap) * a.Cin + c0) * ES + v * 16);
            *reinterpret_cast<u32x4 *>(s_w + row * C::WB + tap * CK * ES + v * 16) = val;
        }
        for (int tap = 0; tap < 9; ++tap) {
            const int ti = tap / 3, tj = tap - ti * 3;
            if (tap) __syncthreads();  // previous tap's MFMAs done reading s_s
            // ---- gather + bilinear blend: sampled[256 px][CK] -> LDS ---------------------------------
#pragma unroll
            for (int pass = 0; pass < C::VPP; ++pass) {
                const int p = pass * PPP + bp;
                const int oy = oy0 + (p >> 4), ox = ox0 + (p & 15);
                float o[NV];
#pragma unroll
                for (int e = 0; e < NV; ++e) o[e] = 0.f;
                if (oy < a.H && ox < a.W) {
                    const float *om = a.om + ((size_t)(b * a.H + oy) * a.W + ox) * a.om_cs;
                    const float h_im = (float)(oy - 1 + ti) + om[2 * tap];
                    const float w_im = (float)(ox - 1 + tj) + om[2 * tap + 1];
                    const Sample s = make_sample(h_im, w_im, sigmoidf_(om[18 + tap]), a.H, a.W);
                    if (s.inside) {
                        float cv[4][NV];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            u32x4 raw = {0u, 0u, 0u, 0u};
                            if (s.off[k] >= 0)
                                raw = *reinterpret_cast<const u32x4 *>(
                                    img + ((size_t)s.off[k] * a.in_cs + c0) * ES + bv * 16);
                            if constexpr (ES == 4) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) cv[k][e] = __uint_as_float(raw[e]);
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    cv[k][2 * e] = __uint_as_float(raw[e] << 16);
                                    cv[k][2 * e + 1] = __uint_as_float(raw[e] & 0xffff0000u);
                                }
                            }
                        }
#pragma unroll
                        for (int e = 0; e < NV; ++e)
                            o[e] = (s.w[0] * cv[0][e] + s.w[1] * cv[1][e] + s.w[2] * cv[2][e] + s.w[3] * cv[3][e]) * s.mask;
                    }
                }
                u32x4 packed;
                if constexpr (ES == 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) packed[e] = __float_as_uint(o[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) packed[e] = pack_bf16x2(o[2 * e], o[2 * e + 1]);
                }
                *reinterpret_cast<u32x4 *>(s_s + p * C::SB + bv * 16) = packed;
            }
            __syncthreads();
            // ---- contraction of this tap -----------------------------------------------------------
#pragma unroll
            for (int kk = 0; kk < CK / 16; ++kk) {
                typename E::frag fa[MT], fb[C::NT];
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    fa[m] = E::lds_frag(s_w + aoff + m * 32 * C::WB + (tap * CK + kk * 16) * ES);
#pragma unroll
                for (int n = 0; n < C::NT; ++n) fb[n] = E::lds_frag(s_s + boff[n] + kk * 16 * ES);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < C::NT; ++n) E::mma(acc[m][n], fa[m], fb[n]);
            }
        }
    }
    EpiArgs e;
    e.bias = a.bias; e.res = nullptr; e.out = a.out; e.Ho = a.H; e.Wo = a.W; e.Cout = a.Cout;
    e.out_cs = a.out_cs; e.res_cs = 0; e.relu = a.relu; e.out_mode = a.out_mode;
    tile_epilogue<T, MT, C::NT>(acc, e, b, oy0, ox0, cout0, wv, r, h);
}

template <typename T, int MT, int CK>
static int launch_dcn_cfg(const DcnArgs &a0, hipStream_t st)
{
    using C = DcnCfg<T, MT, CK>;
    static_assert(C::LDS <= 160 * 1024, "LDS budget");
    DcnArgs a = a0;
    a.tiles_x = cdiv(a.W, 16);
    a.tiles_y = cdiv(a.H, 16);
    dim3 grid(a.B * a.tiles_x * a.tiles_y, cdiv(a.Cout, C::BN));
    if (h3d_note_kernel("dcn_kernel<%s, %d, %d>", sizeof(T) == 2 ? "unsigned short" : "float", MT, CK)) return H3D_OK;
    hipLaunchKernelGGL((dcn_kernel<T, MT, CK>), grid, dim3(256), 0, st, a);
    H3D_CHECK_LAUNCH("dcn_kernel");
    return H3D_OK;
}

int h3d_launch_dcn(const h3d_op &op, hipStream_t st)
{
    if (!op.in || !op.w || !op.bias || !op.out || !op.in2) H3D_FAIL(H3D_ERR_ARG, "dcn: null pointer");
    const int es = op.dtype == H3D_BF16 ? 2 : 4;
    if (op.ksize != 3 || op.stride != 1 || op.Ho != op.H || op.Wo != op.W)
        H3D_FAIL(H3D_ERR_UNSUPPORTED, "dcn op: network path covers 3x3 s1 p1 d1 dg1 only (k=%d s=%d)", op.ksize, op.stride);
    if (op.Cin % 16 || op.in_cs % (16 / es) || op.Cin > op.in_cs || op.in2_cs < 27)
        H3D_FAIL(H3D_ERR_SHAPE, "dcn: Cin=%d (stride %d) must be a multiple of 16; offset stride %d >= 27", op.Cin,
                 op.in_cs, op.in2_cs);
    if (op.wrows < ((op.Cout + 127) / 128) * 128)
        H3D_FAIL(H3D_ERR_SHAPE, "dcn: packed weight rows %d < Cout %d padded to 128", op.wrows, op.Cout);
    if (op.out_mode == H3D_OUT_NHWC && (op.out_cs % 4 || op.Cout > op.out_cs))
        H3D_FAIL(H3D_ERR_SHAPE, "dcn: out channel stride %d", op.out_cs);
    DcnArgs a;
    a.in = (const char *)op.in; a.w = (const char *)op.w; a.bias = op.bias; a.om = (const float *)op.in2;
    a.out = (char *)op.out; a.B = op.B; a.H = op.H; a.W = op.W; a.Cin = op.Cin; a.in_cs = op.in_cs;
    a.om_cs = op.in2_cs; a.Cout = op.Cout; a.out_cs = op.out_cs; a.relu = op.relu; a.out_mode = op.out_mode;
    a.tiles_x = a.tiles_y = 0;
    if (op.dtype == H3D_BF16) {
        if (op.Cin % 32 == 0 && op.Cout <= 64) {
            if (op.Cout <= 32) return launch_dcn_cfg<bf16_t, 1, 32>(a, st);
            return launch_dcn_cfg<bf16_t, 2, 32>(a, st);
        }
        if (op.Cout <= 32) return launch_dcn_cfg<bf16_t, 1, 16>(a, st);
        if (op.Cout <= 64) return launch_dcn_cfg<bf16_t, 2, 16>(a, st);
        return launch_dcn_cfg<bf16_t, 4, 16>(a, st);
    }
    if (op.dtype == H3D_F32) {
        if (op.Cout <= 32) return launch_dcn_cfg<float, 1, 16>(a, st);
        return launch_dcn_cfg<float, 2, 16>(a, st);
    }
    H3D_FAIL(H3D_ERR_DTYPE, "dcn: dtype %d", op.dtype);
}

