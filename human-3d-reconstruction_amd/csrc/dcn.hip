// Modulated deformable convolution (DCNv2) forward on gfx950.
//
// Reference being replaced (all under /root/reference/src/lib/models/DCNv2/src/cuda/):
//   dcn_v2_cuda.cu:43-173            host fn: ones/columns scratch, pointer tables, bias SgemmBatched,
//                                    im2col kernel, weight SgemmBatched
//   dcn_v2_im2col_cuda.cu:25-54      bilinear sample, zero corners outside the image
//   dcn_v2_im2col_cuda.cu:125-195    per (b,c,h,w) 9 taps: position, (>-1, <H) gate, val*mask
// Here the sampled tile goes straight from the gather into LDS as the B operand of the MFMA
// contraction; the `columns` tensor (2.4 GB per 64-channel 128x128 layer at batch 64) never exists.
//
// This file: the OPERATOR BOUNDARY -- dcn_nchw_kernel (`dcn_v2_forward`: contiguous NCHW fp32, any kernel size / stride / pad /
// dilation / deformable groups, fp32 FMA), the workspace / packed-filter forms on csrc/dcn2.hip's kernel, `h3d_dcn_offset_mask`.
// (The first-generation network kernel `dcn_kernel<T>` lives in csrc/dcn1.hip since round 5: an A/B reference, `make EXTRA=1`.)
#include "common.h"
#include "epilogue.h"
#include "dcn_sample.h"
#include <algorithm>

// ================================================================================================
// Operator boundary: general NCHW fp32 kernel.  Block = 64 output pixels x 64 output channels,
// K = C*kh*kw walked in chunks of 16: sampled columns and the weight slab meet in LDS.
struct DcnNchwArgs {
    const float *in, *w, *bias, *off, *mask;   // off == nullptr: zero offsets and a unit mask, i.e. a plain zero-padded convolution
    float *out;
    float *out2;   // split > 0 (h3d_dcn_offset_mask): channels >= split go to out2 [B, Cout - split, Ho, Wo] through a sigmoid
    int split;
    int B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg, Ho, Wo;
};

__global__ __launch_bounds__(256) void dcn_nchw_kernel(DcnNchwArgs a)
{
    constexpr int TP = 64, TC = 64, KC = 16;
    __shared__ float s_col[KC][TP + 4];
    __shared__ float s_wt[KC][TC + 4];
    const int tid = threadIdx.x;
    const int b = blockIdx.z, co0 = blockIdx.y * TC, n0 = blockIdx.x * TP;
    const int HoWo = a.Ho * a.Wo, khw = a.kh * a.kw, K = a.C * khw, cpg = a.C / a.dg;
    const int tp = tid & 15, tc = tid >> 4;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    const int p = tid & 63;
    const int n = n0 + p;
    const int oh = n / a.Wo, ow = n - oh * a.Wo;
    for (int k0 = 0; k0 < K; k0 += KC) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kl = (tid >> 6) + 4 * j, k = k0 + kl;
            float val = 0.f;
            if (k < K && n < HoWo) {
                const int c = k / khw, t = k - c * khw;
                const int i = t / a.kw, jj = t - i * a.kw;
                const int g = c / cpg;
                float d_h = 0.f, d_w = 0.f, m = 1.f;
                if (a.off) {
                    const float *offp = a.off + ((size_t)(b * a.dg + g) * 2 * khw) * HoWo;
                    const float *mp = a.mask + ((size_t)(b * a.dg + g) * khw) * HoWo;
                    d_h = offp[(size_t)(2 * t) * HoWo + n];
                    d_w = offp[(size_t)(2 * t + 1) * HoWo + n];
                    m = mp[(size_t)t * HoWo + n];
                }
                const float h_im = (float)(oh * a.sh - a.ph + i * a.dh) + d_h;
                const float w_im = (float)(ow * a.sw - a.pw + jj * a.dw) + d_w;
                const Sample s = make_sample(h_im, w_im, m, a.H, a.W);
                if (s.inside) {
                    const float *im = a.in + ((size_t)b * a.C + c) * a.H * a.W;
                    const float v1 = s.off[0] >= 0 ? im[s.off[0]] : 0.f;
                    const float v2 = s.off[1] >= 0 ? im[s.off[1]] : 0.f;
                    const float v3 = s.off[2] >= 0 ? im[s.off[2]] : 0.f;
                    const float v4 = s.off[3] >= 0 ? im[s.off[3]] : 0.f;
                    val = (s.w[0] * v1 + s.w[1] * v2 + s.w[2] * v3 + s.w[3] * v4) * s.mask;
                }
            }
            s_col[kl][p] = val;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kl = tid & 15, co = (tid >> 4) + 16 * j;
            const int k = k0 + kl;
            s_wt[kl][co] = (k < K && co0 + co < a.Cout) ? a.w[(size_t)(co0 + co) * K + k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kl = 0; kl < KC; ++kl) {
            float av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = s_wt[kl][tc * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = s_col[kl][tp * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = co0 + tc * 4 + i;
        if (co >= a.Cout) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nn = n0 + tp * 4 + j;
            if (nn >= HoWo) continue;
            const float v = acc[i][j] + a.bias[co];
            if (a.split <= 0) a.out[((size_t)b * a.Cout + co) * HoWo + nn] = v;
            else if (co < a.split) a.out[((size_t)b * a.split + co) * HoWo + nn] = v;
            else a.out2[((size_t)b * (a.Cout - a.split) + co - a.split) * HoWo + nn] = 1.0f / (1.0f + expf(-v));
        }
    }
}

// DCN.forward's own convolution (dcn_v2.py:119-124) for ANY module configuration, on this library's kernel: the general operator kernel
// with zero offsets and a unit mask IS nn.Conv2d(C, 3 dg kh kw, (kh, kw), stride, padding) (bilinear weights (1, 0, 0, 0) at integer
// positions, zero outside the image); its epilogue splits the channels as `torch.chunk(out, 3, dim=1)` / `cat((o1, o2))` / `sigmoid(mask)` do.
extern "C" int h3d_dcn_offset_mask(const float *input, const float *off_weight, const float *off_bias, float *offset, float *mask, int B, int C,
                                   int H, int W, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w,
                                   int deformable_group, void *stream)
{
    if (!input || !off_weight || !off_bias || !offset || !mask) H3D_FAIL(H3D_ERR_ARG, "dcn_offset_mask: null pointer");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || kernel_h <= 0 || kernel_w <= 0 || stride_h <= 0 || stride_w <= 0 || pad_h < 0 || pad_w < 0 ||
        deformable_group <= 0)
        H3D_FAIL(H3D_ERR_SHAPE, "dcn_offset_mask: non-positive dimension");
    const int k = deformable_group * kernel_h * kernel_w;
    DcnNchwArgs a;
    a.in = input; a.w = off_weight; a.bias = off_bias; a.off = nullptr; a.mask = nullptr; a.out = offset; a.out2 = mask; a.split = 2 * k;
    a.B = B; a.C = C; a.H = H; a.W = W; a.Cout = 3 * k; a.kh = kernel_h; a.kw = kernel_w; a.sh = stride_h; a.sw = stride_w;
    a.ph = pad_h; a.pw = pad_w; a.dh = 1; a.dw = 1; a.dg = 1;      // (nn.Conv2d of dcn_v2.py:107-111: no dilation, no groups)
    a.Ho = (H + 2 * pad_h - kernel_h) / stride_h + 1;
    a.Wo = (W + 2 * pad_w - kernel_w) / stride_w + 1;
    if (a.Ho <= 0 || a.Wo <= 0) H3D_FAIL(H3D_ERR_SHAPE, "dcn_offset_mask: empty output %dx%d", a.Ho, a.Wo);
    dim3 grid(cdiv(a.Ho * a.Wo, 64), cdiv(a.Cout, 64), B);
    hipLaunchKernelGGL(dcn_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    H3D_CHECK_LAUNCH("dcn_nchw_kernel");
    return H3D_OK;
}

extern "C" int h3d_dcn_v2_forward(const float *input, const float *weight, const float *bias, const float *offset,
                                  const float *mask, float *output, int B, int C, int H, int W, int Cout, int kernel_h,
                                  int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h,
                                  int dilation_w, int deformable_group, void *stream)
{
    if (!input || !weight || !bias || !offset || !mask || !output) H3D_FAIL(H3D_ERR_ARG, "dcn_v2_forward: null pointer");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cout <= 0 || kernel_h <= 0 || kernel_w <= 0 || stride_h <= 0 ||
        stride_w <= 0 || pad_h < 0 || pad_w < 0 || dilation_h <= 0 || dilation_w <= 0)
        H3D_FAIL(H3D_ERR_SHAPE, "dcn_v2_forward: non-positive dimension");
    if (deformable_group <= 0 || C % deformable_group)
        H3D_FAIL(H3D_ERR_SHAPE, "dcn_v2_forward: channels %d not divisible by deformable_group %d", C, deformable_group);
    DcnNchwArgs a;
    a.in = input; a.w = weight; a.bias = bias; a.off = offset; a.mask = mask; a.out = output; a.out2 = nullptr; a.split = 0;
    a.B = B; a.C = C; a.H = H; a.W = W; a.Cout = Cout; a.kh = kernel_h; a.kw = kernel_w; a.sh = stride_h; a.sw = stride_w;
    a.ph = pad_h; a.pw = pad_w; a.dh = dilation_h; a.dw = dilation_w; a.dg = deformable_group;
    a.Ho = (H + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) / stride_h + 1;
    a.Wo = (W + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) / stride_w + 1;
    if (a.Ho <= 0 || a.Wo <= 0) H3D_FAIL(H3D_ERR_SHAPE, "dcn_v2_forward: empty output %dx%d", a.Ho, a.Wo);
    dim3 grid(cdiv(a.Ho * a.Wo, 64), cdiv(Cout, 64), B);
    hipLaunchKernelGGL(dcn_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    H3D_CHECK_LAUNCH("dcn_nchw_kernel");
    return H3D_OK;
}


// ================================================================================================
// Operator boundary, fast path for the configuration the model uses (model.py:355: 3x3, stride 1, pad 1, dilation 1,
// deformable_group 1) when C % 16 == 0: the reference's operands (NCHW fp32, separate offset / mask tensors, OIHW
// weights) are re-laid once into the network kernels' layout inside a caller-provided workspace -- input -> NHWC,
// (offset | mask) -> [B,H,W,32], weights -> [rows][9][C] -- and the contraction runs on csrc/dcn2.hip's fp32 kernel
// (LDS apron gather, sampling geometry once per pixel instead of once per pixel AND channel as
// dcn_v2_im2col_cuda.cu:170-172 does, exact fmaf chains on v_mfma_f32_32x32x2_f32), which writes the NCHW output itself.
int h3d_launch_dcn2(const h3d_op &op, hipStream_t st);

__global__ void dcn_om_pack_kernel(const float *__restrict__ off, const float *__restrict__ mask, float *__restrict__ om, int HW, size_t total)
{
    // thread = pixel: 27 coalesced channel reads (consecutive threads = consecutive pixels), one 128-byte row written
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t b = i / HW, n = i - b * HW;
    f32x4 row[8];
#pragma unroll
    for (int c = 0; c < 18; ++c) row[c >> 2][c & 3] = off[(b * 18 + c) * HW + n];
#pragma unroll
    for (int c = 0; c < 9; ++c) row[(18 + c) >> 2][(18 + c) & 3] = mask[(b * 9 + c) * HW + n];
#pragma unroll
    for (int c = 27; c < 32; ++c) row[c >> 2][c & 3] = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) reinterpret_cast<f32x4 *>(om + i * 32)[q] = row[q];
}

// max |filter| of an fp32 pack, as the bit pattern of the float (monotonic for non-negative values; a NaN sorts above everything and
// is recognised by the consumer): one atomicMax per wave into `wmax`, which the caller zeroed in front of the pack kernel.  The f16x3
// operator kernel (csrc/dcn2.hip, H3D_F16X3) derives its power-of-two filter scale from it -- on the device, no host round trip.
#define H3D_DCN_AUX_BYTES 256      // behind the bias of an fp32 pack: [0] = max |filter| bits
__device__ __forceinline__ void dcn_wmax_accumulate(unsigned *wmax, float v)
{
    unsigned m = __float_as_uint(fabsf(v));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, d));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(wmax, m);
}

__global__ void dcn_w_pack_kernel(const float *__restrict__ w, const float *__restrict__ bias, float *__restrict__ wp, float *__restrict__ bp,
                                  int Cout, int C, int rows, unsigned *__restrict__ wmax)
{
    // wp [rows][9][C] <- w [Cout][C][3][3]; rows beyond Cout are zero
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)rows * 9 * C;
    if (i < (size_t)rows) bp[i] = i < (size_t)Cout ? bias[i] : 0.f;
    float v = 0.f;
    if (i < total) {
        const int c = (int)(i % C);
        const int tap = (int)((i / C) % 9);
        const int o = (int)(i / ((size_t)9 * C));
        v = o < Cout ? w[((size_t)o * C + c) * 9 + tap] : 0.f;
        wp[i] = v;
    }
    dcn_wmax_accumulate(wmax, v);          // (every lane of the wave takes part in the shuffles)
}

static size_t ws_align(size_t x) { return (x + 255) & ~(size_t)255; }

// fp16 twin of dcn_w_pack_kernel: the filter format of the bf16 DeformConv kernels (csrc/dcn2.hip)
__global__ void dcn_w_pack_f16_kernel(const float *__restrict__ w, const float *__restrict__ bias, _Float16 *__restrict__ wp, float *__restrict__ bp,
                                      int Cout, int C, int rows)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)rows * 9 * C;
    if (i < (size_t)rows) bp[i] = i < (size_t)Cout ? bias[i] : 0.f;
    if (i >= total) return;
    const int c = (int)(i % C);
    const int tap = (int)((i / C) % 9);
    const int o = (int)(i / ((size_t)9 * C));
    wp[i] = (_Float16)(o < Cout ? __builtin_amdgcn_fmed3f(w[((size_t)o * C + c) * 9 + tap], -65504.f, 65504.f) : 0.f);
}

extern "C" size_t h3d_dcn_v2_workspace_bytes(int B, int C, int H, int W, int Cout)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
    const size_t rows = ((size_t)Cout + 127) / 128 * 128, px = (size_t)B * H * W;
    return ws_align(px * C * 4) + ws_align(px * 32 * 4) + ws_align(rows * 9 * C * 4) + ws_align(rows * 4) + H3D_DCN_AUX_BYTES;
}

// The arithmetic of the operator's fp32 fast path: three fp16 MFMAs on split operands per fp32 product (H3D_F16X3: 2^-22 relative per
// product, fp32 accumulation; ~2x the rate of the fp32 matrix instruction on these shapes) unless H3D_DCN_OP_F32=1 is in the environment
// (exact fmaf chains on v_mfma_f32_32x32x2_f32, the round 1-4 behaviour) or the caller passes H3D_DCN_F32_MFMA.
static bool dcn_op_f32_mfma()
{
    static const bool v = [] { const char *e = getenv("H3D_DCN_OP_F32"); return e && e[0] && e[0] != '0'; }();
    return v;
}

extern "C" int h3d_dcn_v2_forward_ws(const float *input, const float *weight, const float *bias, const float *offset,
                                     const float *mask, float *output, int B, int C, int H, int W, int Cout, int kernel_h,
                                     int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h,
                                     int dilation_w, int deformable_group, void *workspace, size_t workspace_bytes, void *stream)
{
    const bool fast = kernel_h == 3 && kernel_w == 3 && stride_h == 1 && stride_w == 1 && pad_h == 1 && pad_w == 1 &&
                      dilation_h == 1 && dilation_w == 1 && deformable_group == 1 && C > 0 && C % 16 == 0 && H <= 32767 && W <= 32767;
    if (!fast || !workspace || workspace_bytes < h3d_dcn_v2_workspace_bytes(B, C, H, W, Cout))
        return h3d_dcn_v2_forward(input, weight, bias, offset, mask, output, B, C, H, W, Cout, kernel_h, kernel_w, stride_h, stride_w,
                                  pad_h, pad_w, dilation_h, dilation_w, deformable_group, stream);
    if (!input || !weight || !bias || !offset || !mask || !output) H3D_FAIL(H3D_ERR_ARG, "dcn_v2_forward: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0) H3D_FAIL(H3D_ERR_SHAPE, "dcn_v2_forward: non-positive dimension");
    hipStream_t st = (hipStream_t)stream;
    const int rows = (Cout + 127) / 128 * 128;
    const size_t px = (size_t)B * H * W;
    char *ws = (char *)workspace;
    float *x_nhwc = (float *)ws;                ws += ws_align(px * C * 4);
    float *om = (float *)ws;                    ws += ws_align(px * 32 * 4);
    float *wp = (float *)ws;                    ws += ws_align((size_t)rows * 9 * C * 4);
    float *bp = (float *)ws;                    // [rows] + H3D_DCN_AUX_BYTES (rows is a multiple of 128: no padding in between)
    unsigned *wmax = (unsigned *)(bp + rows);
    if (hipMemsetAsync(wmax, 0, H3D_DCN_AUX_BYTES, st) != hipSuccess) H3D_FAIL(H3D_ERR_LAUNCH, "dcn_v2_forward: memset");
    int rc = h3d_nchw_f32_to_nhwc(input, x_nhwc, H3D_F32, B, C, H, W, C, stream);
    if (rc != H3D_OK) return rc;
    hipLaunchKernelGGL(dcn_om_pack_kernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, st, offset, mask, om, H * W, px);
    H3D_CHECK_LAUNCH("dcn_om_pack_kernel");
    const size_t wtotal = (size_t)rows * 9 * C;
    hipLaunchKernelGGL(dcn_w_pack_kernel, dim3((unsigned)((wtotal + 255) / 256)), dim3(256), 0, st, weight, bias, wp, bp, Cout, C, rows, wmax);
    H3D_CHECK_LAUNCH("dcn_w_pack_kernel");
    h3d_op op = {};
    op.kind = H3D_OP_DCN; op.dtype = dcn_op_f32_mfma() ? H3D_F32 : H3D_F16X3;
    op.in = x_nhwc; op.in2 = om; op.w = wp; op.bias = bp; op.out = output;
    op.B = B; op.H = H; op.W = W; op.Cin = C; op.in_cs = C; op.in2_cs = 32; op.Ho = H; op.Wo = W; op.Cout = Cout; op.out_cs = Cout;
    op.ksize = 3; op.stride = 1; op.relu = 0; op.out_mode = H3D_OUT_NCHW_F32; op.wrows = rows;
    op.reserved = 0x800 | (op.dtype == H3D_F16X3 ? 0x100000 : 0);      // the mask operand is final (the reference applies the sigmoid in DCN.forward, dcn_v2.py:124); fp32 pack + max |w|
    return h3d_launch_dcn2(op, st);
}

// ---- the operator's THROUGHPUT form (round 3): what a caller that runs the same layer on every batch wants ----------------------
// h3d_dcn_v2_forward / _ws keep the reference's contract literally (dcn_v2.h:9-23: OIHW fp32 weights and NCHW fp32 input on every
// call), so every call re-packs the filters and re-lays the input: on the 16 DeformConv shapes of the network at batch 16 that is 35
// launches each of dcn_w_pack_kernel / dcn_om_pack_kernel / nchw_to_nhwc_kernel and 6.5 ms.  Here the filters are packed ONCE
// (h3d_dcn_v2_pack_weights; the Python shim caches the result per parameter version), the input may be channels-last (torch's
// channels_last memory format IS this library's NHWC: no relayout) and may be bf16 (the network kernels' bf16 path: fp16 filters,
// fp16 blend, f16 MFMA), and the output may be channels-last too.  offset / mask stay the reference's NCHW fp32 tensors.
extern "C" size_t h3d_dcn_v2_packed_weight_bytes(int Cout, int C, int dtype)
{
    if (Cout <= 0 || C <= 0 || (dtype != H3D_F32 && dtype != H3D_BF16)) return 0;
    const size_t rows = ((size_t)Cout + 127) / 128 * 128;
    return ws_align(rows * 9 * C * (dtype == H3D_F32 ? 4 : 2)) + ws_align(rows * 4) + (dtype == H3D_F32 ? H3D_DCN_AUX_BYTES : 0);
}

extern "C" int h3d_dcn_v2_pack_weights(const float *weight, const float *bias, int Cout, int C, int dtype, void *packed, void *stream)
{
    if (!weight || !bias || !packed) H3D_FAIL(H3D_ERR_ARG, "dcn_v2_pack_weights: null pointer");
    if (Cout <= 0 || C <= 0 || C % 16) H3D_FAIL(H3D_ERR_SHAPE, "dcn_v2_pack_weights: C=%d must be a positive multiple of 16", C);
    if (dtype != H3D_F32 && dtype != H3D_BF16) H3D_FAIL(H3D_ERR_DTYPE, "dcn_v2_pack_weights: dtype %d (f32 | bf16)", dtype);
    const int rows = (Cout + 127) / 128 * 128;
    const size_t wtotal = (size_t)rows * 9 * C;
    float *bp = (float *)((char *)packed + ws_align(wtotal * (dtype == H3D_F32 ? 4 : 2)));
    if (dtype == H3D_F32) {
        if (hipMemsetAsync(bp + rows, 0, H3D_DCN_AUX_BYTES, (hipStream_t)stream) != hipSuccess) H3D_FAIL(H3D_ERR_LAUNCH, "dcn_v2_pack_weights: memset");
        hipLaunchKernelGGL(dcn_w_pack_kernel, dim3((unsigned)((wtotal + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight, bias, (float *)packed, bp, Cout, C, rows,
                           (unsigned *)(bp + rows));
    } else
        hipLaunchKernelGGL(dcn_w_pack_f16_kernel, dim3((unsigned)((wtotal + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight, bias, (_Float16 *)packed, bp, Cout, C, rows);
    H3D_CHECK_LAUNCH("dcn_w_pack_kernel");
    return H3D_OK;
}

// ---- packs that VALIDATE themselves on the device (round 4) -------------------------------------------------------------
// A host-side cache keyed on (data_ptr, tensor._version) is not proof that the filters are unchanged: the reference edits its
// parameters through `.data` (dcn_v2.py:80-81 reset_parameters, DCNv2/test.py:21 `weight.data.zero_()`, model.py:462), which
// does not bump `_version`, and an address can be reused.  Asking the device "did the bytes change?" from the host would cost a
// synchronisation per call, so the question is asked AND answered on the device: every call hashes the parameter bytes into
// state[1] (order-independent 64-bit sum of mixed words), the pack kernels return at once when it equals state[0] -- the hash of
// what the packed image was built from -- and a one-thread kernel then commits state[1] to state[0] (and re-zeroes state[1]).  Everything is stream
// ordered on the caller's stream: a consumer on another stream re-validates itself before it reads, and if it finds the pack
// stale it re-packs the SAME bytes (a benign overlap).  state = 2 x uint64 on the device, zero-initialised by the host.
__device__ __forceinline__ unsigned long long sig_mix(unsigned long long x)
{
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
struct SigBufs {
    const uint32_t *p[4];
    unsigned long long nwords[4];
    int n;
};
// one launch hashes every parameter buffer of the pack (the launch, not the bytes, is what a call pays for: two launches + a
// memset per call showed up as 9 us each in the round-4 kernel trace); state[1] is zero on entry: the host zeroes it once,
// sig_commit_kernel re-zeroes it after every call
__global__ void sig_kernel(SigBufs b, unsigned long long *acc)
{
    unsigned long long h = 0;
    for (int k = 0; k < b.n; ++k) {
        const uint32_t *__restrict__ p = b.p[k];
        const unsigned long long salt = 0x9E3779B97F4A7C15ull * (unsigned long long)(k + 1);
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < b.nwords[k]; i += (size_t)gridDim.x * blockDim.x)
            h += sig_mix(((unsigned long long)p[i] << 32 | (unsigned long long)(uint32_t)i) + salt + (i >> 32));
    }
#pragma unroll
    for (int d = 32; d; d >>= 1) h += __shfl_xor(h, d);
    if ((threadIdx.x & 63) == 0 && h) atomicAdd(acc, h);
}
__global__ void sig_commit_kernel(unsigned long long *state) { state[0] = state[1]; state[1] = 0ull; }

static int sig_begin(unsigned long long *state, const void *const *bufs, const size_t *nbytes, int n, hipStream_t st)
{
    SigBufs b;
    size_t most = 0;
    b.n = n;
    for (int k = 0; k < 4; ++k) {
        b.p[k] = k < n ? (const uint32_t *)bufs[k] : nullptr;
        b.nwords[k] = k < n ? nbytes[k] / 4 : 0;
        most = std::max<size_t>(most, (size_t)b.nwords[k]);
    }
    const unsigned blocks = (unsigned)std::min<size_t>(std::max<size_t>((most + 1023) / 1024, 1), 512);
    hipLaunchKernelGGL(sig_kernel, dim3(blocks), dim3(256), 0, st, b, state + 1);
    H3D_CHECK_LAUNCH("sig_kernel");
    return H3D_OK;
}

__global__ void dcn_wmax_reset_if_kernel(const unsigned long long *__restrict__ state, unsigned *__restrict__ wmax)
{
    if (state[0] != state[1]) wmax[0] = wmax[1] = 0u;      // the pack kernel behind this one rebuilds the image and its maxima (two words: main, offset filters)
}

__global__ void dcn_w_pack_if_kernel(const float *__restrict__ w, const float *__restrict__ bias, void *__restrict__ wp_, float *__restrict__ bp,
                                     int Cout, int C, int rows, int f16, const unsigned long long *__restrict__ state, unsigned *__restrict__ wmax)
{
    if (state[0] == state[1]) return;                      // the packed image was built from exactly these bytes
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)rows * 9 * C;
    if (i < (size_t)rows) bp[i] = i < (size_t)Cout ? bias[i] : 0.f;
    float v = 0.f;
    if (i < total) {
        const int c = (int)(i % C);
        const int tap = (int)((i / C) % 9);
        const int o = (int)(i / ((size_t)9 * C));
        v = o < Cout ? w[((size_t)o * C + c) * 9 + tap] : 0.f;
        if (f16) ((_Float16 *)wp_)[i] = (_Float16)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
        else ((float *)wp_)[i] = v;
    }
    if (!f16) dcn_wmax_accumulate(wmax, v);
}

extern "C" int h3d_dcn_v2_pack_weights_cached(const float *weight, const float *bias, int Cout, int C, int dtype, void *packed, void *state_,
                                              void *stream)
{
    if (!weight || !bias || !packed || !state_) H3D_FAIL(H3D_ERR_ARG, "dcn_v2_pack_weights_cached: null pointer");
    if (Cout <= 0 || C <= 0 || C % 16) H3D_FAIL(H3D_ERR_SHAPE, "dcn_v2_pack_weights_cached: C=%d must be a positive multiple of 16", C);
    if (dtype != H3D_F32 && dtype != H3D_BF16) H3D_FAIL(H3D_ERR_DTYPE, "dcn_v2_pack_weights_cached: dtype %d (f32 | bf16)", dtype);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *state = (unsigned long long *)state_;
    const void *bufs[2] = {weight, bias};
    const size_t nb[2] = {(size_t)Cout * C * 9 * 4, (size_t)Cout * 4};
    int rc = sig_begin(state, bufs, nb, 2, st);
    if (rc != H3D_OK) return rc;
    const int rows = (Cout + 127) / 128 * 128;
    const size_t wtotal = (size_t)rows * 9 * C;
    float *bp = (float *)((char *)packed + ws_align(wtotal * (dtype == H3D_F32 ? 4 : 2)));
    unsigned *wmax = dtype == H3D_F32 ? (unsigned *)(bp + rows) : nullptr;
    if (wmax) {
        hipLaunchKernelGGL(dcn_wmax_reset_if_kernel, dim3(1), dim3(1), 0, st, state, wmax);
        H3D_CHECK_LAUNCH("dcn_wmax_reset_if_kernel");
    }
    hipLaunchKernelGGL(dcn_w_pack_if_kernel, dim3((unsigned)((wtotal + 255) / 256)), dim3(256), 0, st, weight, bias, packed, bp, Cout, C, rows,
                       dtype == H3D_F32 ? 0 : 1, state, wmax);
    H3D_CHECK_LAUNCH("dcn_w_pack_if_kernel");
    hipLaunchKernelGGL(sig_commit_kernel, dim3(1), dim3(1), 0, st, state);
    H3D_CHECK_LAUNCH("sig_commit_kernel");
    return H3D_OK;
}

// The stand-alone `DCN` module's four parameters (dcn_v2.py:97-116: weight, bias, conv_offset_mask.{weight,bias}) in the fp32 layout of
// the fused DeformConv kernel (csrc/dcn3.hip): wp [rows][9][C] main filters, wo [128][9][C] offset/mask filters with the 27 channels
// spread over 32 MFMA rows -- value i = 3u + c of lane half h in row (i&3) + 8(i>>2) + 4h; half 0 = taps 0..4, half 1 = taps 5..8;
// c = 0: dh (channel 2 tap), 1: dw (2 tap + 1), 2: mask (18 + tap) -- and bias_out [rows main | 32 offset].  Validated as above.
__global__ void dcn_fused_pack_f32_if_kernel(const float *__restrict__ w, const float *__restrict__ bias, const float *__restrict__ ow,
                                             const float *__restrict__ ob, float *__restrict__ wp, float *__restrict__ wo, float *__restrict__ bo,
                                             int Cout, int C, int rows, const unsigned long long *__restrict__ state)
{
    if (state[0] == state[1]) return;
    unsigned *wmax = (unsigned *)(bo + rows + 32);         // [0] max |main filter|, [1] max |offset / mask filter| (reset by dcn_wmax_reset_if_kernel)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t main_total = (size_t)rows * 9 * C, total = main_total + (size_t)128 * 9 * C;
    auto src_channel = [](int row) {                       // MFMA row of the permuted offset conv -> conv_offset_mask channel (-1: unused)
        if (row >= 32) return -1;
        const int hh = (row >> 2) & 1, i3 = (row & 3) + 4 * (row >> 3);
        const int u = i3 / 3, c = i3 - 3 * u;
        if (u > 4 || (hh == 1 && u > 3)) return -1;
        const int tap = hh ? 5 + u : u;
        return c == 0 ? 2 * tap : c == 1 ? 2 * tap + 1 : 18 + tap;
    };
    if (i < (size_t)rows) bo[i] = i < (size_t)Cout ? bias[i] : 0.f;
    if (i < 32) { const int ch = src_channel((int)i); bo[rows + i] = ch >= 0 ? ob[ch] : 0.f; }
    const bool off = i >= main_total;          // (main_total is a multiple of 64: a wave lies on one side)
    float v = 0.f;
    if (i < total) {
        const size_t j = off ? i - main_total : i;
        const int c = (int)(j % C);
        const int tap = (int)((j / C) % 9);
        const int o = (int)(j / ((size_t)9 * C));
        if (!off) {
            v = o < Cout ? w[((size_t)o * C + c) * 9 + tap] : 0.f;
            wp[j] = v;
        } else {
            const int ch = src_channel(o);
            v = ch >= 0 ? ow[((size_t)ch * C + c) * 9 + tap] : 0.f;
            wo[j] = v;
        }
    }
    dcn_wmax_accumulate(wmax + (off ? 1 : 0), v);      // (every lane of the wave takes part in the shuffles)
}

extern "C" int h3d_dcn_fused_pack_f32_cached(const float *weight, const float *bias, const float *off_weight, const float *off_bias, int Cout, int C,
                                             float *wp, float *wo, float *bias_out, void *state_, void *stream)
{
    if (!weight || !bias || !off_weight || !off_bias || !wp || !wo || !bias_out || !state_) H3D_FAIL(H3D_ERR_ARG, "dcn_fused_pack_f32_cached: null pointer");
    if (Cout <= 0 || C <= 0 || C % 16) H3D_FAIL(H3D_ERR_SHAPE, "dcn_fused_pack_f32_cached: C=%d must be a positive multiple of 16", C);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *state = (unsigned long long *)state_;
    const void *bufs[4] = {weight, bias, off_weight, off_bias};
    const size_t nb[4] = {(size_t)Cout * C * 9 * 4, (size_t)Cout * 4, (size_t)27 * C * 9 * 4, (size_t)27 * 4};
    int rc = sig_begin(state, bufs, nb, 4, st);
    if (rc != H3D_OK) return rc;
    const int rows = (Cout + 127) / 128 * 128;
    const size_t total = ((size_t)rows + 128) * 9 * C;
    hipLaunchKernelGGL(dcn_wmax_reset_if_kernel, dim3(1), dim3(1), 0, st, state, (unsigned *)(bias_out + rows + 32));
    H3D_CHECK_LAUNCH("dcn_wmax_reset_if_kernel");
    hipLaunchKernelGGL(dcn_fused_pack_f32_if_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, weight, bias, off_weight, off_bias, wp, wo,
                       bias_out, Cout, C, rows, state);
    H3D_CHECK_LAUNCH("dcn_fused_pack_f32_if_kernel");
    hipLaunchKernelGGL(sig_commit_kernel, dim3(1), dim3(1), 0, st, state);
    H3D_CHECK_LAUNCH("sig_commit_kernel");
    return H3D_OK;
}

extern "C" size_t h3d_dcn_v2_packed_workspace_bytes(int B, int C, int H, int W, int flags)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    const size_t px = (size_t)B * H * W;
    return ws_align(px * 32 * 4) + ((flags & H3D_DCN_INPUT_NHWC) ? 0 : ws_align(px * C * 4));
}

extern "C" int h3d_dcn_v2_forward_packed(const void *input, const void *packed, const float *offset, const float *mask, void *output, int B,
                                         int C, int H, int W, int Cout, int dtype, int flags, void *workspace, size_t workspace_bytes,
                                         void *stream)
{
    if (!input || !packed || !offset || !mask || !output || !workspace) H3D_FAIL(H3D_ERR_ARG, "dcn_v2_forward_packed: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || C <= 0 || C % 16 || H > 32767 || W > 32767)
        H3D_FAIL(H3D_ERR_SHAPE, "dcn_v2_forward_packed: B=%d C=%d (multiple of 16) H=%d W=%d Cout=%d", B, C, H, W, Cout);
    if (dtype != H3D_F32 && dtype != H3D_BF16) H3D_FAIL(H3D_ERR_DTYPE, "dcn_v2_forward_packed: dtype %d (f32 | bf16)", dtype);
    if (dtype == H3D_BF16 && !(flags & H3D_DCN_INPUT_NHWC)) H3D_FAIL(H3D_ERR_UNSUPPORTED, "dcn_v2_forward_packed: bf16 input must be channels-last");
    if ((flags & H3D_DCN_OUTPUT_NHWC) && Cout % 4) H3D_FAIL(H3D_ERR_SHAPE, "dcn_v2_forward_packed: channels-last output needs Cout %% 4 == 0");
    if (workspace_bytes < h3d_dcn_v2_packed_workspace_bytes(B, C, H, W, flags))
        H3D_FAIL(H3D_ERR_ARG, "dcn_v2_forward_packed: workspace of %zu bytes, %zu needed", workspace_bytes, h3d_dcn_v2_packed_workspace_bytes(B, C, H, W, flags));
    hipStream_t st = (hipStream_t)stream;
    const int rows = (Cout + 127) / 128 * 128;
    const size_t px = (size_t)B * H * W;
    char *ws = (char *)workspace;
    float *om = (float *)ws;                    ws += ws_align(px * 32 * 4);
    const void *x = input;
    if (!(flags & H3D_DCN_INPUT_NHWC)) {
        int rc = h3d_nchw_f32_to_nhwc((const float *)input, ws, H3D_F32, B, C, H, W, C, stream);
        if (rc != H3D_OK) return rc;
        x = ws;
    }
    hipLaunchKernelGGL(dcn_om_pack_kernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, st, offset, mask, om, H * W, px);
    H3D_CHECK_LAUNCH("dcn_om_pack_kernel");
    h3d_op op = {};
    op.kind = H3D_OP_DCN;
    op.dtype = dtype == H3D_F32 && !(flags & H3D_DCN_F32_MFMA) && !dcn_op_f32_mfma() ? H3D_F16X3 : dtype;
    op.in = x; op.in2 = om; op.w = packed;
    op.bias = (const float *)((const char *)packed + ws_align((size_t)rows * 9 * C * (dtype == H3D_F32 ? 4 : 2)));
    op.out = output;
    op.B = B; op.H = H; op.W = W; op.Cin = C; op.in_cs = C; op.in2_cs = 32; op.Ho = H; op.Wo = W; op.Cout = Cout; op.out_cs = Cout;
    op.ksize = 3; op.stride = 1; op.relu = 0; op.out_mode = (flags & H3D_DCN_OUTPUT_NHWC) ? H3D_OUT_NHWC : H3D_OUT_NCHW_F32; op.wrows = rows;
    op.reserved = 0x800 | (op.dtype == H3D_F16X3 ? 0x100000 : 0);      // the mask operand is final; fp32 pack + max |w| behind the bias
    return h3d_launch_dcn2(op, st);
}
