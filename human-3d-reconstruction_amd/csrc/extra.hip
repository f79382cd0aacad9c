// Ops that only the backbones of BASELINE configs 4 and 5 need (Hourglass-104, ResNet-101-DCN; SURVEY 8f-4: the reference
// names them in opts.py:61-63 / experiments/*.sh but ships no source, so these follow the published CenterNet
// definitions).  All of them are data movement (HBM-bound, 16 B per lane); every contraction stays on the conv / DCN
// kernels of the DLA path:
//   H3D_OP_IM2COL       the 7x7 stride-2 stem conv (3 -> 64 / 128 channels) as im2col + the MFMA 1x1 conv: NCHW fp32
//                       images -> NHWC patches [B,Ho,Wo,Kpad], channel k = c*49 + ky*7 + kx (the order of
//                       weight.reshape(Cout, -1)), zero from 147 up to Kpad = 160
//   H3D_OP_MAXPOOL3     nn.MaxPool2d(3, stride 2, padding 1) (ResNet stem)
//   H3D_OP_DEPTH2SPACE  [B,H,W,4C] -> [B,2H,2W,C]: group g = 2*py + px of the channels is the output pixel
//                       (2y+py, 2x+px); with it ConvTranspose2d(C, C, 4, stride 2, padding 1) of the ResNet-DCN up
//                       layers runs as ONE 3x3 conv with 4C output channels (the four 2x2 sub-pixel kernels zero-padded
//                       to 3x3, engine.PackedWeights.deconv4_as_conv3)
#include "common.h"

template <typename T>
__global__ void im2col_kernel(const float *__restrict__ img, T *__restrict__ out, int B, int Cin, int H, int W, int Ho, int Wo,
                              int ks, int stride, int pad, int K, int Kpad, int out_cs)
{
    constexpr int N = 16 / sizeof(T);
    const int vpp = Kpad / N;
    const size_t total = (size_t)B * Ho * Wo * vpp;
    const int kk = ks * ks;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vpp);
        const size_t p = i / vpp;
        const int ox = (int)(p % Wo);
        const size_t q = p / Wo;
        const int oy = (int)(q % Ho), b = (int)(q / Ho);
        float val[N];
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const int k = v * N + e;
            float x = 0.f;
            if (k < K) {
                const int c = k / kk, r = k - c * kk;
                const int ky = r / ks, kx = r - ky * ks;
                const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) x = img[(((size_t)b * Cin + c) * H + iy) * W + ix];
            }
            val[e] = x;
        }
        *reinterpret_cast<u32x4 *>(out + p * out_cs + v * N) = pack16<T>(val);
    }
}

template <typename T>
__global__ void maxpool3_kernel(const T *__restrict__ in, T *__restrict__ out, int B, int H, int W, int C, int in_cs, int Ho, int Wo,
                                int out_cs)
{
    constexpr int N = 16 / sizeof(T);
    const int vpc = C / N;
    const size_t total = (size_t)B * Ho * Wo * vpc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vpc);
        const size_t p = i / vpc;
        const int ox = (int)(p % Wo);
        const size_t q = p / Wo;
        const int oy = (int)(q % Ho), b = (int)(q / Ho);
        float m[N], x[N];
#pragma unroll
        for (int k = 0; k < N; ++k) m[k] = -__builtin_inff();
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if (iy < 0 || iy >= H) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if (ix < 0 || ix >= W) continue;
                unpack16<T>(*reinterpret_cast<const u32x4 *>(in + (((size_t)b * H + iy) * W + ix) * in_cs + v * N), x);
#pragma unroll
                for (int k = 0; k < N; ++k) m[k] = fmaxf(m[k], x[k]);
            }
        }
        *reinterpret_cast<u32x4 *>(out + p * out_cs + v * N) = pack16<T>(m);
    }
}

template <typename T>
__global__ void depth2space_kernel(const T *__restrict__ in, T *__restrict__ out, int B, int H, int W, int C, int in_cs, int out_cs)
{
    constexpr int N = 16 / sizeof(T);
    const int vpc = C / N;
    const size_t total = (size_t)B * (2 * H) * (2 * W) * vpc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vpc);
        const size_t p = i / vpc;
        const int ox = (int)(p % (2 * W));
        const size_t q = p / (2 * W);
        const int oy = (int)(q % (2 * H)), b = (int)(q / (2 * H));
        const int g = (oy & 1) * 2 + (ox & 1);
        *reinterpret_cast<u32x4 *>(out + p * out_cs + v * N) =
            *reinterpret_cast<const u32x4 *>(in + (((size_t)b * H + (oy >> 1)) * W + (ox >> 1)) * in_cs + g * C + v * N);
    }
}

static inline int ex_grid(size_t total) { size_t g = (total + 255) / 256; return (int)(g > 2048 * 8 ? 2048 * 8 : (g ? g : 1)); }

int h3d_launch_extra(const h3d_op &op, hipStream_t st)
{
    if (!op.in || !op.out) H3D_FAIL(H3D_ERR_ARG, "extra op: null pointer");
    const int es = op.dtype == H3D_BF16 ? 2 : (op.dtype == H3D_F32 ? 4 : 0);
    if (!es) H3D_FAIL(H3D_ERR_DTYPE, "extra op: dtype %d", op.dtype);
    const int n = 16 / es;
    const dim3 blk(256);
    if (op.kind == H3D_OP_IM2COL) {
        const int K = op.Cin * op.ksize * op.ksize, pad = op.ksize / 2;
        if (op.ksize < 1 || op.stride < 1 || op.Cout < K || op.Cout % n || op.out_cs % n || op.out_cs < op.Cout)
            H3D_FAIL(H3D_ERR_SHAPE, "im2col: K = %d, padded K (Cout) = %d, stride %d", K, op.Cout, op.out_cs);
        if (op.Ho != (op.H + 2 * pad - op.ksize) / op.stride + 1 || op.Wo != (op.W + 2 * pad - op.ksize) / op.stride + 1)
            H3D_FAIL(H3D_ERR_SHAPE, "im2col: output %dx%d does not match (H+2p-k)/s+1", op.Ho, op.Wo);
        if (h3d_note_kernel("im2col_kernel<%s>", es == 2 ? "unsigned short" : "float")) return H3D_OK;
        const size_t total = (size_t)op.B * op.Ho * op.Wo * (op.Cout / n);
        if (es == 2)
            hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(ex_grid(total)), blk, 0, st, (const float *)op.in, (bf16_t *)op.out, op.B, op.Cin,
                               op.H, op.W, op.Ho, op.Wo, op.ksize, op.stride, pad, K, op.Cout, op.out_cs);
        else
            hipLaunchKernelGGL(im2col_kernel<float>, dim3(ex_grid(total)), blk, 0, st, (const float *)op.in, (float *)op.out, op.B, op.Cin,
                               op.H, op.W, op.Ho, op.Wo, op.ksize, op.stride, pad, K, op.Cout, op.out_cs);
        H3D_CHECK_LAUNCH("im2col_kernel");
        return H3D_OK;
    }
    if (op.Cin % n || op.in_cs % n || op.out_cs % n) H3D_FAIL(H3D_ERR_SHAPE, "extra op: channels %d strides %d/%d must be multiples of %d", op.Cin, op.in_cs, op.out_cs, n);
    if (op.kind == H3D_OP_MAXPOOL3) {
        if (op.Cin != op.Cout || op.Ho != (op.H - 1) / 2 + 1 || op.Wo != (op.W - 1) / 2 + 1)
            H3D_FAIL(H3D_ERR_SHAPE, "maxpool3: output must be floor((H-1)/2)+1 (k3 s2 p1)");
        if (h3d_note_kernel("maxpool3_kernel<%s>", es == 2 ? "unsigned short" : "float")) return H3D_OK;
        const size_t total = (size_t)op.B * op.Ho * op.Wo * (op.Cin / n);
        if (es == 2)
            hipLaunchKernelGGL(maxpool3_kernel<bf16_t>, dim3(ex_grid(total)), blk, 0, st, (const bf16_t *)op.in, (bf16_t *)op.out, op.B, op.H, op.W,
                               op.Cin, op.in_cs, op.Ho, op.Wo, op.out_cs);
        else
            hipLaunchKernelGGL(maxpool3_kernel<float>, dim3(ex_grid(total)), blk, 0, st, (const float *)op.in, (float *)op.out, op.B, op.H, op.W,
                               op.Cin, op.in_cs, op.Ho, op.Wo, op.out_cs);
        H3D_CHECK_LAUNCH("maxpool3_kernel");
        return H3D_OK;
    }
    if (op.kind == H3D_OP_DEPTH2SPACE) {
        if (op.Cin != 4 * op.Cout || op.Cout % n || op.Ho != 2 * op.H || op.Wo != 2 * op.W)
            H3D_FAIL(H3D_ERR_SHAPE, "depth2space: [B,H,W,4C] -> [B,2H,2W,C] expected (Cin=%d Cout=%d)", op.Cin, op.Cout);
        if (h3d_note_kernel("depth2space_kernel<%s>", es == 2 ? "unsigned short" : "float")) return H3D_OK;
        const size_t total = (size_t)op.B * op.Ho * op.Wo * (op.Cout / n);
        if (es == 2)
            hipLaunchKernelGGL(depth2space_kernel<bf16_t>, dim3(ex_grid(total)), blk, 0, st, (const bf16_t *)op.in, (bf16_t *)op.out, op.B, op.H,
                               op.W, op.Cout, op.in_cs, op.out_cs);
        else
            hipLaunchKernelGGL(depth2space_kernel<float>, dim3(ex_grid(total)), blk, 0, st, (const float *)op.in, (float *)op.out, op.B, op.H,
                               op.W, op.Cout, op.in_cs, op.out_cs);
        H3D_CHECK_LAUNCH("depth2space_kernel");
        return H3D_OK;
    }
    H3D_FAIL(H3D_ERR_ARG, "extra op: kind %d", op.kind);
}
