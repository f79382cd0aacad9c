// Ops that only the backbones of BASELINE configs 4 and 5 need (Hourglass-104, ResNet-101-DCN; SURVEY 8f-4: the reference
// names them in opts.py:61-63 / experiments/*.sh but ships no source, so these follow the published CenterNet
// definitions).  All of them are data movement (HBM-bound, 16 B per lane); every contraction stays on the conv / DCN
// kernels of the DLA path:
//   H3D_OP_IM2COL       the 7x7 stride-2 stem conv (3 -> 64 / 128 channels) as im2col + the MFMA 1x1 conv: NCHW fp32
//                       images -> NHWC patches [B,Ho,Wo,Kpad], channel k = c*49 + ky*7 + kx (the order of
//                       weight.reshape(Cout, -1)), zero from 147 up to Kpad = 160
//                       (fp32 plans; bf16 plans run the convolution itself: stem_s2_kernel below, H3D_OP_STEM with stride 2)
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

// The 7x7 stride-2 stem convolution itself (bf16 plans): Conv2d(3, Cout, 7, stride 2, padding 3) + BN + ReLU from NCHW fp32
// images to NHWC bf16 on v_mfma_f32_16x16x32_bf16, csrc/conv.hip's stem_mfma_kernel with stride 2 and more channel tiles.
// im2col + 1x1 conv wrote and re-read a [B,Ho,Wo,160] patch tensor (1.5 GB at 32 x 768 x 768: 1.19 + 0.56 ms per step of
// ResNet-101-DCN); here the image patch of a 8 x 32 output tile (21 x 69 pixels) is staged in LDS as interleaved pixels of
// 4 bf16 (c0, c1, c2, 0), so the 7 taps of a filter row of output pixel ox are ONE contiguous K = 28 (+4 zero-weight) run
// starting at input pixel 2 ox: lane (p, q) reads its 8 K elements with a single aligned ds_read_b128 (pixels 2 ox + 2q,
// + 2q + 1).  Wave w owns 16 output channels (blockIdx.y * 64 + 16 w), keeps their 7 filter-row fragments in registers and
// walks the tile's 16 groups of 16 pixels.  w: bf16 [Cout][7][32] (k = dx*4 + c, engine.PackedWeights.stem_s2), bias fp32.
typedef __attribute__((ext_vector_type(4))) float f32x4_st;
template <typename T>      // bf16_t | f16_t
__global__ __launch_bounds__(256) void stem_s2_kernel(const float *__restrict__ img, const T *__restrict__ w,
                                                      const float *__restrict__ bias, T *__restrict__ out, int B, int H,
                                                      int W, int Ho, int Wo, int Cout, int out_cs, int tiles_x, int tiles_y)
{
    constexpr int TH = 8, TW = 32, IH = (TH - 1) * 2 + 7, IW = 72;      // 21 rows x (62 + 7 + 2 -> 72) pixels
    __shared__ __attribute__((aligned(16))) uint2 s[IH][IW];
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, p = l & 15, q = l >> 4;
    const int tiles = tiles_x * tiles_y;
    const int b = blockIdx.x / tiles;
    const int t = blockIdx.x - b * tiles;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int c0 = blockIdx.y * 64 + wv * 16;
    u32x4 fa[7];
#pragma unroll
    for (int dy = 0; dy < 7; ++dy) fa[dy] = u32x4{0u, 0u, 0u, 0u};
    if (c0 < Cout) {
#pragma unroll
        for (int dy = 0; dy < 7; ++dy) fa[dy] = *reinterpret_cast<const u32x4 *>(w + ((size_t)(c0 + p) * 7 + dy) * 32 + 8 * q);
    }
    const size_t plane = (size_t)H * W;
    const float *im = img + (size_t)b * 3 * plane;
    for (int i = tid; i < IH * IW; i += 256) {
        const int iy = i / IW, ix = i - iy * IW;
        const int gy = 2 * oy0 - 3 + iy, gx = 2 * ox0 - 3 + ix;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const size_t o = (size_t)gy * W + gx;
            v0 = im[o]; v1 = im[plane + o]; v2 = im[2 * plane + o];
        }
        s[iy][ix] = uint2{EP<T>::pack2(v0, v1), EP<T>::pack2(v2, 0.f)};
    }
    __syncthreads();
    if (c0 >= Cout) return;
    float bs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bs[i] = bias[c0 + 4 * q + i];
    // two groups per iteration: two independent MFMA chains
#pragma unroll 1
    for (int g = 0; g < TH * TW / 16; g += 2) {
        f32x4_st acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const int py = g >> 1;
#pragma unroll
        for (int dy = 0; dy < 7; ++dy)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const u32x4 fb = *reinterpret_cast<const u32x4 *>(&s[2 * py + dy][2 * (16 * u + p) + 2 * q]);
                if constexpr (std::is_same_v<T, f16_t>)
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fa[dy]), __builtin_bit_cast(f16x8_t, fb), acc[u], 0, 0, 0);
                else
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[dy]), __builtin_bit_cast(bf16x8_t, fb), acc[u], 0, 0, 0);
            }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int oy = oy0 + py, ox = ox0 + 16 * u + p;
            if (oy < Ho && ox < Wo)
                store4<T>(out + ((size_t)(b * Ho + oy) * Wo + ox) * out_cs + c0 + 4 * q, fmaxf(acc[u][0] + bs[0], 0.f),
                               fmaxf(acc[u][1] + bs[1], 0.f), fmaxf(acc[u][2] + bs[2], 0.f), fmaxf(acc[u][3] + bs[3], 0.f));
        }
    }
}

int h3d_launch_stem_s2(const h3d_op &op, hipStream_t st)
{
    if (op.dtype != H3D_BF16 && op.dtype != H3D_F16) H3D_FAIL(H3D_ERR_DTYPE, "stem (stride 2): bf16 / fp16 plans only (dtype %d)", op.dtype);
    if (op.Cin != 3 || op.ksize != 7 || op.stride != 2 || op.Cout % 16 || op.out_cs % 4 || op.out_cs < op.Cout ||
        op.Ho != (op.H - 1) / 2 + 1 || op.Wo != (op.W - 1) / 2 + 1)
        H3D_FAIL(H3D_ERR_SHAPE, "stem (stride 2): expects 7x7 3->16n, output floor((H-1)/2)+1 (got k=%d %d->%d, %dx%d -> %dx%d)", op.ksize,
                 op.Cin, op.Cout, op.H, op.W, op.Ho, op.Wo);
    if (((uintptr_t)op.w & 15) || ((uintptr_t)op.out & 7)) H3D_FAIL(H3D_ERR_ARG, "stem (stride 2): weights must be 16-byte aligned");
    const int tx = cdiv(op.Wo, 32), ty = cdiv(op.Ho, 8);
    if (h3d_note_kernel("stem_s2_kernel<%s>", op.dtype == H3D_F16 ? "f16_t" : "unsigned short")) return H3D_OK;
    if (op.dtype == H3D_F16)
        hipLaunchKernelGGL(stem_s2_kernel<f16_t>, dim3(op.B * tx * ty, cdiv(op.Cout, 64)), dim3(256), 0, st, (const float *)op.in, (const f16_t *)op.w,
                           op.bias, (f16_t *)op.out, op.B, op.H, op.W, op.Ho, op.Wo, op.Cout, op.out_cs, tx, ty);
    else
        hipLaunchKernelGGL(stem_s2_kernel<bf16_t>, dim3(op.B * tx * ty, cdiv(op.Cout, 64)), dim3(256), 0, st, (const float *)op.in, (const bf16_t *)op.w,
                           op.bias, (bf16_t *)op.out, op.B, op.H, op.W, op.Ho, op.Wo, op.Cout, op.out_cs, tx, ty);
    H3D_CHECK_LAUNCH("stem_s2_kernel");
    return H3D_OK;
}

int h3d_launch_extra(const h3d_op &op, hipStream_t st)
{
    if (!op.in || !op.out) H3D_FAIL(H3D_ERR_ARG, "extra op: null pointer");
    const int es = h3d_dtype_bytes(op.dtype);
    if (!es) H3D_FAIL(H3D_ERR_DTYPE, "extra op: dtype %d", op.dtype);
    const bool f16 = op.dtype == H3D_F16;
    const int n = 16 / es;
    const dim3 blk(256);
    if (op.kind == H3D_OP_IM2COL) {
        const int K = op.Cin * op.ksize * op.ksize, pad = op.ksize / 2;
        if (op.ksize < 1 || op.stride < 1 || op.Cout < K || op.Cout % n || op.out_cs % n || op.out_cs < op.Cout)
            H3D_FAIL(H3D_ERR_SHAPE, "im2col: K = %d, padded K (Cout) = %d, stride %d", K, op.Cout, op.out_cs);
        if (op.Ho != (op.H + 2 * pad - op.ksize) / op.stride + 1 || op.Wo != (op.W + 2 * pad - op.ksize) / op.stride + 1)
            H3D_FAIL(H3D_ERR_SHAPE, "im2col: output %dx%d does not match (H+2p-k)/s+1", op.Ho, op.Wo);
        if (h3d_note_kernel("im2col_kernel<%s>", f16 ? "f16_t" : es == 2 ? "unsigned short" : "float")) return H3D_OK;
        const size_t total = (size_t)op.B * op.Ho * op.Wo * (op.Cout / n);
        if (f16)
            hipLaunchKernelGGL(im2col_kernel<f16_t>, dim3(ex_grid(total)), blk, 0, st, (const float *)op.in, (f16_t *)op.out, op.B, op.Cin,
                               op.H, op.W, op.Ho, op.Wo, op.ksize, op.stride, pad, K, op.Cout, op.out_cs);
        else if (es == 2)
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
        if (h3d_note_kernel("maxpool3_kernel<%s>", f16 ? "f16_t" : es == 2 ? "unsigned short" : "float")) return H3D_OK;
        const size_t total = (size_t)op.B * op.Ho * op.Wo * (op.Cin / n);
        if (f16)
            hipLaunchKernelGGL(maxpool3_kernel<f16_t>, dim3(ex_grid(total)), blk, 0, st, (const f16_t *)op.in, (f16_t *)op.out, op.B, op.H, op.W,
                               op.Cin, op.in_cs, op.Ho, op.Wo, op.out_cs);
        else if (es == 2)
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
        if (h3d_note_kernel("depth2space_kernel<%s>", f16 ? "f16_t" : es == 2 ? "unsigned short" : "float")) return H3D_OK;
        const size_t total = (size_t)op.B * op.Ho * op.Wo * (op.Cout / n);
        if (f16)
            hipLaunchKernelGGL(depth2space_kernel<f16_t>, dim3(ex_grid(total)), blk, 0, st, (const f16_t *)op.in, (f16_t *)op.out, op.B, op.H,
                               op.W, op.Cout, op.in_cs, op.out_cs);
        else if (es == 2)
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
