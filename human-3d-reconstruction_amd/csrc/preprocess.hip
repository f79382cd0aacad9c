// Input pre-process on the GPU: the val branch of the reference's dataset loader,
// datasets/coco_hp.py:151-212 `_get_input`: centre c = (w/2, h/2), scale s = max(w, h), rot = 0 ->
// get_affine_transform (utils/image.py:27-62) -> cv2.warpAffine(INTER_LINEAR, constant border 0) to
// res x res -> /255 -> (x - mean) / std -> HWC -> CHW.
//
// cv2 is not present in this image, so the warp restates OpenCV's published fixed-point scheme
// (modules/imgproc/src/imgwarp.cpp, WarpAffineInvoker + remapBilinear):
//   X = (cvRound((M01*y + M02) * 1024) + 16 + cvRound(M00 * x * 1024)) >> 5      (5 fractional bits)
//   sx = X >> 5, fx = X & 31 (same for Y); weights (32-fx)(32-fy)*32 ... sum 32768;
//   dst = (sum w_i * src_i + 16384) >> 15, neighbours outside the image contribute 0.
// with M the INVERSE 2x3 map (dst -> src) in double, supplied by the host.  Integer arithmetic makes the
// kernel bit-exact with the numpy restatement in oracle/preprocess.py; "parity unpinned" against cv2 itself.
// This file is compiled with -ffp-contract=off: the float tail is the reference's three float32 ops.
#include "common.h"

__global__ void preprocess_kernel(const uint8_t *__restrict__ img, int h, int w, int row_bytes, size_t img_bytes,
                                  const double *__restrict__ minv, const float *__restrict__ mean,
                                  const float *__restrict__ stdv, int res_h, int res_w, float *__restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= res_w) return;
    const double *M = minv + 6 * b;
    const uint8_t *src = img + (size_t)b * img_bytes;
    const int X0 = __double2int_rn((M[1] * y + M[2]) * 1024.0) + 16;
    const int Y0 = __double2int_rn((M[4] * y + M[5]) * 1024.0) + 16;
    const int X = (X0 + __double2int_rn(M[0] * x * 1024.0)) >> 5;
    const int Y = (Y0 + __double2int_rn(M[3] * x * 1024.0)) >> 5;
    const int sx = X >> 5, sy = Y >> 5, fx = X & 31, fy = Y & 31;
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    const bool x0 = sx >= 0 && sx < w, x1 = sx + 1 >= 0 && sx + 1 < w, y0 = sy >= 0 && sy < h, y1 = sy + 1 >= 0 && sy + 1 < h;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int p00 = (x0 && y0) ? src[(size_t)sy * row_bytes + sx * 3 + c] : 0;
        const int p01 = (x1 && y0) ? src[(size_t)sy * row_bytes + (sx + 1) * 3 + c] : 0;
        const int p10 = (x0 && y1) ? src[(size_t)(sy + 1) * row_bytes + sx * 3 + c] : 0;
        const int p11 = (x1 && y1) ? src[(size_t)(sy + 1) * row_bytes + (sx + 1) * 3 + c] : 0;
        const int v = (w00 * p00 + w01 * p01 + w10 * p10 + w11 * p11 + 16384) >> 15;     // 0..255
        const float f = (float)v / 255.0f;
        out[(((size_t)b * 3 + c) * res_h + y) * res_w + x] = (f - mean[c]) / stdv[c];
    }
}

extern "C" int h3d_preprocess(const uint8_t *images, int B, int h, int w, int row_bytes, const double *minv,
                              const float *mean, const float *stdv, int res_h, int res_w, float *out, void *stream)
{
    if (!images || !minv || !mean || !stdv || !out) H3D_FAIL(H3D_ERR_ARG, "preprocess: null pointer");
    if (B <= 0 || h <= 0 || w <= 0 || res_h <= 0 || res_w <= 0 || row_bytes < 3 * w)
        H3D_FAIL(H3D_ERR_SHAPE, "preprocess: B=%d %dx%d (row %d bytes) -> %dx%d", B, h, w, row_bytes, res_h, res_w);
    if (h > 32767 || w > 32767) H3D_FAIL(H3D_ERR_SHAPE, "preprocess: image larger than 32767 (fixed-point warp)");
    hipLaunchKernelGGL(preprocess_kernel, dim3(cdiv(res_w, 128), res_h, B), dim3(128), 0, (hipStream_t)stream, images, h, w,
                       row_bytes, (size_t)h * row_bytes, minv, mean, stdv, res_h, res_w, out);
    H3D_CHECK_LAUNCH("preprocess_kernel");
    return H3D_OK;
}
