// Sampling geometry of one (pixel, tap) in the reference's float arithmetic (dcn_v2_im2col_cuda.cu:25-54, 163-185): shared by the operator
// boundary's general kernel (csrc/dcn.hip) and the first-generation network kernel (csrc/dcn1.hip, `make EXTRA=1`).
#pragma once
#include "common.h"

// sampling geometry of one (pixel, tap): exactly the reference's float arithmetic
struct Sample {
    int off[4];      // element offsets (pixel index) of the 4 corners, -1 = contributes zero
    float w[4];      // hh*hw, hh*lw, lh*hw, lh*lw
    float mask;
    bool inside;
};

__device__ __forceinline__ Sample make_sample(float h_im, float w_im, float mask, int H, int W)
{
    Sample s;
    s.mask = mask;
    s.inside = (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W);
    const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
    const int h_high = h_low + 1, w_high = w_low + 1;
    const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
    const float hh = 1.f - lh, hw = 1.f - lw;
    s.w[0] = hh * hw; s.w[1] = hh * lw; s.w[2] = lh * hw; s.w[3] = lh * lw;
    s.off[0] = (s.inside && h_low >= 0 && w_low >= 0) ? h_low * W + w_low : -1;
    s.off[1] = (s.inside && h_low >= 0 && w_high <= W - 1) ? h_low * W + w_high : -1;
    s.off[2] = (s.inside && h_high <= H - 1 && w_low >= 0) ? h_high * W + w_low : -1;
    s.off[3] = (s.inside && h_high <= H - 1 && w_high <= W - 1) ? h_high * W + w_high : -1;
    return s;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

