/*
 * h3d.h -- C ABI of libh3d_hip.so: the MI355X (gfx950) implementation of the multi_pose
 * inference hot path of Aaron20127/human-3d-reconstruction.
 *
 * Plain C: raw DEVICE pointers, sizes, an explicit hipStream_t (passed as void*), int error
 * codes.  No torch types.  Every entry point is asynchronous on `stream`, allocates nothing,
 * keeps no global state and is re-entrant (the reference launches on the current stream and
 * keeps a global THCState, DCNv2/src/cuda/dcn_v2_cuda.cu:12,108 -- we take the stream instead).
 *
 * Citations (file:line) are relative to /root/reference/src/lib/models/.
 */
#ifndef H3D_H
#define H3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes (the reference raises C++ exceptions via AT_ASSERTM / AT_ERROR,
 *      DCNv2/src/cuda/dcn_v2_cuda.cu:61-85; kernel launch failures are only printf-ed there,
 *      dcn_v2_im2col_cuda.cu:346-350 -- here they are returned) */
#define H3D_OK 0
#define H3D_ERR_SHAPE (-1)       /* "Input shape and kernel shape wont match" class of errors   */
#define H3D_ERR_DTYPE (-2)       /* unsupported element type                                     */
#define H3D_ERR_LAUNCH (-3)      /* hipGetLastError() != hipSuccess after a launch              */
#define H3D_ERR_UNSUPPORTED (-4) /* valid request outside what the kernels cover (message set)  */
#define H3D_ERR_ARG (-5)         /* null pointer / bad enum                                      */

#define H3D_F32 0
#define H3D_BF16 1
#define H3D_F16 2 /* IEEE fp16 activations and weights, fp32 accumulation (BASELINE configs[4]: `--arch resdcn_101` runs in fp16,
                     experiments/ctdet_coco_resdcn101.sh:3); stored values saturate at +-65504 */
#define H3D_F16X3 3 /* the parity arithmetic on the fp16 matrix cores (round 5): fp32 activations in memory exactly as H3D_F32 (same ops,
                       same layouts), every contraction with both fp32 operands split into two fp16 terms x = hi + lo and three products
                       hi.hi + hi.lo + lo.hi accumulated in fp32 (the dropped lo.lo is 2^-22 relative): fp32-level results at ~3x the
                       H3D_F32 plan's rate.  Packed filters hold, per 8 consecutive input channels, 8 fp16 hi terms then 8 fp16 lo terms
                       in the 32 bytes the 8 fp32 values would occupy (engine.PackedWeights / h3d_x3_split).  |activation| <= 65504. */

/* Last error text of the calling thread (thread_local), for the Python shim's RuntimeError. */
const char *h3d_last_error(void);
/* ABI version of this header; the loader checks it. */
int h3d_abi_version(void);
#define H3D_ABI_VERSION 3      /* 2: h3d_op.wexp / wexp2 (120-byte descriptor); 3: fp32 DeformConv packs carry their filter maxima (the sizes
                                  h3d_dcn_v2_packed_weight_bytes / _workspace_bytes return grew by 256 B; bias_out of h3d_dcn_fused_pack_f32_cached
                                  is [rows | 32 | 64] floats) */
/* How this library was built: H3D_BUILD_EXTRA = `make EXTRA=1` (the superseded kernel generations kept as A/B references are in:
 * H3D_OP_DCN_V1, H3D_OP_DCN_FUSED_F16, H3D_OP_UPDCN_F16, the 0x4000 DeformConv variant, h3d_smpl_verts2 -- without it they return
 * H3D_ERR_UNSUPPORTED); H3D_BUILD_ABLATE = `make ABLATE=1` (profiling switches and in-kernel stamps compiled in). */
#define H3D_BUILD_EXTRA 1
#define H3D_BUILD_ABLATE 2
int h3d_build_flags(void);

/* =====================================================================================
 * 1. Operator boundary: replaces pybind module `_ext` (DCNv2/src/vision.cpp:4-9), function
 *    `dcn_v2_forward` (DCNv2/src/dcn_v2.h:9-39 -> dcn_v2_cuda_forward, dcn_v2_cuda.cu:43-173).
 *    Same operands, same positional meaning; layouts as the reference: contiguous NCHW fp32.
 *      input  [B,C,H,W]   weight [Cout,C,kh,kw]   bias [Cout]
 *      offset [B,2*dg*kh*kw,Ho,Wo]  (channel 2t = dh, 2t+1 = dw of tap t, im2col.cu:170-171)
 *      mask   [B,dg*kh*kw,Ho,Wo]
 *      output [B,Cout,Ho,Wo], Ho=(H+2ph-(dh(kh-1)+1))/sh+1 (dcn_v2_cuda.cu:87-88)
 *    The caller allocates `output` (the reference callee allocates it, dcn_v2_cuda.cu:92).
 *    No scratch: the `columns`/`ones` buffers and pointer tables of the reference
 *    (dcn_v2_cuda.cu:90-103) do not exist -- sampling feeds the contraction through LDS.
 * ===================================================================================== */
int h3d_dcn_v2_forward(const float *input, const float *weight, const float *bias,
                       const float *offset, const float *mask, float *output,
                       int B, int C, int H, int W, int Cout,
                       int kernel_h, int kernel_w, int stride_h, int stride_w,
                       int pad_h, int pad_w, int dilation_h, int dilation_w,
                       int deformable_group, void *stream);

/* The convolution in front of the operator inside the `DCN` module (dcn_v2.py:107-111, 119-124), for ANY module configuration:
 *   out = Conv2d(C, 3*dg*kh*kw, (kh,kw), stride, padding)(input);  o1, o2, m = chunk(out, 3, dim=1)
 *   offset [B,2*dg*kh*kw,Ho,Wo] = cat(o1, o2)    mask [B,dg*kh*kw,Ho,Wo] = sigmoid(m)
 * off_weight [3*dg*kh*kw, C, kh, kw], off_bias [3*dg*kh*kw]; Ho = (H + 2*pad_h - kh) / stride_h + 1 (the reference passes no dilation to
 * this convolution).  Runs the general operator kernel with zero offsets and a unit mask -- no vendor library on any DCN path. */
int h3d_dcn_offset_mask(const float *input, const float *off_weight, const float *off_bias, float *offset, float *mask,
                        int B, int C, int H, int W, int kernel_h, int kernel_w, int stride_h, int stride_w,
                        int pad_h, int pad_w, int deformable_group, void *stream);

/* The same operator with a caller-provided device workspace of h3d_dcn_v2_workspace_bytes(...) bytes (the reference
 * callee allocates its own scratch: `columns`, `ones`, pointer tables, dcn_v2_cuda.cu:90-103).  For the configuration the
 * model uses (model.py:355: 3x3, stride 1, pad 1, dilation 1, deformable_group 1) and C % 16 == 0 the operands are re-laid
 * into the network kernels' layout inside the workspace and the contraction runs on the LDS-apron + MFMA kernel (csrc/dcn2.hip: since
 * round 5 three fp16 MFMAs on split operands per fp32 product, see H3D_DCN_F32_MFMA below); every other configuration (or a NULL / short workspace) takes h3d_dcn_v2_forward's general kernel. */
size_t h3d_dcn_v2_workspace_bytes(int B, int C, int H, int W, int Cout);
int h3d_dcn_v2_forward_ws(const float *input, const float *weight, const float *bias,
                          const float *offset, const float *mask, float *output,
                          int B, int C, int H, int W, int Cout,
                          int kernel_h, int kernel_w, int stride_h, int stride_w,
                          int pad_h, int pad_w, int dilation_h, int dilation_w,
                          int deformable_group, void *workspace, size_t workspace_bytes, void *stream);

/* The operator's THROUGHPUT form: the same contraction with the per-call work of the reference contract taken out.  The filters
 * are packed once (h3d_dcn_v2_pack_weights into h3d_dcn_v2_packed_weight_bytes(...) bytes; `dtype` H3D_F32 = fp32 tensors (split-operand
 * fp16 MFMAs, or exact fmaf chains on the fp32 matrix instruction with H3D_DCN_F32_MFMA), H3D_BF16 = fp16 filters + fp16 blend + f16 MFMA on a bf16 input); `input` is NCHW fp32 as in the
 * reference or, with H3D_DCN_INPUT_NHWC, channels-last [B,H,W,C] of `dtype` (torch.channels_last: no relayout); the output is
 * NCHW fp32 or, with H3D_DCN_OUTPUT_NHWC, channels-last of `dtype`.  offset [B,18,H,W] and mask [B,9,H,W] stay the reference's
 * NCHW fp32 operands.  3x3, stride 1, pad 1, dilation 1, deformable_group 1 (model.py:355), C % 16 == 0.
 * workspace: h3d_dcn_v2_packed_workspace_bytes(B, C, H, W, flags) bytes. */
#define H3D_DCN_INPUT_NHWC 1
#define H3D_DCN_OUTPUT_NHWC 2
#define H3D_DCN_F32_MFMA 4      /* H3D_F32 packs: contract on the fp32 matrix instruction (exact fmaf chains) instead of the default since round 5,
                                   three fp16 MFMAs on split operands per fp32 product (2^-22 relative per product, fp32 accumulation: ~2x the
                                   rate).  The same choice for every fp32 fast-path entry point: environment H3D_DCN_OP_F32=1 */
size_t h3d_dcn_v2_packed_weight_bytes(int Cout, int C, int dtype);
int h3d_dcn_v2_pack_weights(const float *weight, const float *bias, int Cout, int C, int dtype, void *packed, void *stream);
/* h3d_dcn_v2_pack_weights for a pack that is KEPT across calls: validated on the device, without a host synchronisation.  Every
 * call hashes the bytes of weight and bias on `stream`; the pack kernel runs only when the hash differs from the one `packed` was
 * built from (state[0]), so an in-place parameter edit that no host-side version counter sees (`weight.data.zero_()`,
 * DCNv2/test.py:21; dcn_v2.py:80-81) is picked up by the next call, and an unchanged layer costs three tiny launches.
 * state: 16 bytes of device memory, zeroed by the caller when `packed` is allocated, owned by this function afterwards. */
int h3d_dcn_v2_pack_weights_cached(const float *weight, const float *bias, int Cout, int C, int dtype, void *packed, void *state,
                                   void *stream);
/* The four parameters of the stand-alone `DCN` module (dcn_v2.py:97-116; conv_offset_mask has 27 output channels) in the fp32 layout
 * of H3D_OP_DCN_FUSED: wp [rows = Cout padded to 128][9][C], wo [128][9][C] (rows permuted as the op expects), bias_out
 * [rows | 32 | 64]: the biases, then (round 5) 64 floats of which the first two words are the bit patterns of max |weight| and
 * max |off_weight| -- what an H3D_OP_DCN_FUSED of dtype H3D_F16X3 with reserved = 0x100000 derives its power-of-two filter scales
 * from (fp32 packs, split while they are staged); validated on the device like h3d_dcn_v2_pack_weights_cached (state: 16 zeroed bytes). */
int h3d_dcn_fused_pack_f32_cached(const float *weight, const float *bias, const float *off_weight, const float *off_bias, int Cout, int C,
                                  float *wp, float *wo, float *bias_out, void *state, void *stream);
size_t h3d_dcn_v2_packed_workspace_bytes(int B, int C, int H, int W, int flags);
int h3d_dcn_v2_forward_packed(const void *input, const void *packed, const float *offset, const float *mask, void *output,
                              int B, int C, int H, int W, int Cout, int dtype, int flags,
                              void *workspace, size_t workspace_bytes, void *stream);

/* =====================================================================================
 * 2. Network ops (DLA-34 + DLAUp/IDAUp + heads, model.py:32-61,148-222,286-292,346-415,475-489)
 *    on the internal layout: activations NHWC (channels-last) of element type f32 or bf16,
 *    addressed as (pointer, channel stride) so a tensor can live inside a wider concat buffer
 *    (Root's torch.cat, model.py:160, costs nothing).  Weights are pre-packed by the host:
 *      conv   : [rows = Cout padded to 128][kh*kw][Cin] elements, BatchNorm folded in,
 *               bias fp32[rows]
 *    One descriptor per launch; h3d_run_ops walks an array of them (the forward "plan").
 * ===================================================================================== */
enum {
    H3D_OP_STEM = 1,    /* base_layer 7x7 3->C0 conv+BN+ReLU from NCHW fp32 images (model.py:231-235): stride 1, C0 = 16;
                           bf16 plans also stride 2 with C0 a multiple of 16 (the stems of ResNet-101-DCN / Hourglass-104):
                           w = bf16 [C0][7][32], k = dx*4 + c, zero for dx = 7 and c = 3 */
    H3D_OP_CONV = 2,    /* kxk (k=1|3, stride 1|2, pad k/2) conv + bias [+residual] [+ReLU]            */
    H3D_OP_DCN = 3,     /* modulated deformable 3x3 s1 p1 d1 dg1 conv + bias [+ReLU] (model.py:346-362);
                           weights [rows][9][Cin] are fp16 when dtype = bf16 (csrc/dcn2.hip), fp32 otherwise */
    H3D_OP_MAXPOOL = 4, /* 2x2 stride-2 max pool (Tree.downsample, model.py:200-201)                   */
    H3D_OP_UPADD = 5,   /* depthwise ConvTranspose2d(k=2f,s=f,p=f/2) + skip add (IDAUp, model.py:375-390) */
    H3D_OP_COPY = 6,    /* strided NHWC copy (y[i] = x[i].clone(), model.py:480-482)                   */
    H3D_OP_DCN_FUSED = 9, /* DeformConv with conv_offset_mask fused in (csrc/dcn3.hip): in2 = offset/mask filters
                             [32 permuted rows][9][Cin] (same element type as w), bias = [wrows main | 32 offset] */
    H3D_OP_CONV_STREAM = 10, /* 3x3 p1 conv, stride 1 or 2 (bf16), fed by LDS-DMA (csrc/conv2.hip): w = stage-major filter image
                                [Cin/16][wrows/32][32 rows][19 slots of 8 elements]: slot 2*tap+h = input channels
                                16*stage + 8h..8h+7 of tap `tap`, slot 18 zero; in2 = optional residual    */
    H3D_OP_DCN_FUSED_F16 = 11, /* H3D_OP_DCN_FUSED for a 64-channel fp16 input, Cout <= 64 (csrc/dcn4.hip): w and in2 are
                                  stage-major fp16 filter images [4][wrows/32 | 1][32][19][8] (layout of H3D_OP_CONV_STREAM) */
    H3D_OP_DCN_FUSED_STREAM = 12, /* H3D_OP_DCN_FUSED (bf16 input) with w / in2 as stage-major fp16 filter images of
                                     CK = h3d_dcn_fused_ck(Cin, Cout) channels per stage: [Cin/CK][wrows/32 | 1][32 rows]
                                     [9*CK/8 + 1 slots][8]: slot tap*CK/8 + j = channels CK*stage + 8j.. of tap `tap` */
    H3D_OP_STEM3 = 13,  /* base_layer + level0 + level1 fused (bf16, csrc/stem3.hip): in = NCHW fp32 images, out = level1 map
                           [B,Ho,Wo,32]; w = bf16 [16][7][32] stem (k = dx*4+c) | [5][16][32] level0 (k = (tap&1)*16+c of tap pair)
                           | [32][9][16] level1; bias = fp32 [16 | 16 | 32]; Cin = 3, Cout = 32.
                           in2 != NULL (round 4): ALSO level2's residual branch, project(max_pool2x2(level1)) (Tree.downsample + Tree.project,
                           model.py:200-207): in2 = that OUTPUT map [B,Ho/2,Wo/2,in2_cs] (64 channels, no ReLU); w continues with the 1x1
                           filters [64][32], bias with their 64 values; Ho, Wo even                                            */
    H3D_OP_DCN_V1 = 8,  /* first-generation DCN kernel (global gather, bf16 weights): kept as an A/B reference */
    H3D_OP_UPDCN_F16 = 14, /* IDAUp's node(up(x) + skip) in one launch (model.py:384-390): depthwise ConvTranspose2d (k = 2f,
                           stride f in {2, 4}) + skip add evaluated while the DeformConv's input tile is staged, then
                           DCN_FUSED_F16's kernel.  in = x at the LOW resolution H x W, Ho x Wo = f * (H x W), stride = f,
                           in2 = HOST pointer to h3d_updcn_desc; bf16 plans, 64 channels                               */
    /* ops of the other backbones (BASELINE configs 4, 5: Hourglass-104, ResNet-101-DCN; csrc/extra.hip) */
    H3D_OP_IM2COL = 15,      /* 7x7 stride-2 pad-3 stem conv as im2col: in = NCHW fp32 images [B,Cin,H,W] -> out [B,Ho,Wo,Cout] patches,
                                channel k = c*ks*ks + ky*ks + kx, zero from Cin*ks*ks up to Cout (a multiple of 16): feeds a 1x1 H3D_OP_CONV */
    H3D_OP_MAXPOOL3 = 16,    /* nn.MaxPool2d(3, stride 2, padding 1): Ho = (H-1)/2+1                                        */
    H3D_OP_DEPTH2SPACE = 17, /* [B,H,W,4*Cout] -> [B,2H,2W,Cout]: channel group 2*py+px -> output pixel (2y+py, 2x+px)        */
    H3D_OP_HEADS = 7    /* all output heads fused: Conv3x3(64->head_conv)+ReLU+Conv1x1(->C) per head
                           (model.py:451-460, 485-489); in2 = HOST pointer to h3d_heads_desc          */
};

/* H3D_OP_HEADS: op.in = y (NHWC, 64 ch), op.w = packed 3x3 weights of all heads
 * [nheads*head_conv][9][64], op.bias = fp32 [nheads*head_conv], op.Cout = head_conv,
 * op.in2 = host pointer to this descriptor.  w2: [96 rows][head_conv] elements with K in MFMA
 * accumulator-row order (h3d_amd/engine.py: pack_head_1x1), b2: fp32 [96], out: NCHW fp32. */
#define H3D_HEADS_MAX 16
typedef struct h3d_heads_desc {
    int32_t nheads;
    int32_t wexp;       /* H3D_F16X3: op.w holds 2^wexp times the 3x3 filters AND op.bias 2^wexp times their biases (h3d_op.wexp) */
    struct {
        const void *w2;
        const float *b2;
        float *out;
        int32_t C;
        int32_t wexp2;  /* H3D_F16X3: w2 holds 2^wexp2 times the 1x1 filters (b2 is unscaled) */
    } head[H3D_HEADS_MAX];
} h3d_heads_desc;
/* H3D_OP_UPDCN_F16: the operands that do not fit h3d_op */
typedef struct h3d_updcn_desc {
    const void *skip;    /* bf16 NHWC [B][Ho][Wo][skip_cs]: layers[i-1]                         */
    const float *w_up;   /* fp32 [k*k][64] tap-major transposed-convolution weights (as UPADD)  */
    const void *w_off;   /* offset/mask filter image (as DCN_FUSED_F16's in2)                   */
    int32_t skip_cs;
    int32_t reserved;
} h3d_updcn_desc;
enum { H3D_OUT_NHWC = 0, H3D_OUT_NCHW_F32 = 1, H3D_OUT_NHWC_F32 = 2,
       H3D_OUT_NHWC_F16 = 3 /* H3D_OP_UPADD in a bf16 plan only: fp16 output, the input of H3D_OP_DCN_FUSED_F16 */ };

typedef struct h3d_op {
    int32_t kind;       /* H3D_OP_*                                                        */
    int32_t dtype;      /* H3D_F32 | H3D_BF16: activation + weight element type           */
    const void *in;     /* input activations (STEM: NCHW fp32 images)                      */
    const void *in2;    /* CONV: residual (or NULL); UPADD: skip tensor; DCN: offset/mask  */
    const void *w;      /* packed weights (UPADD: fp32 [C][k*k])                           */
    const float *bias;  /* fp32 [rows] (NULL for POOL/UPADD/COPY)                          */
    void *out;
    int32_t B, H, W;    /* input batch / height / width                                    */
    int32_t Cin, in_cs; /* input channels, input channel stride (elements per pixel)       */
    int32_t in2_cs;     /* channel stride of in2 (DCN: floats per pixel of offset/mask, >=27) */
    int32_t Ho, Wo;     /* output height / width                                           */
    int32_t Cout, out_cs;
    int32_t ksize;      /* CONV: 1|3; STEM: 7; UPADD: 2f                                   */
    int32_t stride;     /* CONV: 1|2; UPADD: f                                             */
    int32_t relu;       /* 1: ReLU epilogue                                                */
    int32_t out_mode;   /* H3D_OUT_*                                                       */
    int32_t wrows;      /* rows of the packed weight buffer (>= Cout, multiple of 128)     */
    int32_t reserved;   /* 0 in production.  Profiling only: a per-kind tuning override (CONV_STREAM: MT << 8 | WAVES;
                           UPADD: 1 = tap table from global memory, 2 = from LDS) and, in `make ABLATE=1` builds,
                           ablation switches in the high bits (tools/ab_*.py)                              */
    int32_t wexp;       /* H3D_F16X3 plans (ABI 2): the packed filters hold 2^wexp times the layer's filters -- chosen by the packer so
                           that max |w| lands in [2^13, 2^14) and the lo terms of all but vanishing filters are NORMAL fp16 numbers
                           (an unscaled 0.05 has a subnormal lo term: 3e-8 absolute, 2^-20.7 relative, five times the 2^-23 of the
                           split itself) -- and the kernel multiplies its accumulators by 2^-wexp (exact) before the bias.  0 elsewhere */
    int32_t wexp2;      /* ... the same for the offset / mask filters of H3D_OP_DCN_FUSED (in2)                */
} h3d_op;

/* channels per filter stage H3D_OP_DCN_FUSED_STREAM expects for a layer (16) */
int h3d_dcn_fused_ck(int Cin, int Cout);

/* How far a fused DeformConv's samples reach, as the kernel itself sees it: per_tile[b * tiles_y * tiles_x + ty * tiles_x + tx]
 * (tiles of 16x16 output pixels, tiles_x = ceil(W/16)) = the number of (pixel, tap) samples of that tile that lie inside the image
 * but have a bilinear corner outside the tile's LDS apron, for the tile variant `op->reserved` selects (margin 2, or the wide margin
 * with 0x8000).  Phase A + geometry of the production kernel run, nothing else; op->out is not written.  A deterministic function of
 * the layer's input: DLAEngine.calibrate_dcn_margins derives the per-layer variant from these counts (the reference operator,
 * dcn_v2_cuda.cu:43-173, has no data-dependent dispatch at all, so whatever replaces it must not depend on a stopwatch). */
int h3d_dcn_far_samples(const h3d_op *op, int32_t *per_tile, void *stream);

/* Launch ops[0..n) in order on `stream`.  Returns H3D_OK or the first error (index in the
 * message). */
int h3d_run_ops(const h3d_op *ops, int n, void *stream);

/* Same, with a HIP event pair around every op: ms[i] = device time of ops[i] (bench.py's
 * live roofline measurement; events are recorded on `stream`). Synchronises the stream. */
int h3d_run_ops_timed(const h3d_op *ops, int n, void *stream, float *ms);
/* Name of the kernel instantiation `op` dispatches to, as rocprofv3 prints it
 * (e.g. "conv_kernel<unsigned short, 3, 1, 2, 32, 16>"); nothing is launched. */
int h3d_op_kernel_name(const h3d_op *op, char *buf, int buflen);

/* Layout helpers (host interface keeps the reference's NCHW fp32 tensors at the boundary). */
int h3d_nchw_f32_to_nhwc(const float *src, void *dst, int dtype, int B, int C, int H, int W,
                         int dst_cs, void *stream);
int h3d_nhwc_to_nchw_f32(const void *src, int dtype, float *dst, int B, int C, int H, int W,
                         int src_cs, void *stream);

/* =====================================================================================
 * 3. Heat-map decode (decode.py, utils.py).  All tensors contiguous NCHW fp32 as the
 *    reference's heads (model.py:485-489).  Index outputs are int64 like torch.topk's.
 *    Tie rule: equal scores -> lowest flat index first (torch leaves it unspecified).
 * ===================================================================================== */

/* _sigmoid (utils.py:8-10) optionally, then _nms (decode.py:6-13) and the per-channel top-K of
 * _topk_channel / stage 1 of _topk (decode.py:15-24, 26-33) in one pass per (b, c) map.
 *   heat [B,C,H,W]; with H3D_NMS_SIGMOID heat holds logits and scores are
 *   clamp(sigmoid(x), 1e-4, 1-1e-4).  Requires K <= min(H*W, 1024), H*W <= 36864 (larger maps: h3d_nms_topk_large).
 *   out: scores [B,C,K] f32 (descending), inds [B,C,K] i64 (flat y*W+x), ys/xs [B,C,K] f32. */
#define H3D_NMS_SIGMOID 1 /* heat holds logits: apply _sigmoid first            */
#define H3D_NMS_SKIP 2    /* heat is already NMS-ed (plain _topk/_topk_channel) */
int h3d_nms_topk(const float *heat, int B, int C, int H, int W, int K, int flags,
                 float *scores, int64_t *inds, float *ys, float *xs, void *stream);
/* The same for TWO heat-map tensors [B,Ca,H,W] and [B,Cb,H,W] in one launch (the detector's `hm` and `hm_hp`,
 * decode.py:85 and :118: the 1-class map alone occupies B workgroups).  Outputs as h3d_nms_topk, per tensor. */
int h3d_nms_topk2(const float *heat_a, int Ca, float *scores_a, int64_t *inds_a, float *ys_a, float *xs_a,
                  const float *heat_b, int Cb, float *scores_b, int64_t *inds_b, float *ys_b, float *xs_b,
                  int B, int H, int W, int K, int flags, void *stream);

/* h3d_nms_topk for maps of any size (H * W > 36864: e.g. the 320 x 184 output of a --keep_res 1280 x 736 frame,
 * datasets/coco.py:160-163): the map is cut into bands of rows (each with one halo row on either side for the 3x3 max), the bands'
 * top K are merged; same results and the same tie rule as h3d_nms_topk.  workspace: h3d_nms_topk_large_workspace_bytes(...) bytes
 * (0 = the shape is not supported: a band of rows + two halo rows must fit 36864 pixels, bands x K <= 8192). */
size_t h3d_nms_topk_large_workspace_bytes(int B, int C, int H, int W, int K);
int h3d_nms_topk_large(const float *heat, int B, int C, int H, int W, int K, int flags,
                       float *scores, int64_t *inds, float *ys, float *xs,
                       void *workspace, size_t workspace_bytes, void *stream);

/* stand-alone _nms (decode.py:6-13): out = heat * (maxpool3x3(heat) == heat), [B,C,H,W] */
int h3d_nms(const float *heat, int B, int C, int H, int W, float *out, void *stream);
/* stand-alone _sigmoid (utils.py:8-10): out = clamp(sigmoid(in), 1e-4, 1-1e-4); in == out allowed */
int h3d_sigmoid_clamp(const float *in, float *out, size_t n, void *stream);

/* Stage 2 of _topk (decode.py:34-39): top-K over the C*K stage-1 candidates of each image.
 *   in : scores/inds/ys/xs [B,C,K];  out: score [B,K], ind [B,K] i64, cls [B,K] i32, y/x [B,K].
 *   Requires C*K <= 8192. */
int h3d_topk_merge(const float *scores, const int64_t *inds, const float *ys, const float *xs,
                   int B, int C, int K, float *o_score, int64_t *o_ind, int32_t *o_cls,
                   float *o_y, float *o_x, void *stream);

/* _transpose_and_gather_feat (utils.py:23-27): feat [B,C,H,W] (channels_last=0), ind [B,N]
 * -> out [B,N,C], without the NHWC transpose; _gather_feat (utils.py:12-21) itself is the
 * channels_last=1 case: feat [B,HW,C]. */
int h3d_gather_feat(const float *feat, const int64_t *ind, int B, int C, int HW, int N,
                    int channels_last, float *out, void *stream);

/* multi_pose_decode after the two top-k's (decode.py:86-161): gathers, boxes, keypoint
 * matching, assembly of dets [B,K,5+2J+1].  reg / hp_* may be NULL exactly as in the
 * reference signature (reg=None, hm_hp=None, hp_offset=None).
 *   centre top-k  : c_score/c_ind/c_cls/c_y/c_x [B,K]
 *   joint  top-k  : hp_score/hp_ind/hp_y/hp_x [B,J,K] (NULL when hm_hp is None)
 *   wh [B,2,H,W], hps [B,2J,H,W], reg [B,2,H,W]|NULL, hp_offset [B,2,H,W]|NULL */
int h3d_multi_pose_assemble(const float *c_score, const int64_t *c_ind, const int32_t *c_cls,
                            const float *c_y, const float *c_x,
                            const float *hp_score, const int64_t *hp_ind, const float *hp_y,
                            const float *hp_x,
                            const float *wh, const float *hps, const float *reg,
                            const float *hp_offset,
                            int B, int J, int H, int W, int K, float *dets, void *stream);

/* ctdet_decode after _topk (decode.py:52-75): dets [B,K,6]. wh [B,2 or 2C,H,W]. */
int h3d_ctdet_assemble(const float *c_score, const int64_t *c_ind, const int32_t *c_cls,
                       const float *c_y, const float *c_x, const float *wh, const float *reg,
                       int B, int C, int H, int W, int K, int cat_spec_wh, float *dets,
                       void *stream);

/* Pre-process, val branch of datasets/coco_hp.py:151-212: images [B,h,w,3] uint8 (BGR, rows of
 * row_bytes) -> out [B,3,res_h,res_w] fp32 = ((warpAffine(img, M, INTER_LINEAR, border 0) / 255) - mean) / std.
 * minv [B,6] double = the INVERSE (dst -> src) 2x3 affine of get_affine_transform(c, s, 0, [res,res])
 * (utils/image.py:27-62); OpenCV's fixed-point bilinear scheme, see csrc/preprocess.hip. */
int h3d_preprocess(const uint8_t *images, int B, int h, int w, int row_bytes, const double *minv,
                   const float *mean, const float *stdv, int res_h, int res_w, float *out, void *stream);

/* multi_pose_post_process (utils/post_process.py:41-52 + utils/image.py:19-68, inv affine with
 * rot = 0): dets [B,K,40] (output-res px) + c [B,2], s [B] -> out [B,K,39] image px.
 * J = 0 is the transform of ctdet_post_process (utils/post_process.py:24-38): dets [B,K,6] -> out [B,K,5] (box, score). */
int h3d_multi_pose_post_process(const float *dets, const float *c, const float *s, int B, int K,
                                int J, int out_h, int out_w, float *out, void *stream);

/* =====================================================================================
 * 4. SMPL pose/shape -> LBS mesh (north_star; no reference code: published formulation).
 *    Model tensors are packed by the host (h3d_amd/smpl.py: SMPLModel.device_pack):
 *      v_template [3][Vpad], shapedirsT [10][3][Vpad], posedirsT [207][3][Vpad], j_template [24*3],
 *      j_shapedirs [24*3][10], parents i32[24], lbs_idx i32[V][nnz], lbs_w f32[V][nnz]
 * ===================================================================================== */
/* per person: Rodrigues (24), pose feature (207), joints, kinematic chain.
 *   betas [P,10], thetas [P,72] -> pose_feat [P,207], A [P,24,12] (3x4 skinning transforms),
 *   joints [P,24,3] (posed joint positions); if coefT != NULL also the k-major coefficient matrix
 *   coefT [217][Ppad] = [beta | pose_feat]^T consumed by h3d_smpl_verts2 (Ppad multiple of 128). */
int h3d_smpl_pose(const float *betas, const float *thetas, const float *j_template,
                  const float *j_shapedirs, const int32_t *parents, int P,
                  float *pose_feat, float *A, float *joints, float *coefT, int Ppad, void *stream);
/* h3d_smpl_pose + the two gathers in front of it + h3d_smpl_coef_pack behind it in ONE launch (the detector's tail: at a per-GPU shard
 * of 8 images every launch of the tail is a sub-wave-count grid on the critical path): person p = detection p % n of image p / n reads
 * its thetas / betas from the `pose` [B,72,HW] / `shape` [B,10,HW] head maps at pixel inds[(p / n) * K + p % n] -- what
 * _transpose_and_gather_feat (utils.py:23-27) would copy out -- and coefK3 [Ppad][14][3][16] is written by the lanes that hold the
 * values (zero rows for p >= B*n; Ppad multiple of 128).  betas_out [B*n,10] may be NULL.  Bit-identical to the separate launches. */
int h3d_smpl_pose_heads(const float *pose_map, const float *shape_map, const int64_t *inds, int B, int K, int n, int HW,
                        const float *j_template, const float *j_shapedirs, const int32_t *parents, float *betas_out,
                        float *pose_feat, float *A, float *joints, void *coefK3, int Ppad, void *stream);
/* generation 3: blend shapes on the bf16 matrix cores with every fp32 operand split into three bf16 terms
 * (fp32-level accuracy, csrc/smpl.hip).  coefK3 [Ppad][14][3][16] bf16 = per person and K step of 16 the h/m/l terms
 * of [beta | pose_feat | 0] (h3d_smpl_coef_pack; Ppad multiple of 128), dirsK3 [3][Vpad][14][3][16] bf16 = the same
 * split of the K-contiguous direction rows (10 shape + 207 pose, zero padded to 224; Vpad multiple of 64;
 * h3d_amd/smpl.py: _dirs_k3), A from h3d_smpl_pose, <= 4 skinning weights per vertex. */
int h3d_smpl_coef_pack(const float *betas, const float *pose_feat, int P, int Ppad, void *coefK3, void *stream);
int h3d_smpl_verts3(const void *coefK3, const float *A, const float *v_template, const void *dirsK3,
                    const int32_t *lbs_idx, const float *lbs_w, int nnz, int P, int Ppad, int V, int Vpad,
                    float *verts, void *stream);
/* The same with all six products of the three-term split (h3d_smpl_verts3 keeps hh + hm + mh: 2^-16 relative per dropped
 * product, 2e-6 abs on the blend-shape displacement): agrees with the fp32 vector kernels to 2e-6 -- what the f32
 * (parity-mode) detectors run; 1.16x the time. */
int h3d_smpl_verts3_exact(const void *coefK3, const float *A, const float *v_template, const void *dirsK3,
                          const int32_t *lbs_idx, const float *lbs_w, int nnz, int P, int Ppad, int V, int Vpad,
                          float *verts, void *stream);
/* blend shapes + LBS: verts [P,V,3].  Model tensors struct-of-arrays with row stride Vpad:
 * v_template [3][Vpad], shapedirsT [10][3][Vpad], posedirsT [207][3][Vpad]. */
int h3d_smpl_verts(const float *betas, const float *pose_feat, const float *A,
                   const float *v_template, const float *shapedirsT, const float *posedirsT,
                   const int32_t *lbs_idx, const float *lbs_w, int nnz, int P, int V, int Vpad,
                   float *verts, void *stream);
/* same result, LDS-streamed generation 2 (needs nnz <= 4, Vpad % 64 == 0, Ppad % 128 == 0). */
int h3d_smpl_verts2(const float *coefT, const float *A, const float *v_template,
                    const float *shapedirsT, const float *posedirsT, const int32_t *lbs_idx,
                    const float *lbs_w, int nnz, int P, int Ppad, int V, int Vpad, float *verts,
                    void *stream);

#ifdef __cplusplus
}
#endif
#endif /* H3D_H */
