"""Host mirror of the reference's models/DCNv2/dcn_v2.py (forward path): `dcn_v2_forward`
(the `_ext` entry point, DCNv2/src/dcn_v2.h:9-39), `dcn_v2_conv`, `DCNv2`, `DCN` with the same
constructor arguments and state_dict keys (weight, bias, conv_offset_mask.{weight,bias}).

Backward (dcn_v2_backward) and PS-ROI pooling are out of scope:
inference only -- an INPUT that requires grad (with autograd enabled) raises; parameters may require grad (the
nn.Parameter default), results are computed without a graph and returned detached."""
import math

import os

import torch
from torch import nn
from torch.nn.modules.utils import _pair

from . import _lib


_PACKED = {}          # (weight ptr, bias ptr, shape, dtype, device) -> (packed filter image, 16-byte device state of its validation)
_PACKED_MAX = 64


# fp32 tensors in the model's configuration: False (default since round 5) = three fp16 MFMAs on split operands per fp32 product
# (H3D_F16X3: 2^-22 relative per product, fp32 accumulation, ~2x the rate), True = exact fmaf chains on the fp32 matrix instruction
# (H3D_DCN_F32_MFMA; the environment variable H3D_DCN_OP_F32=1 selects the same for every entry point of the library)
OP_F32_MFMA = os.environ.get("H3D_DCN_OP_F32", "") not in ("", "0")

def _packed_weights(weight, bias, dtype):
    """The operator's filters in the kernels' layout, KEPT across calls and validated on the device at every call
    (`h3d_dcn_v2_pack_weights_cached`): the bytes of `weight` and `bias` are hashed on the current stream and the pack kernel runs
    only when the hash differs from the one the image was built from -- no host synchronisation.  Neither `tensor._version` (an
    edit through `.data`, as the reference does in dcn_v2.py:80-81 and DCNv2/test.py:21, does not bump it) nor the address (it can
    be reused) is taken as proof that the filters are unchanged; the host key only finds the buffer.  All of it is stream ordered
    on the stream that OWNS the entry (the stream of the call that created it): the 16-byte validation state is one accumulator, so
    two streams hashing the same layer at once would add into it together (spurious re-packs into a buffer another stream is still
    reading).  A call on any other stream therefore does not touch the cached image: it packs into a buffer of its own
    (`h3d_dcn_v2_pack_weights`, the reference contract's per-call work) -- correct on every stream, cached on one.  The cache holds
    no reference to the parameters.  Call with the tensor's device current (`dcn_v2_forward` does)."""
    key = (weight.data_ptr(), bias.data_ptr(), tuple(weight.shape), dtype, str(weight.device))
    Cout, C = weight.shape[0], weight.shape[1]
    L = _lib.lib()
    stream = torch.cuda.current_stream(weight.device).cuda_stream
    ent = _PACKED.pop(key, None)
    if ent is not None and ent[2] != stream:
        _PACKED[key] = ent
        n = int(L.h3d_dcn_v2_packed_weight_bytes(Cout, C, dtype))
        own = torch.empty(n, dtype=torch.uint8, device=weight.device)
        _lib.check(L.h3d_dcn_v2_pack_weights(_lib.ptr(weight), _lib.ptr(bias), Cout, C, dtype, _lib.ptr(own), _lib.stream_ptr()),
                   "dcn_v2_pack_weights")
        return own
    if ent is None:
        n = int(L.h3d_dcn_v2_packed_weight_bytes(Cout, C, dtype))
        ent = (torch.empty(n, dtype=torch.uint8, device=weight.device), torch.zeros(2, dtype=torch.int64, device=weight.device), stream)
        while len(_PACKED) >= _PACKED_MAX:
            _PACKED.pop(next(iter(_PACKED)))
    _PACKED[key] = ent                                     # (re-inserted: most recently used last)
    _lib.check(L.h3d_dcn_v2_pack_weights_cached(_lib.ptr(weight), _lib.ptr(bias), Cout, C, dtype, _lib.ptr(ent[0]), _lib.ptr(ent[1]),
                                                _lib.stream_ptr()), "dcn_v2_pack_weights_cached")
    return ent[0]


def dcn_v2_forward(input, weight, bias, offset, mask, kernel_h, kernel_w, stride_h, stride_w,
                   pad_h, pad_w, dilation_h, dilation_w, deformable_group):
    """Positional twin of `_ext.dcn_v2_forward` (dcn_v2.py:25-31 call site): fp32 contiguous NCHW CUDA tensors in, a new
    [B,Cout,Ho,Wo] fp32 tensor out.

    Throughput form (same function, nothing to opt into): in the model's configuration (3x3 s1 p1 d1 dg1, C % 16 == 0)
      * the packed filters are cached per (weight, bias) version -- a layer that runs every batch packs once;
      * an `input` in torch.channels_last memory format is read in place (no relayout) and the output comes back
        channels-last as well (torch's convention: the output follows the input's memory format);
      * a bfloat16 channels-last `input` runs the network's bf16 DeformConv path (fp16 filters and blend, f16 MFMA, fp32
        accumulation) and returns bfloat16; offset / mask stay fp32 NCHW as in the reference."""
    _lib.require_cuda(input, weight, bias, offset, mask)
    lowp = input.dtype == torch.bfloat16
    for t in (weight, bias, offset, mask) + (() if lowp else (input,)):
        if t.dtype != torch.float32:
            raise RuntimeError("dcn_v2_forward: expected float32 tensors (reference uses .data<float>())")
    nhwc = input.dim() == 4 and input.is_contiguous(memory_format=torch.channels_last) and not input.is_contiguous()
    if lowp and not (input.dim() == 4 and input.is_contiguous(memory_format=torch.channels_last)):
        raise RuntimeError("dcn_v2_forward: a bfloat16 input must be in torch.channels_last memory format")
    nhwc = nhwc or lowp
    if not nhwc:
        input = input.contiguous()
    weight, bias, offset, mask = [t.contiguous() for t in (weight, bias, offset, mask)]
    B, C, H, W = input.shape
    Cout, Ck, kh_, kw_ = weight.shape
    if kh_ != kernel_h or kw_ != kernel_w:
        raise RuntimeError("Input shape and kernel shape wont match: (%d x %d vs %d x %d)."
                           % (kernel_h, kernel_w, kh_, kw_))
    if C != Ck:
        raise RuntimeError("Input shape and kernel channels wont match: (%d vs %d)." % (C, Ck))
    Ho = (H + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) // stride_h + 1
    Wo = (W + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) // stride_w + 1
    if tuple(offset.shape) != (B, 2 * deformable_group * kernel_h * kernel_w, Ho, Wo):
        raise RuntimeError("offset shape %s does not match [B, 2*dg*kh*kw, Ho, Wo] = %s"
                           % (tuple(offset.shape), (B, 2 * deformable_group * kernel_h * kernel_w, Ho, Wo)))
    if tuple(mask.shape) != (B, deformable_group * kernel_h * kernel_w, Ho, Wo):
        raise RuntimeError("mask shape %s does not match [B, dg*kh*kw, Ho, Wo]" % (tuple(mask.shape),))
    if bias.numel() != Cout:
        raise RuntimeError("bias has %d elements, expected %d" % (bias.numel(), Cout))
    fast = ((kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, deformable_group)
            == (3, 3, 1, 1, 1, 1, 1, 1, 1) and C % 16 == 0 and H <= 32767 and W <= 32767)
    L = _lib.lib()
    with torch.cuda.device(input.device):
        if fast:
            # LDS-apron + MFMA kernel on packed filters (cached); the workspace holds the [B,H,W,32] offset/mask rows and, for an
            # NCHW input, its channels-last copy (the reference allocates `columns` / `ones` itself, dcn_v2_cuda.cu:90-103;
            # here torch's caching allocator does, stream-ordered)
            out_nhwc = nhwc and Cout % 4 == 0
            dtype = _lib.H3D_BF16 if lowp else _lib.H3D_F32
            if lowp and not out_nhwc:
                raise RuntimeError("dcn_v2_forward: bfloat16 needs Cout % 4 == 0")
            flags = (_lib.DCN_INPUT_NHWC if nhwc else 0) | (_lib.DCN_OUTPUT_NHWC if out_nhwc else 0) | (_lib.DCN_F32_MFMA if OP_F32_MFMA and not lowp else 0)
            packed = _packed_weights(weight, bias, dtype)
            out = torch.empty(B, Cout, Ho, Wo, dtype=input.dtype if lowp else torch.float32, device=input.device,
                              memory_format=torch.channels_last if out_nhwc else torch.contiguous_format)
            nws = int(L.h3d_dcn_v2_packed_workspace_bytes(B, C, H, W, flags))
            ws = torch.empty(nws, dtype=torch.uint8, device=input.device)
            rc = L.h3d_dcn_v2_forward_packed(_lib.ptr(input), _lib.ptr(packed), _lib.ptr(offset), _lib.ptr(mask), _lib.ptr(out),
                                             B, C, H, W, Cout, dtype, flags, _lib.ptr(ws), nws, _lib.stream_ptr())
        else:
            if lowp:
                raise RuntimeError("dcn_v2_forward: bfloat16 is implemented for the model's configuration only (3x3 s1 p1 d1 dg1)")
            if nhwc:
                input = input.contiguous()
            out = torch.empty(B, Cout, Ho, Wo, dtype=torch.float32, device=input.device)
            rc = L.h3d_dcn_v2_forward_ws(
                _lib.ptr(input), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(offset), _lib.ptr(mask), _lib.ptr(out),
                B, C, H, W, Cout, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                deformable_group, None, 0, _lib.stream_ptr())
    _lib.check(rc, "dcn_v2_forward")
    return out


def dcn_v2_conv(input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups):
    """`_DCNv2.apply` argument order (dcn_v2.py:18-33), forward only."""
    if torch.is_grad_enabled() and any(t.requires_grad for t in (input, offset, mask)):
        raise RuntimeError("h3d_amd DCNv2 is inference-only (dcn_v2_backward is out of scope): "
                           "an input tensor requires grad")
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    kh, kw = weight.shape[2:4]
    with torch.no_grad():       # (parameters require grad by default; the result carries no graph)
        return dcn_v2_forward(input, weight, bias, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, deformable_groups)


class _InferenceOnly(torch.autograd.Function):
    """Identity whose backward raises: a module in training mode returns its forward result (the reference's plain usage
    `DCN(...).cuda()(x)`, DCNv2/test.py:169-180, is forward-only), and a training loop that calls .backward() through it learns at
    once that dcn_v2_backward (dcn_v2_cuda.cu:175-336) is out of scope instead of silently getting no gradients."""

    @staticmethod
    def forward(ctx, out, *params):
        return out.view_as(out)

    @staticmethod
    def backward(ctx, grad):
        raise RuntimeError("h3d_amd DCNv2 is inference-only (dcn_v2_backward is out of scope): call .eval() or run under torch.no_grad()")


class DCNv2(nn.Module):
    """Parameters + forward(input, offset, mask) (dcn_v2.py:57-94)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.in_channels * self.kernel_size[0] * self.kernel_size[1])
        with torch.no_grad():
            self.weight.uniform_(-stdv, stdv)
            self.bias.zero_()

    def forward(self, input, offset, mask):
        k = self.deformable_groups * self.kernel_size[0] * self.kernel_size[1]
        assert 2 * k == offset.shape[1]
        assert k == mask.shape[1]
        out = dcn_v2_conv(input, offset, mask, self.weight, self.bias, self.stride, self.padding,
                          self.dilation, self.deformable_groups)
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _InferenceOnly.apply(out, *[p for p in self.parameters() if p.requires_grad])    # forward works, backward raises
        return out


class DCN(DCNv2):
    """DCNv2 + its own offset/mask conv (dcn_v2.py:97-128): zero-initialised
    `conv_offset_mask`, offset = first 2/3 of its output, mask = sigmoid(last 1/3)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, deformable_groups)
        ch = self.deformable_groups * 3 * self.kernel_size[0] * self.kernel_size[1]
        self.conv_offset_mask = nn.Conv2d(self.in_channels, ch, kernel_size=self.kernel_size, stride=self.stride,
                                          padding=self.padding, bias=True)
        with torch.no_grad():
            self.conv_offset_mask.weight.zero_()
            self.conv_offset_mask.bias.zero_()

    def _fused_ok(self, input):
        return (self.kernel_size == (3, 3) and self.stride == (1, 1) and self.padding == (1, 1) and self.dilation == (1, 1)
                and self.deformable_groups == 1 and self.in_channels % 16 == 0 and input.is_cuda
                and input.dtype == torch.float32 and input.dim() == 4)

    def _packed(self, device):
        """Weights in the layout of the fused DeformConv kernel (csrc/dcn3.hip, fp32), kept on the module and validated on the
        device at every forward (`h3d_dcn_fused_pack_f32_cached`: a hash of the four parameters' bytes decides, on the stream,
        whether the pack kernel has anything to do) -- so `dcn.weight.data.zero_()` between two forwards (DCNv2/test.py:21) is
        seen although no version counter moves.  `.to(device)` / `load_state_dict` need no hook: new bytes, new hash.
        Called with `device` current.  The kept pack belongs to the stream that created it (one validation accumulator: see
        `_packed_weights`); a forward on another stream packs into buffers of its own."""
        rows = (self.out_channels + 127) // 128 * 128
        C = self.in_channels
        stream = torch.cuda.current_stream(device).cuda_stream

        def fresh():
            return (torch.empty(rows * 9 * C, dtype=torch.float32, device=device), torch.empty(128 * 9 * C, dtype=torch.float32, device=device),
                    torch.zeros(rows + 32 + 64, dtype=torch.float32, device=device), torch.zeros(2, dtype=torch.int64, device=device), (rows, C), stream)
        ent = getattr(self, "_pack", None)
        if ent is None or ent[0].device != device or ent[4] != (rows, C):
            ent = fresh()
            object.__setattr__(self, "_pack", ent)
        elif ent[5] != stream:
            ent = fresh()               # (a zeroed state never matches: the pack kernel runs; nothing is kept)
        ps = [p.detach().contiguous() for p in (self.weight, self.bias, self.conv_offset_mask.weight, self.conv_offset_mask.bias)]
        if any(p.dtype != torch.float32 for p in ps):
            raise RuntimeError("DCN: expected float32 parameters (reference uses .data<float>())")
        _lib.check(_lib.lib().h3d_dcn_fused_pack_f32_cached(*[_lib.ptr(p) for p in ps], self.out_channels, C, _lib.ptr(ent[0]), _lib.ptr(ent[1]),
                                                            _lib.ptr(ent[2]), _lib.ptr(ent[3]), _lib.stream_ptr()), "dcn_fused_pack_f32_cached")
        return ent[0], ent[1], ent[2], rows

    def forward(self, input):
        """conv_offset_mask -> chunk/cat/sigmoid -> dcn_v2_conv (dcn_v2.py:118-128).  In the configuration the model uses
        (model.py:355) all of it is ONE launch of the fused DeformConv kernel in fp32 parity mode (offsets and mask never
        reach memory); other configurations run conv_offset_mask + chunk / cat / sigmoid as one launch of the library's general
        kernel (`h3d_dcn_offset_mask`) and then the operator: no nn.Conv2d / vendor library call on any path."""
        if torch.is_grad_enabled() and input.requires_grad:
            raise RuntimeError("h3d_amd DCNv2 is inference-only (dcn_v2_backward is out of scope): the input requires grad")
        if not input.is_cuda:
            raise RuntimeError("Not implemented on the CPU")
        # (parameters require grad by default: `dcn(x)` outside no_grad must return the forward result as it does in the reference,
        #  DCNv2/test.py:17-30, 169-180)
        with torch.no_grad():
            out = self._forward(input)
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # a fine-tuning loop must not silently get no gradients (the reference implements dcn_v2_backward, out of scope here):
            # the result carries a graph node whose backward raises
            return _InferenceOnly.apply(out, *[p for p in self.parameters() if p.requires_grad])
        return out

    def _forward(self, input):
        if self._fused_ok(input):
            from ._lib import H3dOp
            x = input.contiguous()
            B, C, H, W = x.shape
            with torch.cuda.device(x.device):       # (the pack's launches go to x.device's current stream, like the forward's)
                wp, wo, bias, rows = self._packed(x.device)
                xn = torch.empty(B, H, W, C, dtype=torch.float32, device=x.device)
                out = torch.empty(B, self.out_channels, H, W, dtype=torch.float32, device=x.device)
                L = _lib.lib()
                _lib.check(L.h3d_nchw_f32_to_nhwc(_lib.ptr(x), _lib.ptr(xn), _lib.H3D_F32, B, C, H, W, C, _lib.stream_ptr()), "DCN: to NHWC")
                op = H3dOp()
                # fp32 tensors, fp32 packs; since round 5 every product as three fp16 MFMAs on split operands (filters scaled and split while
                # they are staged: reserved 0x100000), or exact fmaf chains on the fp32 matrix instruction with OP_F32_MFMA
                op.kind, op.dtype, op.B, op.H, op.W, op.Ho, op.Wo = _lib.OP_DCN_FUSED, (_lib.H3D_F32 if OP_F32_MFMA else _lib.H3D_F16X3), B, H, W, H, W
                op.reserved = 0 if OP_F32_MFMA else 0x100000
                op.in_, op.in2, op.w, op.bias, op.out = xn.data_ptr(), wo.data_ptr(), wp.data_ptr(), bias.data_ptr(), out.data_ptr()
                op.Cin, op.in_cs, op.Cout, op.out_cs, op.ksize, op.stride, op.relu = C, C, self.out_channels, self.out_channels, 3, 1, 0
                op.out_mode, op.wrows = _lib.OUT_NCHW_F32, rows
                arr = (H3dOp * 1)(op)
                _lib.check(L.h3d_run_ops(arr, 1, _lib.stream_ptr()), "DCN.forward")
            return out
        # any other configuration (stride / dilation / deformable_groups / kernel size / channel count): conv_offset_mask -> chunk /
        # cat / sigmoid (dcn_v2.py:119-124) in ONE launch of this library's general kernel (`h3d_dcn_offset_mask`: no nn.Conv2d, i.e.
        # no vendor convolution library anywhere in the module), then the operator
        if input.dtype != torch.float32 or input.dim() != 4:
            raise RuntimeError("DCN: expected a float32 [B,C,H,W] input (reference uses .data<float>())")
        x = input.contiguous()
        B, C, H, W = x.shape
        kh, kw = self.kernel_size
        k = self.deformable_groups * kh * kw
        ow, ob = self.conv_offset_mask.weight.detach().contiguous(), self.conv_offset_mask.bias.detach().contiguous()
        if tuple(ow.shape) != (3 * k, C, kh, kw) or ow.dtype != torch.float32 or ob.dtype != torch.float32:
            raise RuntimeError("DCN: conv_offset_mask.weight %s does not match [3*dg*kh*kw, C, kh, kw] = %s (float32)"
                               % (tuple(ow.shape), (3 * k, C, kh, kw)))
        Ho = (H + 2 * self.padding[0] - kh) // self.stride[0] + 1
        Wo = (W + 2 * self.padding[1] - kw) // self.stride[1] + 1
        with torch.cuda.device(x.device):
            offset = torch.empty(B, 2 * k, Ho, Wo, dtype=torch.float32, device=x.device)
            mask = torch.empty(B, k, Ho, Wo, dtype=torch.float32, device=x.device)
            _lib.check(_lib.lib().h3d_dcn_offset_mask(_lib.ptr(x), _lib.ptr(ow), _lib.ptr(ob), _lib.ptr(offset), _lib.ptr(mask), B, C, H, W,
                                                      kh, kw, self.stride[0], self.stride[1], self.padding[0], self.padding[1],
                                                      self.deformable_groups, _lib.stream_ptr()), "DCN: conv_offset_mask")
        return dcn_v2_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding,
                           self.dilation, self.deformable_groups)
