"""Helpers for the -m gpu tests: drive single ops through the C ABI (h3d_run_ops)."""
import ctypes

import numpy as np
import torch

import h3d_amd  # noqa: F401
from h3d_amd import _lib
from h3d_amd._lib import H3dOp

TD = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16, "f16x3": torch.float32}      # storage type of a plan's activations
HD = {"f32": _lib.H3D_F32, "bf16": _lib.H3D_BF16, "f16": _lib.H3D_F16, "f16x3": _lib.H3D_F16X3}
TN = {"f32": "float", "bf16": "unsigned short", "f16": "f16_t", "f16x3": "x3_t"}      # element-type names inside the kernel symbols
DEV = "cuda:0"
WEXP = {}     # f16x3: device pointer of a packed filter bank -> its pre-scale exponent (h3d_op.wexp)
KEEP = []     # ... the banks themselves (a freed bank's address could be handed out again with another exponent)


def pack_conv(w, b, dtype, pad_cout_to=None):
    co, ci, kh, kw = w.shape
    cout = pad_cout_to or co
    rows = ((cout + 127) // 128) * 128
    wp = torch.zeros(rows, kh * kw, ci)
    wp[:co] = w.permute(0, 2, 3, 1).reshape(co, kh * kw, ci)
    bp = torch.zeros(rows)
    if b is not None:
        bp[:co] = b
    if dtype == "f16x3":          # fp32 filters times 2^wexp as (hi | lo) fp16 terms per 8 input channels (engine.x3_split / x3_exp)
        from h3d_amd import engine
        e = engine.x3_exp(wp)
        t = engine.x3_split(wp * 2.0 ** e).contiguous().to(DEV)
        WEXP[t.data_ptr()] = e
        KEEP.append(t)
        return t, bp.to(DEV), cout, rows
    return wp.to(TD[dtype]).contiguous().to(DEV), bp.to(DEV), cout, rows


def nhwc(x, dtype, cs=None, coff=0):
    """NCHW fp32 cpu -> (buffer [B,H,W,cs] on device, element pointer of channel `coff`)."""
    B, C, H, W = x.shape
    cs = cs or C
    buf = torch.zeros(B, H, W, cs, dtype=TD[dtype], device=DEV)
    buf[..., coff:coff + C] = x.permute(0, 2, 3, 1).to(TD[dtype]).to(DEV)
    return buf, buf.data_ptr() + coff * buf.element_size()


def from_nhwc(buf, C, coff=0):
    return buf[..., coff:coff + C].float().permute(0, 3, 1, 2).contiguous().cpu()


def run(op_or_ops):
    ops = op_or_ops if isinstance(op_or_ops, (list, tuple)) else [op_or_ops]
    arr = (H3dOp * len(ops))(*ops)
    rc = _lib.lib().h3d_run_ops(arr, len(ops), _lib.stream_ptr())
    _lib.check(rc, "h3d_run_ops")
    torch.cuda.synchronize()


def mk(kind, dtype, **kw):
    op = H3dOp()
    op.kind, op.dtype = kind, HD[dtype]
    for k, v in kw.items():
        setattr(op, k, v)
    return op


def conv(x, w, b, dtype, stride=1, relu=False, res=None, out_mode=_lib.OUT_NHWC, in_pad=0, out_pad=0,
         pad_cout_to=None, name_only=False, reserved=0):
    """x NCHW fp32 cpu; returns NCHW fp32 cpu result of the HIP conv (input/output living inside wider
    channel-strided buffers when in_pad/out_pad > 0)."""
    B, Ci, H, W = x.shape
    k = w.shape[2]
    wp, bp, cout, rows = pack_conv(w, b, dtype, pad_cout_to)
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    Wo = (W + 2 * (k // 2) - k) // stride + 1
    xin, xptr = nhwc(x, dtype, Ci + in_pad, in_pad // 2 // 8 * 8)
    rptr, rcs = None, 0
    if res is not None:
        rbuf, rptr = nhwc(res, dtype)
        rcs = res.shape[1]
    if out_mode == _lib.OUT_NHWC:
        ocs = cout + out_pad
        coff = out_pad // 2 // 8 * 8
        out = torch.full((B, Ho, Wo, ocs), 7.0, dtype=TD[dtype], device=DEV)
        optr = out.data_ptr() + coff * out.element_size()
    elif out_mode == _lib.OUT_NHWC_F32:
        ocs, coff = cout, 0
        out = torch.full((B, Ho, Wo, ocs), 7.0, dtype=torch.float32, device=DEV)
        optr = out.data_ptr()
    else:
        ocs, coff = cout, 0
        out = torch.full((B, cout, Ho, Wo), 7.0, dtype=torch.float32, device=DEV)
        optr = out.data_ptr()
    op = mk(_lib.OP_CONV, dtype, in_=xptr, in2=rptr, w=wp.data_ptr(), bias=bp.data_ptr(), out=optr, B=B, H=H, W=W,
            Cin=Ci, in_cs=Ci + in_pad, in2_cs=rcs, Ho=Ho, Wo=Wo, Cout=cout, out_cs=ocs, ksize=k, stride=stride,
            relu=int(relu), out_mode=out_mode, wrows=rows, reserved=reserved, wexp=WEXP.get(wp.data_ptr(), 0))
    if name_only:
        return kernel_name(op)
    run(op)
    if out_mode == _lib.OUT_NCHW_F32:
        return out.cpu(), None
    res_t = from_nhwc(out, cout, coff)
    untouched = None
    if out_pad:
        mask = torch.ones(ocs, dtype=torch.bool)
        mask[coff:coff + cout] = False
        untouched = bool((out[..., mask.to(DEV)].float() == 7.0).all().item())
    return res_t, untouched


def bf16_round(t):
    return t.to(torch.bfloat16).float()


def lowp_round(t, dtype):
    """Round to the plan's storage type (f32 / f16x3: unchanged)."""
    return t if dtype in ("f32", "f16x3") else t.to(TD[dtype]).float()


def rnd(key, shape, lo=-1.0, hi=1.0, seed=0):
    from h3d_amd import synth
    return torch.from_numpy(synth.uniform(key, shape, lo, hi, seed))


# ------------------------------------------------------------------------------------------------
# Per-op drivers for the LDS-DMA kernels (csrc/conv2.hip, dcn3.hip, dcn4.hip).  Weights go through the
# PRODUCT's packer (engine.PackedWeights), so the stage-major filter images are tested with it.
def kernel_name(op):
    """Kernel instantiation `op` dispatches to (dry run, nothing is launched)."""
    buf = ctypes.create_string_buffer(200)
    _lib.check(_lib.lib().h3d_op_kernel_name(ctypes.byref(op), buf, 200), "op_kernel_name")
    return buf.value.decode()


def fake_pw(sd, dtype):
    """engine.PackedWeights over an ad-hoc {key: tensor} table (no architecture check)."""
    from h3d_amd import engine
    return engine.PackedWeights.from_tensors(sd, dtype, DEV)


class Built:
    """An op + the tensors it points into + how to read its output back as NCHW fp32 (cpu)."""

    def __init__(self, op, keep, read):
        self.op, self.keep, self.read = op, keep, read

    @property
    def name(self):
        return kernel_name(self.op)

    def run(self):
        run(self.op)
        return self.read()


def conv_stream_op(x, w, b, stride=1, relu=True, res=None, reserved=0, in_pad=0, out_pad=0, dtype="bf16"):
    """H3D_OP_CONV_STREAM (bf16 / fp16 plans) for NCHW fp32 cpu tensors; in_pad/out_pad: live inside wider buffers."""
    B, Ci, H, W = x.shape
    pw = fake_pw({"w": w, "b": b}, dtype)
    wimg, bp, cout, cin, rows = pw.conv_stream("w", "b")
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    xin, xptr = nhwc(x, dtype, Ci + in_pad, in_pad // 2 // 8 * 8)
    keep = [wimg, bp, xin]
    rptr, rcs = None, 0
    if res is not None:
        rbuf, rptr = nhwc(res, dtype)
        rcs = res.shape[1]
        keep.append(rbuf)
    ocs, coff = cout + out_pad, out_pad // 2 // 8 * 8
    out = torch.full((B, Ho, Wo, ocs), 7.0, dtype=TD[dtype], device=DEV)
    op = mk(_lib.OP_CONV_STREAM, dtype, in_=xptr, in2=rptr, w=wimg.data_ptr(), bias=bp.data_ptr(),
            out=out.data_ptr() + coff * 2, B=B, H=H, W=W, Cin=Ci, in_cs=Ci + in_pad, in2_cs=rcs, Ho=Ho, Wo=Wo, Cout=cout,
            out_cs=ocs, ksize=3, stride=stride, relu=int(relu), out_mode=_lib.OUT_NHWC, wrows=rows, reserved=reserved)

    def read():
        if out_pad:
            m = torch.ones(ocs, dtype=torch.bool)
            m[coff:coff + cout] = False
            assert bool((out[..., m.to(DEV)].float() == 7.0).all().item()), "neighbouring channels overwritten"
        return from_nhwc(out, cout, coff)
    return Built(op, keep, read)


def _dcn_sd(w, b, wo, bo):
    co = w.shape[0]
    return {"p.conv.weight": w, "p.conv.bias": b, "p.conv.conv_offset_mask.weight": wo, "p.conv.conv_offset_mask.bias": bo,
            "p.actf.0.weight": torch.ones(co), "p.actf.0.bias": torch.zeros(co), "p.actf.0.running_mean": torch.zeros(co),
            "p.actf.0.running_var": torch.full((co,), 1.0 - 1e-5)}   # + BN_EPS = 1: identity


def dcn_fused_op(kind, x, w, b, wo, bo, dtype="bf16", reserved=0, skip=None, w_up=None):
    """DeformConv with conv_offset_mask fused in, through one of
         'fused'   H3D_OP_DCN_FUSED        (csrc/dcn3.hip, register-staged filters; f32 or bf16)
         'stream'  H3D_OP_DCN_FUSED_STREAM (csrc/dcn3.hip WDMA; bf16)
         'stream16' the same op of a bf16 plan with an fp16 INPUT (reserved | 0x40000: csrc/dcn3.hip F16IN; x is stored as fp16)
         'f16'     H3D_OP_DCN_FUSED_F16    (csrc/dcn4.hip; x is stored as fp16)
         'updcn'   H3D_OP_UPDCN_F16        (csrc/dcn4.hip UP = 1; x is the LOW-resolution bf16 map, skip/w_up given)
    x/skip NCHW fp32 cpu, w [Co,Ci,3,3], wo [27,Ci,3,3], w_up [C,1,2f,2f].  BatchNorm = identity, ReLU on."""
    B, Ci, H, W = x.shape
    pw = fake_pw(_dcn_sd(w, b, wo, bo), dtype)
    keep = []
    Ho, Wo, stride = H, W, 1
    in2 = None
    if kind == "fused":
        wp, bp, cout, cin, k, rows = pw.conv("p.conv.weight", "p.conv.bias", "p.actf.0", as_half=True)
        wop, bop = pw.offset_conv("p.conv.conv_offset_mask.weight", "p.conv.conv_offset_mask.bias", rows)
        bias = torch.cat([bp.cpu(), bop]).contiguous().to(DEV)
        xin, xptr = nhwc(x, dtype)
        opk, wptr, in2 = _lib.OP_DCN_FUSED, wp.data_ptr(), wop.data_ptr()
        keep += [wp, wop, bias, xin]
    else:
        assert dtype in ("bf16", "f16", "f16x3") and (dtype == "bf16" or kind == "stream")
        ck = int(_lib.lib().h3d_dcn_fused_ck(Ci, w.shape[0])) if kind in ("stream", "stream16") else 16
        wimg, woimg, bias, cout, cin, rows = pw.dcn_stream_x3("p") if dtype == "f16x3" else pw.dcn_stream("p", ck)
        wptr = wimg.data_ptr()
        keep += [wimg, woimg, bias]
        if kind == "stream":
            xin, xptr = nhwc(x, dtype)
            opk, in2 = _lib.OP_DCN_FUSED_STREAM, woimg.data_ptr()
        elif kind == "stream16":
            xin = x.permute(0, 2, 3, 1).contiguous().to(torch.float16).to(DEV)
            xptr = xin.data_ptr()
            opk, in2 = _lib.OP_DCN_FUSED_STREAM, woimg.data_ptr()
            reserved |= 0x40000
        elif kind == "f16":
            xin = x.permute(0, 2, 3, 1).contiguous().to(torch.float16).to(DEV)
            xptr = xin.data_ptr()
            opk, in2 = _lib.OP_DCN_FUSED_F16, woimg.data_ptr()
        else:
            k2 = w_up.shape[2]
            f = k2 // 2
            Ho, Wo, stride = H * f, W * f, f
            xin, xptr = nhwc(x, "bf16")
            sbuf, sptr = nhwc(skip, "bf16")
            wup = w_up.reshape(Ci, k2 * k2).t().contiguous().float().to(DEV)
            desc = _lib.H3dUpdcnDesc()
            desc.skip, desc.w_up, desc.w_off, desc.skip_cs = sptr, wup.data_ptr(), woimg.data_ptr(), Ci
            keep += [sbuf, wup, desc]
            opk, in2 = _lib.OP_UPDCN_F16, ctypes.addressof(desc)
        keep.append(xin)
    out = torch.zeros(B, Ho, Wo, cout, dtype=TD[dtype], device=DEV)
    op = mk(opk, dtype, in_=xptr, in2=in2, w=wptr, bias=bias.data_ptr(), out=out.data_ptr(), B=B, H=H, W=W, Cin=Ci, in_cs=Ci,
            Ho=Ho, Wo=Wo, Cout=cout, out_cs=cout, ksize=3, stride=stride, relu=1, out_mode=_lib.OUT_NHWC, wrows=rows,
            reserved=reserved, wexp=pw.wexp.get(wptr, 0), wexp2=pw.wexp.get(in2, 0) if isinstance(in2, int) else 0)
    return Built(op, keep, lambda: from_nhwc(out, cout))


def dcn_fused_reference(x, w, b, wo, bo):
    """fp64-accumulated oracle of DCN.forward + ReLU (dcn_v2.py:118-128; oracle/dcn.py), x NCHW."""
    from oracle import dcn as odcn
    om = torch.nn.functional.conv2d(x.double(), wo.double(), bo.double(), 1, 1).float()
    o1, o2, mask = torch.chunk(om, 3, dim=1)
    offset = torch.cat((o1, o2), dim=1).contiguous()
    y = odcn.dcn_v2_forward(x, w, b, offset, torch.sigmoid(mask).contiguous(), 3, 3, 1, 1, 1, 1, 1, 1, 1,
                            acc_dtype=torch.float64)
    return torch.relu(y), om
