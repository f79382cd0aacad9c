"""Helpers for the -m gpu tests: drive single ops through the C ABI (h3d_run_ops)."""
import ctypes

import numpy as np
import torch

import h3d_amd  # noqa: F401
from h3d_amd import _lib
from h3d_amd._lib import H3dOp

TD = {"f32": torch.float32, "bf16": torch.bfloat16}
HD = {"f32": _lib.H3D_F32, "bf16": _lib.H3D_BF16}
DEV = "cuda:0"


def pack_conv(w, b, dtype, pad_cout_to=None):
    co, ci, kh, kw = w.shape
    cout = pad_cout_to or co
    rows = ((cout + 127) // 128) * 128
    wp = torch.zeros(rows, kh * kw, ci)
    wp[:co] = w.permute(0, 2, 3, 1).reshape(co, kh * kw, ci)
    bp = torch.zeros(rows)
    if b is not None:
        bp[:co] = b
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
         pad_cout_to=None):
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
            relu=int(relu), out_mode=out_mode, wrows=rows)
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


def rnd(key, shape, lo=-1.0, hi=1.0, seed=0):
    from h3d_amd import synth
    return torch.from_numpy(synth.uniform(key, shape, lo, hi, seed))
