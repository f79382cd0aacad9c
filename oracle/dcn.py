"""TEST INFRASTRUCTURE ONLY -- CPU restatement of DCNv2 forward (torch-CPU, vectorised).

Same algorithm as oracle/dcn_ref.c (plain C); this version is vectorised so the whole
DLA-34+DCN graph finishes in seconds and can serve as bench.py's `cpu_baseline` (kind "port").

Reference lines followed:
  DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:25-54   bilinear, zero corners outside the image
  DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:125-195 sampling positions, (>-1, <H) gate, mask multiply
  DCNv2/src/cuda/dcn_v2_cuda.cu:87-88,124-164  output size, bias + W.columns
  DCNv2/dcn_v2.py:118-128                      DCN module: offset/mask split + sigmoid
"""
import ctypes
import os

import numpy as np
import torch


def _r16(t):
    return t.half().to(t.dtype)


def dcn_v2_forward(input, weight, bias, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, dg,
                   acc_dtype=None, blend=None):
    """input [B,C,H,W], weight [Co,C,kh,kw], bias [Co], offset [B,2*dg*kh*kw,Ho,Wo],
    mask [B,dg*kh*kw,Ho,Wo] -> [B,Co,Ho,Wo].  acc_dtype=torch.float64 gives the
    order-independent GEMM value.
    blend='f16': the bilinear blend as the 2-byte GPU plans evaluate it (csrc/dcn_traits.h SE<bf16_t>::blend): the four weights
    times the mask rounded to fp16, then x = fp16(v1 w1); x = fp16(fma(v_k, w_k, x)) for k = 2..4 -- four fp16 roundings per sample
    (the inputs must hold fp16-representable values)."""
    B, C, H, W = input.shape
    Co = weight.shape[0]
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    assert offset.shape == (B, 2 * dg * kh * kw, Ho, Wo), offset.shape
    assert mask.shape == (B, dg * kh * kw, Ho, Wo), mask.shape
    cpg = C // dg
    dt = input.dtype
    cols = torch.zeros(B, C, kh * kw, Ho * Wo, dtype=dt)
    ys = (torch.arange(Ho) * sh - ph).to(dt).view(1, Ho, 1)
    xs = (torch.arange(Wo) * sw - pw).to(dt).view(1, 1, Wo)
    flat = input.reshape(B, C, H * W)
    for g in range(dg):
        src = flat[:, g * cpg:(g + 1) * cpg]                    # [B,cpg,HW]
        for i in range(kh):
            for j in range(kw):
                t = i * kw + j
                off_h = offset[:, g * 2 * kh * kw + 2 * t]
                off_w = offset[:, g * 2 * kh * kw + 2 * t + 1]
                m = mask[:, g * kh * kw + t]
                h_im = (ys + float(i * dh)) + off_h              # [B,Ho,Wo]
                w_im = (xs + float(j * dw)) + off_w
                inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
                h_low = torch.floor(h_im)
                w_low = torch.floor(w_im)
                lh = h_im - h_low
                lw = w_im - w_low
                hh = 1 - lh
                hw = 1 - lw
                h_low = h_low.long()
                w_low = w_low.long()
                h_high = h_low + 1
                w_high = w_low + 1

                def corner(hi, wi, ok):
                    ok = ok & inside
                    idx = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).view(B, 1, Ho * Wo)
                    v = torch.gather(src, 2, idx.expand(B, cpg, Ho * Wo))
                    return v * ok.view(B, 1, Ho * Wo).to(dt)

                v1 = corner(h_low, w_low, (h_low >= 0) & (w_low >= 0))
                v2 = corner(h_low, w_high, (h_low >= 0) & (w_high <= W - 1))
                v3 = corner(h_high, w_low, (h_high <= H - 1) & (w_low >= 0))
                v4 = corner(h_high, w_high, (h_high <= H - 1) & (w_high <= W - 1))
                w1 = (hh * hw).view(B, 1, -1)
                w2 = (hh * lw).view(B, 1, -1)
                w3 = (lh * hw).view(B, 1, -1)
                w4 = (lh * lw).view(B, 1, -1)
                if blend == "f16":
                    mm = (m * inside.to(dt)).reshape(B, 1, -1)
                    q1, q2, q3, q4 = [_r16(wk * mm).double() for wk in (w1, w2, w3, w4)]      # make_geo: (w * mask) -> fp16
                    xv = _r16(v1.double() * q1)                                                   # v_pk_mul_f16
                    for vk, qk in ((v2, q2), (v3, q3), (v4, q4)):
                        xv = _r16(vk.double() * qk + xv)                                          # v_pk_fma_f16 (one rounding: the fp64 sum is exact)
                    cols[:, g * cpg:(g + 1) * cpg, t] = xv.to(dt)
                    continue
                val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
                val = val * inside.view(B, 1, -1).to(dt)
                cols[:, g * cpg:(g + 1) * cpg, t] = val * m.reshape(B, 1, -1)
    a = weight.reshape(Co, C * kh * kw)
    c = cols.reshape(B, C * kh * kw, Ho * Wo)
    if acc_dtype is not None:
        out = torch.matmul(a.to(acc_dtype), c.to(acc_dtype)) + bias.to(acc_dtype).view(1, Co, 1)
        out = out.to(dt)
    else:
        out = torch.matmul(a, c) + bias.view(1, Co, 1)
    return out.view(B, Co, Ho, Wo)


def dcn_module_forward(x, weight, bias, om_weight, om_bias, stride=1, padding=1, dilation=1, dg=1,
                       acc_dtype=None, blend=None):
    """DCN.forward (dcn_v2.py:118-128): conv_offset_mask -> chunk(3) -> cat(o1,o2), sigmoid(mask)."""
    kh, kw = weight.shape[2:]
    out = torch.nn.functional.conv2d(x, om_weight, om_bias, stride, padding)
    o1, o2, mask = torch.chunk(out, 3, dim=1)
    offset = torch.cat((o1, o2), dim=1)
    mask = torch.sigmoid(mask)
    return dcn_v2_forward(x, weight, bias, offset, mask, kh, kw, stride, stride, padding, padding,
                          dilation, dilation, dg, acc_dtype=acc_dtype, blend=blend)


# ---------------------------------------------------------------------------------------------
# plain-C restatement (oracle/dcn_ref.c), built by `make -C oracle` into oracle/_build/
_C = None


def _load_c():
    global _C
    if _C is None:
        here = os.path.dirname(os.path.abspath(__file__))
        path = os.path.join(here, "_build", "libh3d_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle C restatement not built: run `make -C oracle` "
                               "(or __graft_entry__.build())")
        _C = ctypes.CDLL(path)
        _C.h3d_oracle_dcn_v2_forward.restype = ctypes.c_int
    return _C


def dcn_v2_forward_c(input, weight, bias, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, dg):
    """numpy in / numpy out through the plain-C loops."""
    lib = _load_c()
    a = [np.ascontiguousarray(t, dtype=np.float32) for t in (input, weight, bias, offset, mask)]
    B, C, H, W = a[0].shape
    Co = a[1].shape[0]
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    out = np.empty((B, Co, Ho, Wo), dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    rc = lib.h3d_oracle_dcn_v2_forward(*[t.ctypes.data_as(fp) for t in a], out.ctypes.data_as(fp),
                                       B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg)
    if rc != 0:
        raise RuntimeError("h3d_oracle_dcn_v2_forward: bad shapes")
    return out
