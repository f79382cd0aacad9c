"""Host mirror of the reference's models/utils.py (the parts on the inference path):
`_sigmoid`, `_gather_feat`, `_transpose_and_gather_feat` -- same names and argument meaning,
computed by libh3d_hip.so (csrc/decode.hip)."""
import torch

from . import _lib


def _f32c(t):
    _lib.require_cuda(t)
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


def _sigmoid(x):
    """clamp(sigmoid(x), 1e-4, 1-1e-4)  (reference utils.py:8-10).
    The reference also overwrites `x` with the un-clamped sigmoid as a side effect of
    `x.sigmoid_()`; every call site rebinds the name to the return value
    (trains/trainer.py:93,127), so only the returned tensor is produced here."""
    x = _f32c(x)
    y = torch.empty_like(x)
    _lib.check(_lib.lib().h3d_sigmoid_clamp(_lib.ptr(x), _lib.ptr(y), x.numel(), _lib.stream_ptr()),
               "_sigmoid")
    return y


def _gather_feat(feat, ind, mask=None):
    """feat [B,N,C], ind [B,K] -> [B,K,C]  (reference utils.py:12-21)."""
    feat = _f32c(feat)
    _lib.require_cuda(ind)
    ind = ind.contiguous().long()
    B, N, C = feat.shape
    K = ind.shape[1]
    out = torch.empty(B, K, C, dtype=torch.float32, device=feat.device)
    _lib.check(_lib.lib().h3d_gather_feat(_lib.ptr(feat), _lib.ptr(ind), B, C, N, K, 1, _lib.ptr(out),
                                          _lib.stream_ptr()), "_gather_feat")
    if mask is not None:
        mask = mask.unsqueeze(2).expand_as(out)
        out = out[mask].view(-1, C)
    return out


def _transpose_and_gather_feat(feat, ind):
    """feat [B,C,H,W], ind [B,K] (flat y*W+x) -> [B,K,C]  (reference utils.py:23-27); the NHWC
    transpose of the whole map is skipped, the K rows are read in place."""
    feat = _f32c(feat)
    _lib.require_cuda(ind)
    ind = ind.contiguous().long()
    B, C, H, W = feat.shape
    K = ind.shape[1]
    out = torch.empty(B, K, C, dtype=torch.float32, device=feat.device)
    _lib.check(_lib.lib().h3d_gather_feat(_lib.ptr(feat), _lib.ptr(ind), B, C, H * W, K, 0, _lib.ptr(out),
                                          _lib.stream_ptr()), "_transpose_and_gather_feat")
    return out
