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


def flip_tensor(x):
    """reference models/utils.py:29-30: mirror the width axis of an NCHW map."""
    return torch.flip(x, [3])


def _swap_index(n, flip_idx):
    idx = list(range(n))
    for a, b in flip_idx:                      # the reference swaps rows in sequence (utils.py:37-39)
        idx[a], idx[b] = idx[b], idx[a]
    return idx


def flip_lr(x, flip_idx):
    """reference models/utils.py:34-40 (flip test of the joint heat maps): mirror the width axis and swap the
    left/right channel pairs of `flip_idx`; stays on the tensor's device (the reference round-trips through numpy)."""
    y = torch.flip(x, [3])
    return y[:, torch.as_tensor(_swap_index(x.shape[1], flip_idx), device=x.device)].contiguous()


def flip_lr_off(x, flip_idx):
    """reference models/utils.py:42-51 (flip test of the joint offsets `hps` [B,34,H,W]): mirror the width axis,
    negate the x offsets, swap the left/right joint pairs."""
    B, C, H, W = x.shape
    y = torch.flip(x, [3]).reshape(B, 17, 2, H, W).clone()
    y[:, :, 0] *= -1
    y = y[:, torch.as_tensor(_swap_index(17, flip_idx), device=x.device)]
    return y.reshape(B, C, H, W).contiguous()
