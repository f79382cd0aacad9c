"""Host mirror of the reference's models/decode.py: same function names, arguments and return
values; all arithmetic runs in csrc/decode.hip.

Tie rule: where the reference's torch.topk leaves the order of equal scores unspecified, this
implementation orders them lowest flat index first ('Top-k ties')."""
import torch

from . import _lib
from .utils import _f32c

NMS_SIGMOID, NMS_SKIP = 1, 2


def _nms(heat, kernel=3):
    """heat * (max_pool2d(heat, 3, 1, 1) == heat)  (reference decode.py:6-13)."""
    if kernel != 3:
        raise RuntimeError("_nms: only the 3x3 kernel of the reference call sites is implemented")
    heat = _f32c(heat)
    B, C, H, W = heat.shape
    out = torch.empty_like(heat)
    _lib.check(_lib.lib().h3d_nms(_lib.ptr(heat), B, C, H, W, _lib.ptr(out), _lib.stream_ptr()), "_nms")
    return out


MAX_LDS_MAP = 36864      # pixels of a map the one-workgroup kernel holds in LDS; larger maps go through h3d_nms_topk_large


def _map_topk(scores, K, flags):
    scores = _f32c(scores)
    B, C, H, W = scores.shape
    dev = scores.device
    s = torch.empty(B, C, K, dtype=torch.float32, device=dev)
    i = torch.empty(B, C, K, dtype=torch.int64, device=dev)
    y = torch.empty(B, C, K, dtype=torch.float32, device=dev)
    x = torch.empty(B, C, K, dtype=torch.float32, device=dev)
    if H * W > MAX_LDS_MAP:
        # e.g. the 320 x 184 output map of a --keep_res 1280 x 736 frame (datasets/coco.py:160-163): bands of rows + a merge
        L = _lib.lib()
        nws = int(L.h3d_nms_topk_large_workspace_bytes(B, C, H, W, K))
        if nws == 0:
            raise RuntimeError("top-k: a %d x %d map with K = %d is not supported (a band of rows and its two halo rows must fit "
                               "%d pixels)" % (H, W, K, MAX_LDS_MAP))
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        _lib.check(L.h3d_nms_topk_large(_lib.ptr(scores), B, C, H, W, K, flags, _lib.ptr(s), _lib.ptr(i), _lib.ptr(y), _lib.ptr(x),
                                        _lib.ptr(ws), nws, _lib.stream_ptr()), "topk")
        return s, i, y, x
    _lib.check(_lib.lib().h3d_nms_topk(_lib.ptr(scores), B, C, H, W, K, flags, _lib.ptr(s), _lib.ptr(i),
                                       _lib.ptr(y), _lib.ptr(x), _lib.stream_ptr()), "topk")
    return s, i, y, x


def _map_topk2(a, b, K, flags):
    """_map_topk of two tensors with the same B, H, W in ONE launch (h3d_nms_topk2)."""
    a, b = _f32c(a), _f32c(b)
    B, Ca, H, W = a.shape
    Cb = b.shape[1]
    dev = a.device
    outs = []
    for C in (Ca, Cb):
        outs.append((torch.empty(B, C, K, dtype=torch.float32, device=dev), torch.empty(B, C, K, dtype=torch.int64, device=dev),
                     torch.empty(B, C, K, dtype=torch.float32, device=dev), torch.empty(B, C, K, dtype=torch.float32, device=dev)))
    _lib.check(_lib.lib().h3d_nms_topk2(_lib.ptr(a), Ca, *[_lib.ptr(t) for t in outs[0]], _lib.ptr(b), Cb,
                                        *[_lib.ptr(t) for t in outs[1]], B, H, W, K, flags, _lib.stream_ptr()), "topk2")
    return outs


_ZERO_CLS = {}
SINGLE_CLASS_SHORTCUT = True      # False: always launch topk_merge_kernel (A/B and bit-identity tests)


def _merge(s, i, y, x, K):
    B, C, _ = s.shape
    dev = s.device
    if C == 1 and SINGLE_CLASS_SHORTCUT:
        # stage 2 of `_topk` (decode.py:34-39) over ONE class is the identity: the K stage-1 candidates already come sorted by
        # (score descending, flat index ascending) and the merge orders equal scores by position -- the multi_pose task has one
        # class (opts.py:248), so its `_topk` is one launch less (64 workgroups of a bitonic sort that moved nothing)
        # class ids: ONE read-only zero buffer per device for the life of the process (never evicted: captured hipGraphs hold its
        # address), handed out as views -- the public `_topk` clones it (callers may edit their result in place)
        n = B * K
        z = _ZERO_CLS.get(str(dev))
        if z is None or z.numel() < n:
            z = torch.zeros(max(n, 1 << 16), dtype=torch.int32, device=dev)
            torch.cuda.current_stream(dev).synchronize()         # (filled before any other stream reads it; read-only afterwards)
            _ZERO_CLS.setdefault("keep", []).append(z)           # (an outgrown buffer stays alive for the graphs that captured it)
            _ZERO_CLS[str(dev)] = z
        return s.view(B, K), i.view(B, K), z[:n].view(B, K), y.view(B, K), x.view(B, K)
    o_s = torch.empty(B, K, dtype=torch.float32, device=dev)
    o_i = torch.empty(B, K, dtype=torch.int64, device=dev)
    o_c = torch.empty(B, K, dtype=torch.int32, device=dev)
    o_y = torch.empty(B, K, dtype=torch.float32, device=dev)
    o_x = torch.empty(B, K, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().h3d_topk_merge(_lib.ptr(s), _lib.ptr(i), _lib.ptr(y), _lib.ptr(x), B, C, K,
                                         _lib.ptr(o_s), _lib.ptr(o_i), _lib.ptr(o_c), _lib.ptr(o_y),
                                         _lib.ptr(o_x), _lib.stream_ptr()), "_topk")
    return o_s, o_i, o_c, o_y, o_x


def _topk_channel(scores, K=40):
    """Per-channel top-K of an (already NMS-ed) score map (reference decode.py:15-24):
    returns (topk_scores, topk_inds, topk_ys, topk_xs), each [B,C,K]."""
    return _map_topk(scores, K, NMS_SKIP)


def _topk(scores, K=40):
    """Two-stage top-K (reference decode.py:26-41): returns
    (topk_score [B,K], topk_inds [B,K] int64, topk_clses [B,K] int32, topk_ys, topk_xs)."""
    s, i, c, y, x = _merge(*_map_topk(scores, K, NMS_SKIP), K)
    if scores.shape[1] == 1 and SINGLE_CLASS_SHORTCUT:
        c = c.clone()          # (the shortcut's class ids are a shared read-only buffer; a public result must be the caller's own)
    return s, i, c, y, x


def ctdet_decode(heat, wh, reg=None, cat_spec_wh=False, K=100):
    """reference decode.py:44-75 -> detections [B,K,6]."""
    heat, wh = _f32c(heat), _f32c(wh)
    reg = None if reg is None else _f32c(reg)
    B, C, H, W = heat.shape
    s, i, c, y, x = _merge(*_map_topk(heat, K, 0), K)
    dets = torch.empty(B, K, 6, dtype=torch.float32, device=heat.device)
    _lib.check(_lib.lib().h3d_ctdet_assemble(_lib.ptr(s), _lib.ptr(i), _lib.ptr(c), _lib.ptr(y), _lib.ptr(x),
                                             _lib.ptr(wh), _lib.ptr(reg), B, C, H, W, K, int(bool(cat_spec_wh)),
                                             _lib.ptr(dets), _lib.stream_ptr()), "ctdet_decode")
    return dets


def _multi_pose(heat, wh, kps, reg, hm_hp, hp_offset, K, logits, return_aux=False):
    heat, wh, kps = _f32c(heat), _f32c(wh), _f32c(kps)
    reg = None if reg is None else _f32c(reg)
    hm_hp = None if hm_hp is None else _f32c(hm_hp)
    hp_offset = None if hp_offset is None else _f32c(hp_offset)
    B, C, H, W = heat.shape
    J = kps.shape[1] // 2
    flags = NMS_SIGMOID if logits else 0
    if hm_hp is not None and hm_hp.shape[0] == B and hm_hp.shape[2:] == heat.shape[2:] and H * W <= MAX_LDS_MAP:
        (s1, i1, y1, x1), (hs, hi, hy, hx) = _map_topk2(heat, hm_hp, K, flags)      # one launch for both tensors
        s, i, c, y, x = _merge(s1, i1, y1, x1, K)
    else:
        s, i, c, y, x = _merge(*_map_topk(heat, K, flags), K)
        if hm_hp is not None:
            hs, hi, hy, hx = _map_topk(hm_hp, K, flags)
        else:
            hs = hi = hy = hx = None
            hp_offset = None
    dets = torch.empty(B, K, 5 + 2 * J + 1, dtype=torch.float32, device=heat.device)
    _lib.check(_lib.lib().h3d_multi_pose_assemble(
        _lib.ptr(s), _lib.ptr(i), _lib.ptr(c), _lib.ptr(y), _lib.ptr(x),
        _lib.ptr(hs), _lib.ptr(hi), _lib.ptr(hy), _lib.ptr(hx),
        _lib.ptr(wh), _lib.ptr(kps), _lib.ptr(reg), _lib.ptr(hp_offset),
        B, J, H, W, K, _lib.ptr(dets), _lib.stream_ptr()), "multi_pose_decode")
    if return_aux:     # ("clses" of a one-class map is a view of a shared read-only zero buffer: see _merge)
        return dets, {"scores": s, "inds": i, "clses": c, "ys": y, "xs": x, "hm_score": hs, "hm_inds": hi}
    return dets


def multi_pose_decode(heat, wh, kps, reg=None, hm_hp=None, hp_offset=None, K=100):
    """reference decode.py:77-163: `heat`/`hm_hp` are post-`_sigmoid` maps -> detections
    [B,K,40] = [bbox(4), score, 17x(x,y), class]."""
    return _multi_pose(heat, wh, kps, reg, hm_hp, hp_offset, K, logits=False)


def multi_pose_decode_logits(heat, wh, kps, reg=None, hm_hp=None, hp_offset=None, K=100, return_aux=False):
    """Fused form used by the detector: `heat`/`hm_hp` are the raw head outputs; the in-place
    `_sigmoid` the reference's loss module applies before decode (trains/trainer.py:93,127) is
    folded into the NMS kernel."""
    return _multi_pose(heat, wh, kps, reg, hm_hp, hp_offset, K, logits=True, return_aux=return_aux)
