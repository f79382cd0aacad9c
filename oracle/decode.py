"""TEST INFRASTRUCTURE ONLY -- numpy float32 restatement of the reference's heat-map decode.

Reference lines followed (/root/reference/src/lib/models/):
  utils.py:8-10    _sigmoid  = clamp(sigmoid(x), 1e-4, 1-1e-4)
  decode.py:6-13   _nms      = heat * (maxpool3x3(heat) == heat)
  decode.py:15-24  _topk_channel
  decode.py:26-41  _topk (two-stage)
  utils.py:12-27   _gather_feat / _transpose_and_gather_feat
  decode.py:77-163 multi_pose_decode
  decode.py:44-75  ctdet_decode

Tie rule (ours, documented in DESIGN.md): equal scores are ordered lowest flat index first
(torch.topk leaves tie order unspecified, SURVEY 7 'Hard parts').  Wherever the top-(K+1)
scores are distinct this is bit-identical to the reference; tests/golden/decode_*.npz holds the
reference's own outputs and tests compare the strictly-distinct prefix exactly.
"""
import numpy as np

F32 = np.float32


def sigmoid_clamp(x):
    x = np.asarray(x, dtype=F32)
    y = (F32(1) / (F32(1) + np.exp(-x, dtype=F32))).astype(F32)
    return np.clip(y, F32(1e-4), F32(1 - 1e-4)).astype(F32)


def nms(heat):
    """3x3 stride-1 max pool with -inf padding, keep exact-equal positions (plateaus survive)."""
    heat = np.asarray(heat, dtype=F32)
    B, C, H, W = heat.shape
    pad = np.full((B, C, H + 2, W + 2), -np.inf, dtype=F32)
    pad[:, :, 1:-1, 1:-1] = heat
    hmax = pad[:, :, 0:H, 0:W].copy()
    for dy in range(3):
        for dx in range(3):
            np.maximum(hmax, pad[:, :, dy:dy + H, dx:dx + W], out=hmax)
    keep = (hmax == heat).astype(F32)
    return heat * keep


def _topk_rows(scores, K):
    """scores [..., N] -> (values, indices) of the K largest, value desc then index asc."""
    order = np.argsort(-scores, axis=-1, kind="stable")[..., :K]
    return np.take_along_axis(scores, order, axis=-1), order.astype(np.int64)


def topk_channel(scores, K):
    B, C, H, W = scores.shape
    s, inds = _topk_rows(scores.reshape(B, C, H * W), K)
    inds = inds % (H * W)
    ys = (inds // W).astype(F32)
    xs = (inds % W).astype(F32)
    return s, inds, ys, xs


def topk(scores, K):
    B, C, H, W = scores.shape
    s, inds, ys, xs = topk_channel(scores, K)
    s2, ind2 = _topk_rows(s.reshape(B, C * K), K)
    clses = (ind2 // K).astype(np.int32)
    inds = np.take_along_axis(inds.reshape(B, C * K), ind2, axis=1)
    ys = np.take_along_axis(ys.reshape(B, C * K), ind2, axis=1)
    xs = np.take_along_axis(xs.reshape(B, C * K), ind2, axis=1)
    return s2, inds, clses, ys, xs


def gather_nchw(feat, ind):
    """_transpose_and_gather_feat: feat [B,C,H,W], ind [B,N] flat (y*W+x) -> [B,N,C]."""
    B, C, H, W = feat.shape
    f = np.asarray(feat, dtype=F32).reshape(B, C, H * W)
    return np.stack([f[b][:, ind[b]].T for b in range(B)], 0)


def ctdet_decode(heat, wh, reg=None, cat_spec_wh=False, K=100):
    B, C, H, W = heat.shape
    heat = nms(heat)
    scores, inds, clses, ys, xs = topk(heat, K)
    if reg is not None:
        r = gather_nchw(reg, inds)
        xs = xs[..., None] + r[..., 0:1]
        ys = ys[..., None] + r[..., 1:2]
    else:
        xs = xs[..., None] + F32(0.5)
        ys = ys[..., None] + F32(0.5)
    w = gather_nchw(wh, inds)
    if cat_spec_wh:
        w = w.reshape(B, K, C, 2)
        w = np.take_along_axis(w, clses.reshape(B, K, 1, 1).astype(np.int64).repeat(2, 3), axis=2)
        w = w.reshape(B, K, 2)
    half = F32(2)
    bboxes = np.concatenate([xs - w[..., 0:1] / half, ys - w[..., 1:2] / half,
                             xs + w[..., 0:1] / half, ys + w[..., 1:2] / half], axis=2)
    return np.concatenate([bboxes, scores[..., None], clses[..., None].astype(F32)], axis=2).astype(F32)


def multi_pose_decode(heat, wh, kps, reg=None, hm_hp=None, hp_offset=None, K=100,
                      return_aux=False):
    B, C, H, W = heat.shape
    J = kps.shape[1] // 2
    heat = nms(heat)
    scores, inds, clses, ys, xs = topk(heat, K)

    kp = gather_nchw(kps, inds).copy()                      # [B,K,2J]
    kp[..., 0::2] += xs[..., None]
    kp[..., 1::2] += ys[..., None]
    if reg is not None:
        r = gather_nchw(reg, inds)
        cx = xs[..., None] + r[..., 0:1]
        cy = ys[..., None] + r[..., 1:2]
    else:
        cx = xs[..., None] + F32(0.5)
        cy = ys[..., None] + F32(0.5)
    w = gather_nchw(wh, inds)
    half = F32(2)
    bboxes = np.concatenate([cx - w[..., 0:1] / half, cy - w[..., 1:2] / half,
                             cx + w[..., 0:1] / half, cy + w[..., 1:2] / half], axis=2).astype(F32)
    aux = {"scores": scores, "inds": inds, "clses": clses, "ys": ys, "xs": xs}
    if hm_hp is not None:
        hm_hp = nms(hm_hp)
        thresh = F32(0.1)
        kp = kp.reshape(B, K, J, 2).transpose(0, 2, 1, 3).copy()          # B,J,K,2
        hm_score, hm_inds, hm_ys, hm_xs = topk_channel(hm_hp, K)          # B,J,K
        aux.update({"hm_score": hm_score, "hm_inds": hm_inds})
        if hp_offset is not None:
            o = gather_nchw(hp_offset, hm_inds.reshape(B, -1)).reshape(B, J, K, 2)
            hm_xs = hm_xs + o[..., 0]
            hm_ys = hm_ys + o[..., 1]
        else:
            hm_xs = hm_xs + F32(0.5)
            hm_ys = hm_ys + F32(0.5)
        mask = (hm_score > thresh).astype(F32)
        hm_score = (1 - mask) * F32(-1) + mask * hm_score
        hm_ys = (1 - mask) * F32(-10000) + mask * hm_ys
        hm_xs = (1 - mask) * F32(-10000) + mask * hm_xs
        # dist[b,j,k,c] between regressed kp k and heat-map candidate c
        dx = kp[:, :, :, None, 0] - hm_xs[:, :, None, :]
        dy = kp[:, :, :, None, 1] - hm_ys[:, :, None, :]
        dist = np.sqrt(dx * dx + dy * dy, dtype=F32)
        min_ind = dist.argmin(axis=3)                                      # first minimum
        min_dist = np.take_along_axis(dist, min_ind[..., None], axis=3)    # B,J,K,1
        sel_score = np.take_along_axis(hm_score, min_ind, axis=2)[..., None]
        sel_x = np.take_along_axis(hm_xs, min_ind, axis=2)[..., None]
        sel_y = np.take_along_axis(hm_ys, min_ind, axis=2)[..., None]
        l = bboxes[:, None, :, 0:1]
        t = bboxes[:, None, :, 1:2]
        r_ = bboxes[:, None, :, 2:3]
        b_ = bboxes[:, None, :, 3:4]
        bad = ((sel_x < l) | (sel_x > r_) | (sel_y < t) | (sel_y > b_) | (sel_score < thresh) |
               (min_dist > np.maximum(b_ - t, r_ - l) * F32(0.3)))
        m = bad.astype(F32)
        hm_kps = np.concatenate([sel_x, sel_y], axis=3)
        kp = (1 - m) * hm_kps + m * kp
        kp = kp.transpose(0, 2, 1, 3).reshape(B, K, 2 * J)
    dets = np.concatenate([bboxes, scores[..., None], kp, clses[..., None].astype(F32)],
                          axis=2).astype(F32)
    if return_aux:
        return dets, aux
    return dets


def strict_prefix(scores, K):
    """Length of the leading run of a descending score list whose values are strictly greater
    than everything after position K-1 would tie with: the prefix where tie order cannot matter.
    scores: [N>=K] sorted descending (N may be K or K+1)."""
    s = np.asarray(scores)
    n = min(K, len(s))
    # positions i such that s[i] is unique among s[0:len(s)]
    uniq = np.ones(n, dtype=bool)
    for i in range(n):
        if (i > 0 and s[i] == s[i - 1]) or (i + 1 < len(s) and s[i] == s[i + 1]):
            uniq[i] = False
    p = 0
    while p < n and uniq[p]:
        p += 1
    return p


# ---- flip-test helpers (reference models/utils.py:29-51), numpy restatement ------------------------------
def flip_tensor(x):
    return x[..., ::-1].copy()


def flip_lr(x, flip_idx):
    tmp = x[..., ::-1].copy()
    for e in flip_idx:
        tmp[:, e[0], ...], tmp[:, e[1], ...] = tmp[:, e[1], ...].copy(), tmp[:, e[0], ...].copy()
    return tmp


def flip_lr_off(x, flip_idx):
    tmp = x[..., ::-1].copy()
    shape = tmp.shape
    tmp = tmp.reshape(tmp.shape[0], 17, 2, tmp.shape[2], tmp.shape[3])
    tmp[:, :, 0, :, :] *= -1
    for e in flip_idx:
        tmp[:, e[0], ...], tmp[:, e[1], ...] = tmp[:, e[1], ...].copy(), tmp[:, e[0], ...].copy()
    return tmp.reshape(shape)
