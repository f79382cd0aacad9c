"""TEST INFRASTRUCTURE ONLY -- how well the bf16 GPU path's top-k peak indices agree with the fp32 oracle's.

The metric reads "top-k index bit-match" (reference `_topk`, models/decode.py:26-41).  Decode itself is bit-exact on
identical heads (tests/test_gpu_decode.py).  End to end, the bf16 network's `hm` logits differ from the fp32 oracle's by
delta ~ 0.03, so two peaks whose scores are closer than that may swap ranks; what can be asserted is
  * robust_prefix: ranks 1..n of the oracle whose order CANNOT change under a perturbation of delta (see below)
    must be identical on the GPU -- bit-match where bit-match is defined;
  * statistics over all K ranks: positional agreement, set overlap, the longest equal prefix.
Used by tests/test_gpu_fullsize.py and by bench.py's cpu_baseline leg (which already runs the oracle).
"""
import numpy as np

from . import decode as odec


def robust_prefix(score_map, K, delta):
    """Largest n such that the top-n peak indices of `score_map` [H,W] (post-_sigmoid) are the same for every map
    within `delta` (max-norm) of it.  Rank i is safe when, for the i-th peak p:
      (1) p beats its 8 neighbours by more than 2 delta (it stays a peak),
      (2) no other pixel's score is within 2 delta of p's (nothing can cross it),
      (3) every NON-peak pixel scoring above v_p - 2 delta loses to a neighbour by more than 2 delta (no new peak can
          appear above p)."""
    v = np.asarray(score_map, dtype=np.float64)
    H, W = v.shape
    pad = np.full((H + 2, W + 2), -np.inf)
    pad[1:-1, 1:-1] = v
    nb = np.full((H, W), -np.inf)
    for dy in range(3):
        for dx in range(3):
            if dy == 1 and dx == 1:
                continue
            nb = np.maximum(nb, pad[dy:dy + H, dx:dx + W])
    is_peak = v >= nb
    margin = v - nb                                  # > 0: peak by that much; < 0: loses by that much
    flat = v.ravel()
    order = np.argsort(-flat, kind="stable")
    sorted_v = flat[order]
    peaks = [i for i in order if is_peak.ravel()[i]][:K]
    n = 0
    for p in peaks:
        vp = flat[p]
        if margin.ravel()[p] <= 2 * delta:
            break
        pos = np.searchsorted(-sorted_v, -vp)
        lo = sorted_v[pos + 1] if pos + 1 < sorted_v.size else -np.inf
        hi = sorted_v[pos - 1] if pos > 0 else np.inf
        if vp - lo <= 2 * delta or hi - vp <= 2 * delta:
            break
        above = order[:np.searchsorted(-sorted_v, -(vp - 2 * delta), side="right")]
        nonpeak = above[~is_peak.ravel()[above]]
        if nonpeak.size and (margin.ravel()[nonpeak] >= -2 * delta).any():
            break
        n += 1
    return n


_CLAMP_LOGIT = 9.210240366975849          # logit(1 - 1e-4): `_sigmoid` clamps scores to [1e-4, 1 - 1e-4] (models/utils.py:8-10)


def robust_prefix_logit(ref_logits, K, logit_err):
    """`robust_prefix` evaluated on the LOGIT map.  `_sigmoid` is monotone, `_nms` and `_topk` only compare, so the order of
    the peaks is the order of their logits (clamp plateaus aside, and those are ties in both spaces).  Near a score of 1
    the sigmoid compresses gaps (two peaks 0.0066 apart in logits are 7e-6 apart in score), so a bound taken as the
    max-norm of the SCORE error over the whole map (dominated by mid-range pixels, slope 1/4) declares top ranks
    unstable that no logit error of the measured size could swap.  The perturbation allowed here is the measured logit
    error plus what 2 ulp of fp32 score rounding (two sigmoid implementations) amount to in logits at the flattest of
    the top-K peaks."""
    L = np.clip(np.asarray(ref_logits, dtype=np.float64), -_CLAMP_LOGIT, _CLAMP_LOGIT)
    s = 1.0 / (1.0 + np.exp(-L))
    top = np.sort(s.ravel())[::-1][:K]
    slope = float(np.min(top * (1.0 - top)))
    ulp = 2.0 * 5.97e-8 / max(slope, 1e-12)
    return robust_prefix(L, K, float(logit_err) + ulp)


def index_match(gpu_heads, gpu_inds, ref_heads, K=100):
    """gpu_heads / ref_heads: {'hm': [B,1,H,W] logits, ...} numpy; gpu_inds [B,K] from the GPU decode.
    -> dict of plain floats/ints (goes into bench.py's JSON line)."""
    ref_hm = odec.sigmoid_clamp(ref_heads["hm"])
    gpu_hm = odec.sigmoid_clamp(gpu_heads["hm"])
    _, ref_inds, _, _, _ = odec.topk(odec.nms(ref_hm), K)
    B = ref_inds.shape[0]
    gpu_inds = np.asarray(gpu_inds)[:, :K]
    delta = float(np.abs(gpu_hm.astype(np.float64) - ref_hm.astype(np.float64)).max())
    eq = gpu_inds == ref_inds
    prefix = [int(np.argmin(np.r_[e, False])) for e in eq]
    overlap = [len(set(gpu_inds[b].tolist()) & set(ref_inds[b].tolist())) / float(K) for b in range(B)]
    rob = [robust_prefix(ref_hm[b, 0], K, delta) for b in range(B)]
    rob_ok = all(bool((gpu_inds[b, :rob[b]] == ref_inds[b, :rob[b]]).all()) for b in range(B))
    logit_err = float(np.abs(np.asarray(gpu_heads["hm"], np.float64) - np.asarray(ref_heads["hm"], np.float64)).max())
    robl = [robust_prefix_logit(np.asarray(ref_heads["hm"])[b, 0], K, logit_err) for b in range(B)]
    robl_ok = all(bool((gpu_inds[b, :robl[b]] == ref_inds[b, :robl[b]]).all()) for b in range(B))
    head_err = {k: float(np.abs(np.asarray(gpu_heads[k], np.float64) - np.asarray(ref_heads[k], np.float64)).max())
                for k in ref_heads if k in gpu_heads}
    return {"images": B, "K": K,
            "max_abs_head_err": round(max(head_err.values()), 5),
            "max_abs_hm_score_err": round(delta, 6),
            "agreement": round(float(eq.mean()), 4),                  # same index at the same rank
            "set_overlap": round(float(np.mean(overlap)), 4),         # same peak anywhere in the top K
            "equal_prefix": int(min(prefix)),                          # ranks 1..n identical on every image
            "robust_prefix": int(min(rob)),                            # ranks whose order cannot change under that error ...
            "robust_prefix_equal": bool(rob_ok),                       # ... are bit-identical
            "robust_prefix_logit": int(min(robl)),                     # the same analysis on the logit map (sigmoid compresses gaps near 1)
            "robust_prefix_logit_equal": bool(robl_ok)}
