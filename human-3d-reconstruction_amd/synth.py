"""Deterministic synthetic tensors (weights, images, head maps).

There is no network for checkpoints or datasets, so every weight/input used by
bench.py, the tests and the golden-fixture generator is produced by pure
integer/IEEE arithmetic (splitmix64 counter hash -> uniform), independent of
any library RNG stream, so the same bytes come out in every container.

Only `+ - *` on float64/float32 are used (no transcendental), which makes the
values bit-reproducible across numpy builds.
"""
import zlib

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _key_seed(key, seed):
    return ((zlib.crc32(key.encode("utf-8")) << 32) ^ (seed & 0xFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF


def uniform01(key, shape, seed=0, stream=0):
    """float64 uniform in [0,1), a pure function of (key, seed, stream, index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    base0 = (_key_seed(key, seed) + stream * 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF
    base = _splitmix64(np.array([base0], dtype=np.uint64))[0]
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + base
    z = _splitmix64(ctr)
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return u.reshape(shape)


def uniform(key, shape, lo, hi, seed=0, stream=0):
    return (lo + (hi - lo) * uniform01(key, shape, seed, stream)).astype(np.float32)


def normalish(key, shape, mean=0.0, std=1.0, seed=0):
    """Irwin-Hall(4) approximation of a normal: exact IEEE adds only."""
    s = np.zeros(shape, dtype=np.float64)
    for k in range(4):
        s = s + uniform01(key, shape, seed, stream=1 + k)
    # var of sum of 4 U(0,1) = 4/12 -> scale by sqrt(3)
    z = (s - 2.0) * 1.7320508075688772
    return (mean + std * z).astype(np.float32)


def synth_state_dict(shapes, seed=0, offset_scale=0.5, gain=1.0):
    """Synthetic DLA-34 weights for a {key: shape} table (reference state_dict names,
    reference trainer.py:475-509 for the key format).

    conv weights  : U(-a, a) with a = gain * sqrt(3 / fan_in).  gain = 1 (the default, what the committed golden
                    fixtures were generated with) is unit gain per conv, but every ReLU halves the mean square, so after
                    ~50 layers the feature maps are almost constant (head maps: std 0.02-0.1 around their bias) and the
                    top-k peaks are decided by noise.  gain = 1.25 keeps the signal alive through DLA-34 (head maps with
                    std 0.5-1 and separated peaks): bench.py and the full-size / index-match tests use that
    BN            : weight U(0.5,1.5), bias U(-0.2,0.2), running_mean U(-0.2,0.2),
                    running_var U(0.5,1.5)  (non-trivial, so BN folding is exercised)
    conv_offset_mask : U(-a,a)*offset_scale so DCN offsets are non-trivial
                    (the reference zero-inits it, dcn_v2.py:114-116, which would hide gather bugs)
    up_k.weight   : bilinear kernel of fill_up_weights (model.py:334-343) times U(0.9,1.1)
                    per channel (the deconv is trainable: per-channel weights are exercised)
    head biases   : 'hm*' final bias -2.19 (model.py:461-462), others U(-0.1,0.1)
    """
    import math
    sd = {}
    for k, shp in shapes.items():
        shp = tuple(shp)
        if k.endswith("num_batches_tracked"):
            sd[k] = np.zeros((), dtype=np.int64)
            continue
        leaf = k.split(".")[-1]
        parent = k.split(".")[-2] if "." in k else ""
        if leaf == "running_var":
            sd[k] = uniform(k, shp, 0.5, 1.5, seed)
        elif leaf == "running_mean":
            sd[k] = uniform(k, shp, -0.2, 0.2, seed)
        elif len(shp) == 1 and leaf == "weight":       # BN gamma
            sd[k] = uniform(k, shp, 0.5, 1.5, seed)
        elif len(shp) == 1 and leaf == "bias":
            if (parent == "2" or parent == "") and k.split(".")[0].startswith("hm"):
                sd[k] = np.full(shp, -2.19, dtype=np.float32)
            else:
                sd[k] = uniform(k, shp, -0.1, 0.1, seed)
        elif len(shp) == 4 and parent.startswith("up_"):
            kk = shp[2]
            f = math.ceil(kk / 2)
            c = (2 * f - 1 - f % 2) / (2.0 * f)
            w = np.zeros(shp, dtype=np.float64)
            for i in range(kk):
                for j in range(kk):
                    w[:, 0, i, j] = (1 - math.fabs(i / f - c)) * (1 - math.fabs(j / f - c))
            g = uniform01(k, (shp[0],), seed) * 0.2 + 0.9
            sd[k] = (w * g[:, None, None, None]).astype(np.float32)
        elif len(shp) == 4:
            fan_in = shp[1] * shp[2] * shp[3]
            a = math.sqrt(3.0 / fan_in)
            if parent == "conv_offset_mask":
                a *= offset_scale
            else:
                a *= gain
            sd[k] = uniform(k, shp, -a, a, seed)
        else:
            sd[k] = uniform(k, shp, -0.1, 0.1, seed)
    return sd


def synth_images(batch, h=512, w=512, seed=317):
    """Normalised-image-like input [B,3,H,W] fp32 (reference seed opts.py:37;
    normalisation coco_hp.py:206-209 gives roughly unit-variance channels)."""
    return normalish("images", (batch, 3, h, w), 0.0, 1.0, seed)


def synth_image_batch(batch, h=512, w=512, seed=317, first=0):
    """`batch` INDEPENDENT images (image i is a pure function of (seed, first + i), so a rank's shard [lo, hi) of a global
    batch is synth_image_batch(hi - lo, ..., first=lo)); generated one image at a time (small temporaries: 64 images of
    512x512 take 3 s where the one-shot generator needs 18)."""
    return np.concatenate([normalish("images.%d" % (first + i), (1, 3, h, w), 0.0, 1.0, seed) for i in range(batch)])


def synth_heads(batch, h=128, w=128, num_joints=17, seed=0):
    """Decode-only inputs shaped like post-_sigmoid network heads.

    hm / hm_hp = clamp(u^8, 1e-4, 1-1e-4) computed with float32 multiplies only
    (bit-reproducible), giving ~1/9 local maxima and a large clamped plateau at
    1e-4, like real heat maps (hard parts).
    """
    def heat(key, c):
        u = uniform(key, (batch, c, h, w), 0.0, 1.0, seed)
        u2 = u * u
        u4 = u2 * u2
        u8 = u4 * u4
        return np.clip(u8, np.float32(1e-4), np.float32(1 - 1e-4)).astype(np.float32)
    return {
        "hm": heat("hm", 1),
        "wh": uniform("wh", (batch, 2, h, w), 2.0, 40.0, seed),
        "hps": normalish("hps", (batch, 2 * num_joints, h, w), 0.0, 8.0, seed),
        "reg": uniform("reg", (batch, 2, h, w), 0.0, 1.0, seed),
        "hm_hp": heat("hm_hp", num_joints),
        "hp_offset": uniform("hp_offset", (batch, 2, h, w), 0.0, 1.0, seed),
    }
