"""TEST INFRASTRUCTURE ONLY -- torch-CPU functional restatement of Hourglass-104 (`exkp`, nstack 2) driven by a
state_dict with the published CenterNet key names.

PARITY UNPINNED: the reference repository names the backbone (src/lib/opts.py:61-63,
experiments/multi_pose_hg_1x.sh) but contains no source, test or checkpoint for it, so this file follows the published
definition (CenterNet `large_hourglass.py`: convolution = conv + BN + ReLU; residual = conv3-BN-ReLU-conv3-BN (+ 1x1
conv + BN skip when the stride or the width changes), ReLU after the add; kp_module = up1(x) + upsample2(low3(low2(low1(x))))
with nearest-neighbour up-sampling and the pooling layer replaced by the stride-2 first residual of low1).  What it IS
pinned to: structural identities checked in tests/test_oracle_backbones.py (parameter count 191.2 M without heads,
output shapes, the kp_module skip path).
"""
import numpy as np
import torch
import torch.nn.functional as F

N = 5
DIMS = [256, 256, 384, 384, 384, 512]
MODULES = [2, 2, 2, 2, 2, 4]
BN_EPS = 1e-5


class HourglassOracle:
    def __init__(self, state_dict, heads, nstack=2, emulate_bf16=False, emulate=None):
        """emulate ('bf16' | 'f16'; emulate_bf16=True is 'bf16'): an independent low-precision evaluation of the same graph --
        conv weights and every STORED activation (the output of each conv + BN (+ residual) + ReLU unit, of each skip conv,
        of each up-sample + add) rounded to that type, fp32 accumulation: the rounding points of a plan whose activations
        live in HBM as 2-byte elements (the head outputs stay fp32)."""
        self.sd = {k: (v if torch.is_tensor(v) else torch.from_numpy(np.asarray(v))) for k, v in state_dict.items()}
        self.heads, self.nstack = heads, nstack
        emulate = "bf16" if emulate_bf16 else emulate
        td = {None: None, "bf16": torch.bfloat16, "f16": torch.float16}[emulate]
        self.q = (lambda t: t.to(td).float()) if td is not None else (lambda t: t)
        if td is not None:
            self.sd = {k: (self.q(v) if v.dim() == 4 else v) for k, v in self.sd.items()}

    def _bn(self, x, p):
        sd = self.sd
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)

    def _conv(self, x, p, stride=1, pad=0):
        return F.conv2d(self.q(x), self.sd[p + ".weight"], self.sd.get(p + ".bias"), stride, pad)

    def _convolution(self, x, p, k, stride=1, with_bn=True):
        y = self._conv(x, p + ".conv", stride, (k - 1) // 2)
        if with_bn:
            y = self._bn(y, p + ".bn")
        return self.q(F.relu(y))

    def _residual(self, x, p, stride=1):
        y = F.relu(self._bn(self._conv(x, p + ".conv1", stride, 1), p + ".bn1"))
        y = self._bn(self._conv(y, p + ".conv2", 1, 1), p + ".bn2")
        skip = self.q(self._bn(self._conv(x, p + ".skip.0", stride, 0), p + ".skip.1")) if (p + ".skip.0.weight") in self.sd else x
        return self.q(F.relu(y + skip))

    def _seq(self, x, p, n, first_stride=1):
        for j in range(n):
            x = self._residual(x, "%s.%d" % (p, j), first_stride if j == 0 else 1)
        return x

    def _kp(self, x, p, n, modules):
        up1 = self._seq(x, p + ".up1", modules[0])
        low1 = self._seq(x, p + ".low1", modules[0], 2)
        low2 = self._kp(low1, p + ".low2", n - 1, modules[1:]) if n > 1 else self._seq(low1, p + ".low2", modules[1])
        low3 = self._seq(low2, p + ".low3", modules[0])
        return self.q(up1 + F.interpolate(low3, scale_factor=2, mode="nearest"))

    def forward(self, image):
        inter = self._convolution(image, "pre.0", 7, 2)
        inter = self._residual(inter, "pre.1", 2)
        outs = []
        for i in range(self.nstack):
            kp = self._kp(inter, "kps.%d" % i, N, MODULES)
            cnv = self._convolution(kp, "cnvs.%d" % i, 3)
            out = {}
            for head in self.heads:
                y = self._convolution(cnv, "%s.%d.0" % (head, i), 3, with_bn=False)
                out[head] = self._conv(y, "%s.%d.1" % (head, i))
            outs.append(out)
            if i < self.nstack - 1:
                inter = self.q(self._bn(self._conv(inter, "inters_.%d.0" % i), "inters_.%d.1" % i)) + \
                    self._bn(self._conv(cnv, "cnvs_.%d.0" % i), "cnvs_.%d.1" % i)
                inter = self._residual(self.q(F.relu(inter)), "inters.%d" % i)
        return outs

    __call__ = forward
