"""TEST INFRASTRUCTURE ONLY -- torch-CPU functional restatement of ResNet-101-DCN (CenterNet `resnet_dcn.py` PoseResNet)
driven by a state_dict with the published key names.  DCN layers use oracle.dcn (the restatement of the reference's
own DCNv2/dcn_v2.py:118-128 + dcn_v2_im2col_cuda.cu).

PARITY UNPINNED: the reference repository names `resdcn_101` (src/lib/opts.py:61-63, experiments/ctdet_coco_resdcn101.sh:3)
but holds no source, test or checkpoint for it; this follows the published definition: conv7x7/2 + BN + ReLU, maxpool 3x3/2,
bottleneck layers [3, 4, 23, 3] (stride on the 3x3 conv), three (DCN 3x3 -> BN -> ReLU -> ConvTranspose2d 4x4/2 -> BN -> ReLU)
stages, heads conv3x3 + ReLU + conv1x1.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import dcn as _dcn

BN_EPS = 1e-5
LAYERS = (3, 4, 23, 3)
DECONV = (256, 128, 64)


class ResDCNOracle:
    def __init__(self, state_dict, heads, emulate_bf16=False, emulate=None):
        """emulate ('bf16' | 'f16'; emulate_bf16=True is 'bf16'): conv weights and every STORED activation (the output of each
        conv + BN (+ residual) + ReLU unit and of each down-sample conv) rounded to that type, fp32 accumulation -- the
        rounding points of a plan whose activations live in HBM as 2-byte elements (see oracle/hourglass.py)."""
        self.sd = {k: (v if torch.is_tensor(v) else torch.from_numpy(np.asarray(v))) for k, v in state_dict.items()}
        self.heads = heads
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

    def _bottleneck(self, x, p, stride):
        y = F.relu(self._bn(self._conv(x, p + ".conv1"), p + ".bn1"))
        y = F.relu(self._bn(self._conv(y, p + ".conv2", stride, 1), p + ".bn2"))
        y = self._bn(self._conv(y, p + ".conv3"), p + ".bn3")
        if (p + ".downsample.0.weight") in self.sd:
            x = self.q(self._bn(self._conv(x, p + ".downsample.0", stride), p + ".downsample.1"))
        return self.q(F.relu(y + x))

    def forward(self, x):
        sd = self.sd
        x = self.q(F.relu(self._bn(self._conv(x, "conv1", 2, 3), "bn1")))
        x = F.max_pool2d(x, 3, 2, 1)
        for li, n in enumerate(LAYERS, start=1):
            for b in range(n):
                x = self._bottleneck(x, "layer%d.%d" % (li, b), 2 if (b == 0 and li > 1) else 1)
        for i in range(len(DECONV)):
            p = "deconv_layers.%d" % (6 * i)
            x = _dcn.dcn_module_forward(self.q(x), sd[p + ".weight"], sd[p + ".bias"], sd[p + ".conv_offset_mask.weight"],
                                        sd[p + ".conv_offset_mask.bias"])
            x = F.relu(self._bn(x, "deconv_layers.%d" % (6 * i + 1)))
            x = F.conv_transpose2d(self.q(x), sd["deconv_layers.%d.weight" % (6 * i + 3)], None, stride=2, padding=1)
            x = F.relu(self._bn(x, "deconv_layers.%d" % (6 * i + 4)))
        self.feat = x
        out = {}
        for head in self.heads:
            y = F.relu(self._conv(x, head + ".0", 1, 1))
            out[head] = self._conv(y, head + ".2")
        return [out]

    __call__ = forward
