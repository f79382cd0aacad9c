"""TEST INFRASTRUCTURE ONLY -- torch-CPU functional restatement of the reference's
DLA-34 / DLAUp / IDAUp / heads forward, driven by a reference-format state_dict.

Reference lines followed (all /root/reference/src/lib/models/model.py):
  BasicBlock.forward 46-60 | Root.forward 158-166 | Tree.forward 209-222 | DLA.forward 286-292
  dla34 config 309-312 | DeformConv 346-362 | IDAUp 365-390 | DLAUp 393-415 | DLASeg.forward 475-489
DCN layers use oracle.dcn (DCNv2/dcn_v2.py:118-128).

Pinned by tests/golden/dla34_plain_*.npz = outputs of the imported reference model
(plain-conv variant, the only variant the reference can run on a CPU) on synthetic weights.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import dcn as _dcn

LEVELS = [1, 1, 1, 2, 2, 1]
CHANNELS = [16, 32, 64, 128, 256, 512]
BN_EPS = 1e-5


class DLAOracle:
    def __init__(self, state_dict, heads, use_dcn, down_ratio=4, last_level=5, acc_dtype=None, emulate_bf16=False, emulate=None):
        """emulate ('bf16' | 'f16'; emulate_bf16=True is 'bf16'): round every conv / DeformConv input and every conv weight
        to that type (fp32 accumulation, as the MFMA does).  NOT a model of the GPU kernels' exact rounding points (they
        fold BatchNorm into the weights before rounding and keep DeformConv filters in fp16): an independent low-precision
        evaluation of the same graph whose distance from the fp32 result says how much of the GPU's error is the
        arithmetic's, not a bug's (tests/test_gpu_fullsize.py)."""
        self.sd = {k: (v if torch.is_tensor(v) else torch.from_numpy(np.asarray(v)))
                   for k, v in state_dict.items()}
        emulate = "bf16" if emulate_bf16 else emulate
        td = {None: None, "bf16": torch.bfloat16, "f16": torch.float16}[emulate]
        self.q = (lambda t: t.to(td).float()) if td is not None else (lambda t: t)
        if td is not None:
            self.sd = {k: (self.q(v) if v.dim() == 4 else v) for k, v in self.sd.items()}
        self.heads = heads
        self.use_dcn = use_dcn
        self.first_level = int(np.log2(down_ratio))
        self.last_level = last_level
        self.acc_dtype = acc_dtype

    # -- primitives ---------------------------------------------------------------------------
    def _conv(self, x, key, stride=1, padding=0):
        return F.conv2d(self.q(x), self.sd[key + ".weight"], self.sd.get(key + ".bias"), stride, padding)

    def _bn(self, x, key):
        sd = self.sd
        return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"],
                            sd[key + ".weight"], sd[key + ".bias"], False, 0.0, BN_EPS)

    # -- backbone -----------------------------------------------------------------------------
    def _block(self, x, p, stride, residual=None):
        if residual is None:
            residual = x
        y = F.relu(self._bn(self._conv(x, p + ".conv1", stride, 1), p + ".bn1"))
        y = self._bn(self._conv(y, p + ".conv2", 1, 1), p + ".bn2")
        return F.relu(y + residual)

    def _root(self, xs, p):
        y = self._bn(self._conv(torch.cat(xs, 1), p + ".conv"), p + ".bn")
        return F.relu(y)                       # residual_root=False for dla34

    def _tree(self, x, p, levels, cin, cout, stride, level_root, children=None):
        children = [] if children is None else children
        bottom = F.max_pool2d(x, stride, stride) if stride > 1 else x
        if cin != cout:
            residual = self._bn(self._conv(bottom, p + ".project.0"), p + ".project.1")
        else:
            residual = bottom
        if level_root:
            children.append(bottom)
        if levels == 1:
            x1 = self._block(x, p + ".tree1", stride, residual)
            x2 = self._block(x1, p + ".tree2", 1)
            return self._root([x2, x1] + children, p + ".root")
        # levels > 1: the inner tree recomputes its own residual (model.py:212), the outer
        # `residual` is passed but overwritten there -- same result as the reference.
        x1 = self._tree(x, p + ".tree1", levels - 1, cin, cout, stride, False)
        children.append(x1)
        return self._tree(x1, p + ".tree2", levels - 1, cout, cout, 1, False, children)

    def base(self, x):
        y = []
        x = F.relu(self._bn(self._conv(x, "base.base_layer.0", 1, 3), "base.base_layer.1"))
        x = F.relu(self._bn(self._conv(x, "base.level0.0", 1, 1), "base.level0.1"))
        y.append(x)
        x = F.relu(self._bn(self._conv(x, "base.level1.0", 2, 1), "base.level1.1"))
        y.append(x)
        for lv in range(2, 6):
            x = self._tree(x, "base.level%d" % lv, LEVELS[lv], CHANNELS[lv - 1], CHANNELS[lv], 2,
                           level_root=(lv > 2))
            y.append(x)
        return y

    # -- neck ---------------------------------------------------------------------------------
    def _deform_conv(self, x, p):
        if self.use_dcn:
            sd = self.sd
            x = self.q(x)
            y = _dcn.dcn_module_forward(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"],
                                        sd[p + ".conv.conv_offset_mask.weight"],
                                        sd[p + ".conv.conv_offset_mask.bias"],
                                        acc_dtype=self.acc_dtype)
        else:
            y = self._conv(x, p + ".conv", 1, 1)
        return F.relu(self._bn(y, p + ".actf.0"))

    def _ida_up(self, layers, p, startp, endp):
        for i in range(startp + 1, endp):
            k = i - startp
            w = self.sd["%s.up_%d.weight" % (p, k)]
            f = w.shape[2] // 2
            y = self._deform_conv(layers[i], "%s.proj_%d" % (p, k))
            y = F.conv_transpose2d(self.q(y), w, None, stride=f, padding=f // 2, groups=w.shape[0])
            layers[i] = self._deform_conv(y + layers[i - 1], "%s.node_%d" % (p, k))

    def forward(self, x):
        layers = self.base(x)
        # DLAUp (model.py:409-415): ida_i works on layers[len-i-2 : len]
        out = [layers[-1]]
        for i in range(len(layers) - self.first_level - 1):
            self._ida_up(layers, "dla_up.ida_%d" % i, len(layers) - i - 2, len(layers))
            out.insert(0, layers[-1])
        y = [out[i].clone() for i in range(self.last_level - self.first_level)]
        self._ida_up(y, "ida_up", 0, len(y))
        z = {}
        for head in self.heads:
            h = F.relu(self._conv(y[-1], head + ".0", 1, 1))
            z[head] = self._conv(h, head + ".2")
        self.feat = y[-1]
        return [z]

    __call__ = forward


def state_dict_shapes(heads, use_dcn, head_conv=256):
    """{key: shape} of the reference model's state_dict, built without importing the reference
    (key format: SURVEY 5 'Checkpoint / resume'; verified against the imported reference's
    state_dict in tests/test_oracle_golden.py)."""
    shapes = {}

    def bn(p, c):
        shapes[p + ".weight"] = (c,)
        shapes[p + ".bias"] = (c,)
        shapes[p + ".running_mean"] = (c,)
        shapes[p + ".running_var"] = (c,)
        shapes[p + ".num_batches_tracked"] = ()

    def block(p, cin, cout):
        shapes[p + ".conv1.weight"] = (cout, cin, 3, 3)
        bn(p + ".bn1", cout)
        shapes[p + ".conv2.weight"] = (cout, cout, 3, 3)
        bn(p + ".bn2", cout)

    def tree(p, levels, cin, cout, level_root, root_dim=0):
        if root_dim == 0:
            root_dim = 2 * cout
        if level_root:
            root_dim += cin
        if levels == 1:
            block(p + ".tree1", cin, cout)
            block(p + ".tree2", cout, cout)
            shapes[p + ".root.conv.weight"] = (cout, root_dim, 1, 1)
            bn(p + ".root.bn", cout)
        else:
            tree(p + ".tree1", levels - 1, cin, cout, False, 0)
            tree(p + ".tree2", levels - 1, cout, cout, False, root_dim + cout)
        if cin != cout:
            shapes[p + ".project.0.weight"] = (cout, cin, 1, 1)
            bn(p + ".project.1", cout)

    shapes["base.base_layer.0.weight"] = (16, 3, 7, 7)
    bn("base.base_layer.1", 16)
    shapes["base.level0.0.weight"] = (16, 16, 3, 3)
    bn("base.level0.1", 16)
    shapes["base.level1.0.weight"] = (32, 16, 3, 3)
    bn("base.level1.1", 32)
    for lv in range(2, 6):
        tree("base.level%d" % lv, LEVELS[lv], CHANNELS[lv - 1], CHANNELS[lv], lv > 2)

    def deform(p, cin, cout):
        bn(p + ".actf.0", cout)
        shapes[p + ".conv.weight"] = (cout, cin, 3, 3)
        shapes[p + ".conv.bias"] = (cout,)
        if use_dcn:
            shapes[p + ".conv.conv_offset_mask.weight"] = (27, cin, 3, 3)
            shapes[p + ".conv.conv_offset_mask.bias"] = (27,)

    def ida(p, o, chans, ups):
        for i in range(1, len(chans)):
            deform("%s.proj_%d" % (p, i), chans[i], o)
            f = int(ups[i])
            shapes["%s.up_%d.weight" % (p, i)] = (o, 1, 2 * f, 2 * f)
            deform("%s.node_%d" % (p, i), o, o)

    ch = CHANNELS[2:]
    # DLAUp(startp=2, channels=[64,128,256,512], scales=[1,2,4,8])  (model.py:393-407)
    in_ch = list(ch)
    scales = [1, 2, 4, 8]
    for i in range(len(ch) - 1):
        j = -i - 2
        ida("dla_up.ida_%d" % i, ch[j], in_ch[j:], [s // scales[j] for s in scales[j:]])
        scales[j + 1:] = [scales[j]] * len(scales[j + 1:])
        in_ch[j + 1:] = [ch[j]] * len(in_ch[j + 1:])
    ida("ida_up", 64, ch[0:3], [1, 2, 4])
    for head, c in heads.items():
        shapes[head + ".0.weight"] = (head_conv, 64, 3, 3)
        shapes[head + ".0.bias"] = (head_conv,)
        shapes[head + ".2.weight"] = (c, head_conv, 1, 1)
        shapes[head + ".2.bias"] = (c,)
    return shapes
