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
    def __init__(self, state_dict, heads, use_dcn, down_ratio=4, last_level=5, acc_dtype=None, emulate_bf16=False, emulate=None,
                 plan_parts=None):
        """emulate ('bf16' | 'f16'; emulate_bf16=True is 'bf16'): round every conv / DeformConv input and every conv weight
        to that type (fp32 accumulation, as the MFMA does).  NOT a model of the GPU kernels' exact rounding points (they
        fold BatchNorm into the weights before rounding and keep DeformConv filters in fp16): an independent low-precision
        evaluation of the same graph whose distance from the fp32 result says how much of the GPU's error is the
        arithmetic's, not a bug's (tests/test_gpu_fullsize.py)."""
        self.sd = {k: (v if torch.is_tensor(v) else torch.from_numpy(np.asarray(v)))
                   for k, v in state_dict.items()}
        emulate = "bf16" if emulate_bf16 else emulate
        # emulate = 'bf16_plan' | 'f16_plan' (round 5): the rounding points of the GPU's 2-byte launch plans (h3d_amd/engine.py), one
        # by one -- eval BatchNorm folded into the filters in fp64 BEFORE they are rounded (PackedWeights._fold), every activation
        # the plan STORES rounded to the plan's type (so a residual / skip operand is a rounded tensor, not the fp32 value the
        # 'bf16' mode adds), DeformConv filters and samples in fp16 with the blend's four fp16 roundings (csrc/dcn_traits.h), the
        # `node` DeformConvs' input stored as fp16 (engine.node_f16), fp32 accumulation and fp32 bias / residual / ReLU epilogues.
        # What it does not model is the ORDER of the fp32 accumulation (tile shapes), i.e. which way a value that lands within 1e-7
        # of a rounding boundary goes.
        self.plan = emulate in ("bf16_plan", "f16_plan")
        td = {None: None, "bf16": torch.bfloat16, "f16": torch.float16, "bf16_plan": torch.bfloat16, "f16_plan": torch.float16}[emulate]
        self.q = (lambda t: t.to(td).float()) if td is not None else (lambda t: t)
        # plan_parts (analysis aid, tools/hm_tail.py): which of the plan's rounding points are modelled -- "fold" (BatchNorm folded
        # before the filters are rounded), "store" (stored activations rounded: residual / skip operands), "dcn16" (fp16 DeformConv
        # filters, samples and blend), "node16" (fp16 node inputs); the rest falls back to the 'bf16' mode's treatment
        self.parts = set(plan_parts) if plan_parts is not None else {"fold", "store", "dcn16", "node16"}
        ident = lambda t: t
        self.st = self.q if (self.plan and "store" in self.parts) else ident          # rounding of a STORED activation (plan modes)
        self.st_node = ((lambda t: t.clamp(-65504.0, 65504.0).half().float()) if "node16" in self.parts else self.st) if self.plan else ident   # upadd -> fp16 for a node DeformConv
        if td is not None and not self.plan:
            self.sd = {k: (self.q(v) if v.dim() == 4 else v) for k, v in self.sd.items()}
        self._folded = {}
        self.heads = heads
        self.use_dcn = use_dcn
        self.first_level = int(np.log2(down_ratio))
        self.last_level = last_level
        self.acc_dtype = acc_dtype

    # -- primitives ---------------------------------------------------------------------------
    def _conv(self, x, key, stride=1, padding=0):
        if self.plan:       # (no BatchNorm behind it: the heads; x is a stored tensor, already rounded)
            return F.conv2d(self.q(x), self.q(self.sd[key + ".weight"]), self.sd.get(key + ".bias"), stride, padding)
        return F.conv2d(self.q(x), self.sd[key + ".weight"], self.sd.get(key + ".bias"), stride, padding)

    def _fold(self, wkey, bkey, bn):
        """conv (+bias) followed by eval BatchNorm `bn` -> (w', b') as engine.PackedWeights._fold computes them (fp64)."""
        k = (wkey, bn)
        if k not in self._folded:
            sd = self.sd
            w = sd[wkey]
            b = sd[bkey] if bkey is not None and bkey in sd else torch.zeros(w.shape[0])
            scale = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + BN_EPS)
            self._folded[k] = ((w.double() * scale.view(-1, 1, 1, 1)).float(),
                               ((b.double() - sd[bn + ".running_mean"].double()) * scale + sd[bn + ".bias"].double()).float())
        return self._folded[k]

    def _cb(self, x, key, bn, stride=1, padding=0):
        """BatchNorm(conv(x)): plan modes run the folded, rounded filters on the stored (rounded) input."""
        if self.plan and "fold" in self.parts:
            w, b = self._fold(key + ".weight", key + ".bias", bn)
            return F.conv2d(self.q(x), self.q(w), b, stride, padding)       # (q(x) == x when activations are stored rounded)
        return self._bn(self._conv(x, key, stride, padding), bn)

    def _bn(self, x, key):
        sd = self.sd
        return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"],
                            sd[key + ".weight"], sd[key + ".bias"], False, 0.0, BN_EPS)

    # -- backbone -----------------------------------------------------------------------------
    def _block(self, x, p, stride, residual=None):
        if residual is None:
            residual = x
        y = self.st(F.relu(self._cb(x, p + ".conv1", p + ".bn1", stride, 1)))
        y = self._cb(y, p + ".conv2", p + ".bn2", 1, 1)
        return self.st(F.relu(y + residual))

    def _root(self, xs, p):
        y = self._cb(torch.cat(xs, 1), p + ".conv", p + ".bn")
        return self.st(F.relu(y))              # residual_root=False for dla34

    def _tree(self, x, p, levels, cin, cout, stride, level_root, children=None):
        children = [] if children is None else children
        bottom = F.max_pool2d(x, stride, stride) if stride > 1 else x
        if cin != cout:
            residual = self.st(self._cb(bottom, p + ".project.0", p + ".project.1"))
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
        if self.plan:
            x = self.q(x)                      # (the stem kernels convert the fp32 image to the plan's type while staging it)
        x = self.st(F.relu(self._cb(x, "base.base_layer.0", "base.base_layer.1", 1, 3)))
        x = self.st(F.relu(self._cb(x, "base.level0.0", "base.level0.1", 1, 1)))
        y.append(x)
        x = self.st(F.relu(self._cb(x, "base.level1.0", "base.level1.1", 2, 1)))
        y.append(x)
        for lv in range(2, 6):
            x = self._tree(x, "base.level%d" % lv, LEVELS[lv], CHANNELS[lv - 1], CHANNELS[lv], 2,
                           level_root=(lv > 2))
            y.append(x)
        return y

    # -- neck ---------------------------------------------------------------------------------
    def _deform_conv(self, x, p):
        if self.use_dcn and self.plan and "dcn16" in self.parts:
            # csrc/dcn3.hip: fp16 filters (BatchNorm folded into the main ones first), the stored input converted to fp16 (exact for
            # bf16 below 65504), offsets / mask logits from the fp16 offset filters with fp32 accumulation, fp16 blend, f16 MFMA with
            # fp32 accumulation, fp32 bias + ReLU, result stored in the plan's type
            sd = self.sd
            w, b = self._fold(p + ".conv.weight", p + ".conv.bias", p + ".actf.0")
            h = lambda t: t.clamp(-65504.0, 65504.0).half().float()
            y = _dcn.dcn_module_forward(h(x), h(w), b, h(sd[p + ".conv.conv_offset_mask.weight"]), sd[p + ".conv.conv_offset_mask.bias"],
                                        acc_dtype=self.acc_dtype, blend="f16")
            return self.st(F.relu(y))
        if self.use_dcn:
            sd = self.sd
            x = self.q(x)
            qw = self.q if self.plan else (lambda t: t)          # (plan modes keep `sd` unrounded: filters are rounded where they are used)
            y = _dcn.dcn_module_forward(x, qw(sd[p + ".conv.weight"]), sd[p + ".conv.bias"],
                                        qw(sd[p + ".conv.conv_offset_mask.weight"]),
                                        sd[p + ".conv.conv_offset_mask.bias"],
                                        acc_dtype=self.acc_dtype)
        elif self.plan and "fold" in self.parts:
            return self.st(F.relu(self._cb(x, p + ".conv", p + ".actf.0", 1, 1)))
        else:
            y = self._conv(x, p + ".conv", 1, 1)
        return self.st(F.relu(self._bn(y, p + ".actf.0")))

    def _ida_up(self, layers, p, startp, endp):
        for i in range(startp + 1, endp):
            k = i - startp
            w = self.sd["%s.up_%d.weight" % (p, k)]
            f = w.shape[2] // 2
            y = self._deform_conv(layers[i], "%s.proj_%d" % (p, k))
            y = F.conv_transpose2d(self.q(y), w, None, stride=f, padding=f // 2, groups=w.shape[0])
            # (plan modes: the sum is STORED -- as fp16 when only a node DeformConv of a bf16 plan reads it, engine.node_f16)
            s = y + layers[i - 1]
            s = (self.st_node(s) if self.use_dcn else self.st(s)) if self.plan else s
            layers[i] = self._deform_conv(s, "%s.node_%d" % (p, k))

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
            h = self.st(F.relu(self._conv(y[-1], head + ".0", 1, 1)))
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
