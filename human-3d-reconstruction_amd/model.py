"""Host mirror of the reference's model factory `dla_net` (models/model.py:501-516).

`dla_net(heads, num_layers=34, head_conv=256, down_ratio=4, not_use_dcn=False)` returns an
nn.Module whose parameters/buffers carry the reference's state_dict names (so
`load_state_dict(torch.load(ckpt)['state_dict'])` works, trains/trainer.py:475-509) and whose
`forward(x)` returns `[{head: [B,C,H/4,W/4] fp32}]` like `DLASeg.forward` (model.py:475-489).
The forward itself is the HIP engine (engine.py): the module tree only holds parameters.
"""
import math

import torch
from torch import nn

from . import arch, arch_hg, arch_res
from .engine import DLAEngine


class _Holder(nn.Module):
    """Anonymous node of the parameter tree (names come from the dotted state_dict keys)."""


def _init_tensor(key, shape, heads):
    """Default initialisation of the reference modules (nn.Conv2d/BatchNorm2d defaults,
    DCNv2.reset_parameters dcn_v2.py:75-81, init_offset dcn_v2.py:114-116, fill_up_weights
    model.py:334-343, head biases model.py:461-464)."""
    leaf = key.rsplit(".", 1)[-1]
    parts = key.split(".")
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.long)
    if leaf == "running_mean":
        return torch.zeros(shape)
    if leaf == "running_var":
        return torch.ones(shape)
    if "conv_offset_mask" in parts:
        return torch.zeros(shape)
    if len(shape) == 1:
        if leaf == "weight":
            return torch.ones(shape)                       # BN gamma
        if parts[0] in heads:                              # head conv biases
            if "hm" in parts[0]:
                if parts[-2] == "2" or len(parts) == 2:
                    return torch.full(shape, -2.19)
                return None                                # default conv bias init, set by caller
            return torch.zeros(shape)                      # fill_fc_weights
        if len(parts) >= 2 and parts[-2] == "conv" and "proj_" not in key and "node_" not in key:
            return torch.zeros(shape)
        return None if parts[-2] == "conv" else torch.zeros(shape)   # BN beta = 0; conv bias by caller
    if len(shape) == 4 and parts[-2].startswith("up_"):
        k = shape[2]
        f = math.ceil(k / 2)
        c = (2 * f - 1 - f % 2) / (2.0 * f)
        w = torch.zeros(shape)
        for i in range(k):
            for j in range(k):
                w[:, 0, i, j] = (1 - math.fabs(i / f - c)) * (1 - math.fabs(j / f - c))
        return w
    return None


class DLASeg(nn.Module):
    arch_name = "dla34"

    def _shapes(self):
        return arch.state_dict_shapes(self.heads, self.use_dcn, self.head_conv)

    def __init__(self, heads, head_conv=256, use_dcn=True, dtype="bf16"):
        super().__init__()
        self.heads = dict(heads)
        self.head_conv = head_conv
        self.use_dcn = use_dcn
        self.compute_dtype = dtype
        self._engine = None
        shapes = self._shapes()
        pending_bias = {}
        for key, shape in shapes.items():
            t = _init_tensor(key, shape, self.heads)
            parts = key.split(".")
            if t is None and len(shape) == 4:
                fan_in = shape[1] * shape[2] * shape[3]
                bound = 1.0 / math.sqrt(fan_in)            # kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in))
                t = torch.empty(shape).uniform_(-bound, bound)
                pending_bias[".".join(parts[:-1])] = bound
            elif t is None:
                bound = pending_bias.get(".".join(parts[:-1]), 0.0)
                if self.use_dcn and parts[-2] == "conv" and ("proj_" in key or "node_" in key):
                    t = torch.zeros(shape)                 # DCNv2.reset_parameters: bias = 0
                else:
                    t = torch.empty(shape).uniform_(-bound, bound)
            mod = self
            for name in parts[:-1]:
                if name not in mod._modules:
                    mod.add_module(name, _Holder())
                mod = mod._modules[name]
            if parts[-1] in ("running_mean", "running_var", "num_batches_tracked"):
                mod.register_buffer(parts[-1], t)
            else:
                mod.register_parameter(parts[-1], nn.Parameter(t))
        self.eval()

    # any change to the parameters invalidates the packed device weights
    def load_state_dict(self, *a, **kw):
        self._engine = None
        return super().load_state_dict(*a, **kw)

    def _apply(self, fn, *a, **kw):
        self._engine = None
        return super()._apply(fn, *a, **kw)

    def set_compute_dtype(self, dtype):
        self.compute_dtype = dtype
        self._engine = None
        return self

    def engine(self, device):
        if self._engine is None or self._engine.device != torch.device(device):
            sd = {k: v for k, v in self.state_dict().items()}
            self._engine = DLAEngine(sd, self.heads, self.use_dcn, self.compute_dtype, device, self.head_conv, self.arch_name)
        return self._engine

    def forward(self, x, slot=0):
        """slot (extension): which copy of the launch plan's buffers to use; batches on different HIP streams must use
        different slots (the head tensors returned are the plan's output buffers)."""
        if self.training:
            raise RuntimeError("h3d_amd.DLASeg is inference-only (BatchNorm is folded): call .eval()")
        if not x.is_cuda:
            raise RuntimeError("Not implemented on the CPU")
        with torch.no_grad():
            out = self.engine(x.device).forward(x, slot)
            return self._wrap(x, out, slot)

    def _wrap(self, x, out, slot=0):
        return [dict(out)]


def dla_net(heads, num_layers=34, head_conv=256, down_ratio=4, not_use_dcn=False, dtype="bf16"):
    """Same signature as the reference factory (model.py:501-516) plus `dtype`
    ('bf16' throughput mode, 'f16' fp16 activations, 'f32' parity mode, 'f16x3' the parity arithmetic on the fp16 matrix cores:
    fp32 storage, three fp16 MFMAs on split operands per fp32 product -- h3d_amd.detector.Opt)."""
    if num_layers != 34:
        raise ValueError("only dla34 exists in the reference (model.py:309-315)")
    if down_ratio != 4:
        raise ValueError("down_ratio %d: the multi_pose path uses 4 (opts.py)" % down_ratio)
    print("==> Use DeformConv." if not not_use_dcn else "==> Do not use DeformConv.")
    return DLASeg(heads, head_conv=head_conv, use_dcn=not not_use_dcn, dtype=dtype)


class HourglassNet(DLASeg):
    """Hourglass-104 (`exkp`, two stacks) with the published CenterNet state_dict names (arch_hg.py).  `forward(x)` returns
    one head dict per stack, like the published `exkp.forward`; inference uses the last one (`model(x)[-1]`)."""
    arch_name = "hourglass"

    def _shapes(self):
        return arch_hg.state_dict_shapes(self.heads)

    def _wrap(self, x, out, slot=0):
        B, _, H, W = x.shape
        plan = self.engine(x.device).plan(B, H, W, slot)
        return [dict(o) for o in plan.all_outputs]


def hourglass_net(heads, num_stacks=2, dtype="bf16"):
    """`get_large_hourglass_net(num_layers, heads, head_conv)` of the published CenterNet code (the reference parses
    `--arch hourglass`, opts.py:61-63, but has no such factory): Hourglass-104, two stacks, head_conv = 256 built in."""
    if num_stacks != 2:
        raise ValueError("Hourglass-104 as published has 2 stacks")
    return HourglassNet(heads, head_conv=256, use_dcn=False, dtype=dtype)


class PoseResNetDCN(DLASeg):
    """ResNet-101-DCN (`resnet_dcn.py` PoseResNet of the published CenterNet code, arch_res.py): same head dict output as
    DLASeg.forward."""
    arch_name = "resdcn101"

    def _shapes(self):
        return arch_res.state_dict_shapes(self.heads, self.head_conv)


def resdcn_net(heads, num_layers=101, head_conv=64, dtype="bf16"):
    """`get_pose_net(num_layers, heads, head_conv)` of the published `resnet_dcn.py` (`--arch resdcn_101`,
    experiments/ctdet_coco_resdcn101.sh:3); head_conv is 64 for every non-DLA arch there."""
    if num_layers != 101:
        raise ValueError("resdcn_%d: only the 101-layer variant named by BASELINE configs[4] is built" % num_layers)
    return PoseResNetDCN(heads, head_conv=head_conv, use_dcn=True, dtype=dtype)


def create_model(arch_name, heads, head_conv=256, not_use_dcn=False, dtype="bf16"):
    """`--arch` dispatch (opts.py:61-63: 'dla_34 | hourglass | resdcn_101'; the reference ignores it and always builds
    dla_net, trains/trainer.py:165)."""
    name = arch_name.replace("-", "_")
    if name in ("dla_34", "dla34"):
        return dla_net(heads, 34, head_conv, 4, not_use_dcn, dtype=dtype)
    if name in ("hourglass", "hourglass_104", "hg"):
        return hourglass_net(heads, 2, dtype=dtype)
    if name in ("resdcn_101", "resdcn101"):
        return resdcn_net(heads, 101, 64 if head_conv in (256, -1) else head_conv, dtype=dtype)
    raise ValueError("arch %r not supported (dla_34 | hourglass | resdcn_101)" % arch_name)
