"""Launch-plan builder and executor for the DLA-34 (+DCNv2) multi_pose network.

Host-side counterpart of the reference's `DLASeg.forward` (models/model.py:475-489): instead of
walking nn.Modules per call, the network is lowered ONCE per (batch, height, width, dtype) into
an array of `h3d_op` descriptors (include/h3d.h) over pre-allocated NHWC buffers and pre-packed
weights; a forward is a single `h3d_run_ops` call on the current stream.

Lowering decisions (DESIGN.md 'Data layout'):
  * activations NHWC (channels-last) bf16 (throughput) or fp32 (parity mode);
  * eval-mode BatchNorm folded into the preceding conv / DCN weights and bias;
  * Root's torch.cat (model.py:160) is free: producers write into channel slices of one buffer;
  * residual add + ReLU are conv epilogues; IDAUp's depthwise deconv + skip add is one kernel;
  * conv_offset_mask (dcn_v2.py:119-122) writes NHWC fp32 offsets/mask-logits consumed in place
    by the DCN kernel (no chunk/cat/sigmoid passes);
  * the `project` conv of the two-level trees (level3/level4) is dead in the reference
    (model.py:212 recomputes the residual inside tree1) and is not lowered;
  * head outputs are written directly as contiguous NCHW fp32, the reference's head layout.
"""
import ctypes

import numpy as np
import torch

from . import _lib, arch, arch_hg, arch_res
from ._lib import H3dOp

_TORCH_DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "f16x3": torch.float32}
_H3D_DT = {"bf16": _lib.H3D_BF16, "f16": _lib.H3D_F16, "f32": _lib.H3D_F32, "f16x3": _lib.H3D_F16X3}
DCN_F16IN = 0x40000         # h3d_op.reserved of a fused DeformConv in a bf16 plan: its input tensor holds fp16 values (csrc/dcn3.hip F16IN)
LOWP = ("bf16", "f16")      # the 2-byte plans: same kernels, lowering and tile choices; "f16" = BASELINE configs[4]'s arithmetic


def _t(v):
    return v.detach().float().cpu() if torch.is_tensor(v) else torch.from_numpy(np.asarray(v)).float()


def x3_exp(w):
    """Exponent e of the power-of-two pre-scale of an "f16x3" filter bank: 2^e * max|w| lands in [2^13, 2^14), so the fp16 hi terms stay
    far below 65504 and the lo terms (~2^-12 of the value) of every filter above 2^-16 of the largest are NORMAL fp16 numbers -- an
    unscaled 0.05 has a subnormal lo term (3e-8 absolute = 2^-20.7 relative, five times the 2^-23 of the split itself).  The kernels
    multiply their accumulators by 2^-e, which is exact (h3d_op.wexp)."""
    m = float(w.abs().max())
    if not (m > 0.0) or not np.isfinite(m):
        return 0
    return int(max(-60, min(60, 13 - int(np.floor(np.log2(m))))))


def x3_split(w):
    """fp32 filters [..., K] (K % 8 == 0: the contraction index, 8 consecutive elements = one MFMA fragment of a lane) -> the
    operand format of the "f16x3" plans (csrc/common.h ET<x3_t>): per group of 8 elements the 8 fp16 high terms hi = fp16(x)
    followed by the 8 fp16 low terms lo = fp16(x - hi) (round to nearest even), in the 32 bytes the 8 fp32 values occupied --
    returned as a float32-typed tensor of the same shape (raw bytes, not numbers)."""
    w = w.float().contiguous()
    K = w.shape[-1]
    assert K % 8 == 0, K
    hi = w.to(torch.float16)
    lo = (w - hi.float()).to(torch.float16)
    g = torch.stack([hi.reshape(-1, K // 8, 8), lo.reshape(-1, K // 8, 8)], dim=2)            # [rows, K/8, 2, 8]
    return g.reshape(-1, 2 * K).contiguous().view(torch.float32).reshape(w.shape)


class View:
    """A [B,H,W,C] tensor living at channel offset `coff` of an NHWC buffer of channel stride `cs`."""
    __slots__ = ("buf", "H", "W", "C", "cs", "coff", "es")

    def __init__(self, buf, H, W, C, cs, coff, es):
        self.buf, self.H, self.W, self.C, self.cs, self.coff, self.es = buf, H, W, C, cs, coff, es

    @property
    def ptr(self):
        return self.buf.data_ptr() + self.coff * self.es

    def slice(self, c0, c):
        assert c0 + c <= self.C
        return View(self.buf, self.H, self.W, c, self.cs, self.coff + c0, self.es)


class PackedWeights:
    """BN-folded, re-laid-out weights on the device (built once per state_dict/dtype)."""

    def __init__(self, state_dict, heads, use_dcn, dtype, device, head_conv=256, arch_name="dla34"):
        self.heads, self.use_dcn, self.dtype, self.device = dict(heads), use_dcn, dtype, device
        self.head_conv = head_conv
        self.arch = arch_name
        self.sd = {k: _t(v) for k, v in state_dict.items() if not k.endswith("num_batches_tracked")}
        shapes = (arch_hg.state_dict_shapes(heads) if arch_name == "hourglass" else
                  arch_res.state_dict_shapes(heads, head_conv) if arch_name == "resdcn101" else
                  arch.state_dict_shapes(heads, use_dcn, head_conv))
        missing = [k for k in shapes
                   if not k.endswith("num_batches_tracked") and k not in self.sd]
        if missing:
            raise KeyError("state_dict is missing %d keys, e.g. %s" % (len(missing), missing[:3]))
        self.t = {}
        self.wexp = {}              # f16x3: device pointer of a packed filter bank -> its power-of-two pre-scale exponent (x3_exp; h3d_op.wexp)
        self.dcn_variant = {}       # DeformConv layer (state_dict prefix) -> csrc/dcn3.hip variant bits (DLAEngine.calibrate_dcn_margins)
        if arch_name == "resdcn101":
            # the DCN of up-sampling stage i is `deconv_layers.{6i}` (weight, bias, conv_offset_mask.*) followed by the
            # BatchNorm `deconv_layers.{6i+1}`: alias them to the key pattern the DeformConv lowering reads
            # (`p.conv.*`, `p.actf.0.*` of the DLA neck, model.py:346-362)
            for i in range(len(arch_res.DECONV)):
                p, bn = "deconv_layers.%d" % (6 * i), "deconv_layers.%d" % (6 * i + 1)
                for a, b in ((".conv.weight", ".weight"), (".conv.bias", ".bias"),
                             (".conv.conv_offset_mask.weight", ".conv_offset_mask.weight"),
                             (".conv.conv_offset_mask.bias", ".conv_offset_mask.bias")):
                    self.sd[p + a] = self.sd[p + b]
                for leaf in ("weight", "bias", "running_mean", "running_var"):
                    self.sd["%s.actf.0.%s" % (p, leaf)] = self.sd["%s.%s" % (bn, leaf)]

    @classmethod
    def from_tensors(cls, tensors, dtype, device):
        """A packer over a bare {name: tensor} table (no architecture key check): the stand-alone `DCN` module
        (h3d_amd.dcn_v2.DCN) packs its four parameters with the same `conv` / `offset_conv` routines as the network."""
        self = cls.__new__(cls)
        self.heads, self.use_dcn, self.dtype, self.device = {}, True, dtype, torch.device(device)
        self.head_conv, self.arch = 0, "bare"
        self.sd = {k: _t(v) for k, v in tensors.items()}
        self.t = {}
        self.wexp = {}
        self.dcn_variant = {}
        return self

    @property
    def dcn_wide(self):
        return {p for p, v in self.dcn_variant.items() if v == 0x8000}

    def _fold(self, w, b, bn):
        """conv(+bias) followed by eval BatchNorm `bn` -> (w', b')."""
        if b is None:
            b = torch.zeros(w.shape[0])
        if bn is None:
            return w, b
        sd = self.sd
        scale = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + arch.BN_EPS)
        w2 = (w.double() * scale.view(-1, 1, 1, 1)).float()
        b2 = ((b.double() - sd[bn + ".running_mean"].double()) * scale + sd[bn + ".bias"].double()).float()
        return w2, b2

    def conv(self, wkey, bkey=None, bn=None, pad_cout_to=None, as_half=False):
        """-> (packed weights [rows][kh*kw][Cin], bias fp32 [rows], Cout, Cin, k).
        as_half: store fp16 instead of bf16 (DCN layers in bf16 mode, csrc/dcn2.hip)."""
        key = ("conv", wkey, bn, pad_cout_to, as_half)
        if key not in self.t:
            w, b = self._fold(self.sd[wkey], self.sd[bkey] if bkey else None, bn)
            co, ci, kh, kw = w.shape
            cout = pad_cout_to or co
            rows = ((cout + 127) // 128) * 128
            wp = torch.zeros(rows, kh * kw, ci)
            wp[:co] = w.permute(0, 2, 3, 1).reshape(co, kh * kw, ci)
            bp = torch.zeros(rows)
            bp[:co] = b
            td = torch.float16 if (as_half and self.dtype in LOWP) else _TORCH_DT[self.dtype]
            e = x3_exp(wp) if self.dtype == "f16x3" else 0
            wp = x3_split(wp * 2.0 ** e) if self.dtype == "f16x3" else wp.to(td)
            self.t[key] = (wp.contiguous().to(self.device),
                           bp.contiguous().to(self.device), cout, ci, kh, rows)
            self.wexp[self.t[key][0].data_ptr()] = e
        return self.t[key]

    def conv_stream(self, wkey, bkey=None, bn=None):
        """3x3 filter bank as the stage-major LDS image of csrc/conv2.hip: [Cin/16][G][32 rows][19 slots][8]
        bf16, slot 2*tap+h = input channels 16*stage + 8h..8h+7 of tap `tap`, slot 18 zero; G = Cout/32
        row groups padded to a multiple of 4.  -> (image, bias fp32 [32 G], Cout, Cin, rows = 32 G)."""
        key = ("conv_stream", wkey, bn)
        if key not in self.t:
            w, b = self._fold(self.sd[wkey], self.sd[bkey] if bkey else None, bn)
            co, ci, kh, kw = w.shape
            assert kh == 3 and kw == 3 and ci % 16 == 0
            G = ((co + 127) // 128) * 4
            wp = torch.zeros(G * 32, 9, ci)
            wp[:co] = w.permute(0, 2, 3, 1).reshape(co, 9, ci)
            img = self._stage_image(wp)
            bp = torch.zeros(G * 32)
            bp[:co] = b
            self.t[key] = (img.to(_TORCH_DT[self.dtype]).contiguous().to(self.device), bp.contiguous().to(self.device),
                           co, ci, G * 32)
        return self.t[key]

    @staticmethod
    def _stage_image(wp, ck=16):
        """[rows (multiple of 32)][9][Cin] -> stage-major LDS image [Cin/ck][rows/32][32][9*ck/8 + 1 slots][8]:
        slot tap*ck/8 + j = input channels ck*stage + 8j..8j+7 of tap `tap`, last slot zero (csrc/conv2.hip and
        dcn4.hip: ck = 16; csrc/dcn3.hip WDMA: ck = h3d_dcn_fused_ck)."""
        rows, _, ci = wp.shape
        G, spt = rows // 32, ck // 8
        img = torch.zeros(ci // ck, G, 32, 9 * spt + 1, 8)
        v = wp.reshape(G, 32, 9, ci // ck, spt, 8).permute(3, 0, 1, 2, 4, 5)
        img[:, :, :, :9 * spt] = v.reshape(ci // ck, G, 32, 9 * spt, 8)
        return img

    @staticmethod
    def _stage_image_x3(ws):
        """float32-TYPED split filters [rows (multiple of 32)][9][Cin] (x3_split: per 8 channels 8 hi | 8 lo fp16 terms) -> the stage-major
        LDS image of the f16x3 patch-slot DeformConv (csrc/dcn3.hip WDMA, 16 channels per stage): [Cin/16][rows/32][32][37 slots of 16 B]:
        slot 4 * tap + j = bytes 16 j .. 16 j + 15 of the tap's 64 bytes (channels 16 * stage ... + 15), slot 36 zero -- rows of 592 B."""
        rows, _, ci = ws.shape
        G = rows // 32
        img = torch.zeros(ci // 16, G, 32, 37, 4)
        v = ws.reshape(G, 32, 9, ci // 16, 4, 4).permute(3, 0, 1, 2, 4, 5)
        img[:, :, :, :36] = v.reshape(ci // 16, G, 32, 36, 4)
        return img

    def dcn_stream_x3(self, p):
        """Fused DeformConv `p` for the f16x3 patch-slot variant: (main image, offset image, bias [rows | 32], Cout, Cin, rows); the
        power-of-two pre-scale exponents of the two banks land in `self.wexp` under the images' device pointers."""
        key = ("dcn_stream_x3", p)
        if key not in self.t:
            w, b = self._fold(self.sd[p + ".conv.weight"], self.sd[p + ".conv.bias"], p + ".actf.0")
            co, ci = w.shape[:2]
            rows = ((co + 127) // 128) * 128
            wp = torch.zeros(rows, 9, ci)
            wp[:co] = w.permute(0, 2, 3, 1).reshape(co, 9, ci)
            bp = torch.zeros(rows)
            bp[:co] = b
            wo = torch.zeros(32, 9, ci)
            bo = torch.zeros(32)
            ow, ob = self.sd[p + ".conv.conv_offset_mask.weight"], self.sd[p + ".conv.conv_offset_mask.bias"]
            for tap in range(9):                         # rows permuted as offset_conv() does
                hh, u = (0, tap) if tap < 5 else (1, tap - 5)
                for c, ch in enumerate((2 * tap, 2 * tap + 1, 18 + tap)):
                    i = 3 * u + c
                    row = (i & 3) + 8 * (i >> 2) + 4 * hh
                    wo[row] = ow[ch].permute(1, 2, 0).reshape(9, ci)
                    bo[row] = ob[ch]
            e, eo = x3_exp(wp), x3_exp(wo)
            wimg = self._stage_image_x3(x3_split(wp * 2.0 ** e)).contiguous().to(self.device)
            woimg = self._stage_image_x3(x3_split(wo * 2.0 ** eo)).contiguous().to(self.device)
            self.t[key] = (wimg, woimg, torch.cat([bp, bo]).contiguous().to(self.device), co, ci, rows)
            self.wexp[wimg.data_ptr()], self.wexp[woimg.data_ptr()] = e, eo
        return self.t[key]

    def dcn_stream(self, p, ck=16):
        """Fused DeformConv `p` packed as fp16 stage-major images of the main and the offset/mask filters, `ck`
        channels per stage (csrc/dcn4.hip: 16; csrc/dcn3.hip WDMA: h3d_dcn_fused_ck)
        -> (main image, offset image, bias [rows | 32], Cout, Cin, rows)."""
        key = ("dcn_stream", p, ck)
        if key not in self.t:
            w, b = self._fold(self.sd[p + ".conv.weight"], self.sd[p + ".conv.bias"], p + ".actf.0")
            co, ci = w.shape[:2]
            rows = ((co + 127) // 128) * 128
            wp = torch.zeros(rows, 9, ci)
            wp[:co] = w.permute(0, 2, 3, 1).reshape(co, 9, ci)
            bp = torch.zeros(rows)
            bp[:co] = b
            wo, bo = self.offset_conv(p + ".conv.conv_offset_mask.weight", p + ".conv.conv_offset_mask.bias", rows)
            self.t[key] = (self._stage_image(wp, ck).to(torch.float16).contiguous().to(self.device),
                           self._stage_image(wo[:32].float().cpu(), ck).to(torch.float16).contiguous().to(self.device),
                           torch.cat([bp, bo]).contiguous().to(self.device), co, ci, rows)
        return self.t[key]

    def stem(self):
        key = ("stem",)
        if key not in self.t:
            w, b = self._fold(self.sd["base.base_layer.0.weight"], None, "base.base_layer.1")
            e = 0
            if self.dtype in LOWP or self.dtype == "f16x3":
                # MFMA stem (csrc/conv.hip stem_mfma_kernel / stem_x3_kernel): [16][7 dy][32] with k = dx*4 + c
                wp = torch.zeros(16, 7, 8, 4)
                wp[:, :, :7, :3] = w.permute(0, 2, 3, 1)          # [o][dy][dx][c]
                if self.dtype == "f16x3":                        # fp32 bank times 2^e as (hi | lo) fp16 terms per 8 k
                    e = x3_exp(wp)
                    w = x3_split(wp.reshape(16, 7, 32) * 2.0 ** e)
                else:
                    w = wp.reshape(16, 7, 32).to(_TORCH_DT[self.dtype])
            self.t[key] = (w.contiguous().to(self.device), b.contiguous().to(self.device))
            self.wexp[self.t[key][0].data_ptr()] = e
        return self.t[key]

    def stem_s2(self, wkey, bkey, bn):
        """7x7 stride-2 stem filters [Cout,3,7,7] (+ BatchNorm `bn` folded) for csrc/extra.hip stem_s2_kernel:
        bf16 [Cout][7 dy][32] with k = dx*4 + c (zero for dx = 7 and c = 3), bias fp32 [Cout]."""
        key = ("stem_s2", wkey, bn)
        if key not in self.t:
            w, b = self._fold(self.sd[wkey], self.sd[bkey] if bkey else None, bn)
            co = w.shape[0]
            wp = torch.zeros(co, 7, 8, 4)
            wp[:, :, :7, :3] = w.permute(0, 2, 3, 1)          # [o][dy][dx][c]
            self.t[key] = (wp.reshape(co, 7, 32).to(_TORCH_DT[self.dtype]).contiguous().to(self.device), b.float().contiguous().to(self.device))
        return self.t[key]

    def stem3(self, proj=False):
        """base_layer + level0 + level1 packed for csrc/stem3.hip: bf16 [16][7][32] | [5][16][32] | [32][9][16] and
        fp32 biases [16 | 16 | 32] (BatchNorm folded).  proj: + level2's `project` 1x1 conv (model.py:202-207) [64][32] and its
        bias [64], for the launch that also produces level2's residual branch."""
        key = ("stem3", proj)
        if key not in self.t and proj:
            flat, bias = self.stem3(False)
            wp, bp = self._fold(self.sd["base.level2.project.0.weight"], None, "base.level2.project.1")      # [64,32,1,1]
            assert tuple(wp.shape) == (64, 32, 1, 1)
            self.t[key] = (torch.cat([flat.cpu(), wp.reshape(-1).to(_TORCH_DT[self.dtype])]).contiguous().to(self.device),
                           torch.cat([bias.cpu(), bp.float()]).contiguous().to(self.device))
        if key not in self.t:
            w0, b0 = self.stem()
            w1, b1 = self._fold(self.sd["base.level0.0.weight"], None, "base.level0.1")       # [16,16,3,3]
            w2, b2 = self._fold(self.sd["base.level1.0.weight"], None, "base.level1.1")       # [32,16,3,3]
            t1 = torch.zeros(5, 16, 2, 16)
            w1t = w1.permute(0, 2, 3, 1).reshape(16, 9, 16)                                    # [o][tap][c]
            for tap in range(9):
                t1[tap // 2, :, tap % 2, :] = w1t[:, tap, :]
            w2t = w2.permute(0, 2, 3, 1).reshape(32, 9, 16)
            flat = torch.cat([w0.cpu().float().reshape(-1), t1.reshape(-1), w2t.reshape(-1)]).to(_TORCH_DT[self.dtype])
            bias = torch.cat([b0.cpu().float(), b1.float(), b2.float()])
            self.t[key] = (flat.contiguous().to(self.device), bias.contiguous().to(self.device))
        return self.t[key]

    def stem3_x3(self):
        """base_layer + level0 + level1 for csrc/stem3x.hip (f16x3 plans): the three banks of `stem3()` as float32-typed (hi | lo) fp16
        terms per 8 k, each times its own power of two, and fp32 biases [16 | 16 | 32 | 2^-e0, 2^-e1, 2^-e2, 0]."""
        key = ("stem3_x3",)
        if key not in self.t:
            w0, b0 = self._fold(self.sd["base.base_layer.0.weight"], None, "base.base_layer.1")
            w1, b1 = self._fold(self.sd["base.level0.0.weight"], None, "base.level0.1")       # [16,16,3,3]
            w2, b2 = self._fold(self.sd["base.level1.0.weight"], None, "base.level1.1")       # [32,16,3,3]
            wp = torch.zeros(16, 7, 8, 4)
            wp[:, :, :7, :3] = w0.permute(0, 2, 3, 1)                                           # [o][dy][dx][c], k = dx*4 + c
            t0 = wp.reshape(16, 7, 32)
            t1 = torch.zeros(5, 16, 2, 16)
            w1t = w1.permute(0, 2, 3, 1).reshape(16, 9, 16)                                    # [o][tap][c]
            for tap in range(9):
                t1[tap // 2, :, tap % 2, :] = w1t[:, tap, :]
            t1 = t1.reshape(5, 16, 32)
            t2 = w2.permute(0, 2, 3, 1).reshape(32, 9, 16)
            es = [x3_exp(t) for t in (t0, t1, t2)]
            flat = torch.cat([x3_split(t * 2.0 ** e).reshape(-1) for t, e in zip((t0, t1, t2), es)])
            bias = torch.cat([b0.float(), b1.float(), b2.float(), torch.tensor([2.0 ** -es[0], 2.0 ** -es[1], 2.0 ** -es[2], 0.0])])
            self.t[key] = (flat.contiguous().to(self.device), bias.contiguous().to(self.device))
        return self.t[key]

    def offset_conv(self, wkey, bkey, main_rows):
        """conv_offset_mask packed for the fused DeformConv kernel (csrc/dcn3.hip): the 27 filters are
        spread over 32 MFMA rows so that accumulator half h of a pixel owns whole (dh, dw, mask)
        triples -- value i = 3u + c of half h sits in row (i&3) + 8*(i>>2) + 4*h; half 0 holds taps
        0..4, half 1 taps 5..8.  Returns (weights [128 rows][9][Cin] (32 used), permuted bias fp32 [32])."""
        key = ("offconv", wkey)
        if key not in self.t:
            w, b = self.sd[wkey], self.sd[bkey]              # [27,Cin,3,3], [27]
            ci = w.shape[1]
            wp = torch.zeros(128, 9, ci)
            bp = torch.zeros(32)
            for tap in range(9):
                hh, u = (0, tap) if tap < 5 else (1, tap - 5)
                for c, ch in enumerate((2 * tap, 2 * tap + 1, 18 + tap)):
                    i = 3 * u + c
                    row = (i & 3) + 8 * (i >> 2) + 4 * hh
                    wp[row] = w[ch].permute(1, 2, 0).reshape(9, ci)
                    bp[row] = b[ch]
            td = torch.float16 if self.dtype in LOWP else torch.float32
            e = x3_exp(wp) if self.dtype == "f16x3" else 0
            self.t[key] = ((x3_split(wp * 2.0 ** e) if self.dtype == "f16x3" else wp.to(td)).contiguous().to(self.device), bp)
            self.wexp[self.t[key][0].data_ptr()] = e
        return self.t[key]

    def fused_heads(self, names=None):
        """Fused-heads pack: 3x3 weights of all heads stacked [nheads*head_conv][9][64]; per head the
        1x1 weights as [96 rows][head_conv] with K re-ordered to the MFMA accumulator row order
        (csrc/heads.hip): within every 32-channel group, position h*16 + r holds channel
        (r&3) + 8*(r>>2) + 4*h."""
        names = tuple(self.heads) if names is None else names
        key = ("heads", names)
        if key not in self.t:
            hc = self.head_conv
            td = _TORCH_DT[self.dtype]
            w1, b1, per = [], [], []
            perm = torch.tensor([g * 32 + (r & 3) + 8 * (r >> 2) + 4 * h
                                 for g in range(hc // 32) for h in range(2) for r in range(16)])
            for head in names:
                c = self.heads[head]
                w = self.sd[head + ".0.weight"]                      # [hc,64,3,3]
                w1.append(w.permute(0, 2, 3, 1).reshape(hc, 9, w.shape[1]))
                b1.append(self.sd[head + ".0.bias"])
                w2 = torch.zeros(96, hc)
                w2[:c] = self.sd[head + ".2.weight"].reshape(c, hc)[:, perm]
                b2 = torch.zeros(96)
                b2[:c] = self.sd[head + ".2.bias"]
                e2 = x3_exp(w2) if self.dtype == "f16x3" else 0
                per.append((head, c, (x3_split(w2 * 2.0 ** e2) if self.dtype == "f16x3" else w2.to(td)).contiguous().to(self.device), b2.to(self.device)))
                self.wexp[per[-1][2].data_ptr()] = e2
            w1 = torch.cat(w1)
            e1 = x3_exp(w1) if self.dtype == "f16x3" else 0          # (one exponent for the launch's 3x3 bank; its biases are scaled with it)
            self.t[key] = ((x3_split(w1 * 2.0 ** e1) if self.dtype == "f16x3" else w1.to(td)).contiguous().to(self.device),
                           (torch.cat(b1).float() * 2.0 ** e1).contiguous().to(self.device), per)
            self.wexp[self.t[key][0].data_ptr()] = e1
        return self.t[key]

    def nearest_up_key(self, c):
        """Nearest-neighbour x2 up-sampling (Hourglass `nn.Upsample(scale_factor=2)`) as the depthwise
        ConvTranspose2d(k=4, s=2, p=1) the up-sample + add kernel evaluates: taps (1..2, 1..2) = 1, the rest 0 --
        output row y reads input row (y + 1 - ky) / 2 for ky = 1 (y even) or 2 (y odd), i.e. row y // 2."""
        key = "__nearest_up2__.%d" % c
        if key not in self.sd:
            w = torch.zeros(c, 1, 4, 4)
            w[:, 0, 1:3, 1:3] = 1.0
            self.sd[key] = w
        return key

    def deconv4_as_conv3(self, wkey, bn):
        """ConvTranspose2d(C, C, 4, stride 2, padding 1, bias=False) + BatchNorm `bn` as ONE 3x3 conv with 4C output
        channels followed by H3D_OP_DEPTH2SPACE: output pixel (2y+py, 2x+px) only sees inputs (y+dy, x+dx) with
        dy in {-1, 0} (py = 0) or {0, 1} (py = 1) through kernel row ky = py + 1 - 2 dy, so group g = 2 py + px of the
        3x3 filters is that 2x2 sub-kernel, zero elsewhere (2.25x the transposed conv's MACs, all of them on the MFMA conv
        kernel).  -> (weight key [4C, C, 3, 3], BatchNorm prefix with the statistics repeated per group)."""
        key, bkey = wkey + "#conv3", bn + "#x4"
        if key not in self.sd:
            w = self.sd[wkey]                                    # [Cin, Cout, 4, 4]
            ci, co = w.shape[0], w.shape[1]
            w3 = torch.zeros(4, co, ci, 3, 3)
            for py in range(2):
                for px in range(2):
                    for dy in ((-1, 0) if py == 0 else (0, 1)):
                        for dx in ((-1, 0) if px == 0 else (0, 1)):
                            w3[2 * py + px, :, :, dy + 1, dx + 1] = w[:, :, py + 1 - 2 * dy, px + 1 - 2 * dx].t()
            self.sd[key] = w3.reshape(4 * co, ci, 3, 3)
            for leaf in ("weight", "bias", "running_mean", "running_var"):
                self.sd["%s.%s" % (bkey, leaf)] = self.sd["%s.%s" % (bn, leaf)].repeat(4)
        return key, bkey

    def im2col_key(self, wkey, kpad=160):
        """[Cout,3,7,7] stem filters as a 1x1 conv over H3D_OP_IM2COL patches: [Cout,kpad,1,1], k = c*49 + ky*7 + kx."""
        key = wkey + "#im2col"
        if key not in self.sd:
            w = self.sd[wkey]
            co, k = w.shape[0], w.shape[1] * w.shape[2] * w.shape[3]
            wp = torch.zeros(co, kpad, 1, 1)
            wp[:, :k, 0, 0] = w.reshape(co, k)
            self.sd[key] = wp
        return key

    def up(self, wkey):
        key = ("up", wkey)
        if key not in self.t:
            w = self.sd[wkey]                               # [C,1,k,k]
            c, _, k, _ = w.shape
            self.t[key] = (w.reshape(c, k * k).t().contiguous().to(self.device), k)   # [k*k][C] fp32
        return self.t[key]


class Plan:
    """Op array + the buffers it points into, for one (B,H,W)."""

    # lowering switches (DLAEngine mirrors them as attributes; the defaults are the measured-best choices)
    FLAGS = dict(
        fuse_heads=True,       # False: one conv3x3 + conv1x1 launch pair per head (debug/ablation)
        fuse_offsets=True,     # False: conv_offset_mask as its own launch + dcn2_kernel reading NHWC offsets
        stream_convs=True,     # False: 3x3 convs through the register-staged kernel (csrc/conv.hip)
        stream_dcn=False,      # True: 64-channel node DeformConvs through csrc/dcn4.hip (fp16 input, up-sampling folded in): 0.35 ms
                               # per batch-64 step faster while the offsets stay below ~1 px, 5x SLOWER per launch at 2 px
                               # (no patch slots: its LDS is full); default since round 2: csrc/dcn3.hip with patches
        stream_s2=True,        # False: stride-2 3x3 convs (Cin >= 64) through csrc/conv.hip
        stream_dcn3=True,      # ALL remaining fused DeformConvs take their filters by LDS-DMA (csrc/dcn3.hip WDMA: the variants with patches)
        dcn_patches=True,      # False: round 1's WDMA configurations (no patch slots: samples that leave the apron go through pass 2)
        dense_dcn3=True,       # those with <= 64 output channels do: margin-1 apron, two workgroups per CU (csrc/dcn3.hip)
        dense_dcn3_min_tiles=512,   # ... when the layer has at least this many 16x16 tiles (two per CU)
        fuse_upnode=True,      # False: up-sample + add always as its own launch in front of the 64-channel node DeformConvs
        fuse_upnode_min_f=2,   # ... from this up-sampling factor.  Same box, batch 64, up-sampling + node over the five layers:
                               # 1.234 ms as two launches each, 1.194 with the 4x layer folded, 1.156 with all five
        dcn_slots512=0,        # 1: margin 2 on the PACKED apron with 512 patch slots per tile (second 256 filled in a second round per stage)
        dcn_wide_margin=0,     # 1: every fused DeformConv (<= 64-channel workgroups) on the margin-4 packed apron (csrc/dcn3.hip PK): slower
                               # while the offsets stay small (more apron to stage), far faster once many samples of a tile leave a
                               # margin-2 apron; per-layer choices from a calibration batch: DLAEngine.calibrate_dcn_margins
        node_f16=True,         # bf16 plans: the up-sample + add kernel writes the `node` DeformConvs' input as fp16 (only they read it) and those
                               # DeformConvs run csrc/dcn3.hip's fp16-input variants (no conversion while the apron is staged)
        share_pool=True,       # False: level3/level4 max-pool their input twice (outer and inner tree), as the reference does
        fuse_stem=True,        # False: base_layer, level0 and level1 as three launches
        fuse_stem_proj=False,  # True: the fused stem launch also max-pools its output and applies level2's `project` conv (the residual branch
                               # of level2's first block: nothing else reads the pooled map): two HBM-bound launches and a 67 MB map less,
                               # bit-identical -- and measured a wash (round 4, same process, batch 64): the stem launch goes from 0.455 to
                               # 0.622 ms for the 0.113 ms of the two launches it absorbs (0.912 with the 1x1 conv on the waves that pool,
                               # 0.710 with its filters fetched per tile): the kernel sits at its 128-VGPR cap (52 more bytes of scratch)
                               # and its P3 phase, which every wave of the workgroup waits for, gets 16 cross-lane exchanges longer
        mixed_heads=0,         # 1: all heads in ONE launch (the kernel picks the 1 / 2 / 3-tile body per head; the halo tile is staged once).
                               # Measured (batch 64, same process / same box): the heads take 1.838 instead of 1.913 ms, but the STEP with
                               # three steps in flight gets 0.4 % slower (8424 / 8465 vs 8461 / 8499 images/s): the merged kernel
                               # needs 256 VGPRs, two of its waves fill a SIMD's register file, and the other streams' small kernels
                               # (up-sample + add, max-pool, gathers), which the 198-register narrow-heads launch lets onto its
                               # CUs, have to wait
        wide_heads_m2=0,       # 3: heads wider than 32 channels share one launch (measured: no gain)
        x3_dcn_patches=True,      # f16x3 plans: the patch-slot DeformConv variant (filters by LDS-DMA, far samples as patch pixels) for layers with
                                  # Cin % 32 == 0; False: the f32 plan's register-staged tiles, every far sample through pass 2
        stem_s2_direct=True,      # bf16 plans of the other backbones: the 7x7 stride-2 stem conv itself instead of im2col + 1x1 conv
        conv1x1_th16_min_cin=0,   # > 0: 1x1 convs with at least this many input channels (and > 32 outputs) use 16-row tiles
    )

    def __init__(self, pw, B, H, W, **flags):
        unknown = set(flags) - set(self.FLAGS)
        if unknown:
            raise TypeError("unknown lowering flags: %s" % sorted(unknown))
        for k, v in self.FLAGS.items():
            setattr(self, k, flags.get(k, v))
        self.stream_s2_min_cin = 64     # measured: the 32-channel stride-2 layer is faster on csrc/conv.hip (0.136 vs 0.165 ms)
        if pw.arch == "hourglass" and (H % 128 or W % 128):
            raise RuntimeError("Hourglass-104: input height/width must be multiples of 128 (got %dx%d): the published test "
                               "code pads to (x|127)+1" % (H, W))
        if H % 32 or W % 32:
            raise RuntimeError("input height/width must be multiples of 32 (got %dx%d): the reference pads "
                               "to (x|31)+1 (datasets/coco.py:160-163)" % (H, W))
        self.pw, self.B, self.H, self.W = pw, B, H, W
        self.dtype = pw.dtype
        self.es = 2 if pw.dtype in LOWP else 4
        self.ops = []
        self.dcn_layers = []    # (state_dict prefix, op index) of the fused DeformConvs
        self.keep = []          # tensors the ops point into
        self.images = torch.empty(B, 3, H, W, dtype=torch.float32, device=pw.device)
        self.outputs = {}
        self.all_outputs = None         # Hourglass: one head dict per stack (outputs = the last one)
        if pw.arch == "hourglass":
            self._lower_hourglass()
        elif pw.arch == "resdcn101":
            self._lower_resdcn()
        else:
            self._lower()
        self.op_array = (H3dOp * len(self.ops))(*self.ops)

    # -- buffer / op helpers --------------------------------------------------------------------
    def _alloc(self, H, W, C, dtype=None):
        td = _TORCH_DT[self.dtype] if dtype is None else dtype
        buf = torch.empty(self.B, H, W, C, dtype=td, device=self.pw.device)
        self.keep.append(buf)
        return View(buf, H, W, C, C, 0, buf.element_size())

    def _op(self, kind, **kw):
        op = H3dOp()
        op.kind, op.dtype, op.B = kind, _H3D_DT[self.dtype], self.B
        for k, v in kw.items():
            setattr(op, k, v)
        self.ops.append(op)

    def conv(self, x, wkey, out=None, bkey=None, bn=None, stride=1, relu=True, res=None, out_mode=_lib.OUT_NHWC,
             pad_cout_to=None, out_tensor=None):
        wshape = self.pw.sd[wkey].shape
        if (self.stream_convs and self.pw.dtype in LOWP and wshape[2] == 3 and wshape[1] % 16 == 0
                and (stride == 1 or (stride == 2 and wshape[1] >= self.stream_s2_min_cin and self.stream_s2))
                and out_mode == _lib.OUT_NHWC and pad_cout_to is None):
            return self._conv_stream(x, wkey, out, bkey, bn, relu, res, stride)
        wp, bp, cout, cin, k, rows = self.pw.conv(wkey, bkey, bn, pad_cout_to)
        assert cin == x.C, (wkey, cin, x.C)
        Ho = (x.H + 2 * (k // 2) - k) // stride + 1
        Wo = (x.W + 2 * (k // 2) - k) // stride + 1
        if out_mode == _lib.OUT_NHWC:
            if out is None:
                out = self._alloc(Ho, Wo, cout)
            assert (out.H, out.W, out.C) == (Ho, Wo, cout), (wkey, out.H, out.W, out.C, Ho, Wo, cout)
            optr, ocs = out.ptr, out.cs
        elif out_mode == _lib.OUT_NHWC_F32:
            out = self._alloc(Ho, Wo, cout, torch.float32)
            optr, ocs = out.ptr, out.cs
        else:
            optr, ocs = out_tensor.data_ptr(), cout
        tune = 0
        if (k == 1 and stride == 1 and self.conv1x1_th16_min_cin and cin >= self.conv1x1_th16_min_cin and cin % 64 == 0
                and cout > 32 and self.pw.dtype in LOWP):
            tune = 0x1000 | ((4 if cout > 64 else 2) << 4) | 2       # csrc/conv.hip tuning override: MT, TH = 16
        self._op(_lib.OP_CONV, in_=x.ptr, in2=res.ptr if res is not None else None, w=wp.data_ptr(),
                 bias=bp.data_ptr(), out=optr, H=x.H, W=x.W, Cin=cin, in_cs=x.cs,
                 in2_cs=res.cs if res is not None else 0, Ho=Ho, Wo=Wo, Cout=cout, out_cs=ocs, ksize=k,
                 stride=stride, relu=int(relu), out_mode=out_mode, wrows=rows, reserved=tune, wexp=self.pw.wexp.get(wp.data_ptr(), 0))
        return out

    def _conv_stream(self, x, wkey, out, bkey, bn, relu, res, stride=1):
        """3x3 conv (stride 1 or 2) through the LDS-DMA kernel (csrc/conv2.hip)."""
        wimg, bp, cout, cin, rows = self.pw.conv_stream(wkey, bkey, bn)
        assert cin == x.C, (wkey, cin, x.C)
        Ho, Wo = (x.H - 1) // stride + 1, (x.W - 1) // stride + 1
        if out is None:
            out = self._alloc(Ho, Wo, cout)
        assert (out.H, out.W, out.C) == (Ho, Wo, cout), (wkey, out.H, out.W, out.C)
        self._op(_lib.OP_CONV_STREAM, in_=x.ptr, in2=res.ptr if res is not None else None, w=wimg.data_ptr(),
                 bias=bp.data_ptr(), out=out.ptr, H=x.H, W=x.W, Cin=cin, in_cs=x.cs,
                 in2_cs=res.cs if res is not None else 0, Ho=Ho, Wo=Wo, Cout=cout, out_cs=out.cs, ksize=3,
                 stride=stride, relu=int(relu), out_mode=_lib.OUT_NHWC, wrows=rows)
        return out

    def dcn(self, x, om, wkey, bkey, bn, out=None):
        wp, bp, cout, cin, k, rows = self.pw.conv(wkey, bkey, bn, as_half=True)
        assert cin == x.C and k == 3
        if out is None:
            out = self._alloc(x.H, x.W, cout)
        self._op(_lib.OP_DCN, in_=x.ptr, in2=om.ptr, w=wp.data_ptr(), bias=bp.data_ptr(), out=out.ptr, H=x.H,
                 W=x.W, Cin=cin, in_cs=x.cs, in2_cs=om.cs, Ho=x.H, Wo=x.W, Cout=cout, out_cs=out.cs, ksize=3,
                 stride=1, relu=1, out_mode=_lib.OUT_NHWC, wrows=rows)
        return out

    def pool(self, x, out=None):
        if out is None:
            out = self._alloc(x.H // 2, x.W // 2, x.C)
        self._op(_lib.OP_MAXPOOL, in_=x.ptr, out=out.ptr, H=x.H, W=x.W, Cin=x.C, in_cs=x.cs, Ho=out.H, Wo=out.W,
                 Cout=x.C, out_cs=out.cs, ksize=2, stride=2)
        return out

    def upadd(self, x, skip, wkey, f16=False):
        """f16: the sum is written as fp16 (same 2-byte NHWC buffer) for a csrc/dcn4.hip consumer."""
        w, k = self.pw.up(wkey)
        f = k // 2
        out = self._alloc(x.H * f, x.W * f, x.C)
        assert (skip.H, skip.W, skip.C) == (out.H, out.W, out.C), wkey
        self._op(_lib.OP_UPADD, in_=x.ptr, in2=skip.ptr, w=w.data_ptr(), out=out.ptr, H=x.H, W=x.W, Cin=x.C,
                 in_cs=x.cs, in2_cs=skip.cs, Ho=out.H, Wo=out.W, Cout=x.C, out_cs=out.cs, ksize=k, stride=f,
                 out_mode=_lib.OUT_NHWC_F16 if f16 else _lib.OUT_NHWC)
        return out

    # -- network ---------------------------------------------------------------------------------
    def _block(self, x, p, stride, residual, out):
        """BasicBlock (model.py:46-60): conv-bn-relu, conv-bn, +residual, relu."""
        t = self.conv(x, p + ".conv1.weight", bn=p + ".bn1", stride=stride)
        return self.conv(t, p + ".conv2.weight", bn=p + ".bn2", res=residual, out=out)

    def _tree1(self, x, p, cin, cout, stride, level_root, out, cat=None, bottom=None, residual=None):
        """One-level Tree (model.py:209-218).  `cat` = pre-allocated Root input whose trailing
        slices (children) the caller has filled; layout [x2 | x1 | children...].  `bottom`: the
        max-pooled x when the caller already has it (the reference pools the same tensor in the outer
        and in the inner tree, model.py:213)."""
        Ho, Wo = x.H // stride, x.W // stride
        if cat is None:
            cat = self._alloc(Ho, Wo, 2 * cout + (cin if level_root else 0))
        s_x2, s_x1 = cat.slice(0, cout), cat.slice(cout, cout)
        if residual is not None:                             # (the caller already has project(pool(x)): csrc/stem3.hip PROJ)
            assert stride > 1 and not level_root and cin != cout and (residual.H, residual.W, residual.C) == (Ho, Wo, cout)
        elif bottom is not None:
            assert stride > 1 and not level_root and (bottom.H, bottom.W, bottom.C) == (Ho, Wo, cin)
        elif stride > 1:
            bottom = self.pool(x, cat.slice(2 * cout, cin) if level_root else None)
        else:
            bottom = x
        if residual is not None:
            pass
        elif cin != cout:
            residual = self.conv(bottom, p + ".project.0.weight", bn=p + ".project.1", relu=False)
        else:
            residual = bottom
        self._block(x, p + ".tree1", stride, residual, s_x1)
        self._block(s_x1, p + ".tree2", 1, s_x1, s_x2)
        return self.conv(cat, p + ".root.conv.weight", bn=p + ".root.bn", out=out)

    def _tree2(self, x, p, cin, cout, out):
        """Two-level Tree with level_root (level3/level4; model.py:209-222): Root input of the inner
        tree2 = [x2 | x1 | bottom | tree1 output]."""
        Ho, Wo = x.H // 2, x.W // 2
        cat = self._alloc(Ho, Wo, 2 * cout + cin + cout)
        pooled = self.pool(x, cat.slice(2 * cout, cin))
        x1 = self._tree1(x, p + ".tree1", cin, cout, 2, False, cat.slice(2 * cout + cin, cout),
                         bottom=pooled if self.share_pool else None)
        return self._tree1(x1, p + ".tree2", cout, cout, 1, False, out, cat=cat)

    def _dcn_f16_ok(self, p):
        """node DeformConvs with 64 input and <= 64 output channels run on csrc/dcn4.hip (fp16 input)."""
        w = self.pw.sd[p + ".conv.weight"]
        return (self.pw.use_dcn and self.fuse_offsets and self.stream_dcn and self.pw.dtype == "bf16"
                and w.shape[1] == 64 and w.shape[0] <= 64)

    def _node_f16_ok(self, p):
        """`node` DeformConvs of a bf16 plan that take an fp16 input (csrc/dcn3.hip F16IN): the fused patch-slot variants with more
        than 32 output channels."""
        w = self.pw.sd[p + ".conv.weight"]
        return (self.node_f16 and self.pw.use_dcn and self.fuse_offsets and self.pw.dtype == "bf16" and self.stream_dcn3 and self.dcn_patches
                and w.shape[1] % 32 == 0 and w.shape[0] > 32 and w.shape[0] % 8 == 0)

    def _deform(self, x, p, out=None, x_is_f16=False, in_f16=False):
        """DeformConv (model.py:346-362): DCN or plain 3x3 conv, then BN + ReLU (folded).  in_f16: `x` holds fp16 values in a
        bf16 plan (written by `upadd(..., f16=True)`) and the op carries the fp16-input bit."""
        if x_is_f16:
            wimg, woimg, bias, cout, cin, rows = self.pw.dcn_stream(p)
            if out is None:
                out = self._alloc(x.H, x.W, cout)
            self._op(_lib.OP_DCN_FUSED_F16, in_=x.ptr, in2=woimg.data_ptr(), w=wimg.data_ptr(), bias=bias.data_ptr(),
                     out=out.ptr, H=x.H, W=x.W, Cin=cin, in_cs=x.cs, Ho=x.H, Wo=x.W, Cout=cout, out_cs=out.cs, ksize=3,
                     stride=1, relu=1, out_mode=_lib.OUT_NHWC, wrows=rows)
            return out
        w = self.pw.sd[p + ".conv.weight"]
        if (self.pw.use_dcn and self.fuse_offsets and self.pw.dtype in LOWP
                and (self.stream_dcn3 or (self.dense_dcn3 and w.shape[0] <= 64
                                          # two workgroups per CU only pay with >= 2 x 256 tiles (measured: the 32x32 layer
                                          # of a batch-64 plan, 256 tiles, 0.062 -> 0.076 ms)
                                          and self.B * ((x.H + 15) // 16) * ((x.W + 15) // 16) >= self.dense_dcn3_min_tiles))):
            ck = int(_lib.lib().h3d_dcn_fused_ck(int(w.shape[1]), int(w.shape[0])))
            wimg, woimg, bias, cout, cin, rows = self.pw.dcn_stream(p, ck)
            if out is None:
                out = self._alloc(x.H, x.W, cout)
            var = 0
            if self.dcn_patches and cin % 32 == 0:
                var = 0x8000 if self.dcn_wide_margin else 0x10000 if self.dcn_slots512 else self.pw.dcn_variant.get(p, 0)
            self._op(_lib.OP_DCN_FUSED_STREAM, in_=x.ptr, in2=woimg.data_ptr(), w=wimg.data_ptr(), bias=bias.data_ptr(),
                     out=out.ptr, H=x.H, W=x.W, Cin=cin, in_cs=x.cs, Ho=x.H, Wo=x.W, Cout=cout, out_cs=out.cs, ksize=3,
                     stride=1, relu=1, out_mode=_lib.OUT_NHWC, wrows=rows,
                     reserved=(var | (DCN_F16IN if in_f16 else 0)) if self.dcn_patches else 0x1000)
            self.dcn_layers.append((p, len(self.ops) - 1))
            return out
        assert not in_f16, p
        if self.pw.use_dcn and self.fuse_offsets and self.pw.dtype == "f16x3" and self.x3_dcn_patches and w.shape[1] % 32 == 0:
            wimg, woimg, bias, cout, cin, rows = self.pw.dcn_stream_x3(p)
            if out is None:
                out = self._alloc(x.H, x.W, cout)
            self._op(_lib.OP_DCN_FUSED_STREAM, in_=x.ptr, in2=woimg.data_ptr(), w=wimg.data_ptr(), bias=bias.data_ptr(), out=out.ptr,
                     H=x.H, W=x.W, Cin=cin, in_cs=x.cs, Ho=x.H, Wo=x.W, Cout=cout, out_cs=out.cs, ksize=3, stride=1, relu=1,
                     out_mode=_lib.OUT_NHWC, wrows=rows, wexp=self.pw.wexp[wimg.data_ptr()], wexp2=self.pw.wexp[woimg.data_ptr()])
            return out
        if self.pw.use_dcn and self.fuse_offsets:
            wp, bp, cout, cin, k, rows = self.pw.conv(p + ".conv.weight", p + ".conv.bias", p + ".actf.0", as_half=True)
            wo, bo = self.pw.offset_conv(p + ".conv.conv_offset_mask.weight", p + ".conv.conv_offset_mask.bias", rows)
            key = ("dcnbias", p)
            if key not in self.pw.t:
                self.pw.t[key] = torch.cat([bp.cpu(), bo]).contiguous().to(self.pw.device)
            bias = self.pw.t[key]
            if out is None:
                out = self._alloc(x.H, x.W, cout)
            self._op(_lib.OP_DCN_FUSED, in_=x.ptr, in2=wo.data_ptr(), w=wp.data_ptr(), bias=bias.data_ptr(), out=out.ptr,
                     H=x.H, W=x.W, Cin=cin, in_cs=x.cs, Ho=x.H, Wo=x.W, Cout=cout, out_cs=out.cs, ksize=3, stride=1,
                     relu=1, out_mode=_lib.OUT_NHWC, wrows=rows, wexp=self.pw.wexp.get(wp.data_ptr(), 0), wexp2=self.pw.wexp.get(wo.data_ptr(), 0))
            return out
        if self.pw.use_dcn:
            if self.pw.dtype == "f16":
                raise RuntimeError("fp16 plans run the fused DeformConv kernel only (fuse_offsets=False is a bf16 / f32 debugging path)")
            om = self.conv(x, p + ".conv.conv_offset_mask.weight", bkey=p + ".conv.conv_offset_mask.bias",
                           relu=False, out_mode=_lib.OUT_NHWC_F32, pad_cout_to=32)
            return self.dcn(x, om, p + ".conv.weight", p + ".conv.bias", p + ".actf.0", out)
        return self.conv(x, p + ".conv.weight", bkey=p + ".conv.bias", bn=p + ".actf.0", out=out)

    def _ida(self, layers, p, startp, endp):
        """IDAUp.forward (model.py:384-390) on the python list `layers` (mutated like the reference)."""
        for i in range(startp + 1, endp):
            k = i - startp
            y = self._deform(layers[i], "%s.proj_%d" % (p, k))
            f16 = self._dcn_f16_ok("%s.node_%d" % (p, k))
            if f16 and self.fuse_upnode and self.pw.up("%s.up_%d.weight" % (p, k))[1] // 2 >= self.fuse_upnode_min_f:
                layers[i] = self._updcn(y, layers[i - 1], "%s.up_%d.weight" % (p, k), "%s.node_%d" % (p, k))
                continue
            nf16 = not f16 and self._node_f16_ok("%s.node_%d" % (p, k))
            y = self.upadd(y, layers[i - 1], "%s.up_%d.weight" % (p, k), f16=f16 or nf16)
            layers[i] = self._deform(y, "%s.node_%d" % (p, k), x_is_f16=f16, in_f16=nf16)

    def _updcn(self, x, skip, wkey, p):
        """node(up(x) + skip) in one launch (csrc/dcn4.hip UP = 1): the up-sampled sum never reaches HBM."""
        wup, k = self.pw.up(wkey)
        f = k // 2
        wimg, woimg, bias, cout, cin, rows = self.pw.dcn_stream(p)
        assert (skip.H, skip.W, skip.C) == (x.H * f, x.W * f, x.C) and cin == x.C == 64, wkey
        out = self._alloc(skip.H, skip.W, cout)
        desc = _lib.H3dUpdcnDesc()
        desc.skip, desc.w_up, desc.w_off, desc.skip_cs = skip.ptr, wup.data_ptr(), woimg.data_ptr(), skip.cs
        self.keep.append(desc)
        self._op(_lib.OP_UPDCN_F16, in_=x.ptr, in2=ctypes.addressof(desc), w=wimg.data_ptr(), bias=bias.data_ptr(), out=out.ptr,
                 H=x.H, W=x.W, Cin=cin, in_cs=x.cs, Ho=out.H, Wo=out.W, Cout=cout, out_cs=out.cs, ksize=3, stride=f, relu=1,
                 out_mode=_lib.OUT_NHWC, wrows=rows)
        return out

    def _lower(self):
        B, H, W = self.B, self.H, self.W
        C = arch.CHANNELS
        if self.fuse_stem and self.pw.dtype == "f16x3" and C[0] == 16 and C[1] == 32:
            # the f16x3 twin of the fused stem (csrc/stem3x.hip): the two full-resolution maps stay in LDS as split operand fragments
            y0 = None
            y1 = self._alloc((H - 1) // 2 + 1, (W - 1) // 2 + 1, C[1])
            w, b = self.pw.stem3_x3()
            self._op(_lib.OP_STEM3, in_=self.images.data_ptr(), w=w.data_ptr(), bias=b.data_ptr(), out=y1.ptr, H=H, W=W,
                     Cin=3, in_cs=3, Ho=y1.H, Wo=y1.W, Cout=C[1], out_cs=y1.cs, ksize=7, stride=2, relu=1)
            res2 = None
        elif self.fuse_stem and self.pw.dtype in LOWP and C[0] == 16 and C[1] == 32 and W % 4 == 0:
            # (W % 4: csrc/stem3.hip reads the image as aligned float4; any other width takes the three launches)
            # base_layer + level0 + level1 in one launch: the two full-resolution maps never reach HBM (nothing else
            # reads them: DLAUp starts at level 2)
            y0 = None
            y1 = self._alloc((H - 1) // 2 + 1, (W - 1) // 2 + 1, C[1])
            proj = self.fuse_stem_proj and C[2] == 64 and y1.H % 2 == 0 and y1.W % 2 == 0
            w, b = self.pw.stem3(proj)
            res2 = self._alloc(y1.H // 2, y1.W // 2, C[2]) if proj else None
            self._op(_lib.OP_STEM3, in_=self.images.data_ptr(), in2=res2.ptr if proj else None, in2_cs=res2.cs if proj else 0,
                     w=w.data_ptr(), bias=b.data_ptr(), out=y1.ptr, H=H, W=W,
                     Cin=3, in_cs=3, Ho=y1.H, Wo=y1.W, Cout=C[1], out_cs=y1.cs, ksize=7, stride=2, relu=1)
        else:
            w, b = self.pw.stem()
            x = self._alloc(H, W, C[0])
            self._op(_lib.OP_STEM, in_=self.images.data_ptr(), w=w.data_ptr(), bias=b.data_ptr(), out=x.ptr, H=H, W=W,
                     Cin=3, in_cs=3, Ho=H, Wo=W, Cout=C[0], out_cs=x.cs, ksize=7, stride=1, relu=1, wexp=self.pw.wexp.get(w.data_ptr(), 0))
            y0 = self.conv(x, "base.level0.0.weight", bn="base.level0.1")
            y1 = self.conv(y0, "base.level1.0.weight", bn="base.level1.1", stride=2)
            res2 = None
        y2 = self._tree1(y1, "base.level2", C[1], C[2], 2, False, None, residual=res2)
        y3 = self._tree2(y2, "base.level3", C[2], C[3], None)
        y4 = self._tree2(y3, "base.level4", C[3], C[4], None)
        y5 = self._tree1(y4, "base.level5", C[4], C[5], 2, True, None)
        layers = [y0, y1, y2, y3, y4, y5]
        # DLAUp.forward (model.py:409-415)
        outs = [layers[-1]]
        for i in range(3):
            self._ida(layers, "dla_up.ida_%d" % i, len(layers) - i - 2, len(layers))
            outs.insert(0, layers[-1])
        # DLASeg.forward (model.py:480-483): ida_up over the three finest maps
        ys = [outs[0], outs[1], outs[2]]
        self._ida(ys, "ida_up", 0, 3)
        feat = ys[-1]
        self.feat = feat
        self._lower_heads(feat)

    def _lower_heads(self, feat):
        """Output heads on the 64-channel map (model.py:451-460, 485-489; the ResNet-DCN heads have the same form)."""
        B = self.B
        Ho, Wo = feat.H, feat.W
        fused = (self.pw.head_conv > 0 and self.pw.head_conv % 64 == 0 and feat.C == 64 and
                 len(self.pw.heads) <= _lib.HEADS_MAX and max(self.pw.heads.values()) <= 96 and self.fuse_heads)
        if fused:
            # one launch per group of heads with the same number of 32-row output tiles, so the
            # narrow heads do not inherit the register footprint of the 72-channel pose head
            groups = {}
            for head, c in self.pw.heads.items():
                m2 = (c + 31) // 32
                groups.setdefault(0 if self.mixed_heads else 1 if m2 == 1 else self.wide_heads_m2 or m2, []).append(head)
            for m2 in sorted(groups):
                w1, b1, per = self.pw.fused_heads(tuple(groups[m2]))
                desc = _lib.H3dHeadsDesc()
                desc.nheads = len(per)
                desc.wexp = self.pw.wexp.get(w1.data_ptr(), 0)
                for i, (head, c, w2, b2) in enumerate(per):
                    o = torch.empty(B, c, Ho, Wo, dtype=torch.float32, device=self.pw.device)
                    self.outputs[head] = o
                    desc.head[i].w2, desc.head[i].b2, desc.head[i].out, desc.head[i].C = w2.data_ptr(), b2.data_ptr(), o.data_ptr(), c
                    desc.head[i].wexp2 = self.pw.wexp.get(w2.data_ptr(), 0)
                self.keep.append(desc)
                self._op(_lib.OP_HEADS, in_=feat.ptr, in2=ctypes.addressof(desc), w=w1.data_ptr(), bias=b1.data_ptr(),
                         H=Ho, W=Wo, Cin=feat.C, in_cs=feat.cs, Ho=Ho, Wo=Wo, Cout=self.pw.head_conv, ksize=3, stride=1)
            self.outputs = {h: self.outputs[h] for h in self.pw.heads}      # reference head order
            return
        for head, c in self.pw.heads.items():
            o = torch.empty(B, c, Ho, Wo, dtype=torch.float32, device=self.pw.device)
            self.outputs[head] = o
            if self.pw.head_conv > 0:
                t = self.conv(feat, head + ".0.weight", bkey=head + ".0.bias")
                self.conv(t, head + ".2.weight", bkey=head + ".2.bias", relu=False,
                          out_mode=_lib.OUT_NCHW_F32, out_tensor=o)
            else:
                self.conv(feat, head + ".weight", bkey=head + ".bias", relu=False,
                          out_mode=_lib.OUT_NCHW_F32, out_tensor=o)

    def _stem_s2(self, wkey, bkey, bn, cout):
        """Conv2d(3, cout, 7, stride 2, padding 3) + BN + ReLU from the NCHW fp32 images."""
        H, W = self.H, self.W
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        if self.dtype in LOWP and self.stem_s2_direct:
            w, b = self.pw.stem_s2(wkey, bkey, bn)
            x = self._alloc(Ho, Wo, cout)
            self._op(_lib.OP_STEM, in_=self.images.data_ptr(), w=w.data_ptr(), bias=b.data_ptr(), out=x.ptr, H=H, W=W, Cin=3, in_cs=3,
                     Ho=Ho, Wo=Wo, Cout=cout, out_cs=x.cs, ksize=7, stride=2, relu=1)
            return x
        patches = self._alloc(Ho, Wo, 160)                   # fp32 plans: im2col + 1x1 conv (csrc/extra.hip)
        self._op(_lib.OP_IM2COL, in_=self.images.data_ptr(), out=patches.ptr, H=H, W=W, Cin=3, in_cs=3, Ho=Ho, Wo=Wo, Cout=160,
                 out_cs=patches.cs, ksize=7, stride=2)
        return self.conv(patches, self.pw.im2col_key(wkey), bkey=bkey, bn=bn)

    # -- ResNet-101-DCN (arch_res.py; published CenterNet `resnet_dcn.py`) ---------------------------------------------------
    def _lower_resdcn(self):
        B, H, W = self.B, self.H, self.W
        x = self._stem_s2("conv1.weight", None, "bn1", 64)   # conv1 7x7/2 + bn1 + ReLU
        y = self._alloc((x.H - 1) // 2 + 1, (x.W - 1) // 2 + 1, x.C)
        self._op(_lib.OP_MAXPOOL3, in_=x.ptr, out=y.ptr, H=x.H, W=x.W, Cin=x.C, in_cs=x.cs, Ho=y.H, Wo=y.W, Cout=x.C, out_cs=y.cs,
                 ksize=3, stride=2)
        x = y
        for p, cin, planes, stride, down in arch_res.blocks(101):
            t = self.conv(x, p + ".conv1.weight", bn=p + ".bn1")
            t = self.conv(t, p + ".conv2.weight", bn=p + ".bn2", stride=stride)
            res = self.conv(x, p + ".downsample.0.weight", bn=p + ".downsample.1", stride=stride, relu=False) if down else x
            x = self.conv(t, p + ".conv3.weight", bn=p + ".bn3", res=res)
        for i, planes in enumerate(arch_res.DECONV):
            x = self._deform(x, "deconv_layers.%d" % (6 * i))                # DCN + BN + ReLU
            wkey, bn = self.pw.deconv4_as_conv3("deconv_layers.%d.weight" % (6 * i + 3), "deconv_layers.%d" % (6 * i + 4))
            t = self.conv(x, wkey, bn=bn)                                      # [B,H,W,4C], BN + ReLU folded / fused
            x = self._alloc(2 * t.H, 2 * t.W, planes)
            self._op(_lib.OP_DEPTH2SPACE, in_=t.ptr, out=x.ptr, H=t.H, W=t.W, Cin=4 * planes, in_cs=t.cs, Ho=x.H, Wo=x.W,
                     Cout=planes, out_cs=x.cs, ksize=1, stride=1)
        self.feat = x
        self._lower_heads(x)

    # -- Hourglass-104 (arch_hg.py; published CenterNet `exkp`) ---------------------------------------------------------
    def _hg_residual(self, x, p, cin, cout, stride):
        """residual: relu(bn2(conv2(relu(bn1(conv1(x))))) + skip(x)), skip = 1x1 conv + BN when stride / width change."""
        t = self.conv(x, p + ".conv1.weight", bn=p + ".bn1", stride=stride)
        skip = x
        if arch_hg.residual_has_skip(cin, cout, stride):
            skip = self.conv(x, p + ".skip.0.weight", bn=p + ".skip.1", stride=stride, relu=False)
        return self.conv(t, p + ".conv2.weight", bn=p + ".bn2", res=skip)

    def _hg_seq(self, x, p, kind, cin, cout, modules):
        for j, (ci, co, st) in enumerate(arch_hg.layer_specs(kind, cin, cout, modules)):
            x = self._hg_residual(x, "%s.%d" % (p, j), ci, co, st)
        return x

    def _hg_kp(self, x, p, n, dims, modules):
        up1 = self._hg_seq(x, p + ".up1", "layer", dims[0], dims[0], modules[0])
        low1 = self._hg_seq(x, p + ".low1", "hg", dims[0], dims[1], modules[0])
        if n > 1:
            low2 = self._hg_kp(low1, p + ".low2", n - 1, dims[1:], modules[1:])
        else:
            low2 = self._hg_seq(low1, p + ".low2", "layer", dims[1], dims[1], modules[1])
        low3 = self._hg_seq(low2, p + ".low3", "revr", dims[1], dims[0], modules[0])
        return self.upadd(low3, up1, self.pw.nearest_up_key(dims[0]))       # up1 + nearest x2 of low3

    def _lower_hourglass(self):
        B, H, W = self.B, self.H, self.W
        nstack = 2
        # pre.0: Conv2d(3, 128, 7, stride 2, pad 3) + BN + ReLU
        inter = self._stem_s2("pre.0.conv.weight", None, "pre.0.bn", arch_hg.PRE_DIM)
        inter = self._hg_residual(inter, "pre.1", arch_hg.PRE_DIM, arch_hg.DIMS[0], 2)
        self.all_outputs = []
        for i in range(nstack):
            kp = self._hg_kp(inter, "kps.%d" % i, arch_hg.N, arch_hg.DIMS, arch_hg.MODULES)
            cnv = self.conv(kp, "cnvs.%d.conv.weight" % i, bn="cnvs.%d.bn" % i)
            out = {}
            for head, c in self.pw.heads.items():
                o = torch.empty(B, c, cnv.H, cnv.W, dtype=torch.float32, device=self.pw.device)
                t = self.conv(cnv, "%s.%d.0.conv.weight" % (head, i), bkey="%s.%d.0.conv.bias" % (head, i))
                self.conv(t, "%s.%d.1.weight" % (head, i), bkey="%s.%d.1.bias" % (head, i), relu=False,
                          out_mode=_lib.OUT_NCHW_F32, out_tensor=o)
                out[head] = o
            self.all_outputs.append(out)
            if i < nstack - 1:
                a = self.conv(inter, "inters_.%d.0.weight" % i, bn="inters_.%d.1" % i, relu=False)
                inter = self.conv(cnv, "cnvs_.%d.0.weight" % i, bn="cnvs_.%d.1" % i, res=a)       # relu(inters_(inter) + cnvs_(cnv))
                inter = self._hg_residual(inter, "inters.%d" % i, arch_hg.DIMS[0], arch_hg.DIMS[0], 1)
        self.outputs = self.all_outputs[-1]

    def retarget_outputs(self, views):
        """Point the head outputs at caller-provided contiguous [B,C,H,W] fp32 views (sub-batch plans)."""
        for h, v in views.items():
            assert v.is_contiguous() and tuple(v.shape) == tuple(self.outputs[h].shape)
        for i, op in enumerate(self.ops):
            if op.kind == _lib.OP_HEADS:
                d = ctypes.cast(op.in2, ctypes.POINTER(_lib.H3dHeadsDesc)).contents
                for j in range(d.nheads):
                    for h, o in self.outputs.items():
                        if d.head[j].out == o.data_ptr():
                            d.head[j].out = views[h].data_ptr()
                            break
            elif op.kind == _lib.OP_CONV and op.out_mode == _lib.OUT_NCHW_F32:
                for h, o in self.outputs.items():
                    if op.out == o.data_ptr():
                        self.op_array[i].out = views[h].data_ptr()
                        break
        self.outputs = dict(views)

    def run(self):
        rc = _lib.lib().h3d_run_ops(self.op_array, len(self.ops), _lib.stream_ptr())
        _lib.check(rc, "h3d_run_ops")
        return self.outputs


class DLAEngine:
    """state_dict -> packed weights -> cached plans.  `forward(images)` returns the head dict
    (fresh views of the plan's output buffers; they are overwritten by the next forward of the
    same shape, like any static-graph runtime -- clone to keep)."""

    def __init__(self, state_dict, heads, use_dcn, dtype="bf16", device="cuda", head_conv=256, arch_name="dla34"):
        if dtype not in _TORCH_DT:
            raise ValueError("dtype must be 'bf16', 'f16', 'f32' or 'f16x3'")
        _lib.lib()                                     # fail loudly now if the HIP library is missing
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("Not implemented on the CPU")
        self.pw = PackedWeights(state_dict, heads, use_dcn, dtype, self.device, head_conv, arch_name)
        self.plans = {}
        for k, v in Plan.FLAGS.items():   # lowering switches, see Plan.FLAGS (changing one requires plans.clear())
            setattr(self, k, v)
        self.streams = 1                # >1: run that many sub-batches concurrently on their own HIP streams

    def _flags(self):
        return {k: getattr(self, k) for k in Plan.FLAGS}

    def plan(self, B, H, W, slot=0):
        """slot: independent copies of the plan (own activation and output buffers, shared packed weights) so that
        consecutive batches can be in flight on different HIP streams (bench.py --pipeline)."""
        key = (B, H, W) if slot == 0 else (B, H, W, slot)
        if key not in self.plans:
            with torch.cuda.device(self.device):
                self.plans[key] = Plan(self.pw, B, H, W, **self._flags())
        return self.plans[key]

    def forward(self, images, slot=0):
        _lib.require_cuda(images)
        if images.dim() != 4 or images.shape[1] != 3:
            raise RuntimeError("expected images [B,3,H,W], got %s" % (tuple(images.shape),))
        B, _, H, W = images.shape
        if self.streams > 1 and B % self.streams == 0 and B // self.streams >= 8 and self.pw.arch == "dla34":
            return self._forward_split(images, slot)
        plan = self.plan(B, H, W, slot)
        with torch.cuda.device(self.device):
            if images.dtype == torch.float32 and images.is_contiguous():
                plan.op_array[0].in_ = images.data_ptr()       # read the caller's batch in place
            else:
                plan.images.copy_(images)
                plan.op_array[0].in_ = plan.images.data_ptr()
            return plan.run()

    def _forward_split(self, images, slot=0):
        """Sub-batches on separate HIP streams: the small-grid layers (level4/5, the 16x16 / 32x32 neck
        layers: 128-512 workgroups on 256 CUs) and every kernel's tail overlap with the other
        sub-batch's launches.  Outputs are written into one full-batch tensor per head.  Every `slot` has its own
        sub-plans, output tensors and internal streams: two batches in flight on different slots share nothing but
        the packed weights."""
        B, _, H, W = images.shape
        n = self.streams
        sub = B // n
        key = ("split", B, H, W, slot)
        with torch.cuda.device(self.device):
            if key not in self.plans:
                plans = [Plan(self.pw, sub, H, W, **self._flags())
                         for _ in range(n)]
                full = {h: torch.empty((B,) + tuple(o.shape[1:]), dtype=o.dtype, device=o.device)
                        for h, o in plans[0].outputs.items()}
                for i, p in enumerate(plans):          # re-point the head outputs into the full-batch tensors
                    p.retarget_outputs({h: full[h][i * sub:(i + 1) * sub] for h in full})
                self.plans[key] = (plans, full, [torch.cuda.Stream(device=self.device) for _ in range(n)])
            plans, full, streams = self.plans[key]
            if not (images.dtype == torch.float32 and images.is_contiguous()):
                images = images.float().contiguous()
            cur = torch.cuda.current_stream()
            ready = torch.cuda.Event()
            ready.record(cur)
            for i, (p, st) in enumerate(zip(plans, streams)):
                st.wait_event(ready)
                with torch.cuda.stream(st):
                    p.op_array[0].in_ = images[i * sub:(i + 1) * sub].data_ptr()
                    p.run()
                    done = torch.cuda.Event()
                    done.record(st)
                cur.wait_event(done)
            self._keepalive = getattr(self, "_keepalive", {})
            self._keepalive[slot] = images
            return full

    DCN_VARIANTS = {"narrow": 0, "slots512": 0x10000, "wide": 0x8000}      # h3d_op.reserved bits read by csrc/dcn3.hip's launcher

    # Cost model of the three tile variants, in units of "one tile of the `narrow` variant that stays within its slots"
    # (fitted once from round 3's timing table; tools/fit_dcn_rule.py prints model vs stopwatch per layer):
    #   a tile with more far samples than slots re-runs them in pass 2 (offset conv recomputed + serialised global
    #   gathers): it costs ~3.2x a normal tile; the 512-slot variant is 2 % slower on a tile that does not need it and pays
    #   an exposed load latency per stage for a second fill round; the wide margin stages 40 % more apron: 5 % slower.
    #   tail: a launch ends with its slowest workgroup -- on a grid of few rounds (workgroups / resident workgroups) ONE overflowing
    #   tile delays the end by a good part of a tile time, whatever the share of such tiles (measured, round 4: 256 -> 256 @32x32
    #   at batch 64, two rounds, 0.4 % of the tiles over their slots: 0.166 ms narrow vs 0.144 with 512 slots; 256 -> 64 @32x32,
    #   half a round, 7 % of the wide variant's tiles over: 0.108 ms wide vs 0.075 with 512 slots).
    DCN_RULE = {"pass2": 2.2, "slots512": 0.02, "round2": 0.30, "wide": 0.05, "tail": 0.30, "min_gain": 0.03}

    def dcn_far_samples(self, images):
        """Per fused DeformConv layer of the plan for `images`' shape: the kernel's own count of samples per 16x16 tile
        whose corners leave the apron (`h3d_dcn_far_samples`), for the margin-2 apron and for the wide one.
        -> {layer: {"narrow": int32 tensor [tiles], "wide": int32 tensor [tiles]}}.  Runs the plan once (one stream, so
        every layer's input buffer holds real activations), then phase A + geometry of every DeformConv twice."""
        _lib.require_cuda(images)
        if self.pw.dtype not in LOWP:
            raise RuntimeError("dcn_far_samples: the fused DeformConv variants exist for bf16 / f16 plans")
        B, _, H, W = images.shape
        out = {}
        with torch.cuda.device(self.device):
            streams, self.streams = self.streams, 1          # (_forward_split runs sub-plans: plan(B, H, W) would be left untouched)
            try:
                self.forward(images)
            finally:
                self.streams = streams
            plan = self.plan(B, H, W)
            for p, i in plan.dcn_layers:
                src = plan.ops[i]
                if src.Cin % 32 or src.reserved & 0x1000:
                    continue
                tiles = B * (-(-src.H // 16)) * (-(-src.W // 16))
                rec = {}
                for name in ("narrow", "wide"):
                    op = H3dOp()
                    ctypes.memmove(ctypes.byref(op), ctypes.byref(src), ctypes.sizeof(H3dOp))
                    op.reserved = self.DCN_VARIANTS[name] | (src.reserved & DCN_F16IN)
                    cnt = torch.empty(tiles, dtype=torch.int32, device=self.device)
                    _lib.check(_lib.lib().h3d_dcn_far_samples(ctypes.byref(op), cnt.data_ptr(), _lib.stream_ptr()), "h3d_dcn_far_samples")
                    rec[name] = cnt
                out[p] = rec
            torch.cuda.synchronize()
        return out

    def calibrate_dcn_margins(self, images, rule=None):
        """Choose per fused DeformConv layer among the three tile variants of csrc/dcn3.hip:
          narrow    margin-2 apron, 256 patch slots per tile   (default; fastest while almost no tile overflows)
          slots512  margin-2 packed apron, 512 slots in two rounds per stage
          wide      margin-4 packed apron, 256 slots
        by a RULE on what the kernels themselves count on a calibration batch (`dcn_far_samples`: per tile, the samples
        that leave the apron) -- a deterministic function of (weights, images): two processes make the same choice and
        therefore return the same bits (round 3 timed the variants with HIP events, and where two of them were within 3 %
        the choice, and with it the accumulation order of overflowing tiles, differed from run to run).  Cost per layer in
        units of a normal tile (DCN_RULE): narrow = 1 + pass2 * P(n > 256); slots512 = 1 + c + round2 * P(256 < n <= 512) +
        pass2 * P(n > 512); wide = 1 + c' + pass2 * P(n_wide > 256); a layer leaves `narrow` only for a variant cheaper by
        `min_gain`.  Returns {layer: {"cost": {variant: x}, "tiles_over_256": f, ...}}; the choice lands in `pw.dcn_variant`
        and plans built before the call are dropped."""
        rule = dict(self.DCN_RULE, **(rule or {}))
        with torch.cuda.device(self.device):
            self.pw.dcn_variant = {}
            self.plans.clear()
            stats = self.dcn_far_samples(images)
        report = {}
        for p, rec in stats.items():
            n2, n4 = rec["narrow"].float(), rec["wide"].float()
            f256 = float((n2 > 256).float().mean())
            f512 = float((n2 > 512).float().mean())
            w256 = float((n4 > 256).float().mean())
            rounds = self._dcn_rounds(p, n2.numel())

            def over(f):                                    # cost of the tiles that run pass 2: their share, or the launch's tail
                return max(rule["pass2"] * f, min(rule["pass2"], rule["tail"] / rounds) if f > 0 else 0.0)
            cost = {"narrow": 1.0 + over(f256),
                    "slots512": 1.0 + rule["slots512"] + rule["round2"] * (f256 - f512) + over(f512),
                    "wide": 1.0 + rule["wide"] + over(w256)}
            best = min(("narrow", "slots512", "wide"), key=lambda k: (cost[k], k != "narrow"))
            if best != "narrow" and cost[best] < (1.0 - rule["min_gain"]) * cost["narrow"]:
                self.pw.dcn_variant[p] = self.DCN_VARIANTS[best]
            else:
                best = "narrow"
            report[p] = {"choice": best, "cost": {k: round(v, 4) for k, v in cost.items()}, "tiles_over_256": round(f256, 5),
                         "tiles_over_512": round(f512, 5), "tiles_over_256_wide": round(w256, 5), "rounds": round(rounds, 3),
                         "far_samples_per_tile": round(float(n2.mean()), 2)}
        self.plans.clear()
        return report

    def _dcn_rounds(self, p, tiles):
        """Workgroups of DeformConv layer `p` per resident workgroup of the device (csrc/dcn3.hip's launcher: <= 64-channel
        workgroups, two per CU; 128-channel ones, one per CU, unless that grid would leave CUs idle)."""
        cout = int(self.pw.sd[p + ".conv.weight"].shape[0])
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        if cout <= 64:
            groups, per_cu = 1, 2
        elif tiles * (-(-cout // 128)) < 192:
            groups, per_cu = -(-cout // 64), 2
        else:
            groups, per_cu = -(-cout // 128), 1
        return max(tiles * groups / float(cus * per_cu), 1e-3)

    def time_dcn_variants(self, images, reps=3):
        """Stopwatch counterpart of `calibrate_dcn_margins` (what round 3 used to CHOOSE; now only the yardstick the rule's
        constants are fitted against, tools/fit_dcn_rule.py): every fused DeformConv op of the plan timed `reps` times per
        variant with HIP events (`h3d_run_ops_timed`), median.  Changes nothing.  -> {layer: {variant: ms}}."""
        _lib.require_cuda(images)
        if self.pw.dtype not in LOWP:
            raise RuntimeError("time_dcn_variants: the fused DeformConv variants exist for bf16 / f16 plans")
        B, _, H, W = images.shape
        with torch.cuda.device(self.device):
            streams, self.streams = self.streams, 1
            try:
                self.forward(images)
            finally:
                self.streams = streams
            plan = self.plan(B, H, W)
            n = len(plan.ops)
            ms = (ctypes.c_float * n)()
            layers = [(p, i) for p, i in plan.dcn_layers if plan.ops[i].Cin % 32 == 0 and not plan.ops[i].reserved & 0x1000]
            saved = [plan.op_array[i].reserved for _, i in layers]
            times = {p: {} for p, _ in layers}
            for name, bits in self.DCN_VARIANTS.items():
                for (_, i), v in zip(layers, saved):
                    plan.op_array[i].reserved = bits | (v & DCN_F16IN)
                runs = []
                for _ in range(reps + 1):                       # (first run of a variant: code-object load, dropped)
                    _lib.check(_lib.lib().h3d_run_ops_timed(plan.op_array, n, _lib.stream_ptr(), ms), "h3d_run_ops_timed")
                    runs.append([ms[i] for _, i in layers])
                runs = list(zip(*runs[1:]))                    # per layer: its `reps` durations
                for (p, _), r in zip(layers, runs):
                    times[p][name] = float(sorted(r)[len(r) // 2])
            for (_, i), v in zip(layers, saved):
                plan.op_array[i].reserved = v
            torch.cuda.synchronize()
        return times

    __call__ = forward


def _tiles_over_slots(om, margin, slots):
    """Share of 16x16 tiles of an offset/mask map [B,h,w,>=18] (channel 2t = dh, 2t+1 = dw of tap t) with more than `slots`
    samples whose bilinear corners leave the tile's apron of the given margin -- the test of csrc/dcn3.hip, on the device."""
    Bn, h, w = om.shape[0], om.shape[1], om.shape[2]
    dev = om.device
    ys = torch.arange(h, device=dev, dtype=torch.float32).view(1, h, 1)
    xs = torch.arange(w, device=dev, dtype=torch.float32).view(1, 1, w)
    y0, x0 = ys - ys % 16 - 1 - margin, xs - xs % 16 - 1 - margin
    HH = 18 + 2 * margin
    miss = torch.zeros(Bn, h, w, device=dev)
    for t in range(9):
        ti, tj = divmod(t, 3)
        h_im, w_im = ys - 1 + ti + om[..., 2 * t], xs - 1 + tj + om[..., 2 * t + 1]
        inside = (h_im > -1) & (w_im > -1) & (h_im < h) & (w_im < w)
        ry, rx = torch.floor(h_im) - y0, torch.floor(w_im) - x0
        ok = (ry >= 0) & (ry + 1 < HH) & (rx >= 0) & (rx + 1 < HH)
        miss += (inside & ~ok).float()
    th, tw = -(-h // 16), -(-w // 16)
    pad = torch.zeros(Bn, th * 16, tw * 16, device=dev)
    pad[:, :h, :w] = miss
    per_tile = pad.view(Bn, th, 16, tw, 16).sum(dim=(2, 4))
    return (per_tile > slots).float().mean()
