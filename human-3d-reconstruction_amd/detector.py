"""multi_pose detector: the inference call order of the reference's
`HMRTrainer.run_epoch('val')` -> `save_result` (trains/trainer.py:253-261, 456-469), i.e.

    output = model(images)[0]
    output['hm'], output['hm_hp'] = _sigmoid(...)           (loss module side effect, :93,:127)
    dets = multi_pose_decode(hm, wh, hps, reg, hm_hp, hp_offset, K)          [B,K,40]
    dets_out = multi_pose_post_process(dets, c, s, out_h, out_w)             (optional)

plus the north_star's SMPL stage: per-detection pose/shape read from two extra heads at the
detection centres -> 6890-vertex LBS mesh.  One process per GPU; `gather_detections` is the
single RCCL collective of the sharded path.
"""
import torch

from . import _lib, decode, smpl as _smpl
from .model import create_model
from .utils import _transpose_and_gather_feat

MULTI_POSE_HEADS = {"hm": 1, "wh": 2, "hps": 34, "reg": 2, "hm_hp": 17, "hp_offset": 2}   # opts.py:248-258
CTDET_HEADS = {"hm": 80, "wh": 2, "reg": 2}                                                  # opts.py:241-247
SMPL_HEADS = {"pose": 72, "shape": 10}


class Opt:
    """The option names the path consumes (reference opts.py; defaults of the multi_pose task).

    dtype (extension; the reference computes in fp32): the arithmetic of the launch plan.
      "bf16"  bf16 activations and filters, fp32 accumulation (DeformConv samples and blends in fp16).  The throughput headline
              (north_star prices the roofline in bf16).  On noise-like random-weight heat maps the top-k ORDER shuffles
              (head error ~0.75 on maps of std 1-2.5: set overlap 0.72, positional agreement 0.04).
      "f16"   fp16 activations and filters at the same MFMA rate, saturating epilogues: ~2 % slower, head error 7x smaller
              (0.10), set overlap 0.945, agreement 0.40 -- run this when the ranking matters more than 2 % of throughput.
      "f32"   parity mode: exact fmaf chains on v_mfma_f32_32x32x2_f32, 1/8 of the throughput; heads within 2e-4 of the
              reference's own output, top-k indices identical to it (tests/test_gpu_fullsize.py, tests/golden/e2e_plain_512.npz).
      "f16x3" the parity arithmetic on the fp16 matrix cores (round 5): fp32 activations in memory, every fp32 product of a
              contraction as three fp16 MFMAs on split operands (x = hi + lo; hi.hi + hi.lo + lo.hi, fp32 accumulation) -- fp32-level
              heads and the same top-k indices as "f32" at ~3x its rate (csrc/common.h ET<x3_t>)."""

    def __init__(self, **kw):
        self.task = "multi_pose"
        self.arch = "dla_34"              # opts.py:61-63: dla_34 | hourglass | resdcn_101
        self.head_conv = 256              # (the published code uses 64 for the non-DLA backbones: create_model maps 256 -> 64 for resdcn)
        self.down_ratio = 4
        self.K = 100
        self.not_use_dcn = False
        self.input_h = self.input_w = 512
        self.reg_offset = True
        self.hm_hp = True
        self.reg_hp_offset = True
        self.flip_test = False        # opts.py:89 (declared by the reference, used by nothing there): see MultiPoseDetector.run
        self.flip_idx = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]       # opts.py:250
        self.smpl = False             # north_star extension: pose/shape heads + LBS
        self.smpl_people = None       # meshes per image (None = K)
        self.dtype = "bf16"
        for k, v in kw.items():
            setattr(self, k, v)
        self.output_h, self.output_w = self.input_h // self.down_ratio, self.input_w // self.down_ratio
        if self.task == "ctdet":                               # opts.py:241-247
            self.num_classes = getattr(self, "num_classes", 80)
            self.cat_spec_wh = getattr(self, "cat_spec_wh", False)
            heads = {"hm": self.num_classes, "wh": 2 if not self.cat_spec_wh else 2 * self.num_classes}
            if self.reg_offset:
                heads["reg"] = 2
        elif self.task == "multi_pose":                        # opts.py:248-258
            self.num_classes = 1
            heads = {"hm": 1, "wh": 2, "hps": 34}
            if self.reg_offset:
                heads["reg"] = 2
            if self.hm_hp:
                heads["hm_hp"] = 17
            if self.reg_hp_offset:
                heads["hp_offset"] = 2
            if self.smpl:
                heads.update(SMPL_HEADS)
        else:
            raise ValueError("task not defined!")              # opts.py:260
        self.heads = heads


class MultiPoseDetector:
    def __init__(self, opt, state_dict=None, smpl_model=None, device="cuda"):
        if opt.task != "multi_pose":
            raise ValueError("task not defined!")          # trains/trainer.py:472
        self.opt = opt
        self.device = torch.device(device)
        self.model = create_model(opt.arch, opt.heads, opt.head_conv, opt.not_use_dcn, dtype=opt.dtype)
        if state_dict is not None:
            self.model.load_state_dict(state_dict, strict=True)
        self.model.to(self.device).eval()
        self.smpl_model = smpl_model
        self.fused_tail = True         # False: the decode / SMPL tail as round 3 issued it (A/B and bit-identity tests)
        if opt.smpl and smpl_model is None:
            self.smpl_model = _smpl.SMPLModel.synthetic()

    @torch.no_grad()
    def run(self, images, meta=None, slot=0, graph=False):
        """images [B,3,H,W] fp32 on the device -> dict(dets [B,K,40], inds [B,K], optional
        verts [B,N,6890,3] / joints, optional results [B,K,39] in image px when meta={'c','s'}).
        slot: plan-buffer copy (model.forward); consecutive batches issued on different HIP streams alternate slots.
        graph: replay the whole step (network, decode, SMPL: ~75 launches and ~30 host-side allocations) as ONE hipGraph
        captured at the first call of this (shape, slot) -- for small per-GPU shards, where issuing the launches from
        Python takes longer than the GPU needs to run them.  Same kernels, same work; the returned tensors are the
        graph's static outputs (overwritten by the next replay of the same slot: clone to keep)."""
        if graph and meta is None:
            return self._run_graph(images, slot)
        opt = self.opt
        if opt.flip_test:
            out = self._flip_test_heads(images, slot)
        else:
            out = self.model(images, slot)[-1]    # (Hourglass returns one dict per stack: inference uses the last)
        dets, aux = decode.multi_pose_decode_logits(
            out["hm"], out["wh"], out["hps"], reg=out.get("reg") if opt.reg_offset else None,
            hm_hp=out.get("hm_hp") if opt.hm_hp else None,
            hp_offset=out.get("hp_offset") if opt.reg_hp_offset else None, K=opt.K, return_aux=True)
        res = {"dets": dets, "inds": aux["inds"], "heads": out}
        if opt.smpl:
            n = opt.smpl_people or opt.K
            B = aux["inds"].shape[0]
            fusable = all(out[k].dtype == torch.float32 and out[k].is_contiguous() for k in ("pose", "shape"))   # (else: the gather path, any strides)
            if B * n >= 64 and self.fused_tail and fusable:
                # per-detection pose / shape read from the two extra heads at the detection centres INSIDE the SMPL pose kernel, which
                # also writes the blend-shape operand: two launches instead of six (gathers x 2, index copy, pose, pack, verts)
                verts, joints = _smpl.lbs_from_heads(self.smpl_model, out["pose"], out["shape"], aux["inds"], n, return_joints=True,
                                                     exact=opt.dtype in ("f32", "f16x3"))
            else:
                inds = aux["inds"][:, :n].contiguous()
                thetas = _transpose_and_gather_feat(out["pose"], inds).view(B * n, 72)
                betas = _transpose_and_gather_feat(out["shape"], inds).view(B * n, 10)
                # f32 (parity-mode) detectors keep all six products of the blend-shape split (fp32-level accuracy end to end)
                verts, joints = _smpl.lbs(self.smpl_model, betas, thetas, return_joints=True,
                                          kernel="auto_exact" if opt.dtype in ("f32", "f16x3") else "auto")
            res["verts"] = verts.view(B, n, -1, 3)
            res["joints"] = joints.view(B, n, 24, 3)
        if meta is not None:
            res["results"] = multi_pose_post_process(dets, meta["c"], meta["s"], out["hm"].shape[2],
                                                     out["hm"].shape[3])
        return res


def _run_graph(self, images, slot):
    """hipGraph capture / replay of `run` (torch.cuda.CUDAGraph; the C ABI launches on torch's current stream, which is the
    capture stream inside the context).  One graph per (shape, slot); the input lives in a static buffer the caller's
    batch is copied into unless it already IS that buffer's address."""
    _lib.require_cuda(images)
    if not hasattr(self, "_graphs"):
        self._graphs = {}
    key = (tuple(images.shape), slot)
    ent = self._graphs.get(key)
    if ent is None:
        static_in = images.detach().clone().float().contiguous()
        self.run(static_in, slot=slot)                       # eager warm-up: plans, code objects, SMPL model upload
        torch.cuda.current_stream().synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            res = self.run(static_in, slot=slot)
        ent = self._graphs[key] = (g, static_in, res)
    g, static_in, res = ent
    if images.data_ptr() != static_in.data_ptr():
        static_in.copy_(images)
    g.replay()
    return res


MultiPoseDetector._run_graph = _run_graph


class CtdetDetector:
    """The `ctdet` branch of the reference's task dispatch (trains/trainer.py:444-455): model -> `_sigmoid(hm)`
    (loss-module side effect, trainer.py:93) -> `ctdet_decode(hm, wh, reg, cat_spec_wh, K)` [B,K,6] -> optional
    `ctdet_post_process` (utils/post_process.py:24-38).  Same engine, kernels and factory as the multi_pose detector;
    only the head table and the assemble step differ."""

    def __init__(self, opt, state_dict=None, device="cuda"):
        if opt.task != "ctdet":
            raise ValueError("task not defined!")
        self.opt = opt
        self.device = torch.device(device)
        self.model = create_model(opt.arch, opt.heads, opt.head_conv, opt.not_use_dcn, dtype=opt.dtype)
        if state_dict is not None:
            self.model.load_state_dict(state_dict, strict=True)
        self.model.to(self.device).eval()

    @torch.no_grad()
    def run(self, images, meta=None, slot=0):
        """images [B,3,H,W] fp32 on the device -> dict(dets [B,K,6] = box, score, class; heads; optional results =
        the reference's per-image {class id (1-based): [[x1, y1, x2, y2, score], ...]} when meta={'c','s'})."""
        from .utils import _sigmoid
        opt = self.opt
        out = self.model(images, slot)[-1]
        hm = _sigmoid(out["hm"].clone())
        dets = decode.ctdet_decode(hm, out["wh"], reg=out.get("reg") if opt.reg_offset else None,
                                   cat_spec_wh=opt.cat_spec_wh, K=opt.K)
        res = {"dets": dets, "heads": out}
        if meta is not None:
            res["results"] = ctdet_post_process(dets, meta["c"], meta["s"], out["hm"].shape[2], out["hm"].shape[3],
                                                out["hm"].shape[1])
        return res


def make_detector(opt, state_dict=None, device="cuda", **kw):
    """Task plugin dispatch by string, as the reference (`opt.task`, trains/trainer.py:331-342, 444-472)."""
    if opt.task == "multi_pose":
        return MultiPoseDetector(opt, state_dict, device=device, **kw)
    if opt.task == "ctdet":
        return CtdetDetector(opt, state_dict, device=device)
    raise ValueError("task not defined!")


def ctdet_post_process(dets, c, s, h, w, num_classes):
    """reference utils/post_process.py:24-38: dets [B,K,6] (output-res px) -> per image {1-based class id:
    [[x1, y1, x2, y2, score], ...]} in original-image pixels.  The affine map runs on the device
    (h3d_multi_pose_post_process with J = 0); the per-class grouping is the reference's host-side dict."""
    _lib.require_cuda(dets)
    dets = dets.contiguous().float()
    B, K, D = dets.shape
    if D != 6:
        raise RuntimeError("ctdet_post_process: dets [B,K,6] expected, got %s" % (tuple(dets.shape),))
    c = torch.as_tensor(c, dtype=torch.float32, device=dets.device).contiguous().view(B, 2)
    s = torch.as_tensor(s, dtype=torch.float32, device=dets.device).contiguous().view(-1)
    if s.numel() == 2 * B:
        s = s.view(B, 2)[:, 0].contiguous()
    out = torch.empty(B, K, 5, dtype=torch.float32, device=dets.device)
    _lib.check(_lib.lib().h3d_multi_pose_post_process(_lib.ptr(dets), _lib.ptr(c), _lib.ptr(s), B, K, 0, int(h), int(w),
                                                      _lib.ptr(out), _lib.stream_ptr()), "ctdet_post_process")
    boxes = out.cpu().numpy()
    classes = dets[:, :, 5].cpu().numpy()
    ret = []
    for i in range(B):
        ret.append({j + 1: boxes[i, classes[i] == j].tolist() for j in range(num_classes)})
    return ret


def _flip_test_heads(self, images, slot=0):
    """`--flip_test` (opts.py:89; the reference declares the option and ships the helpers `flip_tensor / flip_lr /
    flip_lr_off`, models/utils.py:29-51, but no caller): the published CenterNet multi_pose recipe the helpers were written
    for -- run the batch and its mirror image, average `hm` and `wh` with the mirrored maps flipped back, `hps` with
    `flip_lr_off` (x offsets negated, left/right joints swapped), `hm_hp` with `flip_lr`; `reg` / `hp_offset` come from the
    un-flipped pass.  Heat maps are averaged AFTER `_sigmoid`, so the returned `hm` / `hm_hp` are logits of the average."""
    from .utils import _sigmoid, flip_lr, flip_lr_off, flip_tensor
    B = images.shape[0]
    out = self.model(torch.cat([images, torch.flip(images, [3])], 0).contiguous(), slot)[-1]
    idx = self.opt.flip_idx
    res = {}
    for k, v in out.items():
        a, b = v[:B], v[B:]
        if k in ("hm", "hm_hp"):
            pa = _sigmoid(a.clone())
            pb = _sigmoid(b.clone())
            pb = flip_tensor(pb) if k == "hm" else flip_lr(pb, idx)
            p = (pa + pb) / 2
            res[k] = torch.log(p / (1 - p))
        elif k == "wh":
            res[k] = (a + flip_tensor(b)) / 2
        elif k == "hps":
            res[k] = (a + flip_lr_off(b, idx)) / 2
        else:
            res[k] = a.contiguous()
    return res


MultiPoseDetector._flip_test_heads = _flip_test_heads


def run_frames(detector, frames):
    """Raw uint8 BGR frames [B,h,w,3] on the device -> `MultiPoseDetector.run` results with `results` in
    original-image pixels: pre-process (datasets/coco_hp.py:151-212 val branch, csrc/preprocess.hip) ->
    network -> decode -> post-process, the order of the reference's demo / test loop."""
    from . import preprocess
    inp, c, s = preprocess.pre_process(frames, input_res=detector.opt.input_h)
    return detector.run(inp, meta={"c": c, "s": s})


def multi_pose_post_process(dets, c, s, h, w):
    """reference utils/post_process.py:41-52 on the device: dets [B,K,40], c [B,2], s [B] ->
    [B,K,39] (bbox, score, 17 keypoints in original-image pixels)."""
    _lib.require_cuda(dets)
    dets = dets.contiguous().float()
    B, K, D = dets.shape
    J = (D - 6) // 2
    c = torch.as_tensor(c, dtype=torch.float32, device=dets.device).contiguous().view(B, 2)
    s = torch.as_tensor(s, dtype=torch.float32, device=dets.device).contiguous().view(-1)
    if s.numel() == 2 * B:                                  # per-axis scale: the reference uses scale[0]
        s = s.view(B, 2)[:, 0].contiguous()
    out = torch.empty(B, K, 5 + 2 * J, dtype=torch.float32, device=dets.device)
    _lib.check(_lib.lib().h3d_multi_pose_post_process(_lib.ptr(dets), _lib.ptr(c), _lib.ptr(s), B, K, J, int(h), int(w),
                                                      _lib.ptr(out), _lib.stream_ptr()), "multi_pose_post_process")
    return out


def gather_detections(dets, group=None, n_images=None, force=False):
    """All ranks' [B_rank,K,40] -> [sum B_rank,K,40] with ONE all-gather (RCCL over xGMI when the backend is
    nccl; gloo in the CPU tests).  Nothing else crosses ranks: images are independent, weights are replicated
    (the reference's DataParallel scatter of dim 0, trainer.py:176).

    n_images: the global batch that `shard_batch` split.  None = every rank holds the same number of images.
    When the split is uneven (n_images % world != 0: the lowest ranks hold one image more) every rank pads its
    shard to ceil(n_images / world) rows, so the collective stays ONE fixed-size all_gather_into_tensor, and the
    pad rows are dropped afterwards.
    force: issue the collective even in a group of ONE rank (a rehearsal of the RCCL path on a one-GPU box: communicator
    set-up and the all-gather itself run on the device; `bench.py --rehearse-collective`)."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return dets
    world = dist.get_world_size(group)
    dets = dets.contiguous()
    if n_images is None or n_images % world == 0:
        if n_images is not None and dets.shape[0] != n_images // world:
            raise RuntimeError("gather_detections: this rank holds %d images, shard_batch(%d, ., %d) gives %d"
                               % (dets.shape[0], n_images, world, n_images // world))
        out = torch.empty((world * dets.shape[0],) + tuple(dets.shape[1:]), dtype=dets.dtype, device=dets.device)
        dist.all_gather_into_tensor(out, dets, group=group)
        return out
    rows = -(-n_images // world)
    lo, hi = shard_batch(n_images, dist.get_rank(group), world)
    if dets.shape[0] != hi - lo:
        raise RuntimeError("gather_detections: this rank holds %d images, shard_batch(%d, %d, %d) gives %d"
                           % (dets.shape[0], n_images, dist.get_rank(group), world, hi - lo))
    mine = dets.new_zeros((rows,) + tuple(dets.shape[1:]))
    mine[:hi - lo] = dets
    out = torch.empty((world * rows,) + tuple(dets.shape[1:]), dtype=dets.dtype, device=dets.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    out = out.view((world, rows) + tuple(dets.shape[1:]))
    sizes = [shard_batch(n_images, r, world) for r in range(world)]
    return torch.cat([out[r, :b - a] for r, (a, b) in enumerate(sizes)], 0)


def shard_batch(n_images, rank, world):
    """Contiguous image range of `rank` (B/world contiguous images per rank;
    remainders go to the lowest ranks)."""
    base, rem = divmod(n_images, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
