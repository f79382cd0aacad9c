"""SMPL pose/shape -> linear-blend-skinning mesh stage (north_star).

There is NO SMPL code, model file or test in the reference snapshot, and the real
SMPL model is licence-gated, so this stage follows the published formulation (Loper et al.
2015; axis-angle convention of the public `smplx` package) and ships a *synthetic* model of
the true tensor shapes for benchmarking.  A real model drops in through `SMPLModel.from_npz`.
Parity for this stage is "unpinned by the reference" (DESIGN.md).

Device compute: csrc/smpl.hip via `h3d_smpl_*` (include/h3d.h).
"""
import numpy as np

from . import synth

NUM_VERTS = 6890
NUM_JOINTS = 24
NUM_BETAS = 10
NUM_POSE_FEAT = 207
PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19,
                    20, 21], dtype=np.int32)


class SMPLModel:
    """Host container of the SMPL tensors (float32 numpy) + lazily-built device pack."""

    def __init__(self, v_template, shapedirs, posedirs, J_regressor, weights, parents=PARENTS):
        self.v_template = np.ascontiguousarray(v_template, np.float32)      # [V,3]
        self.shapedirs = np.ascontiguousarray(shapedirs, np.float32)        # [V,3,10]
        self.posedirs = np.ascontiguousarray(posedirs, np.float32)          # [V,3,207]
        self.J_regressor = np.ascontiguousarray(J_regressor, np.float32)    # [24,V]
        self.weights = np.ascontiguousarray(weights, np.float32)            # [V,24]
        self.parents = np.ascontiguousarray(parents, np.int32)
        V = self.v_template.shape[0]
        assert self.shapedirs.shape == (V, 3, NUM_BETAS)
        assert self.posedirs.shape == (V, 3, NUM_POSE_FEAT)
        assert self.J_regressor.shape == (NUM_JOINTS, V)
        assert self.weights.shape == (V, NUM_JOINTS)
        self._dev = None

    @classmethod
    def synthetic(cls, seed=0, num_verts=NUM_VERTS):
        """Seeded model with the true shapes: body-sized template, small blend shapes,
        row-stochastic joint regressor, 4-sparse skinning weights (as the real model has)."""
        V = num_verts
        v = synth.uniform("smpl.v_template", (V, 3), -1.0, 1.0, seed) * np.array([0.45, 0.9, 0.15], np.float32)
        S = synth.uniform("smpl.shapedirs", (V, 3, NUM_BETAS), -0.03, 0.03, seed)
        P = synth.uniform("smpl.posedirs", (V, 3, NUM_POSE_FEAT), -0.01, 0.01, seed)
        Jr = synth.uniform01("smpl.J_regressor", (NUM_JOINTS, V), seed)
        Jr = np.where(Jr > 0.97, Jr, 0.0)
        Jr = (Jr / Jr.sum(1, keepdims=True)).astype(np.float32)
        u = synth.uniform01("smpl.weights", (V, NUM_JOINTS), seed)
        kth = np.sort(u, axis=1)[:, -4][:, None]
        W = np.where(u >= kth, u, 0.0)
        W = (W / W.sum(1, keepdims=True)).astype(np.float32)
        return cls(v, S, P, Jr, W)

    @classmethod
    def from_npz(cls, path):
        d = np.load(path, allow_pickle=False)
        parents = d["parents"] if "parents" in d else PARENTS
        return cls(d["v_template"], d["shapedirs"], d["posedirs"], d["J_regressor"], d["weights"],
                   parents)

    def numpy_dict(self):
        return {"v_template": self.v_template, "shapedirs": self.shapedirs,
                "posedirs": self.posedirs, "J_regressor": self.J_regressor,
                "weights": self.weights, "parents": self.parents}


# ----------------------------------------------------------------------------------------------
# device side (csrc/smpl.hip through the C ABI)
def _device_pack(model, device, nnz=None):
    """Layouts the kernels want (include/h3d.h section 4): K-major blend shapes, the joint
    regressor pre-contracted with template/shapedirs (float64 on the host), sparse LBS weights."""
    import torch
    V = model.v_template.shape[0]
    W = model.weights
    if nnz is None:
        nnz = int(max(1, (W != 0).sum(1).max()))
    order = np.argsort(-W, axis=1, kind="stable")[:, :nnz].astype(np.int32)
    wsel = np.take_along_axis(W, order.astype(np.int64), axis=1).astype(np.float32)
    Jr = model.J_regressor.astype(np.float64)
    j_template = (Jr @ model.v_template.astype(np.float64)).astype(np.float32)                 # [24,3]
    j_dirs = np.einsum("jv,vck->jck", Jr, model.shapedirs.astype(np.float64)).astype(np.float32)  # [24,3,10]
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    Vpad = ((V + 63) // 64) * 64

    def soa(a):                 # [..., V] -> [..., Vpad] zero padded
        out = np.zeros(a.shape[:-1] + (Vpad,), np.float32)
        out[..., :V] = a
        return out
    return {
        "V": V, "Vpad": Vpad, "nnz": nnz,
        # struct-of-arrays with padded rows: [3][Vpad] and [k][3][Vpad] -- a wave's load of one
        # coordinate is one contiguous, 16-byte aligned row
        "v_template": to(soa(model.v_template.T)),
        "shapedirsT": to(soa(model.shapedirs.transpose(2, 1, 0))),
        "posedirsT": to(soa(model.posedirs.transpose(2, 1, 0))),
        "j_template": to(j_template.reshape(-1)),
        "j_shapedirs": to(j_dirs.reshape(NUM_JOINTS * 3, NUM_BETAS)),
        "parents": to(model.parents.astype(np.int32)),
        "lbs_idx": to(order), "lbs_w": to(wsel),
        # generation 3 (MFMA): K-contiguous direction rows [10 shape | 207 pose | 0], each value as three bf16 terms
        "dirsK3": _dirs_k3(model, Vpad, device),
    }


def _dirs_k3(model, Vpad, device, KP=224):
    """[3][Vpad][KP/16][3 terms][16] bf16: x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)
    (round-to-nearest-even; exact to 24 significant bits) -- the operand format of csrc/smpl.hip smpl_verts3."""
    import torch
    V = model.v_template.shape[0]
    d = np.zeros((3, Vpad, KP), np.float32)
    d[:, :V, :NUM_BETAS] = model.shapedirs.transpose(1, 0, 2)                      # [V,3,10] -> [3,V,10]
    d[:, :V, NUM_BETAS:NUM_BETAS + NUM_POSE_FEAT] = model.posedirs.transpose(1, 0, 2)
    x = torch.from_numpy(d)
    h = x.to(torch.bfloat16)
    r1 = x - h.float()
    m = r1.to(torch.bfloat16)
    lo = (r1 - m.float()).to(torch.bfloat16)
    out = torch.stack([t.reshape(3, Vpad, KP // 16, 16) for t in (h, m, lo)], dim=3)     # [3][Vpad][14][3][16]
    return out.contiguous().to(device)


def lbs(model, betas, thetas, return_joints=False, kernel="auto"):
    """betas [P,10], thetas [P,72] (CUDA fp32) -> vertices [P,V,3] (and posed joints [P,24,3]).
    kernel: "gen3" = blend shapes on the matrix cores (3-term bf16 split; the three 2^-16 products dropped: 2e-6 abs on the displacement),
    "gen3x" = the same with all six products (2^-24: what the f32 parity-mode detectors run), "gen2" = LDS-streamed vector kernel (all
    need <= 4 skinning weights per vertex), "gen1" = register kernel, "auto" = gen3 when applicable and P >= 64, "auto_exact" = gen3x."""
    import torch
    from . import _lib
    _lib.require_cuda(betas, thetas)
    dev = betas.device
    if model._dev is None or model._dev["v_template"].device != dev:
        model._dev = _device_pack(model, dev)
    d = model._dev
    betas = betas.contiguous().float()
    thetas = thetas.contiguous().float().view(betas.shape[0], 72)
    P = betas.shape[0]
    pf = torch.empty(P, NUM_POSE_FEAT, dtype=torch.float32, device=dev)
    A = torch.empty(P, NUM_JOINTS, 12, dtype=torch.float32, device=dev)
    joints = torch.empty(P, NUM_JOINTS, 3, dtype=torch.float32, device=dev)
    verts = torch.empty(P, d["V"], 3, dtype=torch.float32, device=dev)
    L = _lib.lib()
    gen3 = kernel in ("gen3", "gen3x") or (kernel in ("auto", "auto_exact") and d["nnz"] <= 4 and P >= 64)
    exact = kernel in ("gen3x", "auto_exact")
    gen2 = kernel == "gen2"
    Ppad = ((P + 127) // 128) * 128
    coefT = torch.zeros(NUM_BETAS + NUM_POSE_FEAT, Ppad, dtype=torch.float32, device=dev) if gen2 else None
    with torch.cuda.device(dev):
        st = _lib.stream_ptr()              # the current stream of `dev` (not of whatever device was current at the call)
        _lib.check(L.h3d_smpl_pose(_lib.ptr(betas), _lib.ptr(thetas), _lib.ptr(d["j_template"]),
                                   _lib.ptr(d["j_shapedirs"]), _lib.ptr(d["parents"]), P, _lib.ptr(pf),
                                   _lib.ptr(A), _lib.ptr(joints), _lib.ptr(coefT), Ppad, st), "smpl_pose")
        if gen3:
            coefK = torch.empty(Ppad, 14, 3, 16, dtype=torch.bfloat16, device=dev)
            _lib.check(L.h3d_smpl_coef_pack(_lib.ptr(betas), _lib.ptr(pf), P, Ppad, _lib.ptr(coefK), st), "smpl_coef_pack")
            fn = L.h3d_smpl_verts3_exact if exact else L.h3d_smpl_verts3
            _lib.check(fn(_lib.ptr(coefK), _lib.ptr(A), _lib.ptr(d["v_template"]), _lib.ptr(d["dirsK3"]),
                          _lib.ptr(d["lbs_idx"]), _lib.ptr(d["lbs_w"]), d["nnz"], P, Ppad, d["V"], d["Vpad"],
                          _lib.ptr(verts), st), "smpl_verts3")
        elif gen2:
            _lib.check(L.h3d_smpl_verts2(_lib.ptr(coefT), _lib.ptr(A), _lib.ptr(d["v_template"]),
                                         _lib.ptr(d["shapedirsT"]), _lib.ptr(d["posedirsT"]), _lib.ptr(d["lbs_idx"]),
                                         _lib.ptr(d["lbs_w"]), d["nnz"], P, Ppad, d["V"], d["Vpad"], _lib.ptr(verts), st),
                       "smpl_verts2")
        else:
            _lib.check(L.h3d_smpl_verts(_lib.ptr(betas), _lib.ptr(pf), _lib.ptr(A), _lib.ptr(d["v_template"]),
                                        _lib.ptr(d["shapedirsT"]), _lib.ptr(d["posedirsT"]), _lib.ptr(d["lbs_idx"]),
                                        _lib.ptr(d["lbs_w"]), d["nnz"], P, d["V"], d["Vpad"], _lib.ptr(verts), st),
                       "smpl_verts")
    return (verts, joints) if return_joints else verts


def lbs_from_heads(model, pose_map, shape_map, inds, n, return_joints=False, exact=False):
    """The detector's SMPL stage straight from the network's outputs: pose_map [B,72,H,W], shape_map [B,10,H,W] (contiguous fp32
    head maps), inds [B,K] int64 (the decode's centre indices), the first `n` detections of every image -> vertices
    [B*n,V,3] (and joints [B*n,24,3]).  Two launches -- `h3d_smpl_pose_heads` (gathers + Rodrigues + kinematic chain + the
    generation-3 coefficient operand) and `h3d_smpl_verts3` -- instead of the five of `_transpose_and_gather_feat` x 2 + `lbs`;
    bit-identical to them (tests/test_gpu_smpl.py).  Needs <= 4 skinning weights per vertex (generation 3)."""
    import torch
    from . import _lib
    _lib.require_cuda(pose_map, shape_map, inds)
    dev = pose_map.device
    if model._dev is None or model._dev["v_template"].device != dev:
        model._dev = _device_pack(model, dev)
    d = model._dev
    if d["nnz"] > 4:
        raise RuntimeError("lbs_from_heads: more than 4 skinning weights per vertex (use lbs)")
    if (pose_map.dtype != torch.float32 or shape_map.dtype != torch.float32 or not pose_map.is_contiguous() or not shape_map.is_contiguous()
            or inds.dtype != torch.int64 or not inds.is_contiguous()):
        raise RuntimeError("lbs_from_heads: contiguous fp32 head maps and int64 indices expected")
    B, K = inds.shape
    HW = pose_map.shape[2] * pose_map.shape[3]
    if pose_map.shape[:2] != (B, 72) or shape_map.shape[:2] != (B, NUM_BETAS) or shape_map.shape[2:] != pose_map.shape[2:] or not 0 < n <= K:
        raise RuntimeError("lbs_from_heads: pose [B,72,H,W], shape [B,10,H,W], inds [B,K], 0 < n <= K")
    P = B * n
    Ppad = ((P + 127) // 128) * 128
    pf = torch.empty(P, NUM_POSE_FEAT, dtype=torch.float32, device=dev)
    A = torch.empty(P, NUM_JOINTS, 12, dtype=torch.float32, device=dev)
    joints = torch.empty(P, NUM_JOINTS, 3, dtype=torch.float32, device=dev)
    verts = torch.empty(P, d["V"], 3, dtype=torch.float32, device=dev)
    coefK = torch.empty(Ppad, 14, 3, 16, dtype=torch.bfloat16, device=dev)
    L = _lib.lib()
    with torch.cuda.device(dev):
        st = _lib.stream_ptr()              # the current stream of `dev`
        _lib.check(L.h3d_smpl_pose_heads(_lib.ptr(pose_map), _lib.ptr(shape_map), _lib.ptr(inds), B, K, n, HW, _lib.ptr(d["j_template"]),
                                         _lib.ptr(d["j_shapedirs"]), _lib.ptr(d["parents"]), None, _lib.ptr(pf), _lib.ptr(A), _lib.ptr(joints),
                                         _lib.ptr(coefK), Ppad, st), "smpl_pose_heads")
        fn = L.h3d_smpl_verts3_exact if exact else L.h3d_smpl_verts3
        _lib.check(fn(_lib.ptr(coefK), _lib.ptr(A), _lib.ptr(d["v_template"]), _lib.ptr(d["dirsK3"]), _lib.ptr(d["lbs_idx"]), _lib.ptr(d["lbs_w"]),
                      d["nnz"], P, Ppad, d["V"], d["Vpad"], _lib.ptr(verts), st), "smpl_verts3")
    return (verts, joints) if return_joints else verts
