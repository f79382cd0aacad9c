"""ctypes binding of csrc/libh3d_hip.so (C ABI: include/h3d.h).

There is deliberately NO fallback: if the HIP library is missing or an entry point is absent,
importing/using the compute path raises.  PyTorch is used by callers only for device memory,
streams and torch.distributed.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libh3d_hip.so")

H3D_F32, H3D_BF16, H3D_F16, H3D_F16X3 = 0, 1, 2, 3
OP_STEM, OP_CONV, OP_DCN, OP_MAXPOOL, OP_UPADD, OP_COPY, OP_HEADS, OP_DCN_V1, OP_DCN_FUSED = 1, 2, 3, 4, 5, 6, 7, 8, 9
OP_CONV_STREAM, OP_DCN_FUSED_F16, OP_DCN_FUSED_STREAM, OP_STEM3, OP_UPDCN_F16 = 10, 11, 12, 13, 14
OP_IM2COL, OP_MAXPOOL3, OP_DEPTH2SPACE = 15, 16, 17
HEADS_MAX = 16
OUT_NHWC, OUT_NCHW_F32, OUT_NHWC_F32, OUT_NHWC_F16 = 0, 1, 2, 3
DCN_INPUT_NHWC, DCN_OUTPUT_NHWC, DCN_F32_MFMA = 1, 2, 4
ABI_VERSION = 3

c_vp, c_i, c_fp = ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p


class H3dOp(ctypes.Structure):
    """Mirror of `struct h3d_op` in include/h3d.h."""
    _fields_ = [
        ("kind", ctypes.c_int32), ("dtype", ctypes.c_int32),
        ("in_", c_vp), ("in2", c_vp), ("w", c_vp), ("bias", c_vp), ("out", c_vp),
        ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
        ("Cin", ctypes.c_int32), ("in_cs", ctypes.c_int32), ("in2_cs", ctypes.c_int32),
        ("Ho", ctypes.c_int32), ("Wo", ctypes.c_int32),
        ("Cout", ctypes.c_int32), ("out_cs", ctypes.c_int32),
        ("ksize", ctypes.c_int32), ("stride", ctypes.c_int32), ("relu", ctypes.c_int32),
        ("out_mode", ctypes.c_int32), ("wrows", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("wexp", ctypes.c_int32), ("wexp2", ctypes.c_int32),
    ]


class _HeadEntry(ctypes.Structure):
    _fields_ = [("w2", c_vp), ("b2", c_vp), ("out", c_vp), ("C", ctypes.c_int32), ("wexp2", ctypes.c_int32)]


class H3dHeadsDesc(ctypes.Structure):
    """Mirror of `struct h3d_heads_desc` (host-side descriptor behind H3D_OP_HEADS)."""
    _fields_ = [("nheads", ctypes.c_int32), ("wexp", ctypes.c_int32), ("head", _HeadEntry * HEADS_MAX)]


class H3dUpdcnDesc(ctypes.Structure):
    """Mirror of `struct h3d_updcn_desc` (host-side descriptor behind H3D_OP_UPDCN_F16)."""
    _fields_ = [("skip", c_vp), ("w_up", c_vp), ("w_off", c_vp), ("skip_cs", ctypes.c_int32), ("reserved", ctypes.c_int32)]


# name -> argtypes (restype is int unless noted); also the export list the CPU test checks
SIGNATURES = {
    "h3d_dcn_v2_forward": [c_vp] * 6 + [c_i] * 14 + [c_vp],
    "h3d_dcn_v2_forward_ws": [c_vp] * 6 + [c_i] * 14 + [c_vp, ctypes.c_size_t, c_vp],
    "h3d_dcn_v2_pack_weights": [c_vp, c_vp, c_i, c_i, c_i, c_vp, c_vp],
    "h3d_dcn_v2_pack_weights_cached": [c_vp, c_vp, c_i, c_i, c_i, c_vp, c_vp, c_vp],
    "h3d_dcn_fused_pack_f32_cached": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_vp, c_vp, c_vp, c_vp, c_vp],
    "h3d_dcn_v2_forward_packed": [c_vp] * 5 + [c_i] * 7 + [c_vp, ctypes.c_size_t, c_vp],
    "h3d_dcn_offset_mask": [c_vp] * 5 + [c_i] * 11 + [c_vp],
    "h3d_build_flags": [],
    "h3d_dcn_fused_ck": [c_i, c_i],
    "h3d_dcn_far_samples": [ctypes.POINTER(H3dOp), c_vp, c_vp],
    "h3d_smpl_pose_heads": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_vp],
    "h3d_smpl_coef_pack": [c_vp, c_vp, c_i, c_i, c_vp, c_vp],
    "h3d_smpl_verts3": [c_vp] * 6 + [c_i] * 5 + [c_vp, c_vp],
    "h3d_smpl_verts3_exact": [c_vp] * 6 + [c_i] * 5 + [c_vp, c_vp],
    "h3d_preprocess": [c_vp, c_i, c_i, c_i, c_i, c_vp, c_vp, c_vp, c_i, c_i, c_vp, c_vp],
    "h3d_run_ops": [ctypes.POINTER(H3dOp), c_i, c_vp],
    "h3d_run_ops_timed": [ctypes.POINTER(H3dOp), c_i, c_vp, c_vp],
    "h3d_op_kernel_name": [ctypes.POINTER(H3dOp), ctypes.c_char_p, c_i],
    "h3d_nchw_f32_to_nhwc": [c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_vp],
    "h3d_nhwc_to_nchw_f32": [c_vp, c_i, c_vp, c_i, c_i, c_i, c_i, c_i, c_vp],
    "h3d_nms_topk": [c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_vp, c_vp, c_vp, c_vp, c_vp],
    "h3d_nms_topk2": [c_vp, c_i, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_vp],
    "h3d_nms_topk_large": [c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp],
    "h3d_nms": [c_vp, c_i, c_i, c_i, c_i, c_vp, c_vp],
    "h3d_topk_merge": [c_vp] * 4 + [c_i] * 3 + [c_vp] * 5 + [c_vp],
    "h3d_gather_feat": [c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_vp, c_vp],
    "h3d_multi_pose_assemble": [c_vp] * 13 + [c_i] * 5 + [c_vp, c_vp],
    "h3d_ctdet_assemble": [c_vp] * 7 + [c_i] * 6 + [c_vp, c_vp],
    "h3d_multi_pose_post_process": [c_vp] * 3 + [c_i] * 5 + [c_vp, c_vp],
    "h3d_smpl_pose": [c_vp] * 5 + [c_i] + [c_vp] * 4 + [c_i, c_vp],
    "h3d_smpl_verts": [c_vp] * 8 + [c_i] * 4 + [c_vp, c_vp],
    "h3d_smpl_verts2": [c_vp] * 7 + [c_i] * 5 + [c_vp, c_vp],
    "h3d_sigmoid_clamp": [c_vp, c_vp, ctypes.c_size_t, c_vp],
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raise loudly when the library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "h3d_amd: HIP library not built (%s). Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C human-3d-reconstruction_amd/csrc`. There is no CPU fallback."
                % LIB_PATH)
        # PyTorch first: its wheel bundles its own HIP / HSA runtime, and libh3d_hip.so must bind to THAT copy (same
        # SONAME once it is loaded).  Loaded the other way round, the process ends up with two runtimes and the first
        # launch fails with "no ROCm-capable device is detected" (seen with build() followed by smoke() in one process).
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        L.h3d_last_error.restype = ctypes.c_char_p
        L.h3d_last_error.argtypes = []
        L.h3d_abi_version.restype = c_i
        if L.h3d_abi_version() != ABI_VERSION:
            raise RuntimeError("h3d_amd: libh3d_hip.so ABI %d != binding ABI %d"
                               % (L.h3d_abi_version(), ABI_VERSION))
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the export is missing
            fn.argtypes = args
            fn.restype = c_i
        L.h3d_dcn_v2_workspace_bytes.argtypes = [c_i] * 5
        L.h3d_dcn_v2_workspace_bytes.restype = ctypes.c_size_t
        L.h3d_dcn_v2_packed_weight_bytes.argtypes = [c_i] * 3
        L.h3d_dcn_v2_packed_weight_bytes.restype = ctypes.c_size_t
        L.h3d_dcn_v2_packed_workspace_bytes.argtypes = [c_i] * 5
        L.h3d_dcn_v2_packed_workspace_bytes.restype = ctypes.c_size_t
        L.h3d_nms_topk_large_workspace_bytes.argtypes = [c_i] * 5
        L.h3d_nms_topk_large_workspace_bytes.restype = ctypes.c_size_t
        _lib = L
    return _lib


def has_extra():
    """True when libh3d_hip.so was built with `make EXTRA=1` (the superseded kernel generations are in: csrc/Makefile)."""
    return bool(lib().h3d_build_flags() & 1)


_ERR_NAMES = {-1: "shape", -2: "dtype", -3: "launch", -4: "unsupported", -5: "argument"}


def check(rc, what):
    if rc != 0:
        msg = lib().h3d_last_error().decode("utf-8", "replace")
        raise RuntimeError("%s failed (%s error): %s" % (what, _ERR_NAMES.get(rc, rc), msg))


def stream_ptr():
    """Raw hipStream_t of torch's current stream (the reference launches on the current
    stream too, dcn_v2_cuda.cu:108)."""
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            # the reference raises "Not implemented on the CPU" (DCNv2/src/dcn_v2.h:38)
            raise RuntimeError("Not implemented on the CPU")
