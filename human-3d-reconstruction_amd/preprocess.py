"""Input pre-process on the device: the val branch of the reference's
`_get_input` (datasets/coco_hp.py:151-212): centre/scale, affine warp to input_res, /255,
standardise, HWC -> CHW.  The warp itself runs in csrc/preprocess.hip (OpenCV's fixed-point
bilinear scheme); the host only builds the 2x3 matrices in float64."""
import numpy as np
import torch

from . import _lib

MEAN = (0.40789654, 0.44719302, 0.47026115)   # datasets/coco_hp.py:35-38 (BGR)
STD = (0.28863828, 0.27408164, 0.27809835)


def _third_point(a, b):
    d = a - b
    return b + np.array([-d[1], d[0]], dtype=np.float32)


def get_affine_transform(center, scale, output_size):
    """utils/image.py:27-62 with rot = 0, shift = 0, inv = 0 -> 2x3 float64 (src -> dst); the 3-point
    solve cv2.getAffineTransform performs is done explicitly."""
    center = np.asarray(center, dtype=np.float32)
    src_w = np.float32(scale)
    dst_w, dst_h = output_size
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0] = center
    src[1] = center + np.array([0, src_w * -0.5], dtype=np.float32)
    dst[0] = [dst_w * 0.5, dst_h * 0.5]
    dst[1] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + np.array([0, dst_w * -0.5], np.float32)
    src[2] = _third_point(src[0], src[1])
    dst[2] = _third_point(dst[0], dst[1])
    A = np.concatenate([src.astype(np.float64), np.ones((3, 1))], axis=1)
    return np.linalg.solve(A, dst.astype(np.float64)).T


def _invert(M):
    """cv::warpAffine's inversion of the 2x3 matrix (double, same operation order)."""
    M = np.array(M, dtype=np.float64).reshape(6).copy()
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11
    M[1] *= -D
    M[3] *= -D
    M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    return M


_CONST = {}      # (B, h, w, res, mean, std, device) -> (minv [B,6] f64, mean [3], std [3]) on the device: uploaded once, not per batch


@torch.no_grad()
def pre_process(images, input_res=512, mean=MEAN, std=STD):
    """images [B,h,w,3] uint8 (BGR, as cv2.imread yields) on the device -> (inp [B,3,res,res] fp32,
    c [B,2] float32, s [B] float32) with c, s the `meta` MultiPoseDetector.run / post-process take."""
    _lib.require_cuda(images)
    if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[3] != 3:
        raise ValueError("pre_process expects a uint8 [B,h,w,3] tensor, got %s %s" % (images.dtype, tuple(images.shape)))
    images = images.contiguous()
    B, h, w, _ = images.shape
    c = np.array([w / 2., h / 2.], dtype=np.float32)
    s = max(w, h) * 1.0
    dev = images.device
    key = (B, h, w, input_res, tuple(mean), tuple(std), str(dev))
    if key not in _CONST:
        minv = np.tile(_invert(get_affine_transform(c, s, [input_res, input_res])), (B, 1))
        if len(_CONST) >= 32:
            _CONST.pop(next(iter(_CONST)))
        _CONST[key] = (torch.from_numpy(minv).to(dev), torch.tensor(mean, dtype=torch.float32, device=dev),
                       torch.tensor(std, dtype=torch.float32, device=dev))
        torch.cuda.current_stream(dev).synchronize()         # (the uploads are complete before any other stream may use them)
    minv_t, mean_t, std_t = _CONST[key]
    out = torch.empty(B, 3, input_res, input_res, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().h3d_preprocess(_lib.ptr(images), B, h, w, 3 * w, _lib.ptr(minv_t), _lib.ptr(mean_t),
                                         _lib.ptr(std_t), input_res, input_res, _lib.ptr(out), _lib.stream_ptr()),
               "preprocess")
    return out, np.tile(c, (B, 1)), np.full((B,), s, dtype=np.float32)
