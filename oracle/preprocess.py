"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's input pre-process (val branch).

Reference lines followed:
  datasets/coco_hp.py:151-212  _get_input: c = image centre, s = max(w, h), rot = 0,
                               warpAffine(INTER_LINEAR) to res x res, /255, (x - mean) / std, HWC -> CHW
  datasets/coco_hp.py:35-38    mean / std (BGR order, float32)
  utils/image.py:27-62         get_affine_transform (restated in oracle/post_process.py)

cv2 is absent from this image, so `warp_affine` restates OpenCV 4's fixed-point bilinear
warp (modules/imgproc/src/imgwarp.cpp: cv::warpAffine inverts M in double; WarpAffineInvoker builds
X = (cvRound((M01*y+M02)*1024) + 16 + cvRound(M00*x*1024)) >> 5; remapBilinear blends the 2x2
neighbourhood with weights (32-fx)(32-fy)*32 ... (sum 2^15) and rounds with +2^14 >> 15; constant
border 0).  PARITY UNPINNED against cv2 itself; pinned by analytic identities in
tests/test_oracle_preprocess.py (identity warp, integer shifts, exact half-pixel averages).
"""
import numpy as np

from oracle.post_process import get_affine_transform

MEAN = np.array([0.40789654, 0.44719302, 0.47026115], dtype=np.float32).reshape(1, 1, 3)
STD = np.array([0.28863828, 0.27408164, 0.27809835], dtype=np.float32).reshape(1, 1, 3)


def invert_affine(M):
    """cv::warpAffine's in-place inversion of the 2x3 matrix (double)."""
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


def warp_affine(img, M, dsize):
    """img [h,w,3] uint8, M 2x3 (src -> dst), dsize (width, height) -> [height,width,3] uint8."""
    h, w = img.shape[:2]
    Mi = invert_affine(M)
    dw, dh = dsize
    xs = np.arange(dw, dtype=np.float64)
    ys = np.arange(dh, dtype=np.float64)
    ad = np.rint(Mi[0] * xs * 1024.0).astype(np.int64)
    bd = np.rint(Mi[3] * xs * 1024.0).astype(np.int64)
    X0 = np.rint((Mi[1] * ys + Mi[2]) * 1024.0).astype(np.int64) + 16
    Y0 = np.rint((Mi[4] * ys + Mi[5]) * 1024.0).astype(np.int64) + 16
    X = (X0[:, None] + ad[None, :]) >> 5
    Y = (Y0[:, None] + bd[None, :]) >> 5
    sx, sy, fx, fy = X >> 5, Y >> 5, X & 31, Y & 31

    def px(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        v = img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.int64)
        return v * ok[..., None]

    w00 = ((32 - fx) * (32 - fy) * 32)[..., None]
    w01 = (fx * (32 - fy) * 32)[..., None]
    w10 = ((32 - fx) * fy * 32)[..., None]
    w11 = (fx * fy * 32)[..., None]
    acc = w00 * px(sy, sx) + w01 * px(sy, sx + 1) + w10 * px(sy + 1, sx) + w11 * px(sy + 1, sx + 1)
    return ((acc + 16384) >> 15).astype(np.uint8)


def get_input(img, res=512):
    """coco_hp.py:151-212, split != 'train': -> (inp [3,res,res] float32, c, s)."""
    c = np.array([img.shape[1] / 2., img.shape[0] / 2.], dtype=np.float32)
    s = max(img.shape[1], img.shape[0]) * 1.0
    trans = get_affine_transform(c, s, 0, [res, res])
    inp = warp_affine(img, trans, (res, res))
    inp = inp.astype(np.float32) / 255.
    inp = (inp - MEAN) / STD
    return inp.transpose(2, 0, 1), c, s
