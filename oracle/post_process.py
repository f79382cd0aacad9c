"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's multi_pose post-process
(map decoded output-resolution coordinates back to original-image pixels).

Reference lines followed:
  utils/post_process.py:41-52  multi_pose_post_process
  utils/image.py:19-24         transform_preds
  utils/image.py:27-68         get_affine_transform(inv=1)  (3-point affine; cv2.getAffineTransform
                               is replaced by an explicit 3-point linear solve -- cv2 is absent here,
                               so the reference module itself is not importable: SURVEY 8c/8f-1)
  utils/image.py:71-79         affine_transform, get_3rd_point

Pinned by analytic identities (tests/test_oracle_post.py): with c = image centre and
s = max(h, w) the map is x_img = (x_out - w_out/2) * s / w_out + c_x (pure scale + shift).
"""
import numpy as np


def _third_point(a, b):
    d = a - b
    return b + np.array([-d[1], d[0]], dtype=np.float32)


def _solve_affine(src, dst):
    """2x3 matrix M with M.[x,y,1]^T = dst for the three src points (float64, as cv2 does)."""
    A = np.concatenate([src.astype(np.float64), np.ones((3, 1))], axis=1)   # 3x3
    return np.linalg.solve(A, dst.astype(np.float64)).T                      # 2x3


def get_affine_transform(center, scale, rot, output_size, inv=0):
    if not isinstance(scale, (np.ndarray, list)):
        scale = np.array([scale, scale], dtype=np.float32)
    center = np.asarray(center, dtype=np.float32)
    src_w = scale[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    p = [0, src_w * -0.5]
    src_dir = np.array([p[0] * cs - p[1] * sn, p[0] * sn + p[1] * cs])
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center
    src[1, :] = center + src_dir
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + dst_dir
    src[2, :] = _third_point(src[0, :], src[1, :])
    dst[2, :] = _third_point(dst[0, :], dst[1, :])
    return _solve_affine(dst, src) if inv else _solve_affine(src, dst)


def transform_preds(coords, center, scale, output_size):
    t = get_affine_transform(center, scale, 0, output_size, inv=1)
    pts = np.concatenate([coords[:, 0:2].astype(np.float32),
                          np.ones((coords.shape[0], 1), np.float32)], axis=1)
    return pts @ t.T                                                           # float64 [N,2]


def multi_pose_post_process(dets, c, s, h, w):
    """dets [B,K,40] -> per image [K,39] float32: 4 box, score, 34 keypoint coords (image px)."""
    out = []
    for i in range(dets.shape[0]):
        bbox = transform_preds(dets[i, :, :4].reshape(-1, 2), c[i], s[i], (w, h))
        pts = transform_preds(dets[i, :, 5:39].reshape(-1, 2), c[i], s[i], (w, h))
        out.append(np.concatenate([bbox.reshape(-1, 4), dets[i, :, 4:5], pts.reshape(-1, 34)],
                                  axis=1).astype(np.float32))
    return out


def ctdet_post_process(dets, c, s, h, w, num_classes):
    """utils/post_process.py:24-38: dets [B,K,6] -> per image {1-based class id: [[x1,y1,x2,y2,score], ...]}."""
    dets = np.array(dets, dtype=np.float32, copy=True)
    ret = []
    for i in range(dets.shape[0]):
        top = {}
        p0 = transform_preds(dets[i, :, 0:2], c[i], s[i], (w, h))
        p1 = transform_preds(dets[i, :, 2:4], c[i], s[i], (w, h))
        box = np.concatenate([p0, p1], axis=1).astype(np.float32)
        classes = dets[i, :, -1]
        for j in range(num_classes):
            inds = classes == j
            top[j + 1] = np.concatenate([box[inds], dets[i, inds, 4:5].astype(np.float32)], axis=1).tolist()
        ret.append(top)
    return ret
