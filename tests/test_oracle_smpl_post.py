"""CPU: analytic known answers for the restatements that have NO importable reference
(SMPL: no reference code at all -> parity unpinned; post-process: reference needs cv2)."""
import numpy as np

import h3d_amd  # noqa: F401
from h3d_amd import smpl as psmpl
from oracle import post_process as opost
from oracle import smpl as osmpl


def _model():
    return psmpl.SMPLModel.synthetic(seed=0).numpy_dict()


def test_rodrigues_properties():
    th = np.array([[0, 0, 0], [0.3, -0.2, 0.9], [np.pi / 2, 0, 0], [0, 0, -3.0]])
    R = osmpl.rodrigues(th)
    np.testing.assert_allclose(R[0], np.eye(3), atol=1e-7)
    for r in R:
        np.testing.assert_allclose(r @ r.T, np.eye(3), atol=1e-7)   # smplx eps convention
        assert abs(np.linalg.det(r) - 1) < 1e-7
    np.testing.assert_allclose(R[2], [[1, 0, 0], [0, 0, -1], [0, 1, 0]], atol=1e-7)


def test_zero_pose_is_shape_blend_only():
    m = _model()
    betas = np.linspace(-1, 1, 20).reshape(2, 10)
    v, j = osmpl.lbs(betas, np.zeros((2, 72)), m)
    v_s = m["v_template"][None] + np.einsum("vck,pk->pvc", m["shapedirs"], betas)
    np.testing.assert_allclose(v, v_s, atol=1e-6)
    np.testing.assert_allclose(j, np.einsum("jv,pvc->pjc", m["J_regressor"], v_s), atol=1e-6)


def test_single_joint_rotation_moves_only_descendants():
    m = _model()
    th = np.zeros((1, 24, 3))
    th[0, 18] = [0, 0, np.pi / 2]          # left elbow: descendants 20, 22
    v0, _ = osmpl.lbs(np.zeros((1, 10)), np.zeros((1, 72)), m)
    mm = dict(m)
    mm["posedirs"] = np.zeros_like(m["posedirs"])      # isolate the skinning effect
    v1, _ = osmpl.lbs(np.zeros((1, 10)), th.reshape(1, 72), mm)
    moved = np.abs(v1 - v0).max(axis=(0, 2)) > 1e-9
    desc = np.zeros(24, bool)
    desc[[18, 20, 22]] = True
    has_desc_weight = (m["weights"][:, desc].sum(1) > 0)
    assert (moved <= has_desc_weight).all()
    assert moved.sum() > 0


def test_global_rotation_rotates_about_root():
    m = _model()
    th = np.zeros((1, 72))
    th[0, :3] = [0, np.pi / 2, 0]
    mm = dict(m)
    mm["posedirs"] = np.zeros_like(m["posedirs"])
    v0, j0 = osmpl.lbs(np.zeros((1, 10)), np.zeros((1, 72)), mm)
    v1, j1 = osmpl.lbs(np.zeros((1, 10)), th, mm)
    R = osmpl.rodrigues(th[0, :3])
    np.testing.assert_allclose(v1[0], (v0[0] - j0[0, 0]) @ R.T + j0[0, 0], atol=1e-6)


def test_post_process_is_scale_and_shift_for_centred_crop():
    # c = image centre, s = max(h,w): x_img = (x_out - w_out/2) * s / w_out + c_x
    dets = np.zeros((1, 5, 40), np.float32)
    rng = np.random.RandomState(0)
    dets[0, :, :4] = rng.uniform(0, 128, (5, 4))
    dets[0, :, 4] = rng.uniform(0, 1, 5)
    dets[0, :, 5:39] = rng.uniform(0, 128, (5, 34))
    c = np.array([[320.0, 240.0]], np.float32)
    s = np.array([640.0], np.float32)
    out = opost.multi_pose_post_process(dets.copy(), c, s, 128, 128)[0]
    exp_box = (dets[0, :, :4].reshape(-1, 2) - 64.0) * 5.0 + c[0]
    np.testing.assert_allclose(out[:, :4], exp_box.reshape(-1, 4), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(out[:, 4], dets[0, :, 4])
    exp_pts = (dets[0, :, 5:39].reshape(-1, 2) - 64.0) * 5.0 + c[0]
    np.testing.assert_allclose(out[:, 5:], exp_pts.reshape(-1, 34), rtol=1e-5, atol=1e-3)


def test_ctdet_post_process_groups_by_class_and_is_scale_and_shift():
    # utils/post_process.py:24-38: box corners through the inverse crop affine, grouped per 1-based class id
    rng = np.random.default_rng(3)
    dets = np.zeros((2, 6, 6), np.float32)
    dets[:, :, :4] = rng.uniform(0, 128, (2, 6, 4))
    dets[:, :, 4] = rng.uniform(0, 1, (2, 6))
    dets[:, :, 5] = np.array([[0, 2, 2, 1, 0, 2], [1, 1, 1, 1, 1, 1]], np.float32)
    c = np.array([[256.0, 256.0], [300.0, 200.0]], np.float32)
    s = np.array([512.0, 640.0], np.float32)
    out = opost.ctdet_post_process(dets, c, s, 128, 128, 3)
    assert [len(out[0][k]) for k in (1, 2, 3)] == [2, 1, 3] and [len(out[1][k]) for k in (1, 2, 3)] == [0, 6, 0]
    row = np.array(out[0][2][0])                               # image 0, class id 2 = detection 3
    exp = (dets[0, 3, :4] - 64.0) * (512.0 / 128.0) + 256.0
    np.testing.assert_allclose(row[:4], exp, rtol=1e-6, atol=1e-3)
    assert abs(row[4] - dets[0, 3, 4]) < 1e-7


def test_rodrigues_agrees_with_scipy_rotation_vectors():
    # no SMPL code exists in the reference; scipy.spatial.transform is an independent implementation of the axis-angle ->
    # matrix map (same convention as SMPL's Rodrigues formula): pins the sign / axis conventions of oracle.smpl.rodrigues
    from scipy.spatial.transform import Rotation
    th = np.random.default_rng(0).normal(0, 1.2, (64, 3))
    np.testing.assert_allclose(osmpl.rodrigues(th), Rotation.from_rotvec(th).as_matrix(), atol=1e-6)


def test_lbs_of_a_one_bone_chain_is_a_rigid_transform_composition():
    # hand-built two-joint model: vertex fully bound to joint 1 -> v' = G0 G1 applied in the kinematic order of the
    # published formulation (world transform of joint 1 = T(j0) R0 T(j1 - j0) R1, minus the rest pose)
    from scipy.spatial.transform import Rotation
    m = _model()
    V = m["v_template"].shape[0]
    mm = dict(m)
    mm["posedirs"] = np.zeros_like(m["posedirs"])
    mm["shapedirs"] = np.zeros_like(m["shapedirs"])
    w = np.zeros_like(m["weights"])
    w[:, 1] = 1.0                                       # every vertex follows joint 1 (child of the root)
    mm["weights"] = w
    th = np.zeros((1, 24, 3))
    th[0, 0] = [0.3, -0.4, 0.2]
    th[0, 1] = [-0.5, 0.1, 0.7]
    v, j = osmpl.lbs(np.zeros((1, 10)), th.reshape(1, 72), mm)
    jr = m["J_regressor"] @ m["v_template"]             # rest joints
    assert int(m["parents"][1]) == 0
    R0, R1 = Rotation.from_rotvec(th[0, 0]).as_matrix(), Rotation.from_rotvec(th[0, 1]).as_matrix()
    want = (R0 @ (R1 @ (m["v_template"] - jr[1]).T + (jr[1] - jr[0])[:, None])).T + jr[0]
    np.testing.assert_allclose(v[0], want, atol=1e-6)
    assert v.shape == (1, V, 3)
