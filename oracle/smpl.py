"""TEST INFRASTRUCTURE ONLY -- float64 numpy restatement of SMPL (Loper et al. 2015) pose/shape
-> linear-blend-skinning mesh.

**Parity unpinned by the reference**: /root/reference contains no SMPL code, model file or
test.  This follows the published formulation, with the axis-angle
convention of the public `smplx` package (angle = ||theta + 1e-8||):

  v_s = v_template + S.beta            S  [6890*3, 10]
  J   = Jreg . v_s                     Jreg [24, 6890]
  R   = rodrigues(theta)               [24,3,3]
  v_p = v_s + P.vec(R[1:] - I)         P  [6890*3, 207]
  G_0 = [R_0 | J_0],  G_j = G_parent(j) . [R_j | J_j - J_parent(j)]
  A_j = [G_j.R | G_j.t - G_j.R J_j]
  v   = sum_j W[v,j] A_j . [v_p; 1]    W  [6890, 24]

Checked by analytic known answers in tests/test_oracle_smpl.py (theta=0 => v = v_s exactly,
R R^T = I, det R = 1, a single-joint rotation moves only vertices weighted to descendants).
"""
import numpy as np

PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19,
                    20, 21], dtype=np.int32)


def rodrigues(theta):
    """theta [...,3] axis-angle -> [...,3,3]."""
    theta = np.asarray(theta, dtype=np.float64)
    angle = np.linalg.norm(theta + 1e-8, axis=-1, keepdims=True)
    d = theta / angle
    s = np.sin(angle)[..., None]
    c = np.cos(angle)[..., None]
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    zero = np.zeros_like(x)
    K = np.stack([zero, -z, y, z, zero, -x, -y, x, zero], -1).reshape(theta.shape[:-1] + (3, 3))
    I = np.eye(3)
    return I + s * K + (1 - c) * (K @ K)


def lbs(betas, thetas, model):
    """betas [P,10], thetas [P,72] -> vertices [P,6890,3], joints [P,24,3] (float64)."""
    v_t = np.asarray(model["v_template"], np.float64)           # [V,3]
    S = np.asarray(model["shapedirs"], np.float64)              # [V,3,10]
    Pd = np.asarray(model["posedirs"], np.float64)              # [V,3,207]
    Jreg = np.asarray(model["J_regressor"], np.float64)         # [24,V]
    W = np.asarray(model["weights"], np.float64)                # [V,24]
    parents = np.asarray(model["parents"])
    betas = np.asarray(betas, np.float64)
    thetas = np.asarray(thetas, np.float64).reshape(-1, 24, 3)
    P = betas.shape[0]
    v_s = v_t[None] + np.einsum("vck,pk->pvc", S, betas)
    J = np.einsum("jv,pvc->pjc", Jreg, v_s)
    R = rodrigues(thetas)                                        # [P,24,3,3]
    pf = (R[:, 1:] - np.eye(3)).reshape(P, 207)
    v_p = v_s + np.einsum("vck,pk->pvc", Pd, pf)
    G = np.zeros((P, 24, 4, 4))
    for j in range(24):
        T = np.zeros((P, 4, 4))
        T[:, :3, :3] = R[:, j]
        T[:, 3, 3] = 1
        if parents[j] < 0:
            T[:, :3, 3] = J[:, j]
            G[:, j] = T
        else:
            T[:, :3, 3] = J[:, j] - J[:, parents[j]]
            G[:, j] = G[:, parents[j]] @ T
    joints = G[:, :, :3, 3].copy()
    A = G.copy()
    A[:, :, :3, 3] -= np.einsum("pjab,pjb->pja", G[:, :, :3, :3], J)
    Tv = np.einsum("vj,pjab->pvab", W, A)                        # [P,V,4,4]
    vh = np.concatenate([v_p, np.ones((P, v_p.shape[1], 1))], -1)
    verts = np.einsum("pvab,pvb->pva", Tv, vh)[..., :3]
    return verts, joints
