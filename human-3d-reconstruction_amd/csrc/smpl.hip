// SMPL pose/shape -> LBS mesh on gfx950 (north_star stage; the reference snapshot has no SMPL
// code, so this follows the published formulation -- see oracle/smpl.py and DESIGN.md).
//
//   smpl_pose_kernel   one lane per person: 24 Rodrigues rotations, pose feature vec(R[1:]-I),
//                      joints J = j_template + j_shapedirs.beta (the joint regressor applied to
//                      the shape blend, pre-contracted on the host), kinematic chain, 3x4
//                      skinning transforms A_j = [G_j.R | G_j.t - G_j.R J_j].
//   smpl_verts_kernel  blend shapes as a [V*3 x 217] x [217 x P] contraction with the person
//                      tile kept in registers (PT persons per lane), then 4-sparse LBS.
//                      Model tensors are stored K-major ([k][V*3]) so lanes (= vertices) read
//                      consecutive addresses for every k.
#include "common.h"

constexpr int SMPL_J = 24;
constexpr int SMPL_NB = 10;
constexpr int SMPL_PF = 207;

__global__ __launch_bounds__(64) void smpl_pose_kernel(const float *__restrict__ betas, const float *__restrict__ thetas,
                                                       const float *__restrict__ j_template,
                                                       const float *__restrict__ j_shapedirs,
                                                       const int32_t *__restrict__ parents, int P,
                                                       float *__restrict__ pose_feat, float *__restrict__ A,
                                                       float *__restrict__ joints)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    float beta[SMPL_NB];
#pragma unroll
    for (int k = 0; k < SMPL_NB; ++k) beta[k] = betas[(size_t)p * SMPL_NB + k];
    // global transforms G_j = [R | t], kept in registers/scratch per lane (24 x 12 floats)
    float G[SMPL_J][12];
    float Jp[SMPL_J][3];
    for (int j = 0; j < SMPL_J; ++j) {
        // rest joint
        float jr[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s = j_template[j * 3 + c];
#pragma unroll
            for (int k = 0; k < SMPL_NB; ++k) s = fmaf(j_shapedirs[(j * 3 + c) * SMPL_NB + k], beta[k], s);
            jr[c] = s;
            Jp[j][c] = s;
        }
        // Rodrigues with the smplx convention: angle = ||theta + 1e-8||, axis = theta / angle
        const float tx = thetas[(size_t)p * 72 + j * 3], ty = thetas[(size_t)p * 72 + j * 3 + 1],
                    tz = thetas[(size_t)p * 72 + j * 3 + 2];
        const float ex = tx + 1e-8f, ey = ty + 1e-8f, ez = tz + 1e-8f;
        const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
        const float inv = 1.f / angle;
        const float x = tx * inv, y = ty * inv, z = tz * inv;
        float sn, cs;
        sincosf(angle, &sn, &cs);
        const float oc = 1.f - cs;
        // R = I + sin K + (1-cos) K^2,  K = skew(x,y,z)
        float R[9];
        R[0] = 1.f + oc * (-(y * y) - z * z);
        R[1] = -sn * z + oc * (x * y);
        R[2] = sn * y + oc * (x * z);
        R[3] = sn * z + oc * (x * y);
        R[4] = 1.f + oc * (-(x * x) - z * z);
        R[5] = -sn * x + oc * (y * z);
        R[6] = -sn * y + oc * (x * z);
        R[7] = sn * x + oc * (y * z);
        R[8] = 1.f + oc * (-(x * x) - y * y);
        if (j > 0) {
            float *pf = pose_feat + (size_t)p * SMPL_PF + (j - 1) * 9;
#pragma unroll
            for (int i = 0; i < 9; ++i) pf[i] = R[i] - ((i == 0 || i == 4 || i == 8) ? 1.f : 0.f);
        }
        const int par = parents[j];
        if (par < 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                G[j][a * 4] = R[a * 3]; G[j][a * 4 + 1] = R[a * 3 + 1]; G[j][a * 4 + 2] = R[a * 3 + 2];
                G[j][a * 4 + 3] = jr[a];
            }
        } else {
            const float rel[3] = {jr[0] - Jp[par][0], jr[1] - Jp[par][1], jr[2] - Jp[par][2]};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float g0 = G[par][a * 4], g1 = G[par][a * 4 + 1], g2 = G[par][a * 4 + 2], g3 = G[par][a * 4 + 3];
                G[j][a * 4] = g0 * R[0] + g1 * R[3] + g2 * R[6];
                G[j][a * 4 + 1] = g0 * R[1] + g1 * R[4] + g2 * R[7];
                G[j][a * 4 + 2] = g0 * R[2] + g1 * R[5] + g2 * R[8];
                G[j][a * 4 + 3] = g0 * rel[0] + g1 * rel[1] + g2 * rel[2] + g3;
            }
        }
    }
    for (int j = 0; j < SMPL_J; ++j) {
        float *Ao = A + ((size_t)p * SMPL_J + j) * 12;
        float *jo = joints + ((size_t)p * SMPL_J + j) * 3;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float g0 = G[j][a * 4], g1 = G[j][a * 4 + 1], g2 = G[j][a * 4 + 2], g3 = G[j][a * 4 + 3];
            Ao[a * 4] = g0; Ao[a * 4 + 1] = g1; Ao[a * 4 + 2] = g2;
            Ao[a * 4 + 3] = g3 - (g0 * Jp[j][0] + g1 * Jp[j][1] + g2 * Jp[j][2]);
            jo[a] = g3;
        }
    }
}

extern "C" int h3d_smpl_pose(const float *betas, const float *thetas, const float *j_template, const float *j_shapedirs,
                             const int32_t *parents, int P, float *pose_feat, float *A, float *joints, void *stream)
{
    if (!betas || !thetas || !j_template || !j_shapedirs || !parents || !pose_feat || !A || !joints)
        H3D_FAIL(H3D_ERR_ARG, "smpl_pose: null pointer");
    if (P <= 0) H3D_FAIL(H3D_ERR_SHAPE, "smpl_pose: P=%d", P);
    hipLaunchKernelGGL(smpl_pose_kernel, dim3(cdiv(P, 64)), dim3(64), 0, (hipStream_t)stream, betas, thetas, j_template,
                       j_shapedirs, parents, P, pose_feat, A, joints);
    H3D_CHECK_LAUNCH("smpl_pose_kernel");
    return H3D_OK;
}

// Block = 256 lanes = 256 vertices, PT persons per block; coefficient vectors [beta | pose_feat]
// (217 floats per person) and the A transforms (288 floats per person) sit in LDS and are read
// as wave-uniform broadcasts.
template <int PT>
__global__ __launch_bounds__(256) void smpl_verts_kernel(const float *__restrict__ betas, const float *__restrict__ pose_feat,
                                                         const float *__restrict__ A, const float *__restrict__ v_template,
                                                         const float *__restrict__ shapedirsT,
                                                         const float *__restrict__ posedirsT,
                                                         const int32_t *__restrict__ lbs_idx, const float *__restrict__ lbs_w,
                                                         int nnz, int P, int V, float *__restrict__ verts)
{
    constexpr int NC = SMPL_NB + SMPL_PF;  // 217
    __shared__ float s_coef[PT][NC + 3];
    __shared__ float s_A[PT][SMPL_J * 12];
    const int tid = threadIdx.x;
    const int v = blockIdx.x * 256 + tid;
    const int p0 = blockIdx.y * PT;
    for (int i = tid; i < PT * NC; i += 256) {
        const int q = i / NC, k = i - q * NC;
        const int p = p0 + q;
        float val = 0.f;
        if (p < P) val = (k < SMPL_NB) ? betas[(size_t)p * SMPL_NB + k] : pose_feat[(size_t)p * SMPL_PF + (k - SMPL_NB)];
        s_coef[q][k] = val;
    }
    for (int i = tid; i < PT * SMPL_J * 12; i += 256) {
        const int q = i / (SMPL_J * 12), k = i - q * (SMPL_J * 12);
        const int p = p0 + q;
        s_A[q][k] = (p < P) ? A[(size_t)p * SMPL_J * 12 + k] : 0.f;
    }
    __syncthreads();
    if (v >= V) return;
    float acc[PT][3];
    const float t0 = v_template[v * 3], t1 = v_template[v * 3 + 1], t2 = v_template[v * 3 + 2];
#pragma unroll
    for (int q = 0; q < PT; ++q) { acc[q][0] = t0; acc[q][1] = t1; acc[q][2] = t2; }
    const size_t V3 = (size_t)V * 3;
    for (int k = 0; k < NC; ++k) {
        const float *dp = (k < SMPL_NB) ? (shapedirsT + (size_t)k * V3) : (posedirsT + (size_t)(k - SMPL_NB) * V3);
        const float d0 = dp[v * 3], d1 = dp[v * 3 + 1], d2 = dp[v * 3 + 2];
#pragma unroll
        for (int q = 0; q < PT; ++q) {
            const float c = s_coef[q][k];
            acc[q][0] = fmaf(d0, c, acc[q][0]);
            acc[q][1] = fmaf(d1, c, acc[q][1]);
            acc[q][2] = fmaf(d2, c, acc[q][2]);
        }
    }
    float T[PT][12];
#pragma unroll
    for (int q = 0; q < PT; ++q)
#pragma unroll
        for (int i = 0; i < 12; ++i) T[q][i] = 0.f;
    for (int s = 0; s < nnz; ++s) {
        const int j = lbs_idx[(size_t)v * nnz + s];
        const float w = lbs_w[(size_t)v * nnz + s];
#pragma unroll
        for (int q = 0; q < PT; ++q)
#pragma unroll
            for (int i = 0; i < 12; ++i) T[q][i] = fmaf(w, s_A[q][j * 12 + i], T[q][i]);
    }
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int p = p0 + q;
        if (p >= P) break;
        float *o = verts + ((size_t)p * V + v) * 3;
#pragma unroll
        for (int a = 0; a < 3; ++a)
            o[a] = T[q][a * 4] * acc[q][0] + T[q][a * 4 + 1] * acc[q][1] + T[q][a * 4 + 2] * acc[q][2] + T[q][a * 4 + 3];
    }
}

extern "C" int h3d_smpl_verts(const float *betas, const float *pose_feat, const float *A, const float *v_template,
                              const float *shapedirsT, const float *posedirsT, const int32_t *lbs_idx, const float *lbs_w,
                              int nnz, int P, int V, float *verts, void *stream)
{
    if (!betas || !pose_feat || !A || !v_template || !shapedirsT || !posedirsT || !lbs_idx || !lbs_w || !verts)
        H3D_FAIL(H3D_ERR_ARG, "smpl_verts: null pointer");
    if (P <= 0 || V <= 0 || nnz <= 0 || nnz > SMPL_J) H3D_FAIL(H3D_ERR_SHAPE, "smpl_verts: P=%d V=%d nnz=%d", P, V, nnz);
    constexpr int PT = 8;
    dim3 grid(cdiv(V, 256), cdiv(P, PT));
    hipLaunchKernelGGL(smpl_verts_kernel<PT>, grid, dim3(256), 0, (hipStream_t)stream, betas, pose_feat, A, v_template,
                       shapedirsT, posedirsT, lbs_idx, lbs_w, nnz, P, V, verts);
    H3D_CHECK_LAUNCH("smpl_verts_kernel");
    return H3D_OK;
}
