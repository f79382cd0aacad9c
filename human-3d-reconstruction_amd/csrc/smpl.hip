// SMPL pose/shape -> LBS mesh on gfx950 (north_star stage; the reference snapshot has no SMPL
// code, so this follows the published formulation -- see oracle/smpl.py and DESIGN.md).
//
//   smpl_pose_kernel   one lane per (person, joint): Rodrigues rotation, pose feature vec(R[1:]-I),
//                      joints J = j_template + j_shapedirs.beta (the joint regressor applied to
//                      the shape blend, pre-contracted on the host), kinematic chain, 3x4
//                      skinning transforms A_j = [G_j.R | G_j.t - G_j.R J_j].
//   smpl_verts_kernel  blend shapes as a [V*3 x 217] x [217 x P] contraction with the person
//                      tile kept in registers (PT persons per lane), then 4-sparse LBS.
//                      Model tensors are stored K-major ([k][V*3]) so lanes (= vertices) read
//                      consecutive addresses for every k.
#include "common.h"
#include <type_traits>

constexpr int SMPL_J = 24;
constexpr int SMPL_NB = 10;
constexpr int SMPL_PF = 207;

// One lane per (person, joint): a wave holds two persons (lanes 32q + j, j < 24).  Rodrigues, the rest joints and
// the outputs are joint-parallel; the kinematic chain walks the joints in their (topological) order and lane j
// fetches its parent's transform by a shuffle -- 23 short steps instead of a 24-joint serial program per lane with
// its 24 x 12 transform table in scratch memory (0.13 ms -> 0.02 ms at 6400 persons).
// HEADS (round 4: the detector's tail as fewer launches): the parameters are read where the network left them -- person p is
// detection q = p % n of image b = p / n, its betas / thetas are the `shape` / `pose` head maps [B,10|72,HW] at pixel inds[b*K + q]
// (what `_transpose_and_gather_feat`, utils.py:23-27, would have copied out in two launches) -- and the generation-3 coefficient
// operand coefK3 [Ppad][14][3][16] (h3d_smpl_coef_pack's output: the h / m / l bf16 terms of [beta | pose_feat | 0], zero rows for
// p >= P) is written by the lanes that hold the values.  Same arithmetic on the same values: bit-identical to the three launches.
struct SmplHeadsSrc {
    const float *pose_map, *shape_map;
    const int64_t *inds;
    int n, K, HW;
    float *betas_out;        // [P,10] (API output; may be NULL)
    bf16_t *coefK3;          // [Ppad][14][3][16]
};
__device__ __forceinline__ void split3(float v, bf16_t &h, bf16_t &m, bf16_t &l);

template <bool HEADS>
__global__ __launch_bounds__(64) void smpl_pose_kernel(const float *__restrict__ betas, const float *__restrict__ thetas,
                                                       const float *__restrict__ j_template,
                                                       const float *__restrict__ j_shapedirs,
                                                       const int32_t *__restrict__ parents, int P,
                                                       float *__restrict__ pose_feat, float *__restrict__ A,
                                                       float *__restrict__ joints, float *__restrict__ coefT, int Ppad, SmplHeadsSrc hs)
{
    const int l = threadIdx.x, j = l & 31, half = l >> 5;
    const int p = blockIdx.x * 2 + half;
    const bool act = p < P && j < SMPL_J;
    const int pc = p < P ? p : P - 1, jc = j < SMPL_J ? j : 0;       // clamped: every lane computes, only `act` lanes store
    float beta[SMPL_NB];
    [[maybe_unused]] size_t hb = 0;
    [[maybe_unused]] int hpix = 0;
    if constexpr (HEADS) {
        hb = (size_t)(pc / hs.n);
        const int64_t pix = hs.inds[hb * hs.K + (pc - (int)hb * hs.n)];
        hpix = (int)(pix < 0 ? 0 : pix >= hs.HW ? hs.HW - 1 : pix);
#pragma unroll
        for (int k = 0; k < SMPL_NB; ++k) beta[k] = hs.shape_map[(hb * SMPL_NB + k) * hs.HW + hpix];
        if (hs.betas_out && p < P && j < SMPL_NB) hs.betas_out[(size_t)p * SMPL_NB + j] = beta[j];
    } else {
#pragma unroll
    for (int k = 0; k < SMPL_NB; ++k) beta[k] = betas[(size_t)pc * SMPL_NB + k];
    }
    if (coefT && p < P && j < SMPL_NB) coefT[(size_t)j * Ppad + p] = beta[j];     // k-major [beta | pose_feat] (gen 2 kernel)
    [[maybe_unused]] auto put_coef = [&](int k, float v) {          // coefK3[p][k >> 4][term][k & 15], zero rows for p >= P
        bf16_t hh, mm, ll;
        split3(p < P ? v : 0.f, hh, mm, ll);
        bf16_t *o = hs.coefK3 + ((size_t)p * 14 + (k >> 4)) * 48 + (k & 15);
        o[0] = hh; o[16] = mm; o[32] = ll;
    };
    if constexpr (HEADS) {
        if (p < Ppad) {
            if (j < SMPL_NB) put_coef(j, beta[j]);
            if (j >= SMPL_J && j < SMPL_J + 7) put_coef(SMPL_NB + SMPL_PF + (j - SMPL_J), 0.f);      // K padding 217..223
        }
    }
    // rest joint of (p, j)
    float jr[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s = j_template[jc * 3 + c];
#pragma unroll
        for (int k = 0; k < SMPL_NB; ++k) s = fmaf(j_shapedirs[(jc * 3 + c) * SMPL_NB + k], beta[k], s);
        jr[c] = s;
    }
    // Rodrigues with the smplx convention: angle = ||theta + 1e-8||, axis = theta / angle
    float tx, ty, tz;
    if constexpr (HEADS) {
        const float *tp = hs.pose_map + (hb * 72 + jc * 3) * hs.HW + hpix;
        tx = tp[0]; ty = tp[hs.HW]; tz = tp[2 * (size_t)hs.HW];
    } else {
        tx = thetas[(size_t)pc * 72 + jc * 3]; ty = thetas[(size_t)pc * 72 + jc * 3 + 1]; tz = thetas[(size_t)pc * 72 + jc * 3 + 2];
    }
    const float ex = tx + 1e-8f, ey = ty + 1e-8f, ez = tz + 1e-8f;
    const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
    const float inv = 1.f / angle;
    const float x = tx * inv, y = ty * inv, z = tz * inv;
    float sn, cs;
    sincosf(angle, &sn, &cs);
    const float oc = 1.f - cs;
    float R[9];          // R = I + sin K + (1-cos) K^2,  K = skew(x,y,z)
    R[0] = 1.f + oc * (-(y * y) - z * z);
    R[1] = -sn * z + oc * (x * y);
    R[2] = sn * y + oc * (x * z);
    R[3] = sn * z + oc * (x * y);
    R[4] = 1.f + oc * (-(x * x) - z * z);
    R[5] = -sn * x + oc * (y * z);
    R[6] = -sn * y + oc * (x * z);
    R[7] = sn * x + oc * (y * z);
    R[8] = 1.f + oc * (-(x * x) - y * y);
    if (act && j > 0) {
        float *pf = pose_feat + (size_t)p * SMPL_PF + (j - 1) * 9;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const float f = R[i] - ((i == 0 || i == 4 || i == 8) ? 1.f : 0.f);
            pf[i] = f;
            if (coefT) coefT[(size_t)(SMPL_NB + (j - 1) * 9 + i) * Ppad + p] = f;
        }
    }
    if constexpr (HEADS) {
        if (p < Ppad && j > 0 && j < SMPL_J) {
#pragma unroll
            for (int i = 0; i < 9; ++i) put_coef(SMPL_NB + (j - 1) * 9 + i, R[i] - ((i == 0 || i == 4 || i == 8) ? 1.f : 0.f));
        }
    }
    // local transform L = [R | jr - jr(parent)] (root: [R | jr]); G starts as L and becomes G(parent) . L
    const int par = parents[jc];
    const int plane = half * 32 + (par < 0 ? 0 : par);
    float G[12];
    {
        const float px = __shfl(jr[0], plane), py = __shfl(jr[1], plane), pz = __shfl(jr[2], plane);
        const float rel[3] = {par < 0 ? jr[0] : jr[0] - px, par < 0 ? jr[1] : jr[1] - py, par < 0 ? jr[2] : jr[2] - pz};
#pragma unroll
        for (int a = 0; a < 3; ++a) { G[a * 4] = R[a * 3]; G[a * 4 + 1] = R[a * 3 + 1]; G[a * 4 + 2] = R[a * 3 + 2]; G[a * 4 + 3] = rel[a]; }
    }
    for (int step = 1; step < SMPL_J; ++step) {          // joints are stored parents-first (SMPL's kintree order)
        float Gp[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) Gp[i] = __shfl(G[i], plane);          // my parent's (already global) transform
        if (j == step) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float g0 = Gp[a * 4], g1 = Gp[a * 4 + 1], g2 = Gp[a * 4 + 2], g3 = Gp[a * 4 + 3];
                const float l0 = G[0], l1 = G[1], l2 = G[2], l3 = G[3], l4 = G[4], l5 = G[5], l6 = G[6], l7 = G[7], l8 = G[8],
                            l9 = G[9], l10 = G[10], l11 = G[11];
                Gp[a * 4] = g0 * l0 + g1 * l4 + g2 * l8;
                Gp[a * 4 + 1] = g0 * l1 + g1 * l5 + g2 * l9;
                Gp[a * 4 + 2] = g0 * l2 + g1 * l6 + g2 * l10;
                Gp[a * 4 + 3] = g0 * l3 + g1 * l7 + g2 * l11 + g3;
            }
#pragma unroll
            for (int i = 0; i < 12; ++i) G[i] = Gp[i];
        }
    }
    if (act) {
        float *Ao = A + ((size_t)p * SMPL_J + j) * 12;
        float *jo = joints + ((size_t)p * SMPL_J + j) * 3;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float g0 = G[a * 4], g1 = G[a * 4 + 1], g2 = G[a * 4 + 2], g3 = G[a * 4 + 3];
            Ao[a * 4] = g0; Ao[a * 4 + 1] = g1; Ao[a * 4 + 2] = g2;
            Ao[a * 4 + 3] = g3 - (g0 * jr[0] + g1 * jr[1] + g2 * jr[2]);
            jo[a] = g3;
        }
    }
}

extern "C" int h3d_smpl_pose(const float *betas, const float *thetas, const float *j_template, const float *j_shapedirs,
                             const int32_t *parents, int P, float *pose_feat, float *A, float *joints, float *coefT,
                             int Ppad, void *stream)
{
    if (!betas || !thetas || !j_template || !j_shapedirs || !parents || !pose_feat || !A || !joints)
        H3D_FAIL(H3D_ERR_ARG, "smpl_pose: null pointer");
    if (P <= 0 || (coefT && Ppad < P)) H3D_FAIL(H3D_ERR_SHAPE, "smpl_pose: P=%d Ppad=%d", P, Ppad);
    hipLaunchKernelGGL(smpl_pose_kernel<false>, dim3(cdiv(P, 2)), dim3(64), 0, (hipStream_t)stream, betas, thetas, j_template,
                       j_shapedirs, parents, P, pose_feat, A, joints, coefT, Ppad, SmplHeadsSrc{});
    H3D_CHECK_LAUNCH("smpl_pose_kernel");
    return H3D_OK;
}

extern "C" int h3d_smpl_pose_heads(const float *pose_map, const float *shape_map, const int64_t *inds, int B, int K, int n, int HW,
                                   const float *j_template, const float *j_shapedirs, const int32_t *parents, float *betas_out,
                                   float *pose_feat, float *A, float *joints, void *coefK3, int Ppad, void *stream)
{
    if (!pose_map || !shape_map || !inds || !j_template || !j_shapedirs || !parents || !pose_feat || !A || !joints || !coefK3)
        H3D_FAIL(H3D_ERR_ARG, "smpl_pose_heads: null pointer");
    const int P = B * n;
    if (B <= 0 || n <= 0 || n > K || HW <= 0 || Ppad < P || Ppad % 128) H3D_FAIL(H3D_ERR_SHAPE, "smpl_pose_heads: B=%d K=%d n=%d HW=%d Ppad=%d", B, K, n, HW, Ppad);
    SmplHeadsSrc hs;
    hs.pose_map = pose_map; hs.shape_map = shape_map; hs.inds = inds; hs.n = n; hs.K = K; hs.HW = HW; hs.betas_out = betas_out;
    hs.coefK3 = (bf16_t *)coefK3;
    hipLaunchKernelGGL(smpl_pose_kernel<true>, dim3(cdiv(Ppad, 2)), dim3(64), 0, (hipStream_t)stream, nullptr, nullptr, j_template, j_shapedirs,
                       parents, P, pose_feat, A, joints, nullptr, Ppad, hs);
    H3D_CHECK_LAUNCH("smpl_pose_kernel");
    return H3D_OK;
}

// Block = 256 lanes = 256 vertices x PT persons.  The blend-shape contraction
//     v_posed[p][v][c] = v_template[v][c] + sum_k dirs[k][c][v] * coef[p][k]      (k = 10 betas + 207 pose feats)
// keeps the PT x 3 accumulators of a lane in registers; direction tensors are stored [k][3][V]
// (struct-of-arrays: each of the 3 loads of a k-step is one fully coalesced 256-byte row per
// wave) and are re-read from L2 once per PT persons; the coefficient block sits in LDS as
// [k][PT] so a k-step reads it with PT/4 broadcast ds_read_b128.  LBS then runs per person with
// the 3x4 joint transforms broadcast from LDS (4-sparse skinning weights).
template <int PT>
__global__ __launch_bounds__(256) void smpl_verts_kernel(const float *__restrict__ betas, const float *__restrict__ pose_feat,
                                                         const float *__restrict__ A, const float *__restrict__ v_template,
                                                         const float *__restrict__ shapedirsT,
                                                         const float *__restrict__ posedirsT,
                                                         const int32_t *__restrict__ lbs_idx, const float *__restrict__ lbs_w,
                                                         int nnz, int P, int V, int Vpad, float *__restrict__ verts)
{
    constexpr int NC = SMPL_NB + SMPL_PF;  // 217
    __shared__ __attribute__((aligned(16))) float s_coef[NC][PT];
    __shared__ __attribute__((aligned(16))) float s_A[PT][SMPL_J * 12];
    const int tid = threadIdx.x;
    const int v = blockIdx.x * 256 + tid;
    const int p0 = blockIdx.y * PT;
    for (int i = tid; i < PT * NC; i += 256) {
        const int q = i / NC, k = i - q * NC;
        const int p = p0 + q;
        float val = 0.f;
        if (p < P) val = (k < SMPL_NB) ? betas[(size_t)p * SMPL_NB + k] : pose_feat[(size_t)p * SMPL_PF + (k - SMPL_NB)];
        s_coef[k][q] = val;
    }
    for (int i = tid; i < PT * SMPL_J * 12; i += 256) {
        const int q = i / (SMPL_J * 12), k = i - q * (SMPL_J * 12);
        const int p = p0 + q;
        s_A[q][k] = (p < P) ? A[(size_t)p * SMPL_J * 12 + k] : 0.f;
    }
    __syncthreads();
    const int vc = v < V ? v : V - 1;      // clamp: out-of-range lanes compute a duplicate and do not store
    float acc[PT][3];
    const float t0 = v_template[vc], t1 = v_template[Vpad + vc], t2 = v_template[2 * Vpad + vc];
#pragma unroll
    for (int q = 0; q < PT; ++q) { acc[q][0] = t0; acc[q][1] = t1; acc[q][2] = t2; }
    const size_t V3 = (size_t)Vpad * 3;
#pragma unroll 2
    for (int k = 0; k < NC; ++k) {
        const float *dp = (k < SMPL_NB) ? (shapedirsT + (size_t)k * V3) : (posedirsT + (size_t)(k - SMPL_NB) * V3);
        const float d0 = dp[vc], d1 = dp[Vpad + vc], d2 = dp[2 * Vpad + vc];
#pragma unroll
        for (int q4 = 0; q4 < PT / 4; ++q4) {
            const f32x4 c = *reinterpret_cast<const f32x4 *>(&s_coef[k][q4 * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[q4 * 4 + e][0] = fmaf(d0, c[e], acc[q4 * 4 + e][0]);
                acc[q4 * 4 + e][1] = fmaf(d1, c[e], acc[q4 * 4 + e][1]);
                acc[q4 * 4 + e][2] = fmaf(d2, c[e], acc[q4 * 4 + e][2]);
            }
        }
    }
    int jidx[4];
    float jw[4];
    const bool sparse4 = (nnz <= 4);
    if (sparse4) {
#pragma unroll
        for (int sI = 0; sI < 4; ++sI) {
            jidx[sI] = sI < nnz ? lbs_idx[(size_t)vc * nnz + sI] : 0;
            jw[sI] = sI < nnz ? lbs_w[(size_t)vc * nnz + sI] : 0.f;
        }
    }
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int p = p0 + q;
        if (p >= P) continue;
        float T[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = 0.f;
        if (sparse4) {
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) {
                const float *Ap = &s_A[q][jidx[sI] * 12];
#pragma unroll
                for (int i4 = 0; i4 < 3; ++i4) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4 *>(Ap + 4 * i4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) T[4 * i4 + e] = fmaf(jw[sI], a4[e], T[4 * i4 + e]);
                }
            }
        } else {
            for (int sI = 0; sI < nnz; ++sI) {
                const int j = lbs_idx[(size_t)vc * nnz + sI];
                const float w = lbs_w[(size_t)vc * nnz + sI];
#pragma unroll
                for (int i = 0; i < 12; ++i) T[i] = fmaf(w, s_A[q][j * 12 + i], T[i]);
            }
        }
        const float x = acc[q][0], y = acc[q][1], z = acc[q][2];
        if (v < V) {
            float *o = verts + ((size_t)p * V + v) * 3;
            o[0] = T[0] * x + T[1] * y + T[2] * z + T[3];
            o[1] = T[4] * x + T[5] * y + T[6] * z + T[7];
            o[2] = T[8] * x + T[9] * y + T[10] * z + T[11];
        }
    }
}

#ifdef H3D_EXTRA      // generation 2 (vector FMA, operands streamed through LDS): superseded by generation 3, kept as an A/B reference
// ------------------------------------------------------------------------------------------------
// Generation 2 of the vertex kernel (gen 1 above was latency-bound on per-lane L2 loads: 42 TFLOP/s).
// Workgroup = 64 vertices x 128 persons: lane = vertex, wave w = persons [32w, 32w+32).  The
// direction tensors [k][3][Vpad] and the k-major coefficients coefT[k][Ppad] are streamed through
// LDS in chunks of 31 k (217 = 7 x 31), double buffered with a register prefetch, so every
// direction element is fetched from L2 once per 128 persons and the inner loop only touches LDS:
// 3 conflict-free ds_read_b32 + 8 broadcast ds_read_b128 per 96 FMAs.  LBS: the 3x4 transforms of
// 16 persons per wave are staged into the (then idle) stream buffers, twice.
constexpr int SV_KC = 31, SV_VT = 64, SV_PT = 32, SV_PB = 128;
__global__ __launch_bounds__(256) void smpl_verts2_kernel(const float *__restrict__ coefT, const float *__restrict__ A,
                                                          const float *__restrict__ v_template,
                                                          const float *__restrict__ shapedirsT,
                                                          const float *__restrict__ posedirsT,
                                                          const int32_t *__restrict__ lbs_idx, const float *__restrict__ lbs_w,
                                                          int nnz, int P, int Ppad, int V, int Vpad, float *__restrict__ verts)
{
    constexpr int NC = SMPL_NB + SMPL_PF;            // 217
    constexpr int DCH = SV_KC * 3 * SV_VT;           // floats of one direction chunk
    constexpr int CCH = SV_KC * SV_PB;               // floats of one coefficient chunk
    constexpr int BUF = DCH + CCH;
    static_assert(NC % SV_KC == 0, "k chunking");
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int v0 = blockIdx.x * SV_VT, p0 = blockIdx.y * SV_PB;
    const int v = v0 + lane;
    constexpr int ND = DCH / 4 / 256 + 1, NCV = CCH / 4 / 256 + 1;   // float4 per thread (rounded up)
    f32x4 sd[ND], sc[NCV];
    auto load_chunk = [&](int ch) {
        const int k0 = ch * SV_KC;
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int i = tid + j * 256;                 // float4 index inside [KC][3][64]
            sd[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < DCH / 4) {
                const int row = i / (SV_VT / 4), q = i - row * (SV_VT / 4);   // row = kl*3 + c
                const int k = k0 + row / 3, c = row - (row / 3) * 3;
                const float *dp = (k < SMPL_NB) ? (shapedirsT + ((size_t)k * 3 + c) * Vpad)
                                                : (posedirsT + ((size_t)(k - SMPL_NB) * 3 + c) * Vpad);
                sd[j] = *reinterpret_cast<const f32x4 *>(dp + v0 + 4 * q);    // Vpad, v0 multiples of 64: aligned, in bounds
            }
        }
#pragma unroll
        for (int j = 0; j < NCV; ++j) {
            const int i = tid + j * 256;                 // float4 index inside [KC][128]
            sc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < CCH / 4) {
                const int kl = i / (SV_PB / 4), q = i - kl * (SV_PB / 4);
                sc[j] = *reinterpret_cast<const f32x4 *>(coefT + (size_t)(k0 + kl) * Ppad + p0 + 4 * q);   // Ppad multiple of 128
            }
        }
    };
    auto store_chunk = [&](int buf) {
        float *d = smem + buf * BUF, *c = d + DCH;
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int i = tid + j * 256;
            if (i < DCH / 4) *reinterpret_cast<f32x4 *>(d + 4 * i) = sd[j];
        }
#pragma unroll
        for (int j = 0; j < NCV; ++j) {
            const int i = tid + j * 256;
            if (i < CCH / 4) *reinterpret_cast<f32x4 *>(c + 4 * i) = sc[j];
        }
    };

    const int vc = v < V ? v : V - 1;
    float acc[SV_PT][3];
    {
        const float t0 = v_template[vc], t1 = v_template[Vpad + vc], t2 = v_template[2 * Vpad + vc];
#pragma unroll
        for (int q = 0; q < SV_PT; ++q) { acc[q][0] = t0; acc[q][1] = t1; acc[q][2] = t2; }
    }
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    constexpr int NCH = NC / SV_KC;
    for (int ch = 0; ch < NCH; ++ch) {
        if (ch + 1 < NCH) load_chunk(ch + 1);
        const float *d = smem + (ch & 1) * BUF, *c = d + DCH + wv * SV_PT;
#pragma unroll 1
        for (int kl = 0; kl < SV_KC; ++kl) {
            const float d0 = d[(kl * 3 + 0) * SV_VT + lane], d1 = d[(kl * 3 + 1) * SV_VT + lane], d2 = d[(kl * 3 + 2) * SV_VT + lane];
#pragma unroll
            for (int q4 = 0; q4 < SV_PT / 4; ++q4) {
                const f32x4 cf = *reinterpret_cast<const f32x4 *>(c + kl * SV_PB + 4 * q4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[q4 * 4 + e][0] = fmaf(d0, cf[e], acc[q4 * 4 + e][0]);
                    acc[q4 * 4 + e][1] = fmaf(d1, cf[e], acc[q4 * 4 + e][1]);
                    acc[q4 * 4 + e][2] = fmaf(d2, cf[e], acc[q4 * 4 + e][2]);
                }
            }
        }
        if (ch + 1 < NCH) store_chunk((ch + 1) & 1);
        __syncthreads();
    }
    // ---- LBS: stage the 3x4 transforms of 16 persons per wave (4 x 16 x 288 floats) into smem, twice ----
    int jidx[4];
    float jw[4];
#pragma unroll
    for (int sI = 0; sI < 4; ++sI) {
        jidx[sI] = sI < nnz ? lbs_idx[(size_t)vc * nnz + sI] : 0;
        jw[sI] = sI < nnz ? lbs_w[(size_t)vc * nnz + sI] : 0.f;
    }
    static_assert(4 * 16 * SMPL_J * 12 <= 2 * BUF, "A staging fits the stream buffers");
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
        // each wave stages its own 16 persons: [16][288] floats = 1152 float4, 18 per lane
        float *sA = smem + wv * (16 * SMPL_J * 12);
        for (int i = lane; i < 16 * SMPL_J * 3; i += 64) {
            const int q = i / (SMPL_J * 3), r = i - q * (SMPL_J * 3);
            const int p = p0 + wv * SV_PT + half * 16 + q;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (p < P) val = *reinterpret_cast<const f32x4 *>(A + (size_t)p * SMPL_J * 12 + 4 * r);
            *reinterpret_cast<f32x4 *>(sA + q * SMPL_J * 12 + 4 * r) = val;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int p = p0 + wv * SV_PT + half * 16 + q;
            float T[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) T[i] = 0.f;
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) {
                const float *Ap = sA + q * SMPL_J * 12 + jidx[sI] * 12;
#pragma unroll
                for (int i4 = 0; i4 < 3; ++i4) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4 *>(Ap + 4 * i4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) T[4 * i4 + e] = fmaf(jw[sI], a4[e], T[4 * i4 + e]);
                }
            }
            const float x = acc[half * 16 + q][0], y = acc[half * 16 + q][1], z = acc[half * 16 + q][2];
            if (v < V && p < P) {
                float *o = verts + ((size_t)p * V + v) * 3;
                o[0] = T[0] * x + T[1] * y + T[2] * z + T[3];
                o[1] = T[4] * x + T[5] * y + T[6] * z + T[7];
                o[2] = T[8] * x + T[9] * y + T[10] * z + T[11];
            }
        }
    }
}
#endif

extern "C" int h3d_smpl_verts2(const float *coefT, const float *A, const float *v_template, const float *shapedirsT,
                               const float *posedirsT, const int32_t *lbs_idx, const float *lbs_w, int nnz, int P, int Ppad,
                               int V, int Vpad, float *verts, void *stream)
{
#ifndef H3D_EXTRA
    (void)coefT; (void)A; (void)v_template; (void)shapedirsT; (void)posedirsT; (void)lbs_idx; (void)lbs_w; (void)nnz; (void)P; (void)Ppad; (void)V; (void)Vpad;
    (void)verts; (void)stream;
    H3D_FAIL(H3D_ERR_UNSUPPORTED, "smpl_verts2: the generation-2 vertex kernel is built only by `make EXTRA=1` (generation 3, h3d_smpl_verts3, is the product path)");
#else
    if (!coefT || !A || !v_template || !shapedirsT || !posedirsT || !lbs_idx || !lbs_w || !verts)
        H3D_FAIL(H3D_ERR_ARG, "smpl_verts2: null pointer");
    if (P <= 0 || V <= 0 || nnz <= 0 || nnz > 4 || Ppad % SV_PB || Ppad < P || Vpad % SV_VT || Vpad < V)
        H3D_FAIL(H3D_ERR_SHAPE, "smpl_verts2: P=%d (pad %d, multiple of %d) V=%d (pad %d, multiple of %d) nnz=%d (<= 4)", P, Ppad,
                 SV_PB, V, Vpad, SV_VT, nnz);
    dim3 grid(Vpad / SV_VT, Ppad / SV_PB);
    hipLaunchKernelGGL(smpl_verts2_kernel, grid, dim3(256), 0, (hipStream_t)stream, coefT, A, v_template, shapedirsT, posedirsT,
                       lbs_idx, lbs_w, nnz, P, Ppad, V, Vpad, verts);
    H3D_CHECK_LAUNCH("smpl_verts2_kernel");
    return H3D_OK;
#endif
}

extern "C" int h3d_smpl_verts(const float *betas, const float *pose_feat, const float *A, const float *v_template,
                              const float *shapedirsT, const float *posedirsT, const int32_t *lbs_idx, const float *lbs_w,
                              int nnz, int P, int V, int Vpad, float *verts, void *stream)
{
    if (!betas || !pose_feat || !A || !v_template || !shapedirsT || !posedirsT || !lbs_idx || !lbs_w || !verts)
        H3D_FAIL(H3D_ERR_ARG, "smpl_verts: null pointer");
    if (P <= 0 || V <= 0 || Vpad < V || nnz <= 0 || nnz > SMPL_J) H3D_FAIL(H3D_ERR_SHAPE, "smpl_verts: P=%d V=%d nnz=%d", P, V, nnz);
    dim3 grid(cdiv(V, 256), 1);
    if (P >= 64) {
        constexpr int PT = 32;
        grid.y = cdiv(P, PT);
        hipLaunchKernelGGL(smpl_verts_kernel<PT>, grid, dim3(256), 0, (hipStream_t)stream, betas, pose_feat, A, v_template,
                           shapedirsT, posedirsT, lbs_idx, lbs_w, nnz, P, V, Vpad, verts);
    } else {
        constexpr int PT = 8;
        grid.y = cdiv(P, PT);
        hipLaunchKernelGGL(smpl_verts_kernel<PT>, grid, dim3(256), 0, (hipStream_t)stream, betas, pose_feat, A, v_template,
                           shapedirsT, posedirsT, lbs_idx, lbs_w, nnz, P, V, Vpad, verts);
    }
    H3D_CHECK_LAUNCH("smpl_verts_kernel");
    return H3D_OK;
}

// ------------------------------------------------------------------------------------------------
// Generation 3 of the vertex kernel: the blend-shape contraction on the matrix cores.
// Gen 2 runs its 57 GFLOP (batch 6400) on the vector ALU at ~40 % of the packed-FMA peak.  The fp32 matrix
// instruction has the same peak as the vector ALU (64 FLOP/clk/SIMD: a first version on v_mfma_f32_32x32x2_f32
// took 1.0 ms against gen 2's 0.9), so the [3*V x 217] x [217 x P] product runs on the bf16 pipe (16x that
// rate) with every fp32 operand split into three bf16 terms x = h + m + l (exact to 24 significant bits) and the
// six products that matter (hh, hm, mh, mm, hl, lh; the dropped ones are <= 2^-24 relative) accumulated in fp32:
// fp32-level accuracy (tests: 2e-6 against gen 2, 1e-4 against the fp64 oracle) at 6/16 of the fp32 cost.
//   operands   K padded to 224 = 14 steps of 16; per row and step the three bf16 terms sit side by side
//              ([16 h][16 m][16 l] = 96 B): dirsK3 [3][Vpad][14][3][16] (host pack), coefK3 [Ppad][14][3][16]
//              (h3d_smpl_coef_pack).  One step per stage streams through a 2-slot LDS ring by LDS-DMA; row stride
//              112 B (96 + 16 pad: odd multiple of 16 B, conflict-free ds_read_b128); one barrier per 36 MFMAs.
//   workgroup  64 vertices x 256 persons, 8 waves; wave w = persons [32w, 32w+32) x 64 vertices x 3 coordinates
//              (6 accumulator tiles, the two tiles of a coordinate alternating in the MFMA stream).  170 B of LDS-DMA
//              per MFMA instead of the 249 B of the first version (128 persons, 4 waves): 0.39 -> 0.30 ms for the
//              contraction, which is bound by its data movement, not by the matrix pipe (0.155 ms at full rate):
//              `make ABLATE=1` + H3D_SMPL_ABLATE: without the MFMAs it still takes 0.18 ms, without the direction
//              reads 0.20, without the DMA 0.24.  Tried and measured neutral: an XCD-aware tile order (directions
//              resident in one L2), a third ring slot (two stages in flight), DMA pieces issued between the MFMA groups.
//   skinning   gen 2's 4-sparse LBS with lane = vertex: the accumulators (lane = person in the MFMA C layout)
//              are transposed through LDS 8 persons at a time (person stride 193 dwords), so the transforms are
//              broadcast-friendly reads and every person's 64 vertices leave as one contiguous 768-byte run.
//              The staging areas are private to a wave: wave-level ordering only, no workgroup barrier.
constexpr int S3_KP = 224, S3_NST = S3_KP / 16, S3_VT = 64, S3_NW = 8, S3_PB = 32 * S3_NW;   // 8 waves x 32 persons
constexpr int S3_PPAD = 128;                                  // the hosts pad the person count to this
constexpr int S3_SPR = 7;                                     // 16-byte slots per row and stage: 6 data + 1 pad
constexpr int S3_ROWB = S3_SPR * 16;                          // 112
// slots of a row the DMA fetches: all six (h, m, l: SIX = true) or h and m only -- the lanes of the l slots then carry an out-of-range
// offset (zeros, no traffic)
template <bool SIX> constexpr int s3_dslots() { return SIX ? 6 : 4; }
constexpr int S3_GROW = S3_NST * 96;                          // bytes of a row in global memory: 1344
constexpr int S3_APIECES = 3 * S3_VT * S3_SPR / 64;           // 21 KiB pieces of direction rows
constexpr int S3_BPIECES = S3_PB * S3_SPR / 64;               // 28 of coefficient rows
constexpr int S3_SLOT = (S3_APIECES + S3_BPIECES) * 1024;     // 50176
constexpr int S3_RP = 8;                                      // persons per skinning round and wave
constexpr int S3_TSTRIDE = 193 * 4;                           // transposed tile: bytes per person (64 v x 3 floats + 1)
constexpr int S3_LBS = S3_NW * S3_RP * S3_TSTRIDE + S3_NW * S3_RP * SMPL_J * 12 * 4;   // 49408 + 73728
constexpr int S3_LDS = 2 * S3_SLOT > S3_LBS ? 2 * S3_SLOT : S3_LBS;

constexpr int S3_AJ = (S3_APIECES + S3_NW - 1) / S3_NW, S3_BJ = (S3_BPIECES + S3_NW - 1) / S3_NW;   // pieces per wave: 3 + 4

// piece j of this wave (0 .. S3_AJ-1 direction rows, then coefficient rows) of stage st -> slot
__device__ __forceinline__ void s3_issue_piece(const char *dirsK, int dbytes, const char *coefK, int cbytes, char *slot,
                                               const int *aoff, const int *boff, int wv, int st, int j)
{
    if (j < S3_AJ) {
        const auto ra = __builtin_amdgcn_make_buffer_rsrc((void *)dirsK, 0, dbytes, 0x00020000);
        const int p = wv + S3_NW * j;
        if (p < S3_APIECES)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void *)(slot + p * 1024), 16, aoff[j], st * 96,
                                                     0, 0);
    } else {
        const auto rb = __builtin_amdgcn_make_buffer_rsrc((void *)coefK, 0, cbytes, 0x00020000);
        const int p = wv + S3_NW * (j - S3_AJ);
        if (p < S3_BPIECES)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void *)(slot + (S3_APIECES + p) * 1024), 16,
                                                     boff[j - S3_AJ], st * 96, 0, 0);
    }
}

__device__ __forceinline__ void s3_issue(const char *dirsK, int dbytes, const char *coefK, int cbytes, char *slot,
                                         const int *aoff, const int *boff, int wv, int st)
{
#pragma unroll
    for (int j = 0; j < S3_AJ + S3_BJ; ++j) s3_issue_piece(dirsK, dbytes, coefK, cbytes, slot, aoff, boff, wv, st, j);
}

template <bool SIX>      // SIX: all six products of the three-term split (2^-24 relative: the f32-mode detectors), else hh + hm + mh (2^-16)
__global__ __launch_bounds__(64 * S3_NW) void smpl_verts3_kernel(const bf16_t *__restrict__ coefK3, const float *__restrict__ A,
                                                          const float *__restrict__ v_template,
                                                          const bf16_t *__restrict__ dirsK3, const int32_t *__restrict__ lbs_idx,
                                                          const float *__restrict__ lbs_w, int nnz, int P, int Ppad, int V,
                                                          int Vpad, float *__restrict__ verts)
{
    using E = ET<bf16_t>;
    constexpr int S3_DSLOTS = s3_dslots<SIX>();
    __shared__ __attribute__((aligned(1024))) char smem[S3_LDS];
    const int tid = threadIdx.x, l = tid & 63, r = l & 31, h = l >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int v0 = blockIdx.x * S3_VT, p0 = blockIdx.y * S3_PB;
#ifdef H3D_ABLATE
    const bool contraction_only = nnz & 0x100;         // profiling flags ride on nnz
    const int ablate_mode = (nnz >> 9) & 3;
    nnz &= 0xff;
#endif

    // per-lane DMA source offsets (stage 0) of my pieces: slot q -> row q / 7, 16-byte column q % 7 (6 = pad)
    static_assert(S3_APIECES % S3_NW != 0 && S3_BPIECES % S3_NW != 0 && S3_AJ + S3_BJ == 7, "piece counts behind the vmcnt immediates");
    int aoff[S3_AJ], boff[S3_BJ];
#pragma unroll
    for (int j = 0; j < (S3_APIECES + S3_NW - 1) / S3_NW; ++j) {
        const int q = (wv + S3_NW * j) * 64 + l;
        const int row = q / S3_SPR, sub = q - S3_SPR * row;           // row = c * 64 + v
        const int c = row >> 6, v = row & 63;
        aoff[j] = (sub < S3_DSLOTS && row < 3 * S3_VT) ? (c * Vpad + v0 + v) * S3_GROW + sub * 16 : 0x7ffffff0;
    }
#pragma unroll
    for (int j = 0; j < (S3_BPIECES + S3_NW - 1) / S3_NW; ++j) {
        const int q = (wv + S3_NW * j) * 64 + l;
        const int row = q / S3_SPR, sub = q - S3_SPR * row;           // row = person inside the workgroup
        boff[j] = (sub < S3_DSLOTS && row < S3_PB) ? (p0 + row) * S3_GROW + sub * 16 : 0x7ffffff0;
    }
    const int dbytes = 3 * Vpad * S3_GROW, cbytes = Ppad * S3_GROW;

    f32x16 acc[3][2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[c][t][i] = 0.f;

    // the 3x4 transforms of a skinning round (8 persons per wave: [8][288] floats = 576 float4, 9 per lane) are fetched
    // into registers one round ahead -- round 0 before the contraction -- so no round waits on global memory
    constexpr int NA4 = S3_RP * SMPL_J * 3 / 64;          // 9
    f32x4 apre[NA4];
    auto load_A = [&](int rnd) {
#pragma unroll
        for (int k = 0; k < NA4; ++k) {
            const int i = l + 64 * k;
            const int q = i / (SMPL_J * 3), rr = i - q * (SMPL_J * 3);
            const int p = p0 + wv * 32 + rnd * S3_RP + q;
            apre[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p < P) apre[k] = *reinterpret_cast<const f32x4 *>(A + (size_t)p * SMPL_J * 12 + 4 * rr);
        }
    };
    load_A(0);
    // this lane's vertex in the skinning phase: template position and (<= 4) joint indices / weights, also fetched early
    const int v = v0 + l;
    const int vc = v < V ? v : V - 1;
    int jidx[4];
    float jw[4];
#pragma unroll
    for (int sI = 0; sI < 4; ++sI) {
        jidx[sI] = sI < nnz ? lbs_idx[(size_t)vc * nnz + sI] : 0;
        jw[sI] = sI < nnz ? lbs_w[(size_t)vc * nnz + sI] : 0.f;
    }
    const float t0 = v_template[vc], t1 = v_template[Vpad + vc], t2 = v_template[2 * Vpad + vc];
    s3_issue((const char *)dirsK3, dbytes, (const char *)coefK3, cbytes, smem, aoff, boff, wv, 0);
    const int fa_off = r * S3_ROWB + h * 16;                                   // + (c * 64 + t * 32) rows, + part * 32
    const int fb_off = S3_APIECES * 1024 + (wv * 32 + r) * S3_ROWB + h * 16;
    // MODE (profiling, ABLATE builds): 0 the contraction, 1 without the MFMAs, 2 without the direction-fragment reads,
    // 3 without the DMA after stage 0
    auto contract = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        for (int st = 0; st < S3_NST; ++st) {
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
            if (st + 1 < S3_NST && MODE != 3)
                s3_issue((const char *)dirsK3, dbytes, (const char *)coefK3, cbytes, smem + ((st + 1) & 1) * S3_SLOT, aoff, boff, wv, st + 1);
            const char *sl = smem + (st & 1) * S3_SLOT;
            // SIX: all six products down to 2^-24 relative.  Otherwise hh + hm + mh: the three dropped products (mm, hl, lh) are
            // 2^-16 relative each -- 2.2e-6 abs on the blend-shape displacement (fp64 emulation, |d| <= 0.34), 45x inside the
            // 1e-4 tolerance -- for half the MFMAs and two thirds of the fragment reads
            const E::frag bh = E::lds_frag(sl + fb_off), bm = E::lds_frag(sl + fb_off + 32);
            E::frag bl = bh;
            if constexpr (SIX) bl = E::lds_frag(sl + fb_off + 64);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                // the two vertex tiles of a coordinate alternate, so consecutive MFMAs do not share an accumulator
                const char *ap0 = sl + fa_off + (c * 64) * S3_ROWB, *ap1 = ap0 + 32 * S3_ROWB;
                E::frag ah0 = bh, am0 = bm, al0 = bl, ah1 = bh, am1 = bm, al1 = bl;
                if constexpr (MODE != 2) {
                    ah0 = E::lds_frag(ap0); am0 = E::lds_frag(ap0 + 32);
                    ah1 = E::lds_frag(ap1); am1 = E::lds_frag(ap1 + 32);
                    if constexpr (SIX) { al0 = E::lds_frag(ap0 + 64); al1 = E::lds_frag(ap1 + 64); }
                }
                if constexpr (MODE == 1) {
                    asm volatile("" ::"v"(ah0.v), "v"(am0.v), "v"(al0.v), "v"(ah1.v), "v"(am1.v), "v"(al1.v), "v"(bh.v), "v"(bm.v), "v"(bl.v));
                } else {
                    if constexpr (SIX) {
                        E::mma(acc[c][0], al0, bh);      // smallest terms first
                        E::mma(acc[c][1], al1, bh);
                        E::mma(acc[c][0], ah0, bl);
                        E::mma(acc[c][1], ah1, bl);
                        E::mma(acc[c][0], am0, bm);
                        E::mma(acc[c][1], am1, bm);
                    }
                    E::mma(acc[c][0], am0, bh);
                    E::mma(acc[c][1], am1, bh);
                    E::mma(acc[c][0], ah0, bm);
                    E::mma(acc[c][1], ah1, bm);
                    E::mma(acc[c][0], ah0, bh);
                    E::mma(acc[c][1], ah1, bh);
                }
            }
        }
    };
#ifdef H3D_ABLATE
    if (ablate_mode == 1) contract(std::integral_constant<int, 1>{});
    else if (ablate_mode == 2) contract(std::integral_constant<int, 2>{});
    else if (ablate_mode == 3) contract(std::integral_constant<int, 3>{});
    else
#endif
        contract(std::integral_constant<int, 0>{});
    __syncthreads();                                   // the ring is free: it becomes the transpose + transform staging area
#ifdef H3D_ABLATE
    if (contraction_only) {
        if (acc[0][0][0] == 12345.678f) verts[0] = acc[1][1][3] + acc[2][0][5] + acc[0][1][7] + acc[1][0][2] + acc[2][1][9];
        return;
    }
#endif

    // ---- skinning, lane = vertex (as gen 2): four rounds of 8 persons per wave -------------------------
    char *sT = smem + wv * (S3_RP * S3_TSTRIDE);                                // [8 persons][64 v][3] (+1)
    float *sA = reinterpret_cast<float *>(smem + S3_NW * S3_RP * S3_TSTRIDE) + wv * (S3_RP * SMPL_J * 12);
#pragma unroll
    for (int rnd = 0; rnd < 32 / S3_RP; ++rnd) {
        // sT and sA are private to the wave and LDS executes a wave's instructions in order: wave-level ordering is all
        // the rounds need, and without workgroup barriers the eight waves drift apart, one's transposition (LDS)
        // under another's blend (vector ALU)
        __builtin_amdgcn_wave_barrier();
        // (a) my accumulators of persons [8 rnd, 8 rnd + 8): C layout row = vertex, column = person
        if ((r / S3_RP) == rnd) {
            const int pl = r % S3_RP;
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int vl = t * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                        *reinterpret_cast<float *>(sT + pl * S3_TSTRIDE + (vl * 3 + c) * 4) = acc[c][t][i];
                    }
        }
        // (b) the 3x4 transforms of those 8 persons (prefetched), then the next round's go in flight
#pragma unroll
        for (int k = 0; k < NA4; ++k) {
            const int i = l + 64 * k;
            const int q = i / (SMPL_J * 3), rr = i - q * (SMPL_J * 3);
            *reinterpret_cast<f32x4 *>(sA + q * SMPL_J * 12 + 4 * rr) = apre[k];
        }
        __builtin_amdgcn_wave_barrier();
        if (rnd + 1 < 32 / S3_RP) load_A(rnd + 1);
        // (c) lane = vertex
#pragma unroll 4
        for (int q = 0; q < S3_RP; ++q) {
            const int p = p0 + wv * 32 + rnd * S3_RP + q;
            float T[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) T[i] = 0.f;
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) {
                const float *Ap = sA + q * SMPL_J * 12 + jidx[sI] * 12;
#pragma unroll
                for (int i4 = 0; i4 < 3; ++i4) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4 *>(Ap + 4 * i4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) T[4 * i4 + e] = fmaf(jw[sI], a4[e], T[4 * i4 + e]);
                }
            }
            const float *xyz = reinterpret_cast<const float *>(sT + q * S3_TSTRIDE + l * 12);
            const float x = t0 + xyz[0], y = t1 + xyz[1], z = t2 + xyz[2];
            if (v < V && p < P) {
                typedef __attribute__((ext_vector_type(3))) float f32x3;     // one 12-byte store per lane
                const f32x3 o3 = {T[0] * x + T[1] * y + T[2] * z + T[3], T[4] * x + T[5] * y + T[6] * z + T[7],
                                  T[8] * x + T[9] * y + T[10] * z + T[11]};
                __builtin_memcpy(verts + ((size_t)p * V + v) * 3, &o3, 12);
            }
        }
    }
}

// three-term bf16 split of an fp32 value: x = h + m + l up to 2^-24 relative (round-to-nearest-even at each step)
__device__ __forceinline__ void split3(float x, bf16_t &hh, bf16_t &mm, bf16_t &ll)
{
    hh = ET<bf16_t>::from_f32(x);
    const float r1 = x - ET<bf16_t>::to_f32(hh);
    mm = ET<bf16_t>::from_f32(r1);
    const float r2 = r1 - ET<bf16_t>::to_f32(mm);
    ll = ET<bf16_t>::from_f32(r2);
}

// coefK3 [Ppad][14][3][16] bf16: per person and K step the h / m / l terms of [beta | pose_feat | 0]; zero rows for p >= P
__global__ void smpl_coef_pack_kernel(const float *__restrict__ betas, const float *__restrict__ pose_feat, int P, int Ppad,
                                      bf16_t *__restrict__ coefK3)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)Ppad * S3_KP) return;
    const int p = (int)(i / S3_KP), k = (int)(i - (size_t)p * S3_KP);
    float v = 0.f;
    if (p < P) {
        if (k < SMPL_NB) v = betas[(size_t)p * SMPL_NB + k];
        else if (k < SMPL_NB + SMPL_PF) v = pose_feat[(size_t)p * SMPL_PF + (k - SMPL_NB)];
    }
    bf16_t hh, mm, ll;
    split3(v, hh, mm, ll);
    bf16_t *o = coefK3 + ((size_t)p * S3_NST + (k >> 4)) * 48 + (k & 15);
    o[0] = hh; o[16] = mm; o[32] = ll;
}

extern "C" int h3d_smpl_coef_pack(const float *betas, const float *pose_feat, int P, int Ppad, void *coefK3, void *stream)
{
    if (!betas || !pose_feat || !coefK3) H3D_FAIL(H3D_ERR_ARG, "smpl_coef_pack: null pointer");
    if (P <= 0 || Ppad < P || Ppad % S3_PPAD) H3D_FAIL(H3D_ERR_SHAPE, "smpl_coef_pack: P=%d Ppad=%d (multiple of %d)", P, Ppad, S3_PPAD);
    const size_t n = (size_t)Ppad * S3_KP;
    hipLaunchKernelGGL(smpl_coef_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, betas, pose_feat, P,
                       Ppad, (bf16_t *)coefK3);
    H3D_CHECK_LAUNCH("smpl_coef_pack_kernel");
    return H3D_OK;
}

static int smpl_verts3_impl(bool six, const void *coefK3, const float *A, const float *v_template, const void *dirsK3,
                            const int32_t *lbs_idx, const float *lbs_w, int nnz, int P, int Ppad, int V, int Vpad,
                            float *verts, void *stream)
{
    if (!coefK3 || !A || !v_template || !dirsK3 || !lbs_idx || !lbs_w || !verts) H3D_FAIL(H3D_ERR_ARG, "smpl_verts3: null pointer");
#ifdef H3D_ABLATE
    const char *abl = getenv("H3D_SMPL_ABLATE");      // bit 0: stop after the contraction; bits 1-2: contraction mode
    const int nnz_flags = abl ? ((atoi(abl) & 1) << 8) | (((atoi(abl) >> 1) & 3) << 9) : 0;
#else
    const int nnz_flags = 0;
#endif
    if (P <= 0 || V <= 0 || nnz <= 0 || nnz > 4 || Ppad % S3_PPAD || Ppad < P || Vpad % S3_VT || Vpad < V)
        H3D_FAIL(H3D_ERR_SHAPE, "smpl_verts3: P=%d (pad %d, multiple of %d) V=%d (pad %d, multiple of %d) nnz=%d (<= 4)", P, Ppad,
                 S3_PPAD, V, Vpad, S3_VT, nnz);
    if ((size_t)3 * Vpad * S3_GROW >= 0x7ffffff0ull || (size_t)Ppad * S3_GROW >= 0x7ffffff0ull)
        H3D_FAIL(H3D_ERR_SHAPE, "smpl_verts3: operand of 2 GiB or more");
    dim3 grid(Vpad / S3_VT, cdiv(Ppad, S3_PB));      // (coefficient rows past Ppad are outside the buffer: zeros)
    if (six)
        hipLaunchKernelGGL(smpl_verts3_kernel<true>, grid, dim3(64 * S3_NW), 0, (hipStream_t)stream, (const bf16_t *)coefK3, A, v_template,
                           (const bf16_t *)dirsK3, lbs_idx, lbs_w, nnz | nnz_flags, P, Ppad, V, Vpad, verts);
    else
        hipLaunchKernelGGL(smpl_verts3_kernel<false>, grid, dim3(64 * S3_NW), 0, (hipStream_t)stream, (const bf16_t *)coefK3, A, v_template,
                           (const bf16_t *)dirsK3, lbs_idx, lbs_w, nnz | nnz_flags, P, Ppad, V, Vpad, verts);
    H3D_CHECK_LAUNCH("smpl_verts3_kernel");
    return H3D_OK;
}

extern "C" int h3d_smpl_verts3(const void *coefK3, const float *A, const float *v_template, const void *dirsK3,
                               const int32_t *lbs_idx, const float *lbs_w, int nnz, int P, int Ppad, int V, int Vpad,
                               float *verts, void *stream)
{
    return smpl_verts3_impl(false, coefK3, A, v_template, dirsK3, lbs_idx, lbs_w, nnz, P, Ppad, V, Vpad, verts, stream);
}

extern "C" int h3d_smpl_verts3_exact(const void *coefK3, const float *A, const float *v_template, const void *dirsK3,
                                     const int32_t *lbs_idx, const float *lbs_w, int nnz, int P, int Ppad, int V, int Vpad,
                                     float *verts, void *stream)
{
    return smpl_verts3_impl(true, coefK3, A, v_template, dirsK3, lbs_idx, lbs_w, nnz, P, Ppad, V, Vpad, verts, stream);
}
