// CenterNet heat-map decode on gfx950 (reference: /root/reference/src/lib/models/decode.py,
// utils.py).  Bit-exact with the reference's CPU results on identical fp32 inputs: every float
// operation below is a single IEEE op in the reference's order (this file is compiled with
// -ffp-contract=off), and index selection is an exact integer radix select.
//
//   nms_topk_kernel      _sigmoid (utils.py:8-10, optional) + _nms (decode.py:6-13) +
//                        per-map top-K (decode.py:18/29) -- one workgroup per (b, c) map;
//                        map keys live in LDS, 4x8-bit radix select for the K-th key, ordered
//                        compaction of ties (lowest index first), bitonic sort of the K survivors.
//   topk_merge_kernel    stage 2 of _topk (decode.py:34-39)
//   gather_feat_kernel   _transpose_and_gather_feat (utils.py:23-27) without the NHWC transpose
//   pose_assemble_kernel multi_pose_decode body (decode.py:86-161)
//   ctdet_assemble_kernel ctdet_decode body (decode.py:52-75)
//   post_process_kernel  multi_pose_post_process (utils/post_process.py:41-52)
#include "common.h"

typedef unsigned long long u64;

__device__ __forceinline__ uint32_t fkey(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// descending bitonic sort of N (power of two) 64-bit keys in LDS, all `nthreads` threads call
__device__ void bitonic_desc(u64 *s, int N, int tid, int nthreads)
{
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < N; i += nthreads) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const u64 x = s[i], y = s[ixj];
                    const bool desc_blk = ((i & k) == 0);
                    if (desc_blk ? (x < y) : (x > y)) { s[i] = y; s[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ float sigmoid_clamp(float x)
{
    // clamp(sigmoid(x), 1e-4, 1-1e-4); expf (not __expf) keeps full fp32 accuracy
    float y = 1.0f / (1.0f + expf(-x));
    return fminf(fmaxf(y, 1e-4f), 1.0f - 1e-4f);
}

constexpr int NMS_THREADS = 1024;   // 16 waves: a (b, c) map is one workgroup, so latency is hidden by waves, not by workgroups
constexpr int NMS_WAVES = NMS_THREADS / 64;
constexpr int NMS_MAXK = 1024;

// exclusive prefix sum of `v` over the workgroup's threads (wave shuffles + one LDS word per wave);
// *total = sum over all threads.  Two barriers; s_wave[NMS_WAVES] is free again on return.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_wave, int tid, uint32_t *total)
{
    const int lane = tid & 63, w = tid >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_wave[w] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int j = 0; j < NMS_WAVES; ++j) {
        const uint32_t x = s_wave[j];
        if (j < w) base += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// One launch can serve two heat-map tensors of the same H x W and K (the detector's `hm` and `hm_hp`: 1 and 17 maps per
// image; on its own the 64-workgroup `hm` launch occupies a quarter of the CUs for a full map latency).
struct NmsJob {
    const float *heat;
    float *o_score;
    int64_t *o_ind;
    float *o_y, *o_x;
    int nblk;       // B * C maps
};

// Maps larger than the LDS (H * W > 36864, e.g. the 320 x 184 output of a --keep_res 1280 x 736 frame, datasets/coco.py:160-163)
// are cut into `nbands` bands of `band_rows` rows: workgroup (map, band) loads its rows plus one halo row on either side (the
// 3x3 max needs them), only the band's own rows are candidates, and the band's top K go to a [maps][nbands][K] scratch that
// topk_merge_kernel reduces (same order: score descending, lowest flat index first).  nbands = 1 is the plain case.
__global__ __launch_bounds__(NMS_THREADS) void nms_topk_kernel(NmsJob j0, NmsJob j1, int H_img, int W, int K, int ncand, int flags,
                                                               int band_rows, int nbands)
{
    int blk = blockIdx.x / nbands;
    const int band = blockIdx.x - blk * nbands;
    NmsJob jb = j0;
    if (blk >= j0.nblk) { blk -= j0.nblk; jb = j1; }     // (workgroup-uniform)
    const int out_blk = blk * nbands + band;             // output row: [map][band]
    const int cy0 = band * band_rows, cy1 = min(H_img, cy0 + band_rows);      // candidate rows
    const int ly0 = max(cy0 - 1, 0), ly1 = min(cy1 + 1, H_img);              // rows held in LDS
    const int H = ly1 - ly0;                                                 // (from here on H, HW and i are LOCAL)
    const int c_lo = (cy0 - ly0) * W, c_hi = (cy1 - ly0) * W;                // local pixel range of the candidates
    const float *__restrict__ heat = jb.heat;
    float *__restrict__ o_score = jb.o_score;
    int64_t *__restrict__ o_ind = jb.o_ind;
    float *__restrict__ o_y = jb.o_y, *__restrict__ o_x = jb.o_x;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_key[];  // [HW] keys, then candidates, then keep bits
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_wave[NMS_WAVES];
    __shared__ uint32_t s_sel[2];  // [0] = chosen bin, [1] = remaining need
    __shared__ uint32_t s_cnt;
    const int tid = threadIdx.x;
    const int HW = H * W;
    const float *map = heat + (size_t)blk * H_img * W + (size_t)ly0 * W;
    u64 *s_cand = reinterpret_cast<u64 *>(s_key + ((HW + 1) & ~1));
    u64 *s_mask = s_cand + ncand;             // keep bits, one word per 64 pixels

    // ---- sigmoid + 3x3 NMS -> order-preserving keys ----------------------------------------------
    // pass 1: (optional) sigmoid of every pixel ONCE, straight into LDS; 8 independent loads per
    //         thread in flight per batch (8192 per workgroup)
    float *s_val = reinterpret_cast<float *>(s_key);
    const int wshift = (W & (W - 1)) == 0 ? __builtin_ctz(W) : -1;             // heat maps are 128 wide in the detector
    for (int base = 0; base < HW; base += 8 * NMS_THREADS) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = base + j * NMS_THREADS + tid;
            v[j] = (i < HW) ? map[i] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = base + j * NMS_THREADS + tid;
            if (i < HW) s_val[i] = (flags & 1) ? sigmoid_clamp(v[j]) : v[j];
        }
    }
    __syncthreads();
    // pass 2: keep = (3x3 max == centre) on the LDS map (the reference pools the sigmoid outputs,
    //         decode.py:6-13); one ballot word per 64 consecutive pixels
    for (int base = 0; base < HW; base += NMS_THREADS) {
        const int i = base + tid;
        bool keep = false;
        if (i < HW) {
            const int y = wshift >= 0 ? (i >> wshift) : i / W, x = i - y * W;     // (a division costs ~25 instructions per pixel)
            const float v = s_val[i];
            float m = v;
            if (!(flags & 2)) {
                const int y0 = max(y - 1, 0), y1 = min(y + 1, H - 1), x0 = max(x - 1, 0), x1 = min(x + 1, W - 1);
                // clamped taps repeat an in-range neighbour (or the centre): the max is unchanged
                const float *r0 = s_val + y0 * W, *r1 = s_val + y * W, *r2 = s_val + y1 * W;
                const float a0 = r0[x0], a1 = r0[x], a2 = r0[x1], b0 = r1[x0], b2 = r1[x1], c0 = r2[x0], c1 = r2[x],
                            c2 = r2[x1];
                m = fmaxf(fmaxf(fmaxf(fmaxf(a0, a1), fmaxf(a2, b0)), fmaxf(fmaxf(b2, c0), fmaxf(c1, c2))), v);
            }
            keep = (m == v);
        }
        const u64 word = __ballot(keep);
        if ((tid & 63) == 0 && i < HW) s_mask[i >> 6] = word;        // i is a multiple of 64 in lane 0
    }
    __syncthreads();
    // pass 3: heat * keep -> key, in place (+0.0f folds -0 into +0 so equal values share one key)
    for (int base = 0; base < HW; base += NMS_THREADS) {
        const int i = base + tid;
        if (i < HW) {
            const float sv = s_val[i];
            const bool keep = (s_mask[i >> 6] >> (i & 63)) & 1ull;
            const float val = (keep ? sv : sv * 0.0f) + 0.0f;
            s_key[i] = (i >= c_lo && i < c_hi) ? fkey(val) : 0u;     // halo rows of a band: below every real key, never selected
        }
    }
    __syncthreads();

#ifdef H3D_ABLATE
    if (flags & 0x100) return;
#endif
    // ---- radix select: key of the K-th largest element ------------------------------------------
    uint32_t prefix = 0, pmask = 0, need = (uint32_t)K;
    for (int pass = 3; pass >= 0; --pass) {
        if (tid < 256) s_hist[tid] = 0;
        __syncthreads();
        const int sh = pass * 8;
        // run-length aggregation: NMS leaves most of a heat map at exactly 0 (one hot bin), so a plain
        // atomic per element would serialise ~HW LDS atomics on one address
        uint32_t run_bin = 0xffffffffu, run_cnt = 0;
        for (int i = tid; i < HW; i += NMS_THREADS) {
            const uint32_t k = s_key[i];
            if ((k & pmask) == prefix) {
                const uint32_t bin = (k >> sh) & 255u;
                if (bin == run_bin) {
                    ++run_cnt;
                } else {
                    if (run_cnt) atomicAdd(&s_hist[run_bin], run_cnt);
                    run_bin = bin;
                    run_cnt = 1;
                }
            }
        }
        if (run_cnt) atomicAdd(&s_hist[run_bin], run_cnt);
        __syncthreads();
        // thread t < 256 owns bin t: elements in bins > t = total - inclusive prefix
        const uint32_t hv = (tid < 256) ? s_hist[tid] : 0u;
        uint32_t total;
        const uint32_t excl = block_excl_scan(hv, s_wave, tid, &total);
        const uint32_t above = total - excl - hv;
        if (tid < 256 && above + hv >= need && above < need) { s_sel[0] = (uint32_t)tid; s_sel[1] = need - above; }
        __syncthreads();
        prefix |= s_sel[0] << sh;
        pmask |= 0xffu << sh;
        need = s_sel[1];
        __syncthreads();
    }
    const uint32_t kth = prefix;            // exact key of the K-th largest
    const uint32_t n_gt = (uint32_t)K - need;  // elements strictly greater; take `need` equal ones

#ifdef H3D_ABLATE
    if (flags & 0x200) return;
#endif
    // ---- compaction: all keys > kth (any order), then `need` keys == kth in index order ----------
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int i = tid; i < HW; i += NMS_THREADS) {
        const uint32_t k = s_key[i];
        if (k > kth) {
            const uint32_t pos = atomicAdd(&s_cnt, 1u);
            s_cand[pos] = ((u64)k << 32) | (u64)(0xffffffffu - (uint32_t)i);
        }
    }
    const int chunk = (HW + NMS_THREADS - 1) / NMS_THREADS;
    const int lo = tid * chunk, hi = min(lo + chunk, HW);
    uint32_t eq = 0;
    for (int i = lo; i < hi; ++i) eq += (s_key[i] == kth);
    uint32_t total_eq;
    uint32_t rank = block_excl_scan(eq, s_wave, tid, &total_eq);
    for (int i = lo; i < hi && rank < need; ++i)
        if (s_key[i] == kth) {
            s_cand[n_gt + rank] = ((u64)kth << 32) | (u64)(0xffffffffu - (uint32_t)i);
            ++rank;
        }
    int N = 1;
    while (N < K) N <<= 1;
    for (int i = K + tid; i < N; i += NMS_THREADS) s_cand[i] = 0ull;
    __syncthreads();
#ifdef H3D_ABLATE
    if (flags & 0x400) return;
#endif
    bitonic_desc(s_cand, N, tid, NMS_THREADS);

    for (int j = tid; j < K; j += NMS_THREADS) {
        const u64 c = s_cand[j];
        const uint32_t idx = 0xffffffffu - (uint32_t)(c & 0xffffffffull) + (uint32_t)(ly0 * W);     // global flat index
        const size_t o = (size_t)out_blk * K + j;
        o_score[o] = fkey_inv((uint32_t)(c >> 32));
        o_ind[o] = (int64_t)idx;
        o_y[o] = (float)(int)(idx / (uint32_t)W);
        o_x[o] = (float)(int)(idx % (uint32_t)W);
    }
}

constexpr long NMS_MAXHW = 36864;      // pixels of a map (or of a band + its halo rows) the LDS holds

static int nms_topk_launch(const NmsJob &j0, const NmsJob &j1, int H_img, int W, int K, int flags, void *stream, int band_rows = 0, int nbands = 1)
{
    if (band_rows <= 0) band_rows = H_img;
    const int H = min(H_img, band_rows + 2);             // rows in LDS (band + halo)
    const long HW = (long)H * W;
    if (K <= 0 || K > (long)min(band_rows, H_img - (nbands - 1) * band_rows) * W)
        H3D_FAIL(H3D_ERR_SHAPE, "nms_topk: selected index k out of range (K=%d, %d x %d pixels per band)", K, band_rows, W);
    if (K > NMS_MAXK || HW > NMS_MAXHW)
        H3D_FAIL(H3D_ERR_UNSUPPORTED, "nms_topk: K=%d (max %d), H*W=%ld (max %ld: larger maps through h3d_nms_topk_large)", K, NMS_MAXK, HW, NMS_MAXHW);
    int N = 1;
    while (N < K) N <<= 1;
    const size_t lds = (size_t)((HW + 1) & ~1L) * 4 + (size_t)N * 8 + (size_t)((HW + 63) / 64) * 8;
    static thread_local size_t max_set = 0;
    if (lds > 64 * 1024 && lds > max_set) {
        if (hipFuncSetAttribute((const void *)nms_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            H3D_FAIL(H3D_ERR_LAUNCH, "nms_topk: cannot reserve %zu bytes of LDS", lds);
        max_set = lds;
    }
    hipLaunchKernelGGL(nms_topk_kernel, dim3((j0.nblk + j1.nblk) * nbands), dim3(NMS_THREADS), lds, (hipStream_t)stream, j0, j1, H_img, W, K, N, flags,
                       band_rows, nbands);
    H3D_CHECK_LAUNCH("nms_topk_kernel");
    return H3D_OK;
}

extern "C" int h3d_nms_topk(const float *heat, int B, int C, int H, int W, int K, int flags, float *scores,
                            int64_t *inds, float *ys, float *xs, void *stream)
{
    if (!heat || !scores || !inds || !ys || !xs) H3D_FAIL(H3D_ERR_ARG, "nms_topk: null pointer");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) H3D_FAIL(H3D_ERR_SHAPE, "nms_topk: bad shape");
    const NmsJob j0 = {heat, scores, inds, ys, xs, B * C}, j1 = {nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    return nms_topk_launch(j0, j1, H, W, K, flags, stream);
}

extern "C" int h3d_nms_topk2(const float *heat_a, int Ca, float *scores_a, int64_t *inds_a, float *ys_a, float *xs_a,
                             const float *heat_b, int Cb, float *scores_b, int64_t *inds_b, float *ys_b, float *xs_b,
                             int B, int H, int W, int K, int flags, void *stream)
{
    if (!heat_a || !scores_a || !inds_a || !ys_a || !xs_a || !heat_b || !scores_b || !inds_b || !ys_b || !xs_b)
        H3D_FAIL(H3D_ERR_ARG, "nms_topk2: null pointer");
    if (B <= 0 || Ca <= 0 || Cb <= 0 || H <= 0 || W <= 0) H3D_FAIL(H3D_ERR_SHAPE, "nms_topk2: bad shape");
    // the many-map tensor first: its workgroups fill the CUs while the few maps of the other ride along
    const NmsJob ja = {heat_a, scores_a, inds_a, ys_a, xs_a, B * Ca}, jb = {heat_b, scores_b, inds_b, ys_b, xs_b, B * Cb};
    return Ca >= Cb ? nms_topk_launch(ja, jb, H, W, K, flags, stream) : nms_topk_launch(jb, ja, H, W, K, flags, stream);
}

// Maps of any size: bands through nms_topk_kernel, then topk_merge_kernel over the bands of every map.  Scratch (caller-provided,
// h3d_nms_topk_large_workspace_bytes): per map and band K (score, index, y, x) candidates + one int32 per output for the band id.
static int nms_bands(int H, int W, int K, int *band_rows, int *nbands)
{
    const long maxrows = NMS_MAXHW / W - 2;              // a band + two halo rows must fit the LDS
    if (maxrows < 1) return H3D_ERR_UNSUPPORTED;
    const int nb = (int)((H + maxrows - 1) / maxrows);
    const int br = (H + nb - 1) / nb;                   // even split: the last band is at most nb - 1 rows shorter
    if ((long)(H - (nb - 1) * br) * W < K) return H3D_ERR_UNSUPPORTED;      // every band must hold K candidates
    *band_rows = br;
    *nbands = nb;
    return H3D_OK;
}

extern "C" size_t h3d_nms_topk_large_workspace_bytes(int B, int C, int H, int W, int K)
{
    int br = 0, nb = 0;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0 || W > NMS_MAXHW / 3 || nms_bands(H, W, K, &br, &nb) != H3D_OK) return 0;
    const size_t n = (size_t)B * C * nb * K;
    return n * (4 + 8 + 4 + 4) + (size_t)B * C * K * 4 + 256;
}

extern "C" int h3d_nms_topk_large(const float *heat, int B, int C, int H, int W, int K, int flags, float *scores, int64_t *inds,
                                  float *ys, float *xs, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!heat || !scores || !inds || !ys || !xs) H3D_FAIL(H3D_ERR_ARG, "nms_topk_large: null pointer");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) H3D_FAIL(H3D_ERR_SHAPE, "nms_topk_large: bad shape");
    if ((long)H * W <= NMS_MAXHW) return h3d_nms_topk(heat, B, C, H, W, K, flags, scores, inds, ys, xs, stream);
    int br = 0, nb = 0;
    if (W > NMS_MAXHW / 3 || nms_bands(H, W, K, &br, &nb) != H3D_OK)
        H3D_FAIL(H3D_ERR_UNSUPPORTED, "nms_topk_large: rows of %d pixels (a band of rows + two halo rows must fit %ld pixels)", W, NMS_MAXHW);
    if ((long)nb * K > 8192) H3D_FAIL(H3D_ERR_UNSUPPORTED, "nms_topk_large: %d bands x K=%d candidates per map (max 8192)", nb, K);
    if (!workspace || workspace_bytes < h3d_nms_topk_large_workspace_bytes(B, C, H, W, K))
        H3D_FAIL(H3D_ERR_ARG, "nms_topk_large: workspace of %zu bytes, %zu needed", workspace_bytes, h3d_nms_topk_large_workspace_bytes(B, C, H, W, K));
    const size_t n = (size_t)B * C * nb * K;
    char *ws = (char *)workspace;
    int64_t *t_ind = (int64_t *)ws;         ws += n * 8;        // (8-byte items first: alignment)
    float *t_score = (float *)ws;           ws += n * 4;
    float *t_y = (float *)ws;               ws += n * 4;
    float *t_x = (float *)ws;               ws += n * 4;
    int32_t *t_band = (int32_t *)ws;
    const NmsJob j0 = {heat, t_score, t_ind, t_y, t_x, B * C}, j1 = {nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    int rc = nms_topk_launch(j0, j1, H, W, K, flags, stream, br, nb);
    if (rc != H3D_OK) return rc;
    // the bands of a map are "classes" of the merge: positions are band-major, i.e. ascending flat index among equal scores
    return h3d_topk_merge(t_score, t_ind, t_y, t_x, B * C, nb, K, scores, inds, t_band, ys, xs, stream);
}

// ------------------------------------------------------------------------------------------------
// stand-alone _nms (decode.py:6-13) and _sigmoid (utils.py:8-10) for API completeness
__global__ void nms_kernel(const float *__restrict__ heat, int H, int W, float *__restrict__ out, size_t total)
{
    const int HW = H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t mapi = i / HW;
        const int p = (int)(i - mapi * HW);
        const int y = p / W, x = p - y * W;
        const float *map = heat + mapi * HW;
        const float v = map[p];
        float m = v;
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= W) continue;
                m = fmaxf(m, map[yy * W + xx]);
            }
        }
        out[i] = (m == v) ? v : v * 0.0f;
    }
}

extern "C" int h3d_nms(const float *heat, int B, int C, int H, int W, float *out, void *stream)
{
    if (!heat || !out) H3D_FAIL(H3D_ERR_ARG, "nms: null pointer");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) H3D_FAIL(H3D_ERR_SHAPE, "nms: bad shape");
    const size_t total = (size_t)B * C * H * W;
    const int grid = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(nms_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, heat, H, W, out, total);
    H3D_CHECK_LAUNCH("nms_kernel");
    return H3D_OK;
}

__global__ void sigmoid_clamp_kernel(const float *__restrict__ in, float *__restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = sigmoid_clamp(in[i]);
}

extern "C" int h3d_sigmoid_clamp(const float *in, float *out, size_t n, void *stream)
{
    if (!in || !out) H3D_FAIL(H3D_ERR_ARG, "sigmoid_clamp: null pointer");
    if (n == 0) return H3D_OK;
    const int grid = (int)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256);
    hipLaunchKernelGGL(sigmoid_clamp_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, out, n);
    H3D_CHECK_LAUNCH("sigmoid_clamp_kernel");
    return H3D_OK;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void topk_merge_kernel(const float *__restrict__ scores, const int64_t *__restrict__ inds,
                                                         const float *__restrict__ ys, const float *__restrict__ xs, int C,
                                                         int K, int N, float *__restrict__ o_score, int64_t *__restrict__ o_ind,
                                                         int32_t *__restrict__ o_cls, float *__restrict__ o_y,
                                                         float *__restrict__ o_x)
{
    extern __shared__ __attribute__((aligned(16))) u64 s_m[];
    const int b = blockIdx.x, tid = threadIdx.x, CK = C * K;
    const size_t base = (size_t)b * CK;
    for (int i = tid; i < N; i += 256)
        s_m[i] = (i < CK) ? (((u64)fkey(scores[base + i] + 0.0f) << 32) | (u64)(0xffffffffu - (uint32_t)i)) : 0ull;
    __syncthreads();
    bitonic_desc(s_m, N, tid, 256);
    for (int j = tid; j < K; j += 256) {
        const u64 c = s_m[j];
        const uint32_t pos = 0xffffffffu - (uint32_t)(c & 0xffffffffull);
        const size_t o = (size_t)b * K + j;
        o_score[o] = scores[base + pos];
        o_cls[o] = (int32_t)(pos / (uint32_t)K);
        o_ind[o] = inds[base + pos];
        o_y[o] = ys[base + pos];
        o_x[o] = xs[base + pos];
    }
}

extern "C" int h3d_topk_merge(const float *scores, const int64_t *inds, const float *ys, const float *xs, int B, int C,
                              int K, float *o_score, int64_t *o_ind, int32_t *o_cls, float *o_y, float *o_x, void *stream)
{
    if (!scores || !inds || !ys || !xs || !o_score || !o_ind || !o_cls || !o_y || !o_x)
        H3D_FAIL(H3D_ERR_ARG, "topk_merge: null pointer");
    if (B <= 0 || C <= 0 || K <= 0) H3D_FAIL(H3D_ERR_SHAPE, "topk_merge: bad shape");
    if ((long)C * K > 8192) H3D_FAIL(H3D_ERR_UNSUPPORTED, "topk_merge: C*K=%ld > 8192", (long)C * K);
    int N = 1;
    while (N < C * K) N <<= 1;
    hipLaunchKernelGGL(topk_merge_kernel, dim3(B), dim3(256), (size_t)N * 8, (hipStream_t)stream, scores, inds, ys, xs, C, K,
                       N, o_score, o_ind, o_cls, o_y, o_x);
    H3D_CHECK_LAUNCH("topk_merge_kernel");
    return H3D_OK;
}

// ------------------------------------------------------------------------------------------------
__global__ void gather_feat_kernel(const float *__restrict__ feat, const int64_t *__restrict__ ind, int C, int HW, int N,
                                   long sb, long sc, long sp, float *__restrict__ out, size_t total)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % C;
        const size_t bn = i / C;
        const size_t b = bn / N;
        const int64_t p = ind[bn];
        out[i] = (p >= 0 && p < HW) ? feat[b * sb + c * sc + p * sp] : 0.f;
    }
}

extern "C" int h3d_gather_feat(const float *feat, const int64_t *ind, int B, int C, int HW, int N, int channels_last,
                               float *out, void *stream)
{
    if (!feat || !ind || !out) H3D_FAIL(H3D_ERR_ARG, "gather_feat: null pointer");
    if (B <= 0 || C <= 0 || HW <= 0 || N <= 0) H3D_FAIL(H3D_ERR_SHAPE, "gather_feat: bad shape");
    const size_t total = (size_t)B * N * C;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    const long sb = (long)C * HW, sc = channels_last ? 1 : HW, sp = channels_last ? C : 1;
    hipLaunchKernelGGL(gather_feat_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, feat, ind, C, HW, N, sb, sc, sp,
                       out, total);
    H3D_CHECK_LAUNCH("gather_feat_kernel");
    return H3D_OK;
}

// ------------------------------------------------------------------------------------------------
// One workgroup per (image, joint) plus one per image for the box / score / class columns, so the
// B*(J+1) independent searches run side by side instead of J rounds inside B workgroups.
// LDS: this joint's regressed keypoints [K][2], boxes [K][4], joint candidates 3x[K].
constexpr int PA_THREADS = 128;
__global__ __launch_bounds__(PA_THREADS) void pose_assemble_kernel(
    const float *__restrict__ c_score, const int64_t *__restrict__ c_ind, const int32_t *__restrict__ c_cls,
    const float *__restrict__ c_y, const float *__restrict__ c_x, const float *__restrict__ hp_score,
    const int64_t *__restrict__ hp_ind, const float *__restrict__ hp_y, const float *__restrict__ hp_x,
    const float *__restrict__ wh, const float *__restrict__ hps, const float *__restrict__ reg,
    const float *__restrict__ hp_offset, int J, int HW, int K, float *__restrict__ dets)
{
    extern __shared__ __attribute__((aligned(16))) float s_f[];
    float *s_kp = s_f;               // [K][2]
    float *s_box = s_kp + K * 2;     // [K][4]
    float *s_hx = s_box + K * 4;     // [K]
    float *s_hy = s_hx + K;
    float *s_hs = s_hy + K;
    const int b = blockIdx.x / (J + 1), j = blockIdx.x - b * (J + 1), tid = threadIdx.x;
    const int D = 5 + 2 * J + 1;
    const float thresh = 0.1f;

    for (int k = tid; k < K; k += PA_THREADS) {
        const size_t o = (size_t)b * K + k;
        const int64_t ind = c_ind[o];
        const float xs = c_x[o], ys = c_y[o];
        float cx, cy;
        if (reg) { cx = xs + reg[((size_t)b * 2) * HW + ind]; cy = ys + reg[((size_t)b * 2 + 1) * HW + ind]; }
        else { cx = xs + 0.5f; cy = ys + 0.5f; }
        const float w = wh[((size_t)b * 2) * HW + ind], hgt = wh[((size_t)b * 2 + 1) * HW + ind];
        const float l = cx - w / 2.f, t = cy - hgt / 2.f, r = cx + w / 2.f, bt = cy + hgt / 2.f;
        if (j == J) {                 // the per-image workgroup: bbox, score, class
            float *d = dets + o * D;
            d[0] = l; d[1] = t; d[2] = r; d[3] = bt;
            d[4] = c_score[o];
            d[5 + 2 * J] = (float)c_cls[o];
        } else {
            s_box[k * 4] = l; s_box[k * 4 + 1] = t; s_box[k * 4 + 2] = r; s_box[k * 4 + 3] = bt;
            const float kx = hps[((size_t)b * 2 * J + 2 * j) * HW + ind] + xs;
            const float ky = hps[((size_t)b * 2 * J + 2 * j + 1) * HW + ind] + ys;
            s_kp[2 * k] = kx; s_kp[2 * k + 1] = ky;
            if (!hp_score) { dets[o * D + 5 + 2 * j] = kx; dets[o * D + 5 + 2 * j + 1] = ky; }
        }
    }
    if (j == J || !hp_score) return;
    for (int c = tid; c < K; c += PA_THREADS) {
        const size_t o = ((size_t)b * J + j) * K + c;
        float hs = hp_score[o], hx = hp_x[o], hy = hp_y[o];
        if (hp_offset) {
            const int64_t ind = hp_ind[o];
            hx = hx + hp_offset[((size_t)b * 2) * HW + ind];
            hy = hy + hp_offset[((size_t)b * 2 + 1) * HW + ind];
        } else { hx = hx + 0.5f; hy = hy + 0.5f; }
        const bool m = hs > thresh;
        s_hs[c] = m ? hs : -1.0f;
        s_hx[c] = m ? hx : -10000.0f;
        s_hy[c] = m ? hy : -10000.0f;
    }
    __syncthreads();
    for (int k = tid; k < K; k += PA_THREADS) {
        const float rx = s_kp[2 * k], ry = s_kp[2 * k + 1];
        float best = 0.f;
        int bi = -1;
        for (int c = 0; c < K; ++c) {
            const float dx = rx - s_hx[c], dy = ry - s_hy[c];
            const float d = __fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
            if (bi < 0 || d < best) { best = d; bi = c; }
        }
        const float sx = s_hx[bi], sy = s_hy[bi], ss = s_hs[bi];
        const float l = s_box[k * 4], t = s_box[k * 4 + 1], r = s_box[k * 4 + 2], bt = s_box[k * 4 + 3];
        const bool bad = (sx < l) || (sx > r) || (sy < t) || (sy > bt) || (ss < thresh) ||
                         (best > __fmul_rn(fmaxf(bt - t, r - l), 0.3f));
        float *d = dets + ((size_t)b * K + k) * D + 5 + 2 * j;
        d[0] = bad ? rx : sx;
        d[1] = bad ? ry : sy;
    }
}

extern "C" int h3d_multi_pose_assemble(const float *c_score, const int64_t *c_ind, const int32_t *c_cls, const float *c_y,
                                       const float *c_x, const float *hp_score, const int64_t *hp_ind, const float *hp_y,
                                       const float *hp_x, const float *wh, const float *hps, const float *reg,
                                       const float *hp_offset, int B, int J, int H, int W, int K, float *dets, void *stream)
{
    if (!c_score || !c_ind || !c_cls || !c_y || !c_x || !wh || !hps || !dets)
        H3D_FAIL(H3D_ERR_ARG, "multi_pose_assemble: null pointer");
    if (hp_score && (!hp_ind || !hp_y || !hp_x)) H3D_FAIL(H3D_ERR_ARG, "multi_pose_assemble: partial joint top-k");
    if (B <= 0 || J <= 0 || H <= 0 || W <= 0 || K <= 0) H3D_FAIL(H3D_ERR_SHAPE, "multi_pose_assemble: bad shape");
    const size_t lds = (size_t)K * 9 * sizeof(float);
    if (lds > 64 * 1024) H3D_FAIL(H3D_ERR_UNSUPPORTED, "multi_pose_assemble: K too large (%d)", K);
    hipLaunchKernelGGL(pose_assemble_kernel, dim3(B * (J + 1)), dim3(PA_THREADS), lds, (hipStream_t)stream, c_score, c_ind, c_cls, c_y, c_x,
                       hp_score, hp_ind, hp_y, hp_x, wh, hps, reg, hp_offset, J, H * W, K, dets);
    H3D_CHECK_LAUNCH("pose_assemble_kernel");
    return H3D_OK;
}

// ------------------------------------------------------------------------------------------------
__global__ void ctdet_assemble_kernel(const float *__restrict__ c_score, const int64_t *__restrict__ c_ind,
                                      const int32_t *__restrict__ c_cls, const float *__restrict__ c_y,
                                      const float *__restrict__ c_x, const float *__restrict__ wh,
                                      const float *__restrict__ reg, int C, int HW, int K, int cat_spec_wh, int total,
                                      float *__restrict__ dets)
{
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total) return;
    const int b = o / K;
    const int64_t ind = c_ind[o];
    const float xs = c_x[o], ys = c_y[o];
    float cx, cy;
    if (reg) { cx = xs + reg[((size_t)b * 2) * HW + ind]; cy = ys + reg[((size_t)b * 2 + 1) * HW + ind]; }
    else { cx = xs + 0.5f; cy = ys + 0.5f; }
    const int cls = c_cls[o];
    const int whc = cat_spec_wh ? 2 * C : 2;
    const int w0 = cat_spec_wh ? 2 * cls : 0;
    const float w = wh[((size_t)b * whc + w0) * HW + ind], hgt = wh[((size_t)b * whc + w0 + 1) * HW + ind];
    float *d = dets + (size_t)o * 6;
    d[0] = cx - w / 2.f; d[1] = cy - hgt / 2.f; d[2] = cx + w / 2.f; d[3] = cy + hgt / 2.f;
    d[4] = c_score[o];
    d[5] = (float)cls;
}

extern "C" int h3d_ctdet_assemble(const float *c_score, const int64_t *c_ind, const int32_t *c_cls, const float *c_y,
                                  const float *c_x, const float *wh, const float *reg, int B, int C, int H, int W, int K,
                                  int cat_spec_wh, float *dets, void *stream)
{
    if (!c_score || !c_ind || !c_cls || !c_y || !c_x || !wh || !dets) H3D_FAIL(H3D_ERR_ARG, "ctdet_assemble: null pointer");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0) H3D_FAIL(H3D_ERR_SHAPE, "ctdet_assemble: bad shape");
    const int total = B * K;
    hipLaunchKernelGGL(ctdet_assemble_kernel, dim3(cdiv(total, 128)), dim3(128), 0, (hipStream_t)stream, c_score, c_ind,
                       c_cls, c_y, c_x, wh, reg, C, H * W, K, cat_spec_wh, total, dets);
    H3D_CHECK_LAUNCH("ctdet_assemble_kernel");
    return H3D_OK;
}

// ------------------------------------------------------------------------------------------------
// Inverse crop affine with rot = 0 (utils/image.py:27-68 with inv=1): the three-point transform
// reduces to  p_img = c + (p_out - [w/2, h/2]) * (s / w_out)  (s = scale[0] = the source width).
__global__ void post_process_kernel(const float *__restrict__ dets, const float *__restrict__ c, const float *__restrict__ s,
                                    int K, int J, int out_h, int out_w, int total, float *__restrict__ out)
{
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total) return;
    const int b = o / K;
    const int D = 5 + 2 * J + 1, DO = 5 + 2 * J;
    const float *d = dets + (size_t)o * D;
    float *q = out + (size_t)o * DO;
    const double sc = (double)s[b] / (double)out_w;
    const double cx = c[b * 2], cy = c[b * 2 + 1];
    const double hx = 0.5 * out_w, hy = 0.5 * out_h;
    for (int i = 0; i < 2; ++i) {
        q[2 * i] = (float)(cx + ((double)d[2 * i] - hx) * sc);
        q[2 * i + 1] = (float)(cy + ((double)d[2 * i + 1] - hy) * sc);
    }
    q[4] = d[4];
    for (int j = 0; j < J; ++j) {
        q[5 + 2 * j] = (float)(cx + ((double)d[5 + 2 * j] - hx) * sc);
        q[5 + 2 * j + 1] = (float)(cy + ((double)d[5 + 2 * j + 1] - hy) * sc);
    }
}

extern "C" int h3d_multi_pose_post_process(const float *dets, const float *c, const float *s, int B, int K, int J,
                                           int out_h, int out_w, float *out, void *stream)
{
    if (!dets || !c || !s || !out) H3D_FAIL(H3D_ERR_ARG, "post_process: null pointer");
    if (B <= 0 || K <= 0 || J < 0 || out_h <= 0 || out_w <= 0) H3D_FAIL(H3D_ERR_SHAPE, "post_process: bad shape");
    const int total = B * K;
    hipLaunchKernelGGL(post_process_kernel, dim3(cdiv(total, 128)), dim3(128), 0, (hipStream_t)stream, dets, c, s, K, J,
                       out_h, out_w, total, out);
    H3D_CHECK_LAUNCH("post_process_kernel");
    return H3D_OK;
}
