// DeformConv with its offset/mask convolution fused in, for fp16 plans (H3D_F16): every operand arrives by LDS-DMA.
// (reference model.py:346-362 DeformConv -> dcn_v2.py:118-128 DCN.forward: conv_offset_mask -> chunk/cat/sigmoid ->
//  dcn_v2_conv -> BN -> ReLU; same math, tile shape, geometry, patch slots and pass 2 as csrc/dcn3.hip, whose header
//  describes them.)
//
// What is different from dcn3: in a bf16 plan the apron must be converted to fp16 on its way into LDS, so it goes through
// registers (two global loads, eight conversions, two ds_write_b128 per thread and 16-channel stage, a single apron
// buffer, TWO barriers per stage: in-kernel stamps put 16 % of a phase-B stage into that store and 27 % into the wait at the
// first barrier).  In an fp16 plan the input already is the sample type:
//   * the apron of a stage is an LDS image that `buffer_load ... lds` writes directly (no staging registers, no VALU, no
//     ds_write): 32 B per pixel, no pad bytes; the 16-byte half a lane-half reads is selected by the ROW parity
//     (half' = half ^ (row & 1)) through the per-lane SOURCE address, which keeps every 16-lane ds_read_b128 group of the
//     offset convolution on 16 distinct bank slots (rows are an even number of 16-byte slots: 44 / 52) at 15.1 KB per
//     stage instead of 28 KB;
//   * so the apron is double buffered like the filters (a ring of two stage slots each), the DMA of stage s+1 is issued
//     right after the barrier that ends stage s-1, and phase A runs ONE barrier per stage; phase B keeps a second, short
//     one only for the patch pixels (blend of four global corners + one ds_write per thread) and only in tiles that have
//     any;
//   * corner (y+1, x) of a sample at LDS byte offset o is (o ^ 16) + ROWB, corner (y, x+1) is o + 32.
// Pieces of 1 KiB; the zeros a piece writes past the end of its image land in the slack of its own slot.
#include "common.h"
#include "epilogue.h"
#include "dcn_traits.h"
#include <type_traits>

struct Dcn5Args {
    const char *in;
    const char *w;      // main filter images (H3D_OP_DCN_FUSED_STREAM: stage-major, CK = 16)
    const char *woff;   // offset/mask filter images
    const float *bias;  // [rows] main bias followed by [32] permuted offset bias
    char *out;
    int B, H, W, Cin, in_cs;
    int Cout, out_cs, relu, out_mode, wrows;
    int tiles_x, tiles_y;
    int dbg;
    int G;     // 32-row groups of the main filter image
    int xcd;   // h3d_tile_id mode
};

template <int MT, int MARGIN, int NP>
struct Dcn5Cfg {
    static constexpr int CK = 16, SS = 2;
    static constexpr int HH = 16 + 2 + 2 * MARGIN;
    static constexpr int PXB = CK * SS;                         // 32 B per apron pixel, no pad
    static constexpr int ROWB = HH * PXB;                       // 704 (margin 2) / 832 (margin 4)
    static constexpr int ASZ = HH * ROWB;
    static constexpr int APIECES = (ASZ + 1023) / 1024;
    static constexpr int ASLOT = APIECES * 1024;
    static constexpr int WB = 9 * CK * SS + 16;                 // 304 B filter rows (stage-major image of engine.PackedWeights.dcn_stream)
    static constexpr int WGRP = 32 * WB;
    static constexpr int WPIECES = (MT * WGRP + 1023) / 1024;
    static constexpr int OPIECES = (WGRP + 1023) / 1024;
    static constexpr int WSLOT = WPIECES * 1024;
    static constexpr int PSLOT = CK * SS;                       // one patch pixel
    static constexpr int PB = (NP * PSLOT + 255) / 256 * 256;   // patch area, in FRONT of the apron slots
    static constexpr int OFF_A = PB, OFF_F = PB + 2 * ASLOT;
    static constexpr int LDS_MAIN = OFF_F + 2 * WSLOT;
    static constexpr int LDS_EPI = 8 * ((32 * (64 * MT + 16) + 1023) / 1024 * 1024);
    static constexpr int LDS = (LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI) + 32;      // + the eight per-wave sample counts
    static constexpr int THREADS = 512;
    static constexpr int AP = (APIECES + 7) / 8;                // apron pieces per wave
    static_assert((ROWB / 16) % 2 == 0 && (ROWB & 16) == 0, "row-parity swizzle: rows of an even number of 16-byte slots");
    static_assert(NP * 16 <= PB, "the sample list is kept in the (not yet used) patch area");
    static_assert(NP * 2 <= 512, "one 16-byte patch unit per thread");
};

typedef __attribute__((address_space(3))) void lds_void5;

// filters of one stage: PIECES KiB pieces, linear copy (lane offsets l * 16), piece p by wave p % 8
template <int PIECES>
__device__ __forceinline__ void dcn5_issue_w(const char *base, int bytes, char *dst, int src, int lane16, int wv)
{
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < (PIECES + 7) / 8; ++j) {
        const int p = wv + 8 * j;
        if (p < PIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void5 *)(dst + p * 1024), 16, lane16, src + p * 1024, 0, 0);
    }
}
// apron of one stage: piece p by wave p % 8, per-lane source offsets `voff` (swizzled half, out of range outside the image)
template <int APIECES>
__device__ __forceinline__ void dcn5_issue_a(const char *img, int bytes, char *dst, const int *voff, int soff, int wv)
{
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)img, 0, bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < (APIECES + 7) / 8; ++j) {
        const int p = wv + 8 * j;
        if (p < APIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void5 *)(dst + p * 1024), 16, voff[j], soff, 0, 0);
    }
}
__device__ __forceinline__ u32x4 dcn5_corner(const char *img, int bytes, int voff, int soff)
{
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)img, 0, bytes, 0x00020000);
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}

// XP: experiment bits (h3d_op.reserved >> 16; tools/ab_dcn5.py).  1: next stage's DMA issued after the second barrier; 2 / 4 / 8
// (timing only, wrong results): no patch pixels / no apron DMA after stage 0 / no filter DMA after stage 0; 16: tap-ahead gathers
template <int MT, int MARGIN, int EPI, int NP, int XP = 0>      // EPI: 0 general, 1 lean NHWC, 2 LDS-transposed
__global__ __launch_bounds__(512, MT <= 2 ? 4 : 2) void dcn5_kernel(Dcn5Args a)
{
    using C = Dcn5Cfg<MT, MARGIN, NP>;
    using T = f16_t;
    using X = SE<f16_t>;
    constexpr int ES = 2, SS = 2, CK = 16;
#ifndef DCN5_TAPAHEAD
#define DCN5_TAPAHEAD 4      // tap-ahead gathers from this many 32-channel output tiles per workgroup
#endif
    constexpr bool TAPAHEAD = MT >= DCN5_TAPAHEAD || (XP & 16);
    __shared__ __attribute__((aligned(1024))) char smem[C::LDS];
    int *s_cnt = reinterpret_cast<int *>(smem + C::LDS - 32);

    const int tid = threadIdx.x;
    const int l = tid & 63, r = l & 31, h = l >> 5;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = a.tiles_x * a.tiles_y;
    const int bid = h3d_tile_id(blockIdx.x, gridDim.x, a.xcd);
    const int b = bid / tiles;
    const int t = bid - b * tiles;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int oy0 = ty * 16, ox0 = tx * 16;
    const int hy0 = oy0 - 1 - MARGIN, hx0 = ox0 - 1 - MARGIN;
    const int cout0 = blockIdx.y * 32 * MT;
    const int py = wvu * 2 + (r >> 4), px = r & 15;          // this lane's pixel inside the tile
    const int oy = oy0 + py, ox = ox0 + px;
    const bool live = (oy < a.H && ox < a.W);
    const char *img = a.in + (size_t)b * a.H * a.W * a.in_cs * ES;
    const int img_bytes = (int)((size_t)a.H * a.W * a.in_cs * ES);
    const int aoff = r * C::WB + 8 * h * SS;
    const int nchunks = a.Cin / CK;
    const int off_bytes = nchunks * C::WGRP, main_bytes = nchunks * a.G * C::WGRP;

    // ---- DMA sources of my apron pieces: LDS slot i of piece p is 16-byte unit p * 64 + i of the image [HH rows][HH px][2
    //      halves]; stored half t holds source half t ^ (row & 1) --------------------------------------------------------
    int avoff[C::AP];
#pragma unroll
    for (int j = 0; j < C::AP; ++j) {
        const int u = (wvu + 8 * j) * 64 + l;
        const int y = u / (C::ROWB / 16), tt = u - y * (C::ROWB / 16);
        const int p = tt >> 1, hs = (tt & 1) ^ (y & 1);
        const int gy = hy0 + y, gx = hx0 + p;
        avoff[j] = (y < C::HH && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? ((gy * a.W + gx) * a.in_cs) * ES + hs * 16 : 0x7ffffff0;
    }
    // stage s: apron chunk s % nchunks -> A[s & 1]; filters (offset conv for s < nchunks, else main) -> F[s & 1]
    auto issue = [&](int s) {
        const int c = s < nchunks ? s : s - nchunks;
        if (!((XP & 4) && s > 0)) dcn5_issue_a<C::APIECES>(img, img_bytes, smem + C::OFF_A + (s & 1) * C::ASLOT, avoff, c * CK * ES, wvu);
        char *dst = smem + C::OFF_F + (s & 1) * C::WSLOT;
        if ((XP & 8) && s > 0) return;
        if (s < nchunks) dcn5_issue_w<C::OPIECES>(a.woff, off_bytes, dst, s * C::WGRP, l * 16, wvu);
        else dcn5_issue_w<C::WPIECES>(a.w, main_bytes, dst, (c * a.G + (int)blockIdx.y * MT) * C::WGRP, l * 16, wvu);
    };

    issue(0);
    if constexpr (NP > 0) {
        // every byte a zero-weighted corner read of a patched sample can touch must hold a finite number: the patch area is
        // cleared once (the apron slots are written whole by the DMA, the filter slots only ever hold finite fp16)
        for (int i = tid * 16; i < C::PB; i += C::THREADS * 16) *reinterpret_cast<u32x4 *>(smem + i) = u32x4{0u, 0u, 0u, 0u};
    }

    // ================= phase A: offsets/mask = conv3x3(x; 27 filters), one barrier per stage ========================
    f32x16 aoffs;
#pragma unroll
    for (int i = 0; i < 16; ++i) aoffs[i] = 0.f;
    // tap (0,0) of the plain conv: row MARGIN + py, pixel MARGIN + px; rows dy = 0, 2 share the half, dy = 1 has the other
    const int bc0 = (MARGIN + py) * C::ROWB + (MARGIN + px) * C::PXB + ((h ^ ((MARGIN + py) & 1)) << 4);
    const int bc1 = (bc0 ^ 16) + C::ROWB;
    auto computeA = [&](int s) {
        const char *s_h = smem + C::OFF_A + (s & 1) * C::ASLOT;
        const char *s_w = smem + C::OFF_F + (s & 1) * C::WSLOT;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            const typename X::frag fa = X::lds(s_w + aoff + tap * CK * SS);
            const typename X::frag fb = X::lds(s_h + (dy == 1 ? bc1 : bc0 + dy * C::ROWB) + dx * C::PXB);
            X::mma(aoffs, fa, fb);
        }
    };
    for (int s = 0; s < nchunks; ++s) {
        __builtin_amdgcn_s_waitcnt(0x0f70);      // my pieces of stage s have landed (nothing younger is in flight)
        h3d_barrier_keep_vmcnt();                // ... everyone's have, and nobody reads the slots of stage s - 1 any more
        issue(s + 1);                            // s + 1 == nchunks is phase B's first stage
        computeA(s);
    }
    {   // + bias (permuted like the rows)
        const float *bo = a.bias + a.wrows;
#pragma unroll
        for (int i = 0; i < 16; ++i) aoffs[i] += bo[(i & 3) + 8 * (i >> 2) + 4 * h];
    }

    // ================= geometry: my taps (h=0: 0..4, h=1: 5..8), then cross-half exchange ===========
    // boff: LDS byte offset of corner (hl, wl)'s 16 bytes for THIS lane half, relative to the stage's apron slot for apron
    // samples; patched samples (bit in pmask) hold the offset of their patch pixel relative to smem
    int boff[9];
    typename X::geo geo[9];
    bool slow = false;
    [[maybe_unused]] int pmask = 0;
    {
        int my_off[5];
        typename X::geo my_geo[5];
        bool my_want[5];
        uint32_t my_hw[5];
        int my_pm = 0;
        const int tb = h ? 5 : 0;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int tap = tb + u;
            const int ti = tap / 3, tj = tap - ti * 3;
            const float h_im = (float)(oy - 1 + ti) + aoffs[3 * u];
            const float w_im = (float)(ox - 1 + tj) + aoffs[3 * u + 1];
            const bool inside = live && tap < 9 && (h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W);
            typename X::geo g = X::zero_geo();
            int off = 0;
            bool want = false;               // inside the image, corners outside the apron
            int hl = 0, wl = 0;
            if (inside) {
                hl = (int)floorf(h_im);
                wl = (int)floorf(w_im);
                const int ry = hl - hy0, rx = wl - hx0;
                const float lh = h_im - (float)hl, lw = w_im - (float)wl;
                const float hh = 1.f - lh, hw = 1.f - lw;
                const float w4[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
                g = X::make_geo(w4, dcn2_sigmoid(aoffs[3 * u + 2]));
                if (ry >= 0 && ry + 1 < C::HH && rx >= 0 && rx + 1 < C::HH) {
                    off = ry * C::ROWB + rx * C::PXB + ((ry & 1) << 4);      // half 0's bytes; half 1 reads off ^ 16
                } else {
                    want = true;
                }
            }
            my_want[u] = want;
            my_hw[u] = ((uint32_t)hl << 16) | ((uint32_t)wl & 0xffffu);
            my_off[u] = off;
            my_geo[u] = g;
        }
        if constexpr (NP > 0) {
            // patch slots in a deterministic order (wave, tap, lane): see csrc/dcn3.hip
            unsigned long long m[5];
            int cnt = 0;
#pragma unroll
            for (int u = 0; u < 5; ++u) { m[u] = __ballot(my_want[u]); cnt += __popcll(m[u]); }
            if (l == 0) s_cnt[wvu] = cnt;
            __syncthreads();                                         // the counts (and the cleared patch area) are visible
            int base = 0;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) base += (w8 < wvu) ? s_cnt[w8] : 0;
            if (cnt) {                                               // wave-uniform
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int slot = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m[u] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[u], 0u));
                    base += __popcll(m[u]);
                    if (my_want[u] && slot < NP) {
                        // the sample list lives in the patch area until the first patch is written (16 B per entry)
                        *reinterpret_cast<u32x4 *>(smem + slot * 16) = u32x4{my_hw[u], my_geo[u].w01, my_geo[u].w23, 0u};
                        my_geo[u].w01 = 0x00003c00u;                 // (1, 0 | 0, 0): the blend is done when the patch is filled
                        my_geo[u].w23 = 0u;
                        my_off[u] = slot * C::PSLOT;                 // the patch pixel (relative to smem), half 0's bytes
                        my_pm |= 1 << u;
                        my_want[u] = false;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 5; ++u)
            if (my_want[u]) { slow = true; my_geo[u] = X::zero_geo(); }
        if constexpr (NP > 0) {
            const int o_pm = __shfl_xor(my_pm, 32);
            pmask = h == 0 ? (my_pm | (o_pm << 5)) : (o_pm | (my_pm << 5));
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int o_off = __shfl_xor(my_off[u], 32);
            const typename X::geo o_geo = X::shfl_xor32(my_geo[u]);
            // tap u (u < 5) belongs to half 0, tap 5 + u (u < 4) to half 1; lane half h reads its 16 bytes at (offset ^ 16 h)
            boff[u] = (h == 0 ? my_off[u] : o_off) ^ (h << 4);
            geo[u] = (h == 0) ? my_geo[u] : o_geo;
            if (u < 4) {
                boff[5 + u] = (h == 1 ? my_off[u] : o_off) ^ (h << 4);
                geo[5 + u] = (h == 1) ? my_geo[u] : o_geo;
            }
        }
    }

    // ---- patches: thread `tid` owns the 16-byte unit tid & 1 of list entry tid >> 1 for the whole of phase B ----------
    [[maybe_unused]] int pbase = 0, pok = 0;
    [[maybe_unused]] typename X::geo pgeo = X::zero_geo();
    [[maybe_unused]] bool phas = false, overflow = false, anyp = false;
    if constexpr (NP > 0) {
        __syncthreads();                                                 // the list is complete
        int nwant = 0;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) nwant += s_cnt[w8];
        overflow = nwant > NP;                                           // workgroup-uniform: some sample found no slot
        anyp = nwant > 0;                                                // workgroup-uniform: the tile has patch pixels
        const int nsl = min(nwant, NP);
        const int ps = tid >> 1, pv = tid & 1;
        phas = ps < nsl;
        if (phas) {
            const u32x4 d = *reinterpret_cast<const u32x4 *>(smem + ps * 16);
            const int hl = (int)d[0] >> 16, wl = (int)(short)(d[0] & 0xffffu);
            pgeo.w01 = d[1]; pgeo.w23 = d[2];
            pbase = ((hl * a.W + wl) * a.in_cs) * ES + pv * 16;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int yy = hl + (k >> 1), xx = wl + (k & 1);
                if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) pok |= 1 << k;
            }
        }
        __syncthreads();                                                 // everyone has read its entry: the area becomes patch pixels
        if (anyp)                                                        // (the list's bit patterns must not be read as fp16 values)
            for (int i = tid * 16; i < nsl * 16; i += C::THREADS * 16) *reinterpret_cast<u32x4 *>(smem + i) = u32x4{0u, 0u, 0u, 0u};
    }
    [[maybe_unused]] u32x4 pst[4];
    auto patch_issue = [&](int s) {
        const int c0 = (s - nchunks) * CK;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int voff = pbase + ((k & 1) + (k >> 1) * a.W) * a.in_cs * ES;
            pst[k] = dcn5_corner(img, img_bytes, ((pok >> k) & 1) ? voff : 0x7ffffff0, c0 * ES);
        }
    };
    auto patch_commit = [&]() {
        if (!phas) return;
        typename X::frag v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k].v = __builtin_bit_cast(half8_t, pst[k]);
        const typename X::frag o = X::blend(v, pgeo);
        *reinterpret_cast<half8_t *>(smem + tid * 16) = o.v;                 // entry tid >> 1, unit tid & 1
    };

    auto tail = [&](auto P2) {
    // ================= phase B: deformable contraction (branch-free) ==================================================
    f32x16 acc[MT][1];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][0][i] = 0.f;
    auto computeB = [&](int s) {
        const int abase = C::OFF_A + (s & 1) * C::ASLOT;
        const char *s_w = smem + C::OFF_F + (s & 1) * C::WSLOT;
        // apron samples are relative to this stage's slot, patch pixels to smem
        auto gather = [&](int tap, typename X::frag (&v)[4]) {
            const int o0 = boff[tap] + (NP > 0 && ((pmask >> tap) & 1) ? 0 : abase);
            const char *p00 = smem + o0;
            const char *p10 = smem + ((o0 ^ 16) + C::ROWB);
            v[0] = X::lds(p00);
            v[1] = X::lds(p00 + C::PXB);
            v[2] = X::lds(p10);
            v[3] = X::lds(p10 + C::PXB);
        };
        if constexpr (TAPAHEAD) {
            // one workgroup per CU = two waves per SIMD: tap t+1's four corner fragments are requested before tap t is blended
            // and multiplied (16 more live registers)
            typename X::frag v[2][4];
            gather(0, v[0]);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                typename X::frag fa[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[m] = X::lds(s_w + aoff + m * 32 * C::WB + tap * CK * SS);
                if (tap + 1 < 9) gather(tap + 1, v[(tap + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);            // (hipcc sinks the reads back in front of their first use otherwise)
                const typename X::frag fb = X::blend(v[tap & 1], geo[tap]);
#pragma unroll
                for (int m = 0; m < MT; ++m) X::mma(acc[m][0], fa[m], fb);
            }
        } else {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                typename X::frag v[4];
                gather(tap, v);
                const typename X::frag fb = X::blend(v, geo[tap]);
                typename X::frag fa[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[m] = X::lds(s_w + aoff + m * 32 * C::WB + tap * CK * SS);
#pragma unroll
                for (int m = 0; m < MT; ++m) X::mma(acc[m][0], fa[m], fb);
            }
        }
    };
    for (int s = nchunks; s < 2 * nchunks; ++s) {
        if (NP > 0 && anyp && !(XP & 2)) {        // workgroup-uniform
            patch_issue(s);
            __builtin_amdgcn_s_waitcnt(0x0f70 | 4);      // vmcnt(4): my DMA pieces of stage s (older than the four corner loads) have landed
            h3d_barrier_keep_vmcnt();             // everyone's have; stage s - 1 is no longer read (its slots and the patch area are free)
            __builtin_amdgcn_s_waitcnt(0x0f70);   // the corners (requested before the barrier wait) have arrived
            patch_commit();
            if (!(XP & 1) && s + 1 < 2 * nchunks) issue(s + 1);
            h3d_barrier_keep_vmcnt();             // the patch pixels are visible (lgkmcnt(0) inside; the DMA stays in flight)
            if ((XP & 1) && s + 1 < 2 * nchunks) issue(s + 1);
        } else {
            __builtin_amdgcn_s_waitcnt(0x0f70);
            h3d_barrier_keep_vmcnt();
            if (s + 1 < 2 * nchunks) issue(s + 1);
        }
        computeB(s);
    }

    // ================= pass 2 (rare): samples that left the apron and found no patch slot ==============================
    bool do_p2 = false;
    if constexpr (decltype(P2)::value) do_p2 = NP > 0 ? overflow : (bool)__syncthreads_or(slow ? 1 : 0);
    if constexpr (decltype(P2)::value) if (do_p2) {
        if constexpr (MT < 4) {
            // under the 128-VGPR cap the phase-A accumulators are not kept alive through phase B for this rare path: the offset
            // convolution is recomputed, unpipelined
#pragma unroll
            for (int i = 0; i < 16; ++i) aoffs[i] = 0.f;
            for (int s = 0; s < nchunks; ++s) {
                __syncthreads();
                issue(s);
                __builtin_amdgcn_s_waitcnt(0x0f70);
                __syncthreads();
                computeA(s);
            }
            const float *bo = a.bias + a.wrows;
#pragma unroll
            for (int i = 0; i < 16; ++i) aoffs[i] += bo[(i & 3) + 8 * (i >> 2) + 4 * h];
        }
        // geometry of the nine taps once; per chunk the corner loads of TB taps are in flight together (range-checked buffer
        // loads: lanes without a pending sample read zeros, no branch); a wave skips the taps none of its lanes needs
        constexpr int TB = 3;
        int qb[9], qok[9], wmask = 0;
        typename X::geo qg[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ti = tap / 3, tj = tap - ti * 3;
            const int src = (tap < 5) ? r : r + 32, u = (tap < 5) ? tap : tap - 5;
            const float d_h = __shfl(aoffs[3 * u], src), d_w = __shfl(aoffs[3 * u + 1], src), d_m = __shfl(aoffs[3 * u + 2], src);
            const float h_im = (float)(oy - 1 + ti) + d_h, w_im = (float)(ox - 1 + tj) + d_w;
            qb[tap] = 0; qok[tap] = 0; qg[tap] = X::zero_geo();
            bool pend = false;
            if (live && h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W) {
                const int hl = (int)floorf(h_im), wl = (int)floorf(w_im);
                const int ry = hl - hy0, rx = wl - hx0;
                if (!(ry >= 0 && ry + 1 < C::HH && rx >= 0 && rx + 1 < C::HH) && !((pmask >> tap) & 1)) {
                    pend = true;
                    const float lh = h_im - (float)hl, lw = w_im - (float)wl;
                    const float hh = 1.f - lh, hw = 1.f - lw;
                    const float w4[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
                    qg[tap] = X::make_geo(w4, dcn2_sigmoid(d_m));
                    qb[tap] = ((hl * a.W + wl) * a.in_cs + 8 * h) * ES;
                    qok[tap] = (hl >= 0 && wl >= 0 ? 1 : 0) | (hl >= 0 && wl + 1 <= a.W - 1 ? 2 : 0) |
                               (hl + 1 <= a.H - 1 && wl >= 0 ? 4 : 0) | (hl + 1 <= a.H - 1 && wl + 1 <= a.W - 1 ? 8 : 0);
                }
            }
            if (__any(pend)) wmask |= 1 << tap;
        }
        const int pxb = a.in_cs * ES, rowb = a.W * pxb;
        char *s_w = smem + C::OFF_F;
        u32x4 pv[TB][4];
        for (int c0 = 0; c0 < a.Cin; c0 += CK) {
            auto fetch = [&](int t0) {
#pragma unroll
                for (int j = 0; j < TB; ++j) {
                    if (!((wmask >> (t0 + j)) & 1)) continue;                // wave-uniform
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        pv[j][k] = dcn5_corner(img, img_bytes, ((qok[t0 + j] >> k) & 1) ? qb[t0 + j] + (k & 1) * pxb + (k >> 1) * rowb : 0x7ffffff0, c0 * ES);
                }
            };
            __syncthreads();
            dcn5_issue_w<C::WPIECES>(a.w, main_bytes, s_w, ((c0 / CK) * a.G + (int)blockIdx.y * MT) * C::WGRP, l * 16, wvu);
            fetch(0);
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
#pragma unroll
            for (int t0 = 0; t0 < 9; t0 += TB) {
                typename X::frag fbs[TB];
#pragma unroll
                for (int j = 0; j < TB; ++j) {
                    if (!((wmask >> (t0 + j)) & 1)) continue;
                    typename X::frag v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k].v = __builtin_bit_cast(half8_t, pv[j][k]);
                    fbs[j] = X::blend(v, qg[t0 + j]);
                }
                if (t0 + TB < 9) fetch(t0 + TB);                               // the next taps fly while these are multiplied
#pragma unroll
                for (int j = 0; j < TB; ++j) {
                    if (!((wmask >> (t0 + j)) & 1)) continue;
                    typename X::frag fa[MT];
#pragma unroll
                    for (int m = 0; m < MT; ++m) fa[m] = X::lds(s_w + aoff + m * 32 * C::WB + ((t0 + j) * CK) * SS);
#pragma unroll
                    for (int m = 0; m < MT; ++m) X::mma(acc[m][0], fa[m], fbs[j]);
                }
            }
        }
    }

    EpiArgs e;
    e.bias = a.bias; e.res = nullptr; e.out = a.out; e.Ho = a.H; e.Wo = a.W; e.Cout = a.Cout;
    e.out_cs = a.out_cs; e.res_cs = 0; e.relu = a.relu; e.out_mode = a.out_mode;
    if constexpr (EPI == 2) {
        __syncthreads();                          // the apron and the filters are no longer read
        tile_epilogue_lds<T, MT>(acc, e, b, oy0, ox0, cout0, wvu, l, smem + wvu * epi_lds_stride<MT>());
    } else {
        tile_epilogue<T, MT, 1, EPI == 1>(acc, e, b, oy0, ox0, cout0, wvu, r, h);
    }
    };
    // (a second copy of phase B without pass 2 only where registers allow: see csrc/dcn3.hip)
    if constexpr (NP > 0 && MT >= 4) {
        if (overflow) tail(std::true_type{});
        else tail(std::false_type{});
    } else {
        tail(std::true_type{});
    }
}

template <int MT, int MARGIN, int NP, int XP = 0>
static int launch_dcn5_cfg(const Dcn5Args &a0, hipStream_t st)
{
    using C = Dcn5Cfg<MT, MARGIN, NP>;
    static_assert(C::LDS * (MT <= 2 ? 2 : 1) <= 160 * 1024, "LDS budget (two workgroups per CU for <= 64 output channels)");
    Dcn5Args a = a0;
    a.tiles_x = cdiv(a.W, 16);
    a.tiles_y = cdiv(a.H, 16);
    a.xcd = h3d_xcd_mode();
    dim3 grid(a.B * a.tiles_x * a.tiles_y, cdiv(a.Cout, 32 * MT));
    const bool lean = a.out_mode == H3D_OUT_NHWC && a.Cout % 4 == 0 && ((uintptr_t)a.bias & 15) == 0;
    const int epi = (MT >= 2 && lean && a.Cout % 8 == 0 && a.out_cs % 8 == 0 && ((uintptr_t)a.out & 15) == 0) ? 2 : lean ? 1 : 0;
    if (h3d_note_kernel(XP ? "dcn5_kernel<%d, %d, %d, %d, %d>" : "dcn5_kernel<%d, %d, %d, %d>", MT, MARGIN, epi, NP, XP)) return H3D_OK;
    if constexpr (MT >= 2) {
        if (epi == 2) {
            hipLaunchKernelGGL((dcn5_kernel<MT, MARGIN, 2, NP, XP>), grid, dim3(C::THREADS), 0, st, a);
            H3D_CHECK_LAUNCH("dcn5_kernel");
            return H3D_OK;
        }
    }
    if constexpr (XP != 0) H3D_FAIL(H3D_ERR_UNSUPPORTED, "dcn5: experiment variants exist for the LDS-transposed epilogue only");
    if (epi == 1) hipLaunchKernelGGL((dcn5_kernel<MT, MARGIN, 1, NP>), grid, dim3(C::THREADS), 0, st, a);
    else hipLaunchKernelGGL((dcn5_kernel<MT, MARGIN, 0, NP>), grid, dim3(C::THREADS), 0, st, a);
    H3D_CHECK_LAUNCH("dcn5_kernel");
    return H3D_OK;
}

// H3D_OP_DCN_FUSED_STREAM of an fp16 plan (called by h3d_launch_dcn3 after its argument checks)
int h3d_launch_dcn5(const h3d_op &op, hipStream_t st)
{
    Dcn5Args a;
    a.in = (const char *)op.in; a.w = (const char *)op.w; a.woff = (const char *)op.in2; a.bias = op.bias;
    a.out = (char *)op.out; a.B = op.B; a.H = op.H; a.W = op.W; a.Cin = op.Cin; a.in_cs = op.in_cs;
    a.Cout = op.Cout; a.out_cs = op.out_cs; a.relu = op.relu; a.out_mode = op.out_mode; a.wrows = op.wrows;
    a.tiles_x = a.tiles_y = 0;
    a.dbg = op.reserved;
    a.G = op.wrows / 32;
    const int xp = (op.reserved >> 16) & 0xff;
    if (xp && op.Cout > 32 && op.Cout <= 64) {
        switch (xp) {
        case 1: return launch_dcn5_cfg<2, 2, 256, 1>(a, st);
        case 2: return launch_dcn5_cfg<2, 2, 256, 2>(a, st);
        case 3: return launch_dcn5_cfg<2, 2, 256, 3>(a, st);
        case 4: return launch_dcn5_cfg<2, 2, 256, 4>(a, st);
        case 8: return launch_dcn5_cfg<2, 2, 256, 8>(a, st);
        case 12: return launch_dcn5_cfg<2, 2, 256, 12>(a, st);
        case 14: return launch_dcn5_cfg<2, 2, 256, 14>(a, st);
        case 17: return launch_dcn5_cfg<2, 2, 256, 17>(a, st);
        default: H3D_FAIL(H3D_ERR_ARG, "dcn5: unknown experiment %d", xp);
        }
    }
    if (op.Cout <= 32) return launch_dcn5_cfg<1, 2, 256>(a, st);
    if (op.Cout <= 64) return launch_dcn5_cfg<2, 2, 256>(a, st);
    // a layer whose 128-channel workgroups would leave CUs idle runs 64-channel workgroups instead (as csrc/dcn3.hip)
    const long wgs4 = (long)op.B * cdiv(op.H, 16) * cdiv(op.W, 16) * cdiv(op.Cout, 128);
    if ((wgs4 < 192 || (op.reserved & 0x200)) && !(op.reserved & 0x400)) return launch_dcn5_cfg<2, 2, 256>(a, st);
    return launch_dcn5_cfg<4, 4, 256>(a, st);
}
