// DeformConv with its offset/mask convolution fused in (reference model.py:346-362 DeformConv ->
// dcn_v2.py:118-128 DCN.forward: conv_offset_mask -> chunk/cat/sigmoid -> dcn_v2_conv -> BN -> ReLU).
//
// One workgroup = 16x16 output pixels x (32*MT) channels, 8 waves x (32 px), two phases that share
// one register-prefetch pipeline over channel chunks:
//   phase A  plain 3x3 conv of the apron tile with the 27 offset/mask filters (32 MFMA rows):
//            the offsets never go to HBM (the reference materialises them as a [B,27,H,W] tensor and
//            re-reads it once per channel in the im2col kernel, dcn_v2_im2col_cuda.cu:170-172).
//            The host permutes the 27 filters over the 32 accumulator rows so that lane half h of
//            a pixel ends up holding complete (dh, dw, mask) triples: h=0 -> taps 0..4, h=1 -> 5..8.
//   geometry each half computes the sampling geometry of ITS taps only (no duplication) and the two
//            halves exchange the results with one cross-half shuffle per value.
//   phase B  dcn2's branch-free gather + fp16 blend + MFMA over the same chunks (csrc/dcn2.hip).
//   patches  (NP > 0, bf16 plans) samples whose corners left the apron do NOT leave the fast path: after the geometry
//            every such (pixel, tap) takes a slot in a per-tile list (wave ballot + one LDS atomic per wave); at every
//            phase-B stage thread e fetches the 4 corners of list entry e for the stage's channels from global memory
//            (range-checked buffer loads: corners outside the image read as zero), blends them with the sample's own
//            weights and drops the result into a "patch" pixel in front of the apron; the sample's gather offset points
//            at that patch with weights (1, 0, 0, 0), so phase B stays branch free.  Cost ~ number of such samples,
//            not "any lane in the wave" x all channels as in pass 2 (at 5 % of the samples pass 2 made the kernel 5x
//            slower).  Samples beyond the NP slots of a tile fall through to pass 2.
//   pass 2   (rare) samples whose corners left the apron and found no patch slot: offsets are re-broadcast from
//            the phase-A accumulators, corners gathered from global memory.
#include "common.h"
#include "epilogue.h"
#include "dcn_traits.h"
#include <type_traits>

struct Dcn3Args {
    const char *in;
    const char *w;      // main weights [rows][9][Cin] of S
    const char *woff;   // offset/mask weights [32][9][Cin] of S, rows permuted (engine.pack_offset_conv)
    const float *bias;  // [rows] main bias followed by [32] permuted offset bias
    char *out;
    int B, H, W, Cin, in_cs;
    int Cout, out_cs, relu, out_mode, wrows;
    int tiles_x, tiles_y;
    int dbg;   // profiling ablation (h3d_op.reserved): 1 no phase-A MFMA, 2 no gather/blend, 4 no phase-B MFMA, 8 stage once, 16 no patch fill
    int G;     // WDMA: 32-row groups of the main filter image
    float wscale, oscale;   // f16x3 plans: 2^-wexp / 2^-wexp2 of the main / offset filters (h3d_op.wexp, wexp2); 1 otherwise
    const unsigned *wmax;   // f16x3, register-staged filters only (the stand-alone `DCN` module): the filters are PLAIN fp32 packs and [0] / [1] hold the
                            // bit patterns of max |main filter| / max |offset filter| (csrc/dcn.hip dcn_fused_pack_f32_if_kernel): scaled by a power of
                            // two and split while they are staged, as csrc/dcn2.hip does for the operator; nullptr = pre-split filters (network plans)
    int xcd;   // h3d_tile_id mode
    unsigned long long *stamps;   // profiling builds: in-kernel phase stamps (common.h H3D_STAMP)
};

// profiling builds with -DDCN3_STAMP_B: the six phase stamps are replaced by sub-step stamps of phase B's second stage
// (wave 0 of every workgroup; tools/stamp_dcn.py --stage-b)
#if defined(H3D_ABLATE) && defined(DCN3_STAMP_B)
#undef H3D_STAMP
#define H3D_STAMP(wg, k) do { if ((k) == 6 && threadIdx.x == 0 && (wg) < 65536 && a.stamps) a.stamps[(wg) * H3D_NSTAMP + 6] = __builtin_readcyclecounter(); } while (0)
#define H3D_STAMP_B(s, k) do { if ((s) == nchunks + 1 && threadIdx.x == 0 && blockIdx.x < 65536 && a.stamps) a.stamps[blockIdx.x * H3D_NSTAMP + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define H3D_STAMP_B(s, k) do { } while (0)
#endif

// PK ("packed" apron, round 3: the wide-margin variants): 32 B per pixel, no pad bytes, rows of HH * 32 B; the 16-byte half a
// lane-half reads is selected by the ROW parity (half' = half ^ (row & 1)), which keeps the 16-lane ds_read_b128 groups of the
// offset convolution on 16 distinct bank slots (rows are an even number of 16-byte slots) -- a margin-4 apron (26 x 26 pixels)
// then takes 21.6 KB where the padded margin-2 one takes 28 KB, so the wide margin still fits two workgroups per CU.  Corner
// (y+1, x) of a sample at byte offset o is (o ^ 16) + RBH.
template <typename T, int MT, int CK, int MARGIN, bool WDMA = false, int NP = 0, bool PK = false>
struct Dcn3Cfg {
    static constexpr int ES = sizeof(T);
    static constexpr int SS = SE<T>::SS;
    static constexpr int HH = 16 + 2 + 2 * MARGIN;
    static constexpr int SBH = PK ? CK * SS : CK * SS + 16;
    static constexpr int RBH = PK ? HH * SBH : ((HH * SBH + 255) / 256) * 256;   // 256 B-aligned rows: the 2-row x 16-px gather of a
                                                                 // 16-lane ds_read_b128 group then covers 16 distinct 16-B slots
    static_assert(!PK || (CK == 16 && SS == 2 && WDMA && NP > 0 && (RBH / 16) % 2 == 0 && (RBH & 16) == 0), "packed apron: fp16 samples, 16-channel stages");
    static constexpr int WB = 9 * CK * SS + 16;
    static constexpr int BN = 32 * MT;
    static constexpr int THREADS = 512;
    static constexpr int VPP = CK * SS / 16;
    static constexpr int LDS_H = HH * RBH;
    static constexpr int WGRP = 32 * WB;                               // one 32-row group of a filter stage
    static constexpr int WPIECES = (MT * WGRP + 1023) / 1024;          // WDMA: KiB pieces of a main-filter stage
    static constexpr int OPIECES = (WGRP + 1023) / 1024;               //       ... of an offset-filter stage
    static constexpr int WSLOT = WDMA ? WPIECES * 1024 : BN * WB;
    // register-staged path: apron and filters are double buffered (stage s+1 is written while stage s is read), which
    // halves the barriers: the SQ counters showed these kernels parked ~43 % of the time on two barriers per stage
    // patches (NP > 0): NP "patch pixels" of CK channels in FRONT of the apron of a stage buffer (a sample's gather offset
    // may then be negative; its three zero-weighted corner reads land in the patch area or the apron: finite data)
    static constexpr int PSLOT = CK * SS;
    static constexpr int PB = NP ? (NP * PSLOT + 255) / 256 * 256 : 0;
    static constexpr int STAGE = PB + LDS_H + WSLOT;                   // one stage buffer (non-WDMA)
    // SINGLE (round 5, f16x3 plans): ONE stage buffer for the register-staged path too -- an fp32 apron with margin 4 (26 x 26 pixels of
    // 80 B: 60 KB) plus its filters is 98 KB, twice that does not exist.  The f32 / f16x3 variants have no patch slots, so every
    // sample outside the apron goes through pass 2 (serialised global gathers: a third of the margin-2 kernel's time at the default
    // offsets, tools/ab_lib.py --offset-scale 0.01 vs 0.5); the wider apron keeps 8 of 10 of them inside for a second barrier per stage.
    static constexpr bool SINGLE = !WDMA && SS == 4 && MARGIN > 2;
    static constexpr int LDS_MAIN = WDMA ? PB + LDS_H + 2 * WSLOT : SINGLE ? STAGE : 2 * STAGE;
    static constexpr int DESC_B = SS == 4 ? 32 : 16;                   // a list entry: (hl | wl), fp16 weight pairs -- or four fp32 weights (f16x3)
    static constexpr int LDS_DESC = NP ? NP * DESC_B + 32 : 0;         // sample list + the eight per-wave sample counts
    static constexpr int LDS_EPI = 8 * ((32 * (64 * MT + 16) + 1023) / 1024 * 1024);   // epilogue.h tile_epilogue_lds regions
    static constexpr int LDS = LDS_MAIN + LDS_DESC > LDS_EPI ? LDS_MAIN + LDS_DESC : LDS_EPI;
    static_assert(NP == 0 || (WDMA && (sizeof(T) == 2 || std::is_same_v<T, x3_t>)), "patches: plans with DMA'd filters (bf16 / fp16 / f16x3)");
    static_assert(NP * (CK * SS / 16) <= 1024, "at most two 16-byte patch units per thread (the second one in a second fill round)");
};

// WDMA (bf16 plans): the filters are stage-major fp16 LDS images (H3D_OP_DCN_FUSED_STREAM) copied by LDS-DMA into a
// two-slot ring one stage ahead; only the apron (which must be converted) still goes through registers.
template <int PIECES>
__device__ __forceinline__ void dcn3_issue_w(const char *base, int bytes, char *dst, int src, int lane16, int wv)
{
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < (PIECES + 7) / 8; ++j) {
        const int p = wv + 8 * j;
        if (p < PIECES)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(dst + p * 1024), 16, lane16, src + p * 1024, 0, 0);
    }
}

// one corner of a patch entry: 16 bytes at byte offset voff (+ soff, the stage's channel offset) of image `img`; an offset
// beyond `bytes` (corner outside the image, idle thread) reads as zero
__device__ __forceinline__ u32x4 dcn3_patch_corner(const char *img, int bytes, int voff, int soff)
{
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)img, 0, bytes, 0x00020000);
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}

// F16IN (round 4, bf16 plans): the INPUT tensor is fp16 although the plan (filters' companion type, output, epilogue) is bf16 -- the
// `node` DeformConvs of IDAUp read a tensor only they consume (`node(up(proj(x)) + skip)`, model.py:384-390), so the up-sample + add
// kernel writes it as fp16 (H3D_OUT_NHWC_F16) and the bf16 -> fp16 conversion of every staged apron vector (3 VALU per pair, between
// the two barriers of a stage where all eight waves do the same thing) disappears; the sample is also more precise (11 significand
// bits instead of 8).  Selected by h3d_op.reserved & 0x40000.
// STATS: the statistics launch of h3d_dcn_far_samples -- its own instantiation (phase A + geometry only; everything behind the
// slot count is compiled out), so that a profiler lists it under its own name and the production kernel carries no switch for it.
template <typename T, int MT, int CK, int MARGIN, int EPI = 0, bool WDMA = false, int NP = 0, bool PK = false, bool F16IN = false, bool STATS = false>   // EPI: 0 general, 1 lean NHWC, 2 LDS-transposed (bf16)
__global__ __launch_bounds__(512, (WDMA && MT <= 2 && sizeof(T) == 2) ? 4 : 2) void dcn3_kernel(Dcn3Args a)
{
    using C = Dcn3Cfg<T, MT, CK, MARGIN, WDMA, NP, PK>;
    using X = SE<std::conditional_t<F16IN, f16_t, T>>;
    static_assert(!F16IN || (std::is_same<T, bf16_t>::value && WDMA && NP > 0), "fp16 input: an option of the bf16 patch-slot variants");
    constexpr int ES = C::ES, SS = C::SS;
    constexpr bool ONEBUF = WDMA || C::SINGLE;     // one apron (and, register-staged, one filter) buffer: a barrier before it is rewritten
    __shared__ __attribute__((aligned(256))) char smem[C::LDS];
    char *s_w = smem + C::PB + C::LDS_H;       // filters of stage buffer 0 (pass 2) / of the WDMA ring

    const int tid = threadIdx.x;
    const int wv = tid >> 6, l = tid & 63, r = l & 31, h = l >> 5;
    const int tiles = a.tiles_x * a.tiles_y;
    const int bid = h3d_tile_id(blockIdx.x, gridDim.x, a.xcd);
    const int b = bid / tiles;
    const int t = bid - b * tiles;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int oy0 = ty * 16, ox0 = tx * 16;
    const int hy0 = oy0 - 1 - MARGIN, hx0 = ox0 - 1 - MARGIN;
    const int cout0 = blockIdx.y * C::BN;
    const int py = wv * 2 + (r >> 4), px = r & 15;          // this lane's pixel inside the tile
    const int oy = oy0 + py, ox = ox0 + px;
    const bool live = (oy < a.H && ox < a.W);
    const char *img = a.in + (size_t)b * a.H * a.W * a.in_cs * ES;
    const int aoff = r * C::WB + 8 * h * SS;
    const int nchunks = a.Cin / CK;
    // filter scales of this launch: the host's exponents (network plans), or derived here from the maxima the pack kernel left (a.wmax)
    [[maybe_unused]] float wsc[2] = {1.f, 1.f}, wun = a.wscale, oun = a.oscale;      // wsc[0]: main, wsc[1]: offset filters
    [[maybe_unused]] bool rawf = false;
    if constexpr (std::is_same_v<T, x3_t> && !WDMA) {
        rawf = a.wmax != nullptr;
        if (rawf) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float m = __uint_as_float(a.wmax[q]);
                int k = 14;
                if (m > 0.f && m < __builtin_inff()) (void)frexpf(m, &k);
                const int e = min(60, max(-60, 14 - k));
                wsc[q] = ldexpf(1.f, e);
                (q ? oun : wun) = ldexpf(1.f, -e);
            }
        }
    }
    [[maybe_unused]] auto store_w = [&](char *dst_row_tap, int v, u32x4 raw, int which) {      // one 16-byte vector of a staged filter row -> LDS
        if constexpr (std::is_same_v<T, x3_t> && !WDMA) {
            if (rawf) {
#pragma unroll
                for (int i = 0; i < 4; ++i) raw[i] = __float_as_uint(__uint_as_float(raw[i]) * wsc[which]);
                x3_store4(dst_row_tap + (v >> 1) * 32, v & 1, raw);
                return;
            }
        }
        *reinterpret_cast<u32x4 *>(dst_row_tap + v * 16) = raw;
    };
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const int off_bytes = nchunks * C::WGRP, main_bytes = nchunks * a.G * C::WGRP;
    // WDMA: filters of stage s -> ring slot s & 1
    auto issue_w = [&](int s) {
        if constexpr (WDMA) {
            char *dst = s_w + (s & 1) * C::WSLOT;
            if (s < nchunks) dcn3_issue_w<C::OPIECES>(a.woff, off_bytes, dst, s * C::WGRP, l * 16, wvu);
            else dcn3_issue_w<C::WPIECES>(a.w, main_bytes, dst, ((s - nchunks) * a.G + (int)blockIdx.y * MT) * C::WGRP, l * 16, wvu);
        }
    };

    // ---- one staging pipeline for both phases: stage s < nchunks = (apron chunk s, offset filters),
    //      stage s >= nchunks = (apron chunk s - nchunks, main filters) ---------------------------------
    constexpr int WV = 9 * C::VPP;
    constexpr int NH = C::HH * C::HH * C::VPP, NW = WDMA ? 0 : C::BN * WV;
    constexpr int NV = (NH + NW + C::THREADS - 1) / C::THREADS;
    u32x4 stg[NV];
    auto load_stage = [&](int s) {
        if ((H3D_DBG(a) & 8) && s > 0) return;
        const bool phaseA = s < nchunks;
        const int c0 = (phaseA ? s : s - nchunks) * CK;
        const char *wsrc = phaseA ? a.woff : a.w;
        const int wr0 = phaseA ? 0 : cout0;
        const int nrows = phaseA ? 32 : C::BN;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + j * C::THREADS;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (i < NH) {
                const int v = i % C::VPP, pix = i / C::VPP;
                const int iy = pix / C::HH, ix = pix - iy * C::HH;
                const int gy = hy0 + iy, gx = hx0 + ix;
                if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                    val = *reinterpret_cast<const u32x4 *>(img + ((size_t)(gy * a.W + gx) * a.in_cs + c0) * ES + v * 16);
            } else if (i < NH + NW) {
                const int q0 = i - NH;
                const int row = q0 / WV, q = q0 - row * WV;
                const int tap = q / C::VPP, v = q - tap * C::VPP;
                if (row < nrows)
                    val = *reinterpret_cast<const u32x4 *>(wsrc + (((size_t)(wr0 + row) * 9 + tap) * a.Cin + c0) * SS + v * 16);
            }
            stg[j] = val;
        }
    };
    auto store_stage = [&](int s) {
        if ((H3D_DBG(a) & 8) && s > 0) return;
        char *s_h = smem + C::PB + (ONEBUF ? 0 : (s & 1) * C::STAGE);
        char *s_w = s_h + C::LDS_H;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + j * C::THREADS;
            if (i < NH) {
                const int v = i % C::VPP, pix = i / C::VPP;
                const int iy = pix / C::HH, ix = pix - iy * C::HH;
                if constexpr (X::SPLIT_A) {
                    if (s < nchunks) {                              // (f16x3, phase A: operand fragments; phase B: the fp32 values the blend reads)
                        x3_store4(s_h + iy * C::RBH + ix * C::SBH + (v >> 1) * 32, v & 1, stg[j]);
                        continue;
                    }
                }
                *reinterpret_cast<u32x4 *>(s_h + iy * C::RBH + ix * C::SBH + ((PK ? (v ^ (iy & 1)) : v) * 16)) = X::convert16(stg[j]);
            } else if (i < NH + NW) {
                const int q0 = i - NH;
                const int row = q0 / WV, q = q0 - row * WV;
                const int tap = q / C::VPP, v = q - tap * C::VPP;
                store_w(s_w + row * C::WB + tap * CK * SS, v, stg[j], s < nchunks ? 1 : 0);
            }
        }
    };

    // ---- D2 (WDMA with patches): the apron loads run TWO stages ahead in phase A.  A stage of phase A is 9 MFMAs per
    //      wave (~0.3k cycles), a global load takes 2-4k: one stage of distance (the pipeline above) left every stage
    //      waiting for its apron (in-kernel stamps: phase A 26 % of a tile for 33 % of its MFMAs).  Two register sets,
    //      stage s in set s & 1 (the loops are unrolled by two, so Cin % 32 == 0); loads are range-checked buffer loads
    //      (outside the image: zeros, no branch), so every wave issues exactly NV of them per stage and the wait for the
    //      filter DMA of stage s can be counted: only the NV loads of stage s+1 were issued after it.
    constexpr bool D2 = WDMA && NP > 0;
#ifndef DCN3_TAPAHEAD
#define DCN3_TAPAHEAD 1
#endif
    constexpr bool TAPAHEAD = DCN3_TAPAHEAD && D2 && MT >= 4 && CK == 16;   // phase B: gathers one tap ahead (see computeB)
    [[maybe_unused]] int avoff[NV], adst[NV];
    [[maybe_unused]] u32x4 stg2[D2 ? 2 : 1][NV];
    if constexpr (D2) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + j * C::THREADS;
            const int v = i % C::VPP, pix = i / C::VPP;
            const int iy = pix / C::HH, ix = pix - iy * C::HH;
            const int gy = hy0 + iy, gx = hx0 + ix;
            avoff[j] = (i < NH && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? ((gy * a.W + gx) * a.in_cs) * ES + v * 16 : 0x7ffffff0;
            adst[j] = i < NH ? iy * C::RBH + ix * C::SBH + ((PK ? (v ^ (iy & 1)) : v) * 16) : -1;
        }
    }
    [[maybe_unused]] const int img_bytes = (int)((size_t)a.H * a.W * a.in_cs * ES);
    auto load2 = [&](int s, auto P) {
        if constexpr (D2) {
            constexpr int q = decltype(P)::value;
            const int c0 = (s < nchunks ? s : s - nchunks) * CK;
#pragma unroll
            for (int j = 0; j < NV; ++j) stg2[q][j] = dcn3_patch_corner(img, img_bytes, avoff[j], c0 * ES);
        }
    };
    auto store2 = [&](auto P, auto PHASE_A) {
        if constexpr (D2) {
            constexpr int q = decltype(P)::value;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if constexpr (X::SPLIT_A && decltype(PHASE_A)::value) {      // f16x3, phase A: finished operand fragments (x3_store4; the vector's
                    if (adst[j] >= 0) x3_store4(smem + C::PB + adst[j] - (tid & 1) * 16, tid & 1, stg2[q][j]);   // half of its 8-channel group = tid & 1: VPP = 4)
                } else {
                    if (adst[j] >= 0) *reinterpret_cast<u32x4 *>(smem + C::PB + adst[j]) = X::convert16(stg2[q][j]);      // (converting in front of the barrier instead, in place: +0.5 %, tools/ab_lib.py)
                }
            }
        }
    };
    [[maybe_unused]] constexpr std::integral_constant<int, 0> I0{};
    [[maybe_unused]] constexpr std::integral_constant<int, 1> I1{};
    [[maybe_unused]] constexpr int WAIT_NV = 0x0f70 | (NV & 15) | ((NV >> 4) << 14);     // s_waitcnt vmcnt(NV)

    H3D_STAMP(blockIdx.x, 6);
    if constexpr (NP > 0) {
        // every byte a zero-weighted corner read of a patched sample can touch must hold a finite number (0 x NaN = NaN):
        // the patch area and the apron are cleared once (pad slots and row tails are never written afterwards; the filter
        // slots only ever hold finite fp16); the barriers of phase A order this before any use
        for (int i = tid * 16; i < C::PB + C::LDS_H; i += C::THREADS * 16) *reinterpret_cast<u32x4 *>(smem + i) = u32x4{0u, 0u, 0u, 0u};
        __syncthreads();                       // ... and before stage 0's apron is stored
    }
    H3D_STAMP(blockIdx.x, 0);
    // ================= phase A: offsets/mask = conv3x3(x; 27 filters) ===============================
    f32x16 aoffs;   // rows (i&3)+8(i>>2)+4h of the permuted offset conv for this lane's pixel
#pragma unroll
    for (int i = 0; i < 16; ++i) aoffs[i] = 0.f;
    // tap (0,0) of the plain conv (PK: rows dy = 0, 2 share the lane half's position, dy = 1 has the other one)
    const int bconv = PK ? (MARGIN + py) * C::RBH + (MARGIN + px) * C::SBH + ((h ^ ((MARGIN + py) & 1)) << 4)
                         : (MARGIN + py) * C::RBH + (MARGIN + px) * C::SBH + 8 * h * SS;
    [[maybe_unused]] const int bconv1 = (bconv ^ 16) + C::RBH;
    auto computeA = [&](int s) {
        const char *s_h = smem + C::PB + (ONEBUF ? 0 : (s & 1) * C::STAGE);
        const char *s_w = s_h + C::LDS_H + (WDMA ? (s & 1) * C::WSLOT : 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
            for (int kk = 0; kk < CK / 16; ++kk) {
                const typename X::wfrag fa = X::lds_w(s_w + aoff + (tap * CK + kk * 16) * SS);
                const typename X::bfrag fb = X::lds_a(s_h + ((PK && dy == 1) ? bconv1 : bconv + dy * C::RBH) + dx * C::SBH + kk * 16 * SS);
                if (H3D_DBG(a) & 1) { X::keep(fa); X::keep(fb); } else X::mma(aoffs, fa, fb);
            }
        }
#ifndef DCN3_ADEPTH
#define DCN3_ADEPTH 4
#endif
        if constexpr (DCN3_ADEPTH > 0 && CK == 16 && sizeof(T) == 2) {
            // the nine MFMAs of a stage form one dependent chain, and hipcc feeds it one tap at a time: two fragment reads, lgkmcnt(1), MFMA
            // -- a read issued two instructions earlier is waited for in front of every MFMA.  Pinned order: the fragments of DCN3_ADEPTH
            // taps requested up front, then one tap's reads behind each MFMA, so a read has DCN3_ADEPTH MFMAs of time to land.
            constexpr int RPT = 2, MPT = 1;        // two 16-byte fragments, one MFMA per tap
            __builtin_amdgcn_sched_group_barrier(0x100, RPT * DCN3_ADEPTH, 0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                __builtin_amdgcn_sched_group_barrier(0x008, MPT, 0);
                if (tap + DCN3_ADEPTH < 9) __builtin_amdgcn_sched_group_barrier(0x100, RPT, 0);
            }
        }
    };
    if constexpr (D2) {
        issue_w(0);
        load2(0, I0);
        load2(1, I1);                            // (nchunks is even: stage 1 exists)
        auto stepA = [&](int s, auto P) {
            if (s) __syncthreads();              // (single apron buffer)
            store2(P, std::true_type{});
            __builtin_amdgcn_s_waitcnt(WAIT_NV); // the filters of stage s have landed; the apron of stage s+1 may still fly
            __syncthreads();
            issue_w(s + 1);
            load2(s + 2, P);                     // s + 2 < 2 nchunks: phase B's first two stages are fetched here too
            computeA(s);
        };
        for (int s = 0; s < nchunks; s += 2) { stepA(s, I0); stepA(s + 1, I1); }
    } else {
    issue_w(0);
    load_stage(0);
    for (int s = 0; s < nchunks; ++s) {
        if (ONEBUF && s) __syncthreads();        // (single apron buffer)
        store_stage(s);
        if constexpr (WDMA) __builtin_amdgcn_s_waitcnt(0x0f70);   // the filters of stage s have landed too
        __syncthreads();
        issue_w(s + 1);
        load_stage(s + 1);                       // s + 1 == nchunks is phase B's first stage
        computeA(s);
    }
    }
    {   // + bias (permuted like the rows)
        const float *bo = a.bias + a.wrows;
        if constexpr (std::is_same_v<T, x3_t>) {
#pragma unroll
            for (int i = 0; i < 16; ++i) aoffs[i] = fmaf(aoffs[i], oun, bo[(i & 3) + 8 * (i >> 2) + 4 * h]);     // (exact power-of-two unscale)
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) aoffs[i] += bo[(i & 3) + 8 * (i >> 2) + 4 * h];
        }
    }

    H3D_STAMP(blockIdx.x, 1);
    // ================= geometry: my taps (h=0: 0..4, h=1: 5..8), then cross-half exchange ===========
    int boff[9];
    typename X::geo geo[9];
    bool slow = false;
    [[maybe_unused]] int pmask = 0;          // NP: bit `tap` = this pixel's sample of that tap lives in a patch
    {
        int my_off[5];
        typename X::geo my_geo[5];
        bool my_want[5];
        uint32_t my_hw[5];
        int my_pm = 0;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            // tap = tb + u: row / column of the tap as a select between the two halves' compile-time values (no per-lane division)
            const int ti = h ? (5 + u) / 3 : u / 3, tj = h ? (5 + u) % 3 : u % 3;
            const float h_im = (float)(oy - 1 + ti) + aoffs[3 * u];
            const float w_im = (float)(ox - 1 + tj) + aoffs[3 * u + 1];
            const bool inside = live & (u < 4 || h == 0) & (h_im > -1.f) & (w_im > -1.f) & (h_im < (float)a.H) & (w_im < (float)a.W);
            // branch free (the divergent `if (inside)` cost five exec-mask regions and ~150 register moves per wave and tile):
            // everything is computed for every lane and SELECTED -- a sample outside the image keeps offset 0 with zero weights
            const float fh = floorf(h_im), fw = floorf(w_im);
            const int hl = (int)fh, wl = (int)fw;
            const int ry = hl - hy0, rx = wl - hx0;
            const float lh = h_im - fh, lw = w_im - fw;
            const float hh = 1.f - lh, hw = 1.f - lw;
            const float w4[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
            typename X::geo g = X::select_geo(inside, X::make_geo(w4, dcn2_sigmoid(aoffs[3 * u + 2])));
            const bool inap = (unsigned)ry < (unsigned)(C::HH - 1) && (unsigned)rx < (unsigned)(C::HH - 1);   // all four corners in the apron
            const int off = (inside && inap) ? ry * C::RBH + rx * C::SBH + (PK ? ((ry & 1) << 4) : 0) : 0;    // (PK: half 0's bytes; half 1 reads off ^ 16)
            const bool want = inside && !inap;               // inside the image, corners outside the apron
            my_want[u] = want;
            my_hw[u] = ((uint32_t)hl << 16) | ((uint32_t)wl & 0xffffu);
            my_off[u] = off;
            my_geo[u] = g;
        }
        if constexpr (NP > 0) {
            // patch slots, in a DETERMINISTIC order (wave, tap, lane): every wave publishes its sample count, and after a barrier
            // takes the slots behind those of the lower-numbered waves.  (Round 2 used one LDS atomic per wave: the order in which
            // the waves arrived decided WHICH samples of a tile with more than NP of them went to pass 2 instead of a patch --
            // two accumulation orders, so such tiles differed in the last bit from run to run: found by the batch-8 full-size
            // test, where the 256-channel 32 x 32 layer runs this variant.)
            unsigned long long m[5];
            int cnt = 0;
#pragma unroll
            for (int u = 0; u < 5; ++u) { m[u] = __ballot(my_want[u]); cnt += __popcll(m[u]); }
            int *s_cnt = reinterpret_cast<int *>(smem + C::LDS_MAIN + NP * C::DESC_B);
            if (l == 0) s_cnt[wv] = cnt;
            __syncthreads();
            int base = 0;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) base += (w8 < wv) ? s_cnt[w8] : 0;
            if (cnt) {                                               // wave-uniform
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int slot = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m[u] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[u], 0u));
                    base += __popcll(m[u]);
                    if (my_want[u] && slot < NP) {
                        if constexpr (sizeof(T) == 2) {
                            *reinterpret_cast<u32x4 *>(smem + C::LDS_MAIN + slot * 16) = u32x4{my_hw[u], my_geo[u].w01, my_geo[u].w23, 0u};
                            my_geo[u].w01 = 0x00003c00u;             // (1, 0 | 0, 0): the blend is done when the patch is filled
                            my_geo[u].w23 = 0u;
                        } else {                                     // f16x3: (hl | wl) and the four fp32 weights (mask folded in)
                            *reinterpret_cast<uint32_t *>(smem + C::LDS_MAIN + slot * C::DESC_B) = my_hw[u];
                            *reinterpret_cast<f32x4 *>(smem + C::LDS_MAIN + slot * C::DESC_B + 16) = f32x4{my_geo[u].w[0], my_geo[u].w[1], my_geo[u].w[2], my_geo[u].w[3]};
                            my_geo[u].w[0] = 1.f; my_geo[u].w[1] = 0.f; my_geo[u].w[2] = 0.f; my_geo[u].w[3] = 0.f;
                        }
                        my_off[u] = slot * C::PSLOT - C::PB;         // the patch pixel, relative to the apron base
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
            const int o_pm = (int)h3d_xor32((uint32_t)my_pm);
            pmask = h == 0 ? (my_pm | (o_pm << 5)) : (o_pm | (my_pm << 5));
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int o_off = (int)h3d_xor32((uint32_t)my_off[u]);
            const typename X::geo o_geo = X::shfl_xor32(my_geo[u]);
            // tap u (u < 5) belongs to half 0, tap 5 + u (u < 4) to half 1
            boff[u] = PK ? ((h == 0 ? my_off[u] : o_off) ^ (h << 4)) : (h == 0 ? my_off[u] : o_off) + 8 * h * SS;
            geo[u] = (h == 0) ? my_geo[u] : o_geo;
            if (u < 4) {
                boff[5 + u] = PK ? ((h == 1 ? my_off[u] : o_off) ^ (h << 4)) : (h == 1 ? my_off[u] : o_off) + 8 * h * SS;
                geo[5 + u] = (h == 1) ? my_geo[u] : o_geo;
            }
        }
    }

    // ---- patches: thread `tid` owns the 16-byte unit tid % VPP of list entry tid / VPP for the whole of phase B ----------
    [[maybe_unused]] int pbase = 0, pok = 0;       // byte offset of corner (hl, wl) of my entry's unit; bit k of pok: corner k inside the image
    [[maybe_unused]] typename X::geo pgeo = X::zero_geo();
    [[maybe_unused]] bool phas = false;
    [[maybe_unused]] bool overflow = false;
    [[maybe_unused]] int nsl2 = 0;                 // slots in use (workgroup-uniform); entries >= 512 / VPP are filled in a second round
    if constexpr (NP > 0) {
        __syncthreads();                                                 // the list is complete
        int nwant = 0;
        {
            const int *s_cnt = reinterpret_cast<const int *>(smem + C::LDS_MAIN + NP * C::DESC_B);
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) nwant += s_cnt[w8];
        }
        overflow = nwant > NP;                                           // workgroup-uniform: some sample found no slot
        if constexpr (STATS) {
            // statistics launch (h3d_dcn_far_samples): `out` is an int32 array with one entry per 16x16 tile; the tile's number of
            // samples whose corners leave THIS variant's apron is all that is produced (what DLAEngine.calibrate_dcn_margins
            // chooses the per-layer variant from -- a deterministic function of the layer's input, not a stopwatch)
            if (tid == 0 && blockIdx.y == 0) reinterpret_cast<int *>(a.out)[bid] = nwant;
            return;
        }
        const int nsl = min(nwant, NP);
        nsl2 = nsl;
        const int ps = tid / C::VPP, pv = tid - ps * C::VPP;
        phas = ps < nsl;
        // (round 5, measured and dropped: a wave-uniform skip of patch_issue / patch_commit for the waves that own no patch unit -- at
        //  the default offsets a tile has 20-90 far samples, i.e. only the first one to three waves do -- made the 16 launches of the
        //  batch-64 plan 3.5 % SLOWER (2.205 vs 2.129 ms, tools/ab_lib.py): the scalar branch splits the block the four loads are
        //  scheduled in, and the loads of an idle thread cost nothing but their issue slot)
        if (phas) {
            const u32x4 d = *reinterpret_cast<const u32x4 *>(smem + C::LDS_MAIN + ps * C::DESC_B);
            const int hl = (int)d[0] >> 16, wl = (int)(short)(d[0] & 0xffffu);
            if constexpr (sizeof(T) == 2) { pgeo.w01 = d[1]; pgeo.w23 = d[2]; }
            else {
                const f32x4 w4 = *reinterpret_cast<const f32x4 *>(smem + C::LDS_MAIN + ps * C::DESC_B + 16);
                pgeo.w[0] = w4[0]; pgeo.w[1] = w4[1]; pgeo.w[2] = w4[2]; pgeo.w[3] = w4[3];
            }
            pbase = ((hl * a.W + wl) * a.in_cs) * ES + pv * 16;          // (may be negative: only used for corners inside the image)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int yy = hl + (k >> 1), xx = wl + (k & 1);
                if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) pok |= 1 << k;
            }
        }
    }
    // my patch unit for stage s (c0 = its first channel): the 4 corner loads are issued BEFORE the barrier that frees the
    // apron buffer (they touch no LDS), so their latency overlaps the barrier wait and the apron stores; blend + store after
    [[maybe_unused]] u32x4 pst[4];
    auto patch_issue = [&](int s) {
        if constexpr (NP > 0) {
            if (H3D_DBG(a) & 16) return;
            const int c0 = (s - nchunks) * CK;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int voff = pbase + ((k & 1) + (k >> 1) * a.W) * a.in_cs * ES;
                pst[k] = dcn3_patch_corner(img, img_bytes, ((pok >> k) & 1) ? voff : 0x7ffffff0, c0 * ES);   // idle threads / corners outside: zeros
            }
        }
    };
    // f16x3: a patch unit is 4 fp32 channels of the sample, blended in fp32 with the entry's four weights
    auto blend4 = [&](const u32x4 (&c)[4], const typename X::geo &g) -> u32x4 {
        u32x4 o = {0u, 0u, 0u, 0u};
        if constexpr (SS == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o[e] = __float_as_uint(__builtin_amdgcn_fmed3f(fmaf(g.w[3], __uint_as_float(c[3][e]), fmaf(g.w[2], __uint_as_float(c[2][e]), fmaf(g.w[1], __uint_as_float(c[1][e]), g.w[0] * __uint_as_float(c[0][e])))),
                                                               -65504.f, 65504.f));      // (the slot is read by X::prep, which does not clamp)
        }
        return o;
    };
    auto patch_commit = [&](char *s_h) {
        if constexpr (NP > 0 && SS == 4) {
            if (!phas || (H3D_DBG(a) & 16)) return;
            *reinterpret_cast<u32x4 *>(s_h - C::PB + tid * 16) = blend4(pst, pgeo);      // entry tid / VPP, unit tid % VPP: PSLOT = VPP * 16
        }
        if constexpr (NP > 0 && sizeof(T) == 2) {
            if (!phas || (H3D_DBG(a) & 16)) return;
            typename X::frag v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k].v = __builtin_bit_cast(half8_t, X::convert16(pst[k]));
            const typename X::frag o = X::blend(v, pgeo);
            *reinterpret_cast<half8_t *>(s_h - C::PB + tid * 16) = o.v;      // entry tid/VPP, unit tid%VPP: PSLOT = VPP * 16
        }
    };

    // NP > 256 (the packed-apron variants): a tile with more far samples than one round of 512 / VPP entries fills the rest in a
    // SECOND round per stage -- entry 256 + tid / VPP, its constants re-read from the list (rare path: no registers kept for
    // it), its four corner loads exposed (they are issued behind the barrier, ~2k cycles per stage) -- instead of sending
    // those samples through pass 2, which costs the tile 3x its time
    auto patch_round2 = [&](int s, char *s_h) {
        if constexpr (NP * C::VPP > 512 && SS == 4) {            // f16x3: 4 units per entry, so entries 128 ... NP - 1 are the second round
            constexpr int R = 512 / C::VPP;
            if (nsl2 <= R) return;                                       // workgroup-uniform
            const int ps = R + tid / C::VPP, pv = tid % C::VPP;
            if (ps >= nsl2) return;
            const uint32_t hw = *reinterpret_cast<const uint32_t *>(smem + C::LDS_MAIN + ps * C::DESC_B);
            const f32x4 w4 = *reinterpret_cast<const f32x4 *>(smem + C::LDS_MAIN + ps * C::DESC_B + 16);
            const int hl = (int)hw >> 16, wl = (int)(short)(hw & 0xffffu);
            typename X::geo g2;
            g2.w[0] = w4[0]; g2.w[1] = w4[1]; g2.w[2] = w4[2]; g2.w[3] = w4[3];
            const int base2 = ((hl * a.W + wl) * a.in_cs) * ES + pv * 16;
            const int c0 = (s - nchunks) * CK;
            u32x4 c[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int yy = hl + (k >> 1), xx = wl + (k & 1);
                const bool ok = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
                const int voff = base2 + ((k & 1) + (k >> 1) * a.W) * a.in_cs * ES;
                c[k] = dcn3_patch_corner(img, img_bytes, ok ? voff : 0x7ffffff0, c0 * ES);
            }
            *reinterpret_cast<u32x4 *>(s_h - C::PB + (R * C::VPP + tid) * 16) = blend4(c, g2);
        }
        if constexpr (NP * C::VPP > 512 && sizeof(T) == 2) {
            constexpr int R = 512 / C::VPP;
            if (nsl2 <= R) return;                                       // workgroup-uniform
            const int ps = R + tid / C::VPP, pv = tid % C::VPP;
            if (ps >= nsl2) return;
            const u32x4 d = *reinterpret_cast<const u32x4 *>(smem + C::LDS_MAIN + ps * 16);
            const int hl = (int)d[0] >> 16, wl = (int)(short)(d[0] & 0xffffu);
            typename X::geo g2;
            g2.w01 = d[1]; g2.w23 = d[2];
            const int base2 = ((hl * a.W + wl) * a.in_cs) * ES + pv * 16;
            const int c0 = (s - nchunks) * CK;
            typename X::frag v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int yy = hl + (k >> 1), xx = wl + (k & 1);
                const bool ok = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
                const int voff = base2 + ((k & 1) + (k >> 1) * a.W) * a.in_cs * ES;
                v[k].v = __builtin_bit_cast(half8_t, X::convert16(dcn3_patch_corner(img, img_bytes, ok ? voff : 0x7ffffff0, c0 * ES)));
            }
            const typename X::frag o = X::blend(v, g2);
            *reinterpret_cast<half8_t *>(s_h - C::PB + (R * C::VPP + tid) * 16) = o.v;
        }
    };

    // Everything from here to the stores, as a function of "is pass 2 compiled in": with patches (NP > 0) a tile needs
    // pass 2 only when it ran out of slots, which is known NOW (workgroup-uniform) -- and only pass 2 reads the phase-A
    // accumulators `aoffs` again.  Instantiated twice, the common branch does not keep them alive through phase B (they
    // were spilled around it: 13 scratch stores + loads per thread and tile, half of the kernel's HBM write traffic).
    auto tail = [&](auto P2) {
    H3D_STAMP(blockIdx.x, 2);
    // ================= phase B: deformable contraction (branch-free, apron samples) ==================
    f32x16 acc[MT][1];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][0][i] = 0.f;
    auto computeB = [&](int s) {
        const char *s_h = smem + C::PB + (ONEBUF ? 0 : (s & 1) * C::STAGE);
        const char *s_w = s_h + C::LDS_H + (WDMA ? (s & 1) * C::WSLOT : 0);
#ifndef DCN3_X3_PIPE
#define DCN3_X3_PIPE 1
#endif
        if constexpr (DCN3_X3_PIPE && std::is_same_v<T, x3_t> && CK == 16 && WDMA) {
            // f16x3, one workgroup per CU (two waves a SIMD): per launch the vector work (blend + split: ~55 instructions per tap), the six
            // MFMAs of a tap and its twelve LDS reads each cost about the same -- and hipcc's order runs them one after the other in every
            // wave (gather -> wait -> blend -> split -> filter reads -> wait -> MFMAs): the kernel took the SUM of the three.  A wave issues in
            // order, so vector work only overlaps matrix work when it sits BETWEEN the MFMAs: tap t+1 is blended and split while tap t is
            // multiplied (second operand register set), one MFMA per ~6 vector instructions, the order pinned by sched_group_barrier; tap
            // t+2's corners are requested as soon as tap t+1 is blended.  (tools/ab_lib.py, batch 64, same process: the margin-2 launches
            // 1.048 -> 0.986 ms for four of them, the margin-3 ones +0.7 % at 144 instead of 112 bytes of scratch; a first version that only
            // moved the corner request of tap t+1 behind the blend of tap t, one register set: -4 % / +2 %)
            typename X::frag v[4];
            typename X::bfrag pb[2];
            auto gather = [&](int tap) {
                const char *p00 = s_h + boff[tap], *p10 = p00 + C::RBH;
                v[0] = X::lds(p00);
                v[1] = X::lds(p00 + C::SBH);
                v[2] = X::lds(p10);
                v[3] = X::lds(p10 + C::SBH);
            };
            gather(0);
            {
                const typename X::frag fb = X::blend(v, geo[0]);
                __builtin_amdgcn_sched_barrier(0);
                gather(1);
                pb[0] = X::prep(fb);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                typename X::wfrag fa[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[m] = X::lds_w(s_w + aoff + m * 32 * C::WB + tap * CK * SS);
                if (tap + 1 < 9) {
                    const typename X::frag fb = X::blend(v, geo[tap + 1]);
                    if (tap + 2 < 9) gather(tap + 2);
                    pb[(tap + 1) & 1] = X::prep(fb);
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) X::mma(acc[m][0], fa[m], pb[tap & 1]);
                // pinned order: filter reads, then MFMA / vector / MFMA / vector ...; the corner reads of tap t+2 behind the blend
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * MT, 0);
                if (tap + 1 < 9) {
#pragma unroll
                    for (int i = 0; i < 3 * MT; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (i == 3 * MT - 1) __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);
                        else __builtin_amdgcn_sched_group_barrier(0x002, 36 / (3 * MT - 1) + 1, 0);
                        if (i == (MT > 1 ? 2 : 1) && tap + 2 < 9) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                    }
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 3 * MT, 0);
                }
            }
            return;
        }
        if constexpr (TAPAHEAD) {
            // one workgroup per CU = two waves per SIMD: they cannot cover the gather -> blend -> MFMA chain of a tap by
            // themselves, so tap t+1's four corner fragments
            // are requested before tap t is blended and multiplied (16 more live registers: only the 128-channel variant has
            // them)
            static_assert(CK == 16, "one 16-channel fragment per tap");
            typename X::frag v[2][4];
            auto gather = [&](int tap, int q) {
                const char *p00 = s_h + boff[tap];
                const char *p10 = PK ? s_h + ((boff[tap] ^ 16) + C::RBH) : p00 + C::RBH;
                v[q][0] = X::lds(p00);
                v[q][1] = X::lds(p00 + C::SBH);
                v[q][2] = X::lds(p10);
                v[q][3] = X::lds(p10 + C::SBH);
            };
            gather(0, 0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                // LDS returns in order: tap t's filter fragments first (they arrive while tap t is blended), then the
                // corners of tap t+1 (they arrive while tap t is multiplied)
                typename X::wfrag fa[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[m] = X::lds_w(s_w + aoff + m * 32 * C::WB + tap * CK * SS);
                if (tap + 1 < 9) gather(tap + 1, (tap + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);            // (hipcc sinks the reads back in front of their first use otherwise)
                const typename X::frag fb = X::blend(v[tap & 1], geo[tap]);
#pragma unroll
                for (int m = 0; m < MT; ++m) X::mma(acc[m][0], fa[m], fb);
            }
            return;
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            typename X::frag fb[CK / 16];
            const char *p00 = s_h + boff[tap];
            const char *p10 = PK ? s_h + ((boff[tap] ^ 16) + C::RBH) : p00 + C::RBH;
#pragma unroll
            for (int kk = 0; kk < CK / 16; ++kk) {
                if (H3D_DBG(a) & 2) { fb[kk] = X::lds(p00 + kk * 16 * SS); continue; }
                typename X::frag v[4];
                v[0] = X::lds(p00 + kk * 16 * SS);
                v[1] = X::lds(p00 + C::SBH + kk * 16 * SS);
                v[2] = X::lds(p10 + kk * 16 * SS);
                v[3] = X::lds(p10 + C::SBH + kk * 16 * SS);
                fb[kk] = X::blend(v, geo[tap]);
            }
#pragma unroll
            for (int kk = 0; kk < CK / 16; ++kk) {
                typename X::wfrag fa[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[m] = X::lds_w(s_w + aoff + m * 32 * C::WB + (tap * CK + kk * 16) * SS);
                if (H3D_DBG(a) & 4) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) { X::keep(fa[m]); X::keep(fb[kk]); }
                    continue;
                }
                const typename X::bfrag pb = X::prep(fb[kk]);       // (f16x3 plans: the fp32 sample split into fp16 terms, once for all M-tiles)
#pragma unroll
                for (int m = 0; m < MT; ++m) X::mma(acc[m][0], fa[m], pb);
            }
        }
    };
    if constexpr (D2) {
        // one stage of distance from here on (the accumulators are live now); stage nchunks+1 is already in flight
        auto stepB = [&](int s, auto P, auto Q) {
            H3D_STAMP_B(s, 0);
            patch_issue(s);
            __syncthreads();
            H3D_STAMP_B(s, 1);
            store2(P, std::false_type{});
            patch_commit(smem + C::PB);
            patch_round2(s, smem + C::PB);
            __builtin_amdgcn_s_waitcnt(0x0f70);  // (the patch loads were the youngest: nothing older is pending either)
            H3D_STAMP_B(s, 2);
            __syncthreads();
            H3D_STAMP_B(s, 3);
            if (s + 1 < 2 * nchunks) {
                issue_w(s + 1);
                if (s != nchunks) load2(s + 1, Q);
            }
            H3D_STAMP_B(s, 4);
            computeB(s);
            H3D_STAMP_B(s, 5);
        };
        for (int s = nchunks; s < 2 * nchunks; s += 2) { stepB(s, I0, I1); stepB(s + 1, I1, I0); }
    } else {
    for (int s = nchunks; s < 2 * nchunks; ++s) {
        patch_issue(s);
        if (ONEBUF) __syncthreads();
        store_stage(s);
        patch_commit(smem + C::PB + (ONEBUF ? 0 : (s & 1) * C::STAGE));
        if constexpr (WDMA) __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        if (s + 1 < 2 * nchunks) { issue_w(s + 1); load_stage(s + 1); }
        computeB(s);
    }
    }

    H3D_STAMP(blockIdx.x, 3);
    // ================= pass 2 (rare): samples whose corners left the apron ===========================
    bool do_p2 = false;
    // (round 5, measured and dropped: the far samples of the fp32-storage variants -- which have no patch slots -- taken one at a time
    //  by the wave that owns the pixel, 64 lanes = 64 output channels walking Cin as a dot product against filter rows read from global
    //  memory, no barrier, no LDS: 2.4x SLOWER on the f32 plan's 16 launches (42.1 vs 18.3 ms) and 1.8x on the f16x3 plan's margin-4
    //  tiles (19.4 vs 10.9): a sample is ~5 us of dependent global round trips, and some layers have hundreds per tile)
    if constexpr (decltype(P2)::value) do_p2 = NP > 0 ? overflow : (bool)__syncthreads_or(slow ? 1 : 0);
    if constexpr (decltype(P2)::value) if (do_p2) {
        if constexpr (D2 && MT < 4) {
            // Under the 128-VGPR cap the phase-A accumulators would be spilled around phase B just for this rare path
            // (13 scratch stores + loads per thread and tile: half of the kernel's HBM write traffic).  Recomputed instead:
            // the offset convolution once more, unpipelined -- only tiles that ran out of patch slots come here.
#pragma unroll
            for (int i = 0; i < 16; ++i) aoffs[i] = 0.f;
            for (int s = 0; s < nchunks; ++s) {
                __syncthreads();
                issue_w(s);
                load2(s, I0);
                store2(I0, std::true_type{});
                __builtin_amdgcn_s_waitcnt(0x0f70);
                __syncthreads();
                computeA(s);
            }
            const float *bo = a.bias + a.wrows;
            if constexpr (std::is_same_v<T, x3_t>) {
#pragma unroll
                for (int i = 0; i < 16; ++i) aoffs[i] = fmaf(aoffs[i], oun, bo[(i & 3) + 8 * (i >> 2) + 4 * h]);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) aoffs[i] += bo[(i & 3) + 8 * (i >> 2) + 4 * h];
            }
        }
        if constexpr (D2) {
            // The tile ran out of patch slots.  Pass 2 as round 1 wrote it takes one global round trip per tap and chunk
            // (9 x Cin/16 serialised latencies: 5x the tile's time at 5 % of the samples).  Here the geometry of the nine taps
            // is computed once, and per chunk the corner loads of TB taps are in flight together (range-checked buffer
            // loads: lanes without a pending sample read zeros, no branch); a wave skips the taps none of its lanes needs.
            // Registers spill in this branch under the 128-VGPR cap -- it is the rare path.
            constexpr int TB = SS == 4 ? 1 : 3;          // (an fp32 corner fragment is 32 bytes per lane: two loads, twice the registers)
            constexpr int CV = SS == 4 ? 2 : 1;          // 16-byte loads per corner fragment
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
            u32x4 pv[TB][4][CV];
            for (int c0 = 0; c0 < a.Cin; c0 += CK) {
                auto fetch = [&](int t0) {
#pragma unroll
                    for (int j = 0; j < TB; ++j) {
                        if (!((wmask >> (t0 + j)) & 1)) continue;                // wave-uniform
#pragma unroll
                        for (int k = 0; k < 4; ++k)
#pragma unroll
                            for (int cv = 0; cv < CV; ++cv)
                                pv[j][k][cv] = dcn3_patch_corner(img, img_bytes, ((qok[t0 + j] >> k) & 1) ? qb[t0 + j] + (k & 1) * pxb + (k >> 1) * rowb + cv * 16 : 0x7ffffff0, c0 * ES);
                    }
                };
                __syncthreads();
                dcn3_issue_w<C::WPIECES>(a.w, main_bytes, s_w, ((c0 / CK) * a.G + (int)blockIdx.y * MT) * C::WGRP, l * 16, wvu);
                fetch(0);
                __builtin_amdgcn_s_waitcnt(0x0f70);
                __syncthreads();
#pragma unroll
                for (int t0 = 0; t0 < 9; t0 += TB) {
                    typename X::bfrag fbs[TB];
#pragma unroll
                    for (int j = 0; j < TB; ++j) {
                        if (!((wmask >> (t0 + j)) & 1)) continue;
                        typename X::frag v[4];
                        if constexpr (SS == 4) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) { v[k].lo = __builtin_bit_cast(f32x4, pv[j][k][0]); v[k].hi = __builtin_bit_cast(f32x4, pv[j][k][CV - 1]); }
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[k].v = __builtin_bit_cast(half8_t, X::convert16(pv[j][k][0]));
                        }
                        fbs[j] = X::prep_raw(X::blend(v, qg[t0 + j]));
                    }
                    if (t0 + TB < 9) fetch(t0 + TB);                               // the next taps fly while these are multiplied
#pragma unroll
                    for (int j = 0; j < TB; ++j) {
                        if (!((wmask >> (t0 + j)) & 1)) continue;
                        typename X::wfrag fa[MT];
#pragma unroll
                        for (int m = 0; m < MT; ++m) fa[m] = X::lds_w(s_w + aoff + m * 32 * C::WB + ((t0 + j) * CK) * SS);
#pragma unroll
                        for (int m = 0; m < MT; ++m) X::mma(acc[m][0], fa[m], fbs[j]);
                    }
                }
            }
        } else
        for (int c0 = 0; c0 < a.Cin; c0 += CK) {
            __syncthreads();
            if constexpr (WDMA) {
                dcn3_issue_w<C::WPIECES>(a.w, main_bytes, s_w, ((c0 / CK) * a.G + (int)blockIdx.y * MT) * C::WGRP, l * 16, wvu);
                __builtin_amdgcn_s_waitcnt(0x0f70);
            } else {
                for (int i = tid; i < C::BN * WV; i += C::THREADS) {
                    const int row = i / WV, q = i - row * WV;
                    const int tap = q / C::VPP, v = q - tap * C::VPP;
                    store_w(s_w + row * C::WB + tap * CK * SS, v, *reinterpret_cast<const u32x4 *>(
                        a.w + (((size_t)(cout0 + row) * 9 + tap) * a.Cin + c0) * SS + v * 16), 0);
                }
            }
            __syncthreads();
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ti = tap / 3, tj = tap - ti * 3;
                // raw (dh, dw, mask logit) of this tap live in the half that owns it: broadcast to both
                const int src = (tap < 5) ? r : r + 32, u = (tap < 5) ? tap : tap - 5;
                const float d_h = __shfl(aoffs[3 * u], src), d_w = __shfl(aoffs[3 * u + 1], src),
                            d_m = __shfl(aoffs[3 * u + 2], src);
                typename X::frag fb[CK / 16];
#pragma unroll
                for (int kk = 0; kk < CK / 16; ++kk) fb[kk] = X::zero();
                bool any = false;
                const float h_im = (float)(oy - 1 + ti) + d_h;
                const float w_im = (float)(ox - 1 + tj) + d_w;
                if (live && h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W) {
                    const int hl = (int)floorf(h_im), wl = (int)floorf(w_im);
                    const int ry = hl - hy0, rx = wl - hx0;
                    if (!(ry >= 0 && ry + 1 < C::HH && rx >= 0 && rx + 1 < C::HH) && !((pmask >> tap) & 1)) {
                        any = true;
                        const float lh = h_im - (float)hl, lw = w_im - (float)wl;
                        const float hh = 1.f - lh, hw = 1.f - lw;
                        const float w4[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
                        const typename X::geo g = X::make_geo(w4, dcn2_sigmoid(d_m));
                        const bool okh0 = hl >= 0, okh1 = hl + 1 <= a.H - 1, okw0 = wl >= 0, okw1 = wl + 1 <= a.W - 1;
                        const bool ok[4] = {okh0 && okw0, okh0 && okw1, okh1 && okw0, okh1 && okw1};
                        const int pix[4] = {hl * a.W + wl, hl * a.W + wl + 1, (hl + 1) * a.W + wl, (hl + 1) * a.W + wl + 1};
#pragma unroll
                        for (int kk = 0; kk < CK / 16; ++kk) {
                            typename X::frag v[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                v[k] = ok[k] ? X::global8(img + ((size_t)pix[k] * a.in_cs + c0 + kk * 16 + 8 * h) * ES) : X::zero();
                            fb[kk] = X::blend(v, g);
                        }
                    }
                }
                if (!__any(any)) continue;
#pragma unroll
                for (int kk = 0; kk < CK / 16; ++kk) {
                    typename X::wfrag fa[MT];
#pragma unroll
                    for (int m = 0; m < MT; ++m) fa[m] = X::lds_w(s_w + aoff + m * 32 * C::WB + (tap * CK + kk * 16) * SS);
                    const typename X::bfrag pb = X::prep_raw(fb[kk]);
#pragma unroll
                    for (int m = 0; m < MT; ++m) X::mma(acc[m][0], fa[m], pb);
                }
            }
        }
    }

    H3D_STAMP(blockIdx.x, 4);
    if constexpr (std::is_same_v<T, x3_t>) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][0][i] *= wun;
    }
    EpiArgs e;
    e.bias = a.bias; e.res = nullptr; e.out = a.out; e.Ho = a.H; e.Wo = a.W; e.Cout = a.Cout;
    e.out_cs = a.out_cs; e.res_cs = 0; e.relu = a.relu; e.out_mode = a.out_mode;
    if constexpr (EPI == 2) {
        __syncthreads();                          // the apron and the filters are no longer read
        tile_epilogue_lds<T, MT>(acc, e, b, oy0, ox0, cout0, wv, l, smem + wv * epi_lds_stride<MT>());
    } else {
        tile_epilogue<typename StoreT<T>::type, MT, 1, EPI == 1>(acc, e, b, oy0, ox0, cout0, wv, r, h);
    }
    };
    // (only where registers allow: under the 128-VGPR cap of the two-workgroups-per-CU variants the second copy of
    //  phase B made the allocation worse -- 64 -> 64 @128x128: 0.238 -> 0.283 ms -- while the 128-channel variants gained
    //  13-18 %: 0.184 -> 0.160 ms on 256 -> 256 @32x32)
    if constexpr (NP > 0 && MT >= 4) {
        if (overflow) tail(std::true_type{});
        else tail(std::false_type{});
    } else {
        tail(std::true_type{});
    }
    H3D_STAMP(blockIdx.x, 5);
}

template <typename T, int MT, int CK, int MARGIN, bool WDMA = false, int NP = 0, bool PK = false, bool F16IN = false>
static int launch_dcn3_cfg(const Dcn3Args &a0, hipStream_t st)
{
    using C = Dcn3Cfg<T, MT, CK, MARGIN, WDMA, NP, PK>;
    static_assert(!(WDMA && MT <= 2 && sizeof(T) == 2) || C::LDS * 2 <= 160 * 1024, "two workgroups per CU");
    static_assert(C::LDS <= 160 * 1024, "LDS budget");
    Dcn3Args a = a0;
    a.tiles_x = cdiv(a.W, 16);
    a.tiles_y = cdiv(a.H, 16);
    a.xcd = h3d_xcd_mode();
#ifdef H3D_ABLATE
    a.stamps = h3d_stamp_buffer();
#else
    a.stamps = nullptr;
#endif
    dim3 grid(a.B * a.tiles_x * a.tiles_y, cdiv(a.Cout, C::BN));
    const bool lean = a.out_mode == H3D_OUT_NHWC && a.Cout % 4 == 0 && ((uintptr_t)a.bias & 15) == 0;
    const int epi = (sizeof(T) == 2 && MT >= 2 && lean && a.Cout % 8 == 0 && a.out_cs % 8 == 0 && ((uintptr_t)a.out & 15) == 0) ? 2 : lean ? 1 : 0;
    if (h3d_note_kernel(F16IN ? (PK ? "dcn3_kernel<%s, %d, %d, %d, %d, %s, %d, true, true>" : "dcn3_kernel<%s, %d, %d, %d, %d, %s, %d, false, true>")
                              : PK ? "dcn3_kernel<%s, %d, %d, %d, %d, %s, %d, true>" : "dcn3_kernel<%s, %d, %d, %d, %d, %s, %d>", h3d_tname<T>(), MT, CK, MARGIN, epi,
                        WDMA ? "true" : "false", NP))
        return H3D_OK;
    if constexpr (NP > 0 && sizeof(T) == 2) {
        if (a.dbg & 0x20000) {                  // h3d_dcn_far_samples
            hipLaunchKernelGGL((dcn3_kernel<T, MT, CK, MARGIN, 1, WDMA, NP, PK, F16IN, true>), dim3(grid.x), dim3(C::THREADS), 0, st, a);
            H3D_CHECK_LAUNCH("dcn3_kernel<stats>");
            return H3D_OK;
        }
    }
    if constexpr (sizeof(T) == 2 && MT >= 2) {
        if (epi == 2) {
            hipLaunchKernelGGL((dcn3_kernel<T, MT, CK, MARGIN, 2, WDMA, NP, PK, F16IN>), grid, dim3(C::THREADS), 0, st, a);
            H3D_CHECK_LAUNCH("dcn3_kernel");
            return H3D_OK;
        }
    }
    if constexpr (F16IN) {
        // (only the LDS-transposed epilogue is instantiated for the fp16-input variants: every layer of the network that uses them
        //  has Cout % 8 == 0 and an aligned NHWC output)
        H3D_FAIL(H3D_ERR_UNSUPPORTED, "dcn_fused_stream with an fp16 input: needs an NHWC output with Cout %% 8 == 0 (Cout=%d)", a.Cout);
    } else {
    if (epi == 1)
        hipLaunchKernelGGL((dcn3_kernel<T, MT, CK, MARGIN, 1, WDMA, NP, PK>), grid, dim3(C::THREADS), 0, st, a);
    else
        hipLaunchKernelGGL((dcn3_kernel<T, MT, CK, MARGIN, 0, WDMA, NP, PK>), grid, dim3(C::THREADS), 0, st, a);
    }
    H3D_CHECK_LAUNCH("dcn3_kernel");
    return H3D_OK;
}

// channels per filter stage of H3D_OP_DCN_FUSED_STREAM: hosts pack the stage-major filter images with this CK
extern "C" int h3d_dcn_fused_ck(int Cin, int Cout) { (void)Cin; (void)Cout; return 16; }

// 2-byte plans (bf16_t: the apron is converted to fp16 while it is staged; f16_t: it is fp16 already)
template <typename T, bool F16IN = false>
static int launch_dcn3_lowp(const h3d_op &op, const Dcn3Args &a, bool wdma, hipStream_t st)
{
    if constexpr (F16IN) {
        if (!wdma || (op.reserved & 0x1000) || op.Cin % 32 || op.Cout <= 32)
            H3D_FAIL(H3D_ERR_UNSUPPORTED, "dcn_fused_stream: the fp16-input option (reserved & 0x40000) exists for the patch-slot variants with > 32 output channels");
    }
    if (wdma) {
        if ((op.reserved & 0x1000) || op.Cin % 32) {     // tuning override: round 1's configurations (no patches: every sample that
                                                         // leaves the apron goes through pass 2); also Cin = 16 (mod 32): the
                                                         // patch variants' pipeline is unrolled by two stages
            if (op.Cout <= 32) return launch_dcn3_cfg<T, 1, 16, 1, true>(a, st);
            if (op.Cout <= 64) return launch_dcn3_cfg<T, 2, 16, 1, true>(a, st);
            return launch_dcn3_cfg<T, 4, 16, 2, true>(a, st);
        }
        const long wgs4w = (long)op.B * cdiv(op.H, 16) * cdiv(op.W, 16) * cdiv(op.Cout, 128);
        if (op.reserved & 0x8000) {
            // wide margin on the packed apron (engine.dcn_wide_margin / DLAEngine.calibrate_dcn_margins: layers whose offsets send
            // many samples outside a margin-2 apron): margin 4 at two workgroups per CU (73 KB).  The 128-channel variant has
            // margin 4 anyway (a margin-6 packed apron needs a fourth staging register set: 14 spilled registers)
            if (op.Cout <= 32) return launch_dcn3_cfg<T, 1, 16, 4, true, 256, true>(a, st);
            if (op.Cout <= 64 || ((wgs4w < 192 || (op.reserved & 0x200)) && !(op.reserved & 0x400))) return launch_dcn3_cfg<T, 2, 16, 4, true, 256, true, F16IN>(a, st);
            return launch_dcn3_cfg<T, 4, 16, 4, true, 256, false, F16IN>(a, st);
        }
        // <= 64 output channels: margin-2 apron, 16-channel stages, <= 128 VGPRs and 78 KB of LDS -> two workgroups
        // (16 waves) per CU, one computing while the other waits at its stage barriers; 256 patch slots per tile.
        // > 64: one workgroup per CU has the LDS for a margin-4 apron (26 x 26 pixels)
        const long wgs4 = (long)op.B * cdiv(op.H, 16) * cdiv(op.W, 16) * cdiv(op.Cout, 128);
        if (op.reserved & 0x10000) {
            // experiment / candidate default: margin 2 on the PACKED apron (15 KB instead of 28) with 512 patch slots per tile, the second
            // 256 filled in a second round per stage
            if (op.Cout <= 32) return launch_dcn3_cfg<T, 1, 16, 2, true, 512, true>(a, st);
            if (op.Cout <= 64 || ((wgs4 < 192 || (op.reserved & 0x200)) && !(op.reserved & 0x400))) return launch_dcn3_cfg<T, 2, 16, 2, true, 512, true, F16IN>(a, st);
            return launch_dcn3_cfg<T, 4, 16, 4, true, 512, false, F16IN>(a, st);
        }
        if (op.Cout <= 32) return launch_dcn3_cfg<T, 1, 16, 2, true, 256>(a, st);
        if (op.Cout <= 64) return launch_dcn3_cfg<T, 2, 16, 2, true, 256, false, F16IN>(a, st);
        // a layer whose 128-channel workgroups would leave CUs idle (16 x 16 maps at batch 64: 128 workgroups on 256 CUs)
        // runs 64-channel workgroups instead: twice the gather / blend work, on CUs that had nothing to do
        if ((wgs4 < 192 || (op.reserved & 0x200)) && !(op.reserved & 0x400)) return launch_dcn3_cfg<T, 2, 16, 2, true, 256, false, F16IN>(a, st);
        return launch_dcn3_cfg<T, 4, 16, 4, true, 256, false, F16IN>(a, st);
    }
    if (op.Cin % 32 == 0 && op.Cout <= 64) {
        if (op.Cout <= 32) return launch_dcn3_cfg<T, 1, 32, 2>(a, st);
        return launch_dcn3_cfg<T, 2, 32, 2>(a, st);
    }
    if (op.Cout <= 32) return launch_dcn3_cfg<T, 1, 16, 2>(a, st);
    if (op.Cout <= 64) return launch_dcn3_cfg<T, 2, 16, 2>(a, st);
    // a layer whose 128-channel workgroups would leave CUs idle (16x16 maps at batch 64: 128 workgroups) runs
    // 64-channel workgroups instead: twice the gather / blend work, but on CUs that had nothing to do
    const long wgs4 = (long)op.B * cdiv(op.H, 16) * cdiv(op.W, 16) * cdiv(op.Cout, 128);
    if ((wgs4 < 192 || (op.reserved & 0x200)) && !(op.reserved & 0x400)) {
        if (op.Cin % 32 == 0) return launch_dcn3_cfg<T, 2, 32, 2>(a, st);
        return launch_dcn3_cfg<T, 2, 16, 2>(a, st);
    }
    return launch_dcn3_cfg<T, 4, 16, 2>(a, st);
}

int h3d_launch_dcn5(const h3d_op &op, hipStream_t st);      // csrc/dcn5.hip: fp16 plans, every operand by LDS-DMA

int h3d_launch_dcn3(const h3d_op &op, hipStream_t st)
{
    const bool wdma = op.kind == H3D_OP_DCN_FUSED_STREAM;
    if (wdma && op.dtype != H3D_BF16 && op.dtype != H3D_F16 && op.dtype != H3D_F16X3) H3D_FAIL(H3D_ERR_DTYPE, "dcn_fused_stream: bf16 / fp16 / f16x3 plans only");
    if (!op.in || !op.w || !op.bias || !op.out || !op.in2) H3D_FAIL(H3D_ERR_ARG, "dcn_fused: null pointer");
    const int es = h3d_dtype_bytes(op.dtype);
    if (!es) H3D_FAIL(H3D_ERR_DTYPE, "dcn_fused: dtype %d", op.dtype);
    if (op.ksize != 3 || op.stride != 1 || op.Ho != op.H || op.Wo != op.W)
        H3D_FAIL(H3D_ERR_UNSUPPORTED, "dcn_fused: covers 3x3 s1 p1 d1 dg1 only (k=%d s=%d)", op.ksize, op.stride);
    if (op.Cin % 16 || op.in_cs % (16 / es) || op.Cin > op.in_cs)
        H3D_FAIL(H3D_ERR_SHAPE, "dcn_fused: Cin=%d (stride %d) must be a multiple of 16", op.Cin, op.in_cs);
    if (op.H > 32767 || op.W > 32767) H3D_FAIL(H3D_ERR_SHAPE, "dcn_fused: image larger than 32767");
    if (op.wrows < ((op.Cout + 127) / 128) * 128)
        H3D_FAIL(H3D_ERR_SHAPE, "dcn_fused: packed weight rows %d < Cout %d padded to 128", op.wrows, op.Cout);
    if (op.out_mode != H3D_OUT_NCHW_F32 && (op.out_cs % 4 || op.Cout > op.out_cs))
        H3D_FAIL(H3D_ERR_SHAPE, "dcn_fused: out channel stride %d", op.out_cs);
    Dcn3Args a;
    a.in = (const char *)op.in; a.w = (const char *)op.w; a.woff = (const char *)op.in2; a.bias = op.bias;
    a.out = (char *)op.out; a.B = op.B; a.H = op.H; a.W = op.W; a.Cin = op.Cin; a.in_cs = op.in_cs;
    a.Cout = op.Cout; a.out_cs = op.out_cs; a.relu = op.relu; a.out_mode = op.out_mode; a.wrows = op.wrows;
    a.tiles_x = a.tiles_y = 0;
    a.dbg = op.reserved;
    a.G = op.wrows / 32;
    if (op.wexp < -60 || op.wexp > 60 || op.wexp2 < -60 || op.wexp2 > 60 || ((op.wexp || op.wexp2) && op.dtype != H3D_F16X3))
        H3D_FAIL(H3D_ERR_ARG, "dcn_fused: wexp %d / %d (H3D_F16X3 filter exponents)", op.wexp, op.wexp2);
    a.wscale = ldexpf(1.f, -op.wexp);
    a.oscale = ldexpf(1.f, -op.wexp2);
    a.wmax = nullptr;
    if (wdma && (size_t)op.H * op.W * op.in_cs * es >= 0x7ffffff0ull) H3D_FAIL(H3D_ERR_SHAPE, "dcn_fused_stream: image of 2 GiB or more");
    if (op.dtype == H3D_BF16 && (op.reserved & 0x40000)) return launch_dcn3_lowp<bf16_t, true>(op, a, wdma, st);
    if (op.dtype == H3D_BF16) return launch_dcn3_lowp<bf16_t>(op, a, wdma, st);
    // fp16 plans: the apron needs no conversion while it is staged.  csrc/dcn5.hip also moves it by LDS-DMA (double buffered, one
    // barrier per phase-A stage): measured 5.7 % SLOWER on the ten <= 64-channel launches of the batch-64 plan (1.519 vs 1.437 ms,
    // tools/ab_dcn5.py: an LDS-DMA piece costs its wave more issue cycles than two global loads + two ds_write_b128, and the
    // kernel is bound by LDS reads and vector issue, not by the staging), so it runs only on request (tuning override 0x4000)
    if (op.dtype == H3D_F16 && wdma && (op.reserved & 0x4000) && !(op.reserved & 0x3000)) {
#ifdef H3D_EXTRA
        return h3d_launch_dcn5(op, st);
#else
        H3D_FAIL(H3D_ERR_UNSUPPORTED, "dcn_fused_stream: the LDS-DMA apron variant (csrc/dcn5.hip, reserved & 0x4000) is built only by `make EXTRA=1`");
#endif
    }
    if (op.dtype == H3D_F16) return launch_dcn3_lowp<f16_t>(op, a, wdma, st);
    if (op.dtype == H3D_F32) {
        if (op.Cout <= 32) return launch_dcn3_cfg<float, 1, 16, 2>(a, st);
        return launch_dcn3_cfg<float, 2, 16, 2>(a, st);
    }
    if (op.dtype == H3D_F16X3 && wdma) {
        // f16x3 with PATCH SLOTS (round 5): the 2-byte plans' pipeline -- filters as stage-major images by LDS-DMA (pre-split (hi | lo)
        // terms: 592-byte rows), the apron two stages ahead through registers, 256 patch slots per tile (32-byte list entries: four
        // fp32 weights; a patch unit is four fp32 channels; entries 128 ... 255 in a second fill round) -- on the fp32 apron of margin 2:
        // 141 KB of LDS, one workgroup per CU.  A far sample costs its four corner loads instead of a share of pass 2's re-walk.
        if (op.Cin % 32) H3D_FAIL(H3D_ERR_UNSUPPORTED, "dcn_fused_stream (f16x3): Cin %d must be a multiple of 32 (two stages per pipeline turn); use H3D_OP_DCN_FUSED", op.Cin);
        // The in-kernel stamps of the margin-3 tiles (make ABLATE=1, tools/stamp_dcn.py --f16x3) showed pass 2 -- the tiles with more far
        // samples than patch slots -- at 21 % of the 256 -> 256 layer and 12 % of a 128 -> 128 one: the `node` DeformConvs (Cin == Cout: their
        // input is the up-sampled sum) have the largest offsets.  Margin 4 (26 x 26 apron, 160 KB) for them, margin 2 for the rest
        // (tools/ab_op_reserved.py --kind 12 --codes 0x4000 0x8000 0x10000, batch 64, same process, margin 2 / 3 / 4: 256 -> 256 @32x32
        // 0.791 / 0.757 / 0.541 ms, 128 -> 128 @64x64 1.343 / 1.222 / 1.112 for the two launches, 64 -> 64 @128x128 2.849 / 3.001 / 3.178 for
        // the five (margin 3 was ahead there before phase B interleaved its vector work with the MFMAs), 256 -> 128 0.446 / 0.451 / 0.472,
        // 512 -> 256 @16x16 0.186 / 0.199 / 0.215).  0x4000 / 0x8000 / 0x10000 force margin 2 / 3 / 4.
        if (((op.Cin == op.Cout && op.Cout > 64) || (op.reserved & 0x10000)) && !(op.reserved & 0xc000)) {
            if (op.Cout <= 32) return launch_dcn3_cfg<x3_t, 1, 16, 4, true, 256>(a, st);
            return launch_dcn3_cfg<x3_t, 2, 16, 4, true, 256>(a, st);
        }
        if ((op.reserved & 0x8000) && !(op.reserved & 0x4000)) {
            if (op.Cout <= 32) return launch_dcn3_cfg<x3_t, 1, 16, 3, true, 256>(a, st);
            return launch_dcn3_cfg<x3_t, 2, 16, 3, true, 256>(a, st);
        }
        if (op.Cout <= 32) return launch_dcn3_cfg<x3_t, 1, 16, 2, true, 256>(a, st);
        return launch_dcn3_cfg<x3_t, 2, 16, 2, true, 256>(a, st);
    }
    if (op.dtype == H3D_F16X3) {            // the f32 plan's tiles (fp32 apron, register-staged pre-split filters) on 3 fp16 MFMAs per step
        // 0x100000 (the stand-alone `DCN` module, h3d_amd/dcn_v2.py): plain fp32 filter packs of h3d_dcn_fused_pack_f32_cached, the two filter
        // maxima behind the biases (bias[wrows + 32 ...]); the kernel scales and splits the filters while it stages them
        if (op.reserved & 0x100000) {
            if (op.wexp || op.wexp2) H3D_FAIL(H3D_ERR_ARG, "dcn_fused (f16x3, raw filters): wexp must be 0 (the scale comes from the pack's maxima)");
            a.wmax = (const unsigned *)(op.bias + op.wrows + 32);
        }
        if (op.reserved & 0x2000) {         // tuning override (tools/ab_flag.py): the f32 plan's margin-2 double-buffered tile
            if (op.Cout <= 32) return launch_dcn3_cfg<x3_t, 1, 16, 2>(a, st);
            return launch_dcn3_cfg<x3_t, 2, 16, 2>(a, st);
        }
        // margin 6 (30 x 30 apron, 115 KB with its filters) on the large maps with <= 64 output channels (tools/ab_op_reserved.py, batch
        // 64, same process, margin 4 / 2 / 6: 64 -> 64 @128x128 4.79 / 4.60 / 4.32 ms for the five launches, 128 -> 64 @64x64 1.58 / 1.76 /
        // 1.49; the 128- and 256-channel layers 3.73 / 3.88 / 3.79: they stay on margin 4); 0x4000 / 0x8000: force margin 6 / margin 4
        if (((op.Cout <= 64 && op.H >= 64) || (op.reserved & 0x4000)) && !(op.reserved & 0x8000)) {
            if (op.Cout <= 32) return launch_dcn3_cfg<x3_t, 1, 16, 6>(a, st);
            return launch_dcn3_cfg<x3_t, 2, 16, 6>(a, st);
        }
        if (op.Cout <= 32) return launch_dcn3_cfg<x3_t, 1, 16, 4>(a, st);      // margin 4, one stage buffer (Dcn3Cfg::SINGLE)
        return launch_dcn3_cfg<x3_t, 2, 16, 4>(a, st);
    }
    H3D_FAIL(H3D_ERR_DTYPE, "dcn_fused: dtype %d", op.dtype);
}

// Per-tile count of bilinear samples that leave the LDS apron of the variant `op` dispatches to (see include/h3d.h).
extern "C" int h3d_dcn_far_samples(const h3d_op *op_in, int32_t *per_tile, void *stream)
{
    if (!op_in || !per_tile) H3D_FAIL(H3D_ERR_ARG, "dcn_far_samples: null pointer");
    if (op_in->kind != H3D_OP_DCN_FUSED_STREAM || (op_in->dtype != H3D_BF16 && op_in->dtype != H3D_F16) || op_in->Cin % 32 || (op_in->reserved & 0x1000))
        H3D_FAIL(H3D_ERR_UNSUPPORTED, "dcn_far_samples: a 2-byte H3D_OP_DCN_FUSED_STREAM op of a patch-slot variant (Cin %% 32 == 0)");
    h3d_op op = *op_in;
    op.reserved = (op.reserved & 0x58600) | 0x20000;      // variant bits (margin / slots / workgroup width) + the statistics switch
    op.out = per_tile;
    op.out_mode = H3D_OUT_NHWC;
    return h3d_launch_dcn3(op, (hipStream_t)stream);
}
