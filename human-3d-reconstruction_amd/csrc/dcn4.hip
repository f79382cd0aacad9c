// DeformConv with fused offsets, generation 4: 64-channel fp16 input, every operand by LDS-DMA.
// (reference model.py:346-362 DeformConv -> dcn_v2.py:118-128 DCN.forward; same math as csrc/dcn3.hip,
//  which keeps the other channel counts, bf16 inputs and the fp32 parity mode.)
//
// The five IDAUp `node` layers at the finest level (64 -> 64 channels at H/4 x W/4) are the most
// expensive launches of the network and their input is produced by the up-sample + add kernel for
// them alone, so that kernel writes fp16 (H3D_OUT_NHWC_F16) and nothing has to be converted here:
//   apron   the (16+2+2*MARGIN)^2 pixel neighbourhood of the 16x16 tile with ALL 64 channels is DMA'd
//           once: [22 rows][4 KiB], pixel stride 144 B (8 data slots + 1 pad slot of 16 B), so both the
//           plain-conv fragments of the offset convolution and the 4-corner gathers are conflict free
//           (odd multiple of 16 B between pixels, 0 mod 256 B between rows).  Pixels outside the image
//           carry an out-of-range buffer offset and arrive as zeros.
//   filters stream through a two-slot ring in stages of 16 input channels: 4 stages of the 27 (32 rows)
//           offset/mask filters, then 4 stages of the 32*MT main filters; the host packs both stage-major
//           in LDS image order (19 slots per row: 2 per tap + 1 pad), so each stage is a linear copy.
// No staging registers, no ds_write, no address arithmetic inside the loops; six barriers per tile.
// Phases as in dcn3: A (offset conv) -> geometry (split across half-waves) -> B (gather + fp16 blend +
// MFMA) -> rare pass 2 for samples whose corners left the apron (global gather).
#include "common.h"
#include "epilogue.h"
#include "dcn_traits.h"

struct Dcn4Args {
    const char *in;     // fp16 NHWC
    const char *wimg;   // main filters  [4][G][32][19][8] fp16
    const char *woff;   // offset filters [4][32][19][8] fp16, rows permuted (engine.offset_conv)
    const float *bias;  // [wrows] main bias followed by [32] permuted offset bias
    char *out;          // bf16 NHWC
    int B, H, W, in_cs;
    int Cout, out_cs, relu, out_mode, wrows;
    int G, tiles_x, tiles_y;
    int dbg;            // profiling (ABLATE builds): leave after 1 the DMA prologue, 2 phase A, 3 geometry, 4 phase B
    // UP = 1 (H3D_OP_UPDCN_F16): the input is  skip + ConvTranspose2d_depthwise(xlo)  computed into the apron
    const char *xlo;    // bf16 NHWC [B][Hl][Wl][xlo_cs], Hl = H / f
    const char *skip;   // bf16 NHWC [B][H][W][skip_cs]
    const float *wup;   // fp32 [k*k][64] tap-major (k = 2f), as H3D_OP_UPADD
    int f, Hl, Wl, xlo_cs, skip_cs;
};

// DENSE = 0: MARGIN 2, apron rows of 4 KiB, three-slot main-filter ring, one 8-wave workgroup per CU (148 KiB).
// DENSE = 1: MARGIN 1, apron rows of 3 KiB, ONE filter slot, 79 KiB -> two workgroups per CU (<= 128 VGPRs):
//            the second workgroup computes while the first waits for its apron / filter stage / stores,
//            which the phase timings showed to be 40 % of a tile (prologue DMA 18 %, stores 22 %).
template <int MT, int DENSE>
struct Dcn4Cfg {
    static constexpr int MARGIN = DENSE ? 1 : 2;
    static constexpr int HH = 16 + 2 + 2 * MARGIN;     // 22 / 20
    static constexpr int PXB = 144;
    static constexpr int ROWB = (HH * PXB + 1023) / 1024 * 1024;   // 4096 / 3072: 0 mod 256 B
    static constexpr int RPARTS = ROWB / 1024;
    static constexpr int APIECES = HH * RPARTS;        // 88 / 60
    static constexpr int APRON = HH * ROWB;            // 90112 / 61440
    static constexpr int WROW = 19 * 16, WGRP = 32 * WROW;
    static constexpr int WPIECES = (MT * WGRP + 1023) / 1024;
    static constexpr int WSLOT = WPIECES * 1024;
    static constexpr int NSLOT = DENSE ? 1 : 3;        // main-filter ring: stage s+NSLOT-1 is in flight while stage s is consumed
    static constexpr int OROUND = DENSE ? 2 : 4;       // offset-filter stages that land together
    static constexpr int OPIECES = (OROUND * WGRP + 1023) / 1024;
    static constexpr int RING = (NSLOT * WSLOT > OPIECES * 1024) ? NSLOT * WSLOT : OPIECES * 1024;
    static constexpr int LDS = APRON + RING;
    static constexpr int NSTAGE = 4;                   // 64 input channels / 16
};

typedef __attribute__((address_space(3))) void lds_void4;

// one filter stage -> ring slot `dst` (linear copy of `pieces` KiB starting at byte `src` of buffer `base`)
template <int PIECES>
__device__ __forceinline__ void dcn4_issue_w(const char *base, int bytes, char *dst, int src, int woff, int wv)
{
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < (PIECES + 7) / 8; ++j) {
        const int p = wv + 8 * j;
        if (p < PIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void4 *)(dst + p * 1024), 16, woff, src + p * 1024, 0, 0);
    }
}

// UP, pass 2 only: 8 channels (c0 .. c0+7) of  skip + up(xlo)  at pixel (gy, gx) inside the image, straight from global
// memory (tap table included): the same arithmetic as the apron prologue
__device__ __forceinline__ u32x4 dcn4_up8(const Dcn4Args &a, const char *xb, const char *sb, int gy, int gx, int c0)
{
    const int fs = a.f >> 1, k = 2 * a.f, pd = a.f >> 1;
    float acc[8], x[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[n] = 0.f;
    const int ry = (gy + pd) & (a.f - 1), rx = (gx + pd) & (a.f - 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ky = ry + (q >> 1) * a.f, kx = rx + (q & 1) * a.f;
        const int ny = gy + pd - ky, nx = gx + pd - kx;
        const int iy = ny >> fs, ix = nx >> fs;
        if (!(ny >= 0 && iy < a.Hl && nx >= 0 && ix < a.Wl)) continue;
        unpack16<bf16_t>(*reinterpret_cast<const u32x4 *>(xb + ((size_t)(iy * a.Wl + ix) * a.xlo_cs + c0) * 2), x);
        const float *wp = a.wup + (ky * k + kx) * 64 + c0;
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[n] = fmaf(wp[n], x[n], acc[n]);
    }
    unpack16<bf16_t>(*reinterpret_cast<const u32x4 *>(sb + ((size_t)(gy * a.W + gx) * a.skip_cs + c0) * 2), x);
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[n] += x[n];
    return pack16_f16(acc);
}

template <int MT, int EPI, int DENSE, int UP = 0>
__global__ __launch_bounds__(512, DENSE ? 4 : 2) void dcn4_kernel(Dcn4Args a)
{
    using C = Dcn4Cfg<MT, DENSE>;
    using X = SE<bf16_t>;
    __shared__ __attribute__((aligned(1024))) char smem[C::LDS];
    char *s_ring = smem + C::APRON;

    const int tid = threadIdx.x;
    const int l = tid & 63, r = l & 31, h = l >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = a.tiles_x * a.tiles_y;
    const int b = blockIdx.x / tiles;
    const int t = blockIdx.x - b * tiles;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int oy0 = ty * 16, ox0 = tx * 16;
    const int hy0 = oy0 - 1 - C::MARGIN, hx0 = ox0 - 1 - C::MARGIN;
    const int g0 = blockIdx.y * MT, cout0 = g0 * 32;
    const int py = wv * 2 + (r >> 4), px = r & 15;
    const int oy = oy0 + py, ox = ox0 + px;
    const bool live = (oy < a.H && ox < a.W);
    const size_t img_bytes = (size_t)a.H * a.W * a.in_cs * 2;
    const char *img = a.in + (size_t)b * img_bytes;
    const int woffl = l * 16;
    const int off_bytes = C::NSTAGE * C::WGRP, main_bytes = C::NSTAGE * a.G * C::WGRP;

    if constexpr (!UP) {
    // ---- apron: HH rows x RPARTS KiB pieces, wave w takes pieces w, w+8, ...; a lane's pixel column and channel
    //      slot follow from the piece's KiB within the row ---------------------------------------------
    {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)img, 0, (int)img_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < (C::APIECES + 7) / 8; ++j) {
            const int q = wv + 8 * j;
            if (q < C::APIECES) {
                const int row = q / C::RPARTS, part = q - row * C::RPARTS;
                const int slot = part * 64 + l;
                const int ix = slot / 9, sub = slot - 9 * ix;
                const int gx = hx0 + ix, gy = hy0 + row;
                const bool ok = ix < C::HH && sub < 8 && gx >= 0 && gx < a.W && gy >= 0 && gy < a.H;
                const int voff = ok ? ((gy * a.W + gx) * a.in_cs + sub * 8) * 2 : 0x7ffffff0;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void4 *)(smem + q * 1024), 16, voff, 0, 0, 0);
            }
        }
    }
    // the first OROUND offset-filter stages land in the (still idle) main-filter ring together with the apron; with
    // all four (DENSE = 0) phase A runs its 36 MFMAs per wave without a single wait (staged one by one, each 9-MFMA
    // stage exposed a full DMA round trip: the SQ counters showed the waves parked 52 % of the time)
    dcn4_issue_w<C::OPIECES>(a.woff, off_bytes, s_ring, 0, woffl, wv);
    } else {
    // ---- UP: the apron is computed, not copied: apron[y][x][c] = skip[y][x][c] + sum of the 2x2 taps of the depthwise
    //      transposed convolution of xlo that reach (y, x) -- the arithmetic of upadd_kernel (csrc/conv.hip), same
    //      order, same single rounding to fp16 -- so the up-sampled sum (134 MB at batch 64) never exists in HBM.
    //      Thread = (channel vector tid & 7, apron pixel (tid >> 3) + 64 j); the k*k x 64 tap table sits in the idle ring.
        const int fs = a.f >> 1;                                // f = 2 or 4: log2
        const int k = 2 * a.f, pd = a.f >> 1;
        const int vec = tid & 7, pp = tid >> 3;
        const char *xb = a.xlo + (size_t)b * a.Hl * a.Wl * a.xlo_cs * 2;
        const char *sb = a.skip + (size_t)b * a.H * a.W * a.skip_cs * 2;
        constexpr int NIT = (C::HH * C::HH + 63) / 64;          // 7
        u32x4 xv[2][4], sv[2];
        auto geom = [&](int j, int &gy, int &gx, int &ldso) -> bool {
            const int pidx = pp + 64 * j;
            const int row = pidx / C::HH, col = pidx - row * C::HH;
            gy = hy0 + row; gx = hx0 + col;
            ldso = row * C::ROWB + col * C::PXB + vec * 16;
            return pidx < C::HH * C::HH;
        };
        auto taps = [&](int g, int lim, int (&ii)[2], int (&kk)[2], bool (&ok)[2]) {   // one axis: input index, kernel index, valid
            const int rr = (g + pd) & (a.f - 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                kk[t] = rr + t * a.f;
                const int num = g + pd - kk[t];
                ii[t] = num >> fs;                               // exact when num >= 0 (multiple of f)
                ok[t] = num >= 0 && ii[t] < lim;
            }
        };
        auto fetch = [&](int j, int buf) {
            int gy, gx, ldso;
            const bool live = geom(j, gy, gx, ldso) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            int iy[2], ky[2], ix[2], kx[2];
            bool oky[2], okx[2];
            taps(gy, a.Hl, iy, ky, oky);
            taps(gx, a.Wl, ix, kx, okx);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                xv[buf][q] = u32x4{0u, 0u, 0u, 0u};
                if (live && oky[q >> 1] && okx[q & 1])
                    xv[buf][q] = *reinterpret_cast<const u32x4 *>(xb + ((size_t)(iy[q >> 1] * a.Wl + ix[q & 1]) * a.xlo_cs + vec * 8) * 2);
            }
            sv[buf] = u32x4{0u, 0u, 0u, 0u};
            if (live) sv[buf] = *reinterpret_cast<const u32x4 *>(sb + ((size_t)(gy * a.W + gx) * a.skip_cs + vec * 8) * 2);
        };
        const float *s_w = reinterpret_cast<const float *>(s_ring);
        auto blend = [&](int j, int buf) {
            int gy, gx, ldso;
            if (!geom(j, gy, gx, ldso)) return;
            const bool live = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            int iy[2], ky[2], ix[2], kx[2];
            bool oky[2], okx[2];
            taps(gy, a.Hl, iy, ky, oky);
            taps(gx, a.Wl, ix, kx, okx);
            float acc[8], x[8];
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[n] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (!(live && oky[q >> 1] && okx[q & 1])) continue;     // taps (a, c2) ascending, as upadd_kernel
                unpack16<bf16_t>(xv[buf][q], x);
                const float *wp = s_w + (ky[q >> 1] * k + kx[q & 1]) * 64 + vec * 8;
                const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wp), w1 = *reinterpret_cast<const f32x4 *>(wp + 4);
#pragma unroll
                for (int n = 0; n < 4; ++n) { acc[n] = fmaf(w0[n], x[n], acc[n]); acc[4 + n] = fmaf(w1[n], x[4 + n], acc[4 + n]); }
            }
            unpack16<bf16_t>(sv[buf], x);
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[n] += x[n];
            u32x4 o = pack16_f16(acc);
            if (!live) o = u32x4{0u, 0u, 0u, 0u};                       // outside the image: the zero padding
            *reinterpret_cast<u32x4 *>(smem + ldso) = o;
        };
        auto issue_table = [&]() {   // tap table -> ring (KiB pieces), behind the first fetches in the same vmcnt queue
            const int tbytes = k * k * 64 * 4;
            const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.wup, 0, tbytes, 0x00020000);
            for (int p = wv; p * 1024 < tbytes; p += 8)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void4 *)(s_ring + p * 1024), 16, woffl, p * 1024, 0, 0);
        };
        if (a.f == 2) {
            // 2x: the output pixels (2m+1 .. 2m+2) x (2n+1 .. 2n+2) share their four inputs (m .. m+1) x (n .. n+1), so a
            // thread takes such a QUAD: 8 loads and 4 input conversions for 4 apron pixels instead of 20 and 16.  The apron
            // (rows -2 .. 17 of the tile) is covered by 11 x 11 quads; quad rows / columns -1 and 20 are outside it.
            constexpr int QN = C::HH / 2 + 1, QIT = (QN * QN + 63) / 64;      // 11, 2
            u32x4 qv[QIT][4], qs[QIT][4];
            auto qgeom = [&](int j, int &m, int &n) -> bool {
                const int item = pp + 64 * j;
                const int qy = item / QN, qx = item - qy * QN;
                m = (hy0 >> 1) - 1 + qy; n = (hx0 >> 1) - 1 + qx;             // hy0, hx0 are even
                return item < QN * QN;
            };
            auto qfetch = [&](int j) {
                int m, n;
                const bool act = qgeom(j, m, n);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int iy = m + 1 - (q >> 1), ix = n + 1 - (q & 1);     // tap order of upadd_kernel
                    qv[j][q] = u32x4{0u, 0u, 0u, 0u};
                    if (act && iy >= 0 && iy < a.Hl && ix >= 0 && ix < a.Wl)
                        qv[j][q] = *reinterpret_cast<const u32x4 *>(xb + ((size_t)(iy * a.Wl + ix) * a.xlo_cs + vec * 8) * 2);
                    const int gy = 2 * m + 1 + (q >> 1), gx = 2 * n + 1 + (q & 1);
                    qs[j][q] = u32x4{0u, 0u, 0u, 0u};
                    if (act && (unsigned)(gy - hy0) < (unsigned)C::HH && (unsigned)(gx - hx0) < (unsigned)C::HH && gy >= 0 && gy < a.H &&
                        gx >= 0 && gx < a.W)
                        qs[j][q] = *reinterpret_cast<const u32x4 *>(sb + ((size_t)(gy * a.W + gx) * a.skip_cs + vec * 8) * 2);
                }
            };
            auto qblend = [&](int j) {
                int m, n;
                if (!qgeom(j, m, n)) return;
                float xf[4][8];
                bool okin[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int iy = m + 1 - (q >> 1), ix = n + 1 - (q & 1);
                    okin[q] = iy >= 0 && iy < a.Hl && ix >= 0 && ix < a.Wl;
                    unpack16<bf16_t>(qv[j][q], xf[q]);
                }
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int gy = 2 * m + 1 + (o >> 1), gx = 2 * n + 1 + (o & 1);
                    const int row = gy - hy0, col = gx - hx0;
                    if (!((unsigned)row < (unsigned)C::HH && (unsigned)col < (unsigned)C::HH)) continue;
                    const bool live = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                    float acc[8], x[8];
#pragma unroll
                    for (int c8 = 0; c8 < 8; ++c8) acc[c8] = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (!okin[q]) continue;
                        const int ky = (o >> 1) + 2 * (q >> 1), kx = (o & 1) + 2 * (q & 1);   // (gy + 1) & 1 == o >> 1
                        const float *wp = s_w + (ky * 4 + kx) * 64 + vec * 8;
                        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wp), w1 = *reinterpret_cast<const f32x4 *>(wp + 4);
#pragma unroll
                        for (int c8 = 0; c8 < 4; ++c8) {
                            acc[c8] = fmaf(w0[c8], xf[q][c8], acc[c8]);
                            acc[4 + c8] = fmaf(w1[c8], xf[q][4 + c8], acc[4 + c8]);
                        }
                    }
                    unpack16<bf16_t>(qs[j][o], x);
#pragma unroll
                    for (int c8 = 0; c8 < 8; ++c8) acc[c8] += x[c8];
                    u32x4 ov = pack16_f16(acc);
                    if (!live) ov = u32x4{0u, 0u, 0u, 0u};
                    *reinterpret_cast<u32x4 *>(smem + row * C::ROWB + col * C::PXB + vec * 16) = ov;
                }
            };
            static_assert(QIT == 2, "quad schedule");
            qfetch(0);
            issue_table();
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
            qblend(0);                                          // (fetching quad 1 up front as well costs 32 more live registers
            qfetch(1);                                          //  under the 128-VGPR cap: spills in the main loop)
            qblend(1);
        } else {
            fetch(0, 0);
            fetch(1, 1);
            issue_table();
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NIT; ++j) {
                blend(j, j & 1);
                if (j + 2 < NIT) fetch(j + 2, j & 1);
            }
        }
        __syncthreads();                                        // apron complete, tap table no longer read
        dcn4_issue_w<C::OPIECES>(a.woff, off_bytes, s_ring, 0, woffl, wv);
    }

    // ================= phase A: offsets/mask = conv3x3(x; 27 filters) ==================================
    f32x16 aoffs;                                 // accumulators start at the (permuted) offset bias: its loads retire with the
    {                                             // apron, not between the main-filter DMAs and their first use
        const float *bo = a.bias + a.wrows;
#pragma unroll
        for (int i = 0; i < 16; ++i) aoffs[i] = bo[(i & 3) + 8 * (i >> 2) + 4 * h];
    }
    const int bconv = (C::MARGIN + py) * C::ROWB + (C::MARGIN + px) * C::PXB + h * 16;
    const int aoff = r * C::WROW + h * 16;
    __builtin_amdgcn_s_waitcnt(0x0f70);           // vmcnt(0)
    __syncthreads();
    if (H3D_DBG(a) == 1) return;
#pragma unroll
    for (int s = 0; s < C::NSTAGE; ++s) {
        if (s > 0 && s % C::OROUND == 0) {        // DENSE: stages 2,3 replace 0,1 (the CU's other workgroup covers the wait)
            __syncthreads();
            dcn4_issue_w<C::OPIECES>(a.woff, off_bytes, s_ring, s * C::WGRP, woffl, wv);
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
        }
        const char *sw = s_ring + (s % C::OROUND) * C::WGRP;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            const typename X::frag fa = X::lds(sw + aoff + tap * 32);
            const typename X::frag fb = X::lds(smem + bconv + dy * C::ROWB + dx * C::PXB + s * 32);
            X::mma(aoffs, fa, fb);
        }
    }
    __syncthreads();                              // the ring is free: the first main stages fly while the geometry is computed
    if (H3D_DBG(a) == 2) { if (aoffs[0] == 1234.5f) a.out[0] = 1; return; }
    dcn4_issue_w<C::WPIECES>(a.wimg, main_bytes, s_ring, g0 * C::WGRP, woffl, wv);
    if (C::NSLOT > 1) dcn4_issue_w<C::WPIECES>(a.wimg, main_bytes, s_ring + C::WSLOT, (a.G + g0) * C::WGRP, woffl, wv);

    // ================= geometry (branch free): my taps (h=0: 0..4, h=1: 5..8), cross-half exchange ====
    int boff[9];
    typename X::geo geo[9];
    bool slow = false;
    {
        int my_off[5];
        typename X::geo my_geo[5];
        const int tb = h ? 5 : 0;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int tap = tb + u;
            const int ti = tap / 3, tj = tap - ti * 3;
            const float h_im = (float)(oy - 1 + ti) + aoffs[3 * u];
            const float w_im = (float)(ox - 1 + tj) + aoffs[3 * u + 1];
            const bool inside = live && tap < 9 && (h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W);
            const float fh = floorf(h_im), fw = floorf(w_im);
            const int ry = (int)fh - hy0, rx = (int)fw - hx0;
            const bool inap = (unsigned)ry < (unsigned)(C::HH - 1) && (unsigned)rx < (unsigned)(C::HH - 1);
            const bool use = inside && inap;
            slow |= inside && !inap;
            const float lh = h_im - fh, lw = w_im - fw;
            const float hh = 1.f - lh, hw = 1.f - lw;
            const float w4[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
            typename X::geo g = X::make_geo(w4, dcn2_sigmoid(aoffs[3 * u + 2]));
            g.w01 = use ? g.w01 : 0u;
            g.w23 = use ? g.w23 : 0u;
            my_off[u] = use ? ry * C::ROWB + rx * C::PXB : 0;
            my_geo[u] = g;
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int o_off = __shfl_xor(my_off[u], 32);
            const typename X::geo o_geo = X::shfl_xor32(my_geo[u]);
            boff[u] = (h == 0 ? my_off[u] : o_off) + h * 16;
            geo[u].w01 = (h == 0) ? my_geo[u].w01 : o_geo.w01;
            geo[u].w23 = (h == 0) ? my_geo[u].w23 : o_geo.w23;
            if (u < 4) {
                boff[5 + u] = (h == 1 ? my_off[u] : o_off) + h * 16;
                geo[5 + u].w01 = (h == 1) ? my_geo[u].w01 : o_geo.w01;
                geo[5 + u].w23 = (h == 1) ? my_geo[u].w23 : o_geo.w23;
            }
        }
    }

    if (H3D_DBG(a) == 3) { int q = 0; for (int u = 0; u < 9; ++u) q += boff[u] + geo[u].w01 + geo[u].w23; if (q == 12345) a.out[0] = 1; __builtin_amdgcn_s_waitcnt(0x0f70); return; }
    // ================= phase B: deformable contraction (branch-free, apron samples) ==================
    f32x16 acc[MT][1];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][0][i] = 0.f;
#pragma unroll
    for (int s = 0; s < C::NSTAGE; ++s) {
        constexpr int PMIN = C::WPIECES / 8;
        static_assert(PMIN >= 1 && PMIN <= 15, "vmcnt immediate");
        if constexpr (C::NSLOT > 1) {
            // stage s has landed once at most the pieces of stage s+1 are outstanding: every wave issues at least
            // PMIN pieces per stage and vmcnt retires in order, so vmcnt(PMIN) is safe for all waves
            if (s + 1 < C::NSTAGE) __builtin_amdgcn_s_waitcnt(0x0f70 | PMIN);
            else __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
            if (s + 2 < C::NSTAGE)
                dcn4_issue_w<C::WPIECES>(a.wimg, main_bytes, s_ring + ((s + 2) % C::NSLOT) * C::WSLOT, ((s + 2) * a.G + g0) * C::WGRP, woffl, wv);
        } else {
            if (s > 0) {
                __syncthreads();
                dcn4_issue_w<C::WPIECES>(a.wimg, main_bytes, s_ring, (s * a.G + g0) * C::WGRP, woffl, wv);
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
        }
        const char *sw = s_ring + (s % C::NSLOT) * C::WSLOT;
        if constexpr (!DENSE) {
            // software pipeline: tap t+1's four corner fragments and filter fragments are in flight while tap t is
            // blended and multiplied (2 waves per SIMD cannot hide the LDS latency by themselves: the SQ counters
            // showed the waves parked on s_waitcnt half of the time)
            typename X::frag v[2][4], fa[2][MT];
            auto gather = [&](int tap, int buf) {
                const char *p00 = smem + boff[tap] + s * 32;
                v[buf][0] = X::lds(p00);
                v[buf][1] = X::lds(p00 + C::PXB);
                v[buf][2] = X::lds(p00 + C::ROWB);
                v[buf][3] = X::lds(p00 + C::ROWB + C::PXB);
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[buf][m] = X::lds(sw + aoff + m * C::WGRP + tap * 32);
            };
            gather(0, 0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap + 1 < 9) gather(tap + 1, (tap + 1) & 1);
                const typename X::frag fb = X::blend(v[tap & 1], geo[tap]);
#pragma unroll
                for (int m = 0; m < MT; ++m) X::mma(acc[m][0], fa[tap & 1][m], fb);
            }
        } else {
            // four waves per SIMD: the other waves cover the LDS latency, no register double buffer
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const char *p00 = smem + boff[tap] + s * 32;
                typename X::frag v[4], fa[MT];
                v[0] = X::lds(p00);
                v[1] = X::lds(p00 + C::PXB);
                v[2] = X::lds(p00 + C::ROWB);
                v[3] = X::lds(p00 + C::ROWB + C::PXB);
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[m] = X::lds(sw + aoff + m * C::WGRP + tap * 32);
                const typename X::frag fb = X::blend(v, geo[tap]);
#pragma unroll
                for (int m = 0; m < MT; ++m) X::mma(acc[m][0], fa[m], fb);
            }
        }
    }

    if (H3D_DBG(a) == 4) { if (acc[0][0][0] + acc[MT - 1][0][5] == 1234.5f) a.out[0] = 1; return; }
    // ================= pass 2 (rare): samples whose corners left the apron ===========================
    if (__syncthreads_or(slow ? 1 : 0)) {
        for (int s = 0; s < C::NSTAGE; ++s) {
            __syncthreads();
            dcn4_issue_w<C::WPIECES>(a.wimg, main_bytes, s_ring, (s * a.G + g0) * C::WGRP, woffl, wv);
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ti = tap / 3, tj = tap - ti * 3;
                const int src = (tap < 5) ? r : r + 32, u = (tap < 5) ? tap : tap - 5;
                const float d_h = __shfl(aoffs[3 * u], src), d_w = __shfl(aoffs[3 * u + 1], src),
                            d_m = __shfl(aoffs[3 * u + 2], src);
                typename X::frag fb = X::zero();
                bool any = false;
                const float h_im = (float)(oy - 1 + ti) + d_h;
                const float w_im = (float)(ox - 1 + tj) + d_w;
                if (live && h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W) {
                    const int hl = (int)floorf(h_im), wl = (int)floorf(w_im);
                    const int ry = hl - hy0, rx = wl - hx0;
                    if (!((unsigned)ry < (unsigned)(C::HH - 1) && (unsigned)rx < (unsigned)(C::HH - 1))) {
                        any = true;
                        const float lh = h_im - (float)hl, lw = w_im - (float)wl;
                        const float hh = 1.f - lh, hw = 1.f - lw;
                        const float w4[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
                        const typename X::geo g = X::make_geo(w4, dcn2_sigmoid(d_m));
                        const bool okh0 = hl >= 0, okh1 = hl + 1 <= a.H - 1, okw0 = wl >= 0, okw1 = wl + 1 <= a.W - 1;
                        const bool ok[4] = {okh0 && okw0, okh0 && okw1, okh1 && okw0, okh1 && okw1};
                        const int pix[4] = {hl * a.W + wl, hl * a.W + wl + 1, (hl + 1) * a.W + wl, (hl + 1) * a.W + wl + 1};
                        typename X::frag v[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                        {
                            if constexpr (UP) {
                                v[k] = X::zero();
                                if (ok[k]) {
                                    const int yy = hl + (k >> 1), xx = wl + (k & 1);
                                    v[k].v = __builtin_bit_cast(decltype(v[k].v), dcn4_up8(a, a.xlo + (size_t)b * a.Hl * a.Wl * a.xlo_cs * 2,
                                                                                            a.skip + (size_t)b * a.H * a.W * a.skip_cs * 2, yy, xx, s * 16 + 8 * h));
                                }
                            } else {
                                v[k] = ok[k] ? X::lds(img + ((size_t)pix[k] * a.in_cs + s * 16 + 8 * h) * 2) : X::zero();   // fp16 input: plain 16-byte load
                            }
                        }
                        fb = X::blend(v, g);
                    }
                }
                if (!__any(any)) continue;
                typename X::frag fa[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[m] = X::lds(s_ring + aoff + m * C::WGRP + tap * 32);
#pragma unroll
                for (int m = 0; m < MT; ++m) X::mma(acc[m][0], fa[m], fb);
            }
        }
    }

    EpiArgs e;
    e.bias = a.bias; e.res = nullptr; e.out = a.out; e.Ho = a.H; e.Wo = a.W; e.Cout = a.Cout;
    e.out_cs = a.out_cs; e.res_cs = 0; e.relu = a.relu; e.out_mode = a.out_mode;
    if constexpr (EPI == 2) {
        __syncthreads();
        tile_epilogue_lds<bf16_t, MT>(acc, e, b, oy0, ox0, cout0, wv, l, smem + wv * epi_lds_stride<MT>());
    } else {
        tile_epilogue<bf16_t, MT, 1, EPI == 1>(acc, e, b, oy0, ox0, cout0, wv, r, h);
    }
}

template <int MT, int DENSE, int UP = 0>
static int launch_dcn4_cfg(const Dcn4Args &a0, hipStream_t st)
{
    using C = Dcn4Cfg<MT, DENSE>;
    static_assert(C::LDS <= (DENSE ? 80 : 160) * 1024, "LDS budget");
    static_assert(8 * epi_lds_stride<MT>() <= C::LDS, "epilogue staging");
    static_assert(!UP || C::RING >= 8 * 8 * 64 * 4, "tap table of the 4x up-sampling in the ring");
    Dcn4Args a = a0;
    a.tiles_x = cdiv(a.W, 16);
    a.tiles_y = cdiv(a.H, 16);
    dim3 grid(a.B * a.tiles_x * a.tiles_y, cdiv(cdiv(a.Cout, 32), MT));
    const bool lean = a.out_mode == H3D_OUT_NHWC && a.Cout % 4 == 0 && ((uintptr_t)a.bias & 15) == 0;
    const int epi = (MT >= 2 && lean && a.Cout % 8 == 0 && a.out_cs % 8 == 0 && ((uintptr_t)a.out & 15) == 0) ? 2 : lean ? 1 : 0;
    if (h3d_note_kernel("dcn4_kernel<%d, %d, %d, %d>", MT, epi, DENSE, UP)) return H3D_OK;
    if (epi == 2)
        hipLaunchKernelGGL((dcn4_kernel<MT, 2, DENSE, UP>), grid, dim3(512), 0, st, a);
    else if (epi == 1)
        hipLaunchKernelGGL((dcn4_kernel<MT, 1, DENSE, UP>), grid, dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((dcn4_kernel<MT, 0, DENSE, UP>), grid, dim3(512), 0, st, a);
    H3D_CHECK_LAUNCH("dcn4_kernel");
    return H3D_OK;
}

int h3d_launch_dcn4(const h3d_op &op, hipStream_t st)
{
    if (!op.in || !op.w || !op.bias || !op.out || !op.in2) H3D_FAIL(H3D_ERR_ARG, "dcn_fused_f16: null pointer");
    if (op.dtype != H3D_BF16) H3D_FAIL(H3D_ERR_DTYPE, "dcn_fused_f16: bf16 plans only (dtype %d)", op.dtype);
    if (op.ksize != 3 || op.stride != 1 || op.Ho != op.H || op.Wo != op.W)
        H3D_FAIL(H3D_ERR_UNSUPPORTED, "dcn_fused_f16: covers 3x3 s1 p1 d1 dg1 only (k=%d s=%d)", op.ksize, op.stride);
    if (op.Cin != 64 || op.in_cs % 8 || op.Cin > op.in_cs) H3D_FAIL(H3D_ERR_SHAPE, "dcn_fused_f16: Cin=%d (stride %d), 64 expected", op.Cin, op.in_cs);
    if (op.Cout > 64) H3D_FAIL(H3D_ERR_SHAPE, "dcn_fused_f16: Cout=%d > 64", op.Cout);
    if (op.H > 32767 || op.W > 32767 || (size_t)op.H * op.W * op.in_cs * 2 >= 0x7ffffff0ull)
        H3D_FAIL(H3D_ERR_SHAPE, "dcn_fused_f16: image too large");
    if (op.wrows % 128 || op.wrows < op.Cout) H3D_FAIL(H3D_ERR_SHAPE, "dcn_fused_f16: packed weight rows %d for Cout %d", op.wrows, op.Cout);
    if (op.out_mode != H3D_OUT_NCHW_F32 && (op.out_cs % 4 || op.Cout > op.out_cs))
        H3D_FAIL(H3D_ERR_SHAPE, "dcn_fused_f16: out channel stride %d", op.out_cs);
    Dcn4Args a;
    a.in = (const char *)op.in; a.wimg = (const char *)op.w; a.woff = (const char *)op.in2; a.bias = op.bias;
    a.out = (char *)op.out; a.B = op.B; a.H = op.H; a.W = op.W; a.in_cs = op.in_cs;
    a.Cout = op.Cout; a.out_cs = op.out_cs; a.relu = op.relu; a.out_mode = op.out_mode; a.wrows = op.wrows;
    a.G = op.wrows / 32; a.tiles_x = a.tiles_y = 0; a.dbg = op.reserved & 0xff;
    a.xlo = a.skip = nullptr; a.wup = nullptr; a.f = a.Hl = a.Wl = a.xlo_cs = a.skip_cs = 0;
    const bool dense = !(op.reserved & 0x100);    // tuning override (tools/ab_conv.py): 0x100 = one workgroup per CU
    if (op.Cout <= 32) return dense ? launch_dcn4_cfg<1, 1>(a, st) : launch_dcn4_cfg<1, 0>(a, st);
    return dense ? launch_dcn4_cfg<2, 1>(a, st) : launch_dcn4_cfg<2, 0>(a, st);
}

// H3D_OP_UPDCN_F16: IDAUp's  node(up(x) + skip)  in one launch (model.py:384-390): the depthwise transposed convolution and
// the skip add are evaluated while the apron is filled (dcn4_kernel UP = 1).  op.in = x (bf16, H x W = the LOW resolution),
// op.Ho x op.Wo = f * (H x W), op.stride = f, op.in2 = HOST pointer to h3d_updcn_desc, op.w / op.bias as DCN_FUSED_F16.
int h3d_launch_updcn(const h3d_op &op, hipStream_t st)
{
    if (!op.in || !op.w || !op.bias || !op.out || !op.in2) H3D_FAIL(H3D_ERR_ARG, "updcn: null pointer");
    const h3d_updcn_desc *d = (const h3d_updcn_desc *)op.in2;
    if (!d->skip || !d->w_up || !d->w_off) H3D_FAIL(H3D_ERR_ARG, "updcn: null pointer in the descriptor");
    if (op.dtype != H3D_BF16) H3D_FAIL(H3D_ERR_DTYPE, "updcn: bf16 plans only (dtype %d)", op.dtype);
    const int f = op.stride;
    if ((f != 2 && f != 4) || op.Ho != op.H * f || op.Wo != op.W * f)
        H3D_FAIL(H3D_ERR_UNSUPPORTED, "updcn: up-sampling factor %d (2 or 4), %dx%d -> %dx%d", f, op.H, op.W, op.Ho, op.Wo);
    if (op.Cin != 64 || op.in_cs % 8 || op.Cin > op.in_cs || d->skip_cs % 8 || d->skip_cs < 64)
        H3D_FAIL(H3D_ERR_SHAPE, "updcn: Cin=%d (strides %d, %d), 64 expected", op.Cin, op.in_cs, d->skip_cs);
    if (op.Cout > 64) H3D_FAIL(H3D_ERR_SHAPE, "updcn: Cout=%d > 64", op.Cout);
    if (op.Ho > 32767 || op.Wo > 32767) H3D_FAIL(H3D_ERR_SHAPE, "updcn: image too large");
    if (op.wrows % 128 || op.wrows < op.Cout) H3D_FAIL(H3D_ERR_SHAPE, "updcn: packed weight rows %d for Cout %d", op.wrows, op.Cout);
    if (op.out_mode != H3D_OUT_NCHW_F32 && (op.out_cs % 4 || op.Cout > op.out_cs))
        H3D_FAIL(H3D_ERR_SHAPE, "updcn: out channel stride %d", op.out_cs);
    Dcn4Args a;
    a.in = nullptr; a.wimg = (const char *)op.w; a.woff = (const char *)d->w_off; a.bias = op.bias;
    a.out = (char *)op.out; a.B = op.B; a.H = op.Ho; a.W = op.Wo; a.in_cs = 64;
    a.Cout = op.Cout; a.out_cs = op.out_cs; a.relu = op.relu; a.out_mode = op.out_mode; a.wrows = op.wrows;
    a.G = op.wrows / 32; a.tiles_x = a.tiles_y = 0; a.dbg = op.reserved & 0xff;
    a.xlo = (const char *)op.in; a.skip = (const char *)d->skip; a.wup = d->w_up;
    a.f = f; a.Hl = op.H; a.Wl = op.W; a.xlo_cs = op.in_cs; a.skip_cs = d->skip_cs;
    if (op.Cout <= 32) return launch_dcn4_cfg<1, 1, 1>(a, st);
    return launch_dcn4_cfg<2, 1, 1>(a, st);
}
