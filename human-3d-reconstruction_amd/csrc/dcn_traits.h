// Sample-element traits shared by the DCN kernels (dcn2.hip, dcn3.hip): type of the LDS apron tile,
// of the bilinear blend and of the MFMA operands (fp32 in parity mode, fp16 in bf16 mode).
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(8))) _Float16 half8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 half2_t;

template <typename T> struct SE;  // "sample element": type of the LDS halo, the blend and the MFMA operands

template <> struct SE<float> {
    using S = float;
    static constexpr int SS = 4;
    struct frag { f32x4 lo, hi; };
    static __device__ __forceinline__ frag lds(const char *p)
    {
        frag f;
        f.lo = *reinterpret_cast<const f32x4 *>(p);
        f.hi = *reinterpret_cast<const f32x4 *>(p + 16);
        return f;
    }
    static __device__ __forceinline__ frag zero() { frag f; f.lo = f32x4{0, 0, 0, 0}; f.hi = f.lo; return f; }
    static __device__ __forceinline__ void keep(const frag &f) { asm volatile("" ::"v"(f.lo), "v"(f.hi)); }
    static __device__ __forceinline__ frag global8(const char *p) { return lds(p); }   // 8 floats
    // filter fragments (`wfrag`, read by lds_w) and prepared sample fragments (`bfrag` = prep(frag)) are the sample type itself
    // here and in the 2-byte plans; f16x3 plans (SE<x3_t> below) split them into fp16 terms
    using wfrag = frag;
    using bfrag = frag;
    static __device__ __forceinline__ wfrag lds_w(const char *p) { return lds(p); }
    static __device__ __forceinline__ bfrag prep(const frag &f) { return f; }
    static __device__ __forceinline__ bfrag prep_raw(const frag &f) { return f; }
    // apron fragment of the OFFSET convolution (phase A of csrc/dcn3.hip), ready for the MFMA
    static __device__ __forceinline__ bfrag lds_a(const char *p) { return lds(p); }
    static constexpr bool SPLIT_A = false;
    static __device__ __forceinline__ void mma(f32x16 &acc, const frag &a, const frag &b)
    {
        ET<float>::frag fa, fb;
        fa.lo = a.lo; fa.hi = a.hi; fb.lo = b.lo; fb.hi = b.hi;
        ET<float>::mma(acc, fa, fb);
    }
    struct geo { float w[4]; float mask; };     // per (pixel, tap) blend coefficients kept in registers
    static __device__ __forceinline__ geo make_geo(const float (&w)[4], float mask)
    {
        geo g;
        g.w[0] = w[0]; g.w[1] = w[1]; g.w[2] = w[2]; g.w[3] = w[3]; g.mask = mask;
        return g;
    }
    static __device__ __forceinline__ geo zero_geo() { geo g; g.w[0] = g.w[1] = g.w[2] = g.w[3] = 0.f; g.mask = 0.f; return g; }
    static __device__ __forceinline__ geo select_geo(bool keep, const geo &g)      // keep ? g : zero_geo(), as selects
    {
        geo o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o.w[i] = keep ? g.w[i] : 0.f;
        o.mask = keep ? g.mask : 0.f;
        return o;
    }
    static __device__ __forceinline__ geo shfl_xor32(const geo &g)
    {
        geo o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o.w[i] = __shfl_xor(g.w[i], 32);
        o.mask = __shfl_xor(g.mask, 32);
        return o;
    }
    static __device__ __forceinline__ frag blend(const frag (&v)[4], const geo &g) { return blend(v, g.w, g.mask); }
    // reference order: (w1*v1 + w2*v2 + w3*v3 + w4*v4) * mask
    static __device__ __forceinline__ frag blend(const frag (&v)[4], const float (&w)[4], float mask)
    {
        frag o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o.lo[e] = (w[0] * v[0].lo[e] + w[1] * v[1].lo[e] + w[2] * v[2].lo[e] + w[3] * v[3].lo[e]) * mask;
            o.hi[e] = (w[0] * v[0].hi[e] + w[1] * v[1].hi[e] + w[2] * v[2].hi[e] + w[3] * v[3].hi[e]) * mask;
        }
        return o;
    }
    // staging: 16 bytes of T=float -> 16 bytes of S
    static __device__ __forceinline__ u32x4 convert16(u32x4 raw) { return raw; }
};

template <> struct SE<bf16_t> {
    using S = _Float16;
    static constexpr int SS = 2;
    struct frag { half8_t v; };
    static __device__ __forceinline__ frag lds(const char *p)
    {
        frag f;
        f.v = *reinterpret_cast<const half8_t *>(p);
        return f;
    }
    static __device__ __forceinline__ frag zero()
    {
        frag f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f.v[e] = (_Float16)0.f;
        return f;
    }
    static __device__ __forceinline__ void keep(const frag &f) { asm volatile("" ::"v"(f.v)); }
    using wfrag = frag;
    using bfrag = frag;
    static __device__ __forceinline__ wfrag lds_w(const char *p) { return lds(p); }
    static __device__ __forceinline__ bfrag prep(const frag &f) { return f; }
    static __device__ __forceinline__ bfrag prep_raw(const frag &f) { return f; }
    // apron fragment of the OFFSET convolution (phase A of csrc/dcn3.hip), ready for the MFMA
    static __device__ __forceinline__ bfrag lds_a(const char *p) { return lds(p); }
    static constexpr bool SPLIT_A = false;
    static __device__ __forceinline__ _Float16 cvt(uint32_t bits_hi)   // bf16 in the high half of an f32 pattern
    {
        const float x = __uint_as_float(bits_hi);
        return (_Float16)__builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);  // v_med3_f32: no inf from bf16's wider range
    }
    static __device__ __forceinline__ u32x4 convert16(u32x4 raw)        // 8 bf16 -> 8 fp16
    {
        // v_cvt_pkrtz_f16_f32: three instructions per pair (shift, and, convert + pack).  A bf16 value inside fp16's normal range
        // converts exactly under any rounding (8 significand bits into 11); round-toward-zero differs from nearest-even only
        // (a) beyond +-65504, where it SATURATES to the largest finite fp16 instead of producing an infinity -- the clamp this
        // conversion used to spend a v_med3_f32 per element on -- and (b) below 2^-14 (fp16 subnormals: at most 6e-8 off).
        // CONTRACT: activations are finite.  +-inf / NaN pass through unchanged (cvt() below would clamp an inf), and the
        // branch-free blend multiplies zero weights with whatever the corner read returns, so a non-finite activation can leak
        // NaN into samples that should contribute 0.  Every producer of a 2-byte plan saturates (bf16 epilogues cannot make inf
        // from finite fp32 sums below 3.4e38; fp16 epilogues clamp), so only a non-finite INPUT image gets here.
        typedef __fp16 pk2_t __attribute__((ext_vector_type(2)));
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const pk2_t q = __builtin_amdgcn_cvt_pkrtz(__uint_as_float(raw[e] << 16), __uint_as_float(raw[e] & 0xffff0000u));
            o[e] = __builtin_bit_cast(uint32_t, q);
        }
        return o;
    }
    static __device__ __forceinline__ frag global8(const char *p)       // 8 bf16 from global -> fp16
    {
        frag f;
        f.v = __builtin_bit_cast(half8_t, convert16(*reinterpret_cast<const u32x4 *>(p)));
        return f;
    }
    static __device__ __forceinline__ void mma(f32x16 &acc, const frag &a, const frag &b)
    {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, b.v, acc, 0, 0, 0);
    }
    struct geo { uint32_t w01, w23; };          // fp16 pairs (w0 | w1 << 16), (w2 | w3 << 16); mask folded in
    static __device__ __forceinline__ geo make_geo(const float (&w)[4], float mask)
    {
        geo g;
        g.w01 = __builtin_bit_cast(uint32_t, half2_t{(_Float16)(w[0] * mask), (_Float16)(w[1] * mask)});
        g.w23 = __builtin_bit_cast(uint32_t, half2_t{(_Float16)(w[2] * mask), (_Float16)(w[3] * mask)});
        return g;
    }
    static __device__ __forceinline__ geo zero_geo() { geo g; g.w01 = 0u; g.w23 = 0u; return g; }
    static __device__ __forceinline__ geo select_geo(bool keep, const geo &g) { geo o; o.w01 = keep ? g.w01 : 0u; o.w23 = keep ? g.w23 : 0u; return o; }
    static __device__ __forceinline__ geo shfl_xor32(const geo &g)
    {
        geo o;
        o.w01 = h3d_xor32(g.w01);
        o.w23 = h3d_xor32(g.w23);
        return o;
    }
    // out = v0*w0 + v1*w1 + v2*w2 + v3*w3 on 8 fp16 channels: 4 packed ops per dword, the per-pixel
    // weight is broadcast to both halves by op_sel (no duplicated weight registers).  The trailing
    // s_nop covers the VALU-write -> MFMA-operand wait states hipcc does not pad inside asm.
    static __device__ __forceinline__ frag blend(const frag (&v)[4], const geo &g)
    {
        const u32x4 a = __builtin_bit_cast(u32x4, v[0].v), b = __builtin_bit_cast(u32x4, v[1].v),
                    c = __builtin_bit_cast(u32x4, v[2].v), d = __builtin_bit_cast(u32x4, v[3].v);
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t x;
            asm("v_pk_mul_f16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(x) : "v"(a[i]), "v"(g.w01));
            asm("v_pk_fma_f16 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(x) : "v"(b[i]), "v"(g.w01));
            asm("v_pk_fma_f16 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(x) : "v"(c[i]), "v"(g.w23));
            asm("v_pk_fma_f16 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(x) : "v"(d[i]), "v"(g.w23));
            o[i] = x;
        }
        asm volatile("s_nop 1" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
        frag f;
        f.v = __builtin_bit_cast(half8_t, u32x4{o[0], o[1], o[2], o[3]});
        return f;
    }
};

// fp16 plans: the input already IS the sample type -- no conversion while staging, the blend and the MFMA operands as in
// bf16 plans
template <> struct SE<f16_t> : SE<bf16_t> {
    static __device__ __forceinline__ u32x4 convert16(u32x4 raw) { return raw; }
    static __device__ __forceinline__ frag global8(const char *p) { return lds(p); }
};

// f16x3 plans (common.h ET<x3_t>): the apron in LDS, the sampling geometry and the bilinear blend are the fp32 ones of SE<float>
// (reference operation order, dcn_v2_im2col_cuda.cu:25-54); the filters arrive pre-split into (hi | lo) fp16 terms, the blended
// sample (and, for the offset convolution, the apron fragment) is split in registers, and each fp32 product becomes three fp16 MFMAs
template <> struct SE<x3_t> : SE<float> {
    using wfrag = ET<x3_t>::frag;
    using bfrag = ET<x3_t>::frag;
    static __device__ __forceinline__ wfrag lds_w(const char *p) { return ET<x3_t>::lds_frag(p); }
    // prep: a sample blended from values that went through convert16 (clamped to the fp16 range while staging: a convex combination
    // times a mask in (0, 1) stays inside it), so the split needs no clamp of its own -- 8 of its 28 vector instructions, nine times per
    // staged element.  prep_raw: a sample blended from values read straight from memory (pass 2).
    static __device__ __forceinline__ bfrag prep(const frag &f)
    {
        const float x[8] = {f.lo[0], f.lo[1], f.lo[2], f.lo[3], f.hi[0], f.hi[1], f.hi[2], f.hi[3]};
        return ET<x3_t>::split8<false>(x);
    }
    static __device__ __forceinline__ bfrag prep_raw(const frag &f)
    {
        const float x[8] = {f.lo[0], f.lo[1], f.lo[2], f.lo[3], f.hi[0], f.hi[1], f.hi[2], f.hi[3]};
        return ET<x3_t>::split8<true>(x);
    }
    static __device__ __forceinline__ u32x4 convert16(u32x4 raw)
    {
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __float_as_uint(__builtin_amdgcn_fmed3f(__uint_as_float(raw[i]), -65504.f, 65504.f));
        return o;
    }
    // phase A only multiplies the apron (no blend), so the thread that stages a phase-A chunk stores it already split (x3_store4:
    // once per element) and the nine taps read finished operand fragments; phase B stages the same chunk again as plain fp32
    static __device__ __forceinline__ bfrag lds_a(const char *p) { return ET<x3_t>::lds_frag(p); }
    static constexpr bool SPLIT_A = true;
    // blend: the mask folded into the four bilinear weights (as the 2-byte plans do), then sum_k w_k v_k on PAIRS of channels
    // (v_pk_mul_f32 / v_pk_fma_f32: 16 packed instructions per 8-channel fragment instead of 40 scalar ones).  Differs from the
    // reference's (w1 v1 + w2 v2 + w3 v3 + w4 v4) * mask by fp32 rounding order only.
    struct geo { float w[4]; };
    static __device__ __forceinline__ geo make_geo(const float (&w)[4], float mask)
    {
        geo g;
#pragma unroll
        for (int i = 0; i < 4; ++i) g.w[i] = w[i] * mask;
        return g;
    }
    static __device__ __forceinline__ geo zero_geo() { geo g; g.w[0] = g.w[1] = g.w[2] = g.w[3] = 0.f; return g; }
    static __device__ __forceinline__ geo select_geo(bool keep_, const geo &g)
    {
        geo o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o.w[i] = keep_ ? g.w[i] : 0.f;
        return o;
    }
    static __device__ __forceinline__ geo shfl_xor32(const geo &g)
    {
        geo o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o.w[i] = __uint_as_float(h3d_xor32(__float_as_uint(g.w[i])));
        return o;
    }
    static __device__ __forceinline__ frag blend(const frag (&v)[4], const geo &g)
    {
        frag o;
        const f32x2 w0 = {g.w[0], g.w[0]}, w1 = {g.w[1], g.w[1]}, w2 = {g.w[2], g.w[2]}, w3 = {g.w[3], g.w[3]};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const f32x2 a0 = {v[0].lo[2 * e], v[0].lo[2 * e + 1]}, a1 = {v[1].lo[2 * e], v[1].lo[2 * e + 1]};
            const f32x2 a2 = {v[2].lo[2 * e], v[2].lo[2 * e + 1]}, a3 = {v[3].lo[2 * e], v[3].lo[2 * e + 1]};
            const f32x2 x = __builtin_elementwise_fma(a3, w3, __builtin_elementwise_fma(a2, w2, __builtin_elementwise_fma(a1, w1, a0 * w0)));
            o.lo[2 * e] = x[0]; o.lo[2 * e + 1] = x[1];
            const f32x2 b0 = {v[0].hi[2 * e], v[0].hi[2 * e + 1]}, b1 = {v[1].hi[2 * e], v[1].hi[2 * e + 1]};
            const f32x2 b2 = {v[2].hi[2 * e], v[2].hi[2 * e + 1]}, b3 = {v[3].hi[2 * e], v[3].hi[2 * e + 1]};
            const f32x2 y = __builtin_elementwise_fma(b3, w3, __builtin_elementwise_fma(b2, w2, __builtin_elementwise_fma(b1, w1, b0 * w0)));
            o.hi[2 * e] = y[0]; o.hi[2 * e + 1] = y[1];
        }
        return o;
    }
    using SE<float>::keep;
    static __device__ __forceinline__ void keep(const wfrag &f) { asm volatile("" ::"v"(f.hi), "v"(f.lo)); }
    static __device__ __forceinline__ void mma(f32x16 &acc, const wfrag &a, const bfrag &b) { ET<x3_t>::mma(acc, a, b); }
};

__device__ __forceinline__ float dcn2_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }   // v_exp + v_rcp (1 ulp each)
