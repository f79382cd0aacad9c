// Shared device/host helpers for libh3d_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "../../include/h3d.h"

#include <type_traits>
typedef uint16_t bf16_t;  // raw bf16 bits
struct f16_t { uint16_t bits; };   // raw IEEE fp16 bits: a distinct type, so that templates can tell the two 2-byte element types apart
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
// element-type name as it appears in the kernel symbols rocprofv3 prints
struct x3_t;                       // fp32 storage, fp16 x 3 contraction (below)
template <typename T> constexpr const char *h3d_tname() { return std::is_same_v<T, float> ? "float" : std::is_same_v<T, f16_t> ? "f16_t" : std::is_same_v<T, x3_t> ? "x3_t" : "unsigned short"; }
// host side: element size of an h3d_op dtype (0 = unknown)
static inline int h3d_dtype_bytes(int dtype) { return (dtype == H3D_F32 || dtype == H3D_F16X3) ? 4 : (dtype == H3D_BF16 || dtype == H3D_F16) ? 2 : 0; }
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// ---- error plumbing -------------------------------------------------------------------------
void h3d_set_error(const char *fmt, ...);
// Records the kernel instantiation an op maps to; returns true when the caller must NOT launch
// (h3d_op_kernel_name's dry run).  Names match the kernel symbols rocprofv3 reports.
bool h3d_note_kernel(const char *fmt, ...);

// Profiling ablation switches (h3d_op.reserved) exist only in `make ABLATE=1` builds: as run-time
// tests they put every MFMA in its own basic block and wreck the schedule of the production kernel.
#ifdef H3D_ABLATE
#define H3D_DBG(a) ((a).dbg)
// in-kernel phase stamps (profiling builds only): wave 0 of workgroup `wg` stores s_memtime at point `k`
#define H3D_NSTAMP 8
unsigned long long *h3d_stamp_buffer();     // device buffer [65536][H3D_NSTAMP] (ops.hip); kernels take it in their args
#define H3D_STAMP(wg, k)                                                                                   \
    do {                                                                                                   \
        if (threadIdx.x == 0 && (wg) < 65536 && a.stamps) a.stamps[(wg) * H3D_NSTAMP + (k)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define H3D_DBG(a) 0
#define H3D_STAMP(wg, k) do { } while (0)
#endif
#define H3D_FAIL(code, ...)        \
    do {                           \
        h3d_set_error(__VA_ARGS__); \
        return (code);             \
    } while (0)
#define H3D_CHECK_LAUNCH(name)                                                          \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) H3D_FAIL(H3D_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

// ---- element traits: the same kernel source runs in f32 ("parity mode", exact fmaf chain on
//      v_mfma_f32_32x32x2_f32) and bf16 (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//      A fragment = 8 consecutive K elements of one row/column per lane:
//        lane l: row/col = l & 31, K offset = 8 * (l >> 5)   (both operands use the same map,
//        so for f32 the 8 k=2 MFMAs simply walk the 8 elements; K order is permuted
//        identically on A and B).
template <typename T> struct ET;

template <> struct ET<float> {
    static constexpr int BYTES = 4;
    struct frag { f32x4 lo, hi; };
    static __device__ __forceinline__ frag lds_frag(const char *p)
    {
        frag f;
        f.lo = *reinterpret_cast<const f32x4 *>(p);
        f.hi = *reinterpret_cast<const f32x4 *>(p + 16);
        return f;
    }
    // 8 K elements whose two 16-byte halves sit at different (swizzled) LDS addresses
    static __device__ __forceinline__ frag lds_frag2(const char *lo, const char *hi)
    {
        frag f;
        f.lo = *reinterpret_cast<const f32x4 *>(lo);
        f.hi = *reinterpret_cast<const f32x4 *>(hi);
        return f;
    }
    static __device__ __forceinline__ void mma(f32x16 &acc, const frag &a, const frag &b)
    {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[0], b.lo[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[1], b.lo[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[2], b.lo[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[3], b.lo[3], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[0], b.hi[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[1], b.hi[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[2], b.hi[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[3], b.hi[3], acc, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f32(float v) { return v; }
    static __device__ __forceinline__ float from_f32(float v) { return v; }
};

template <> struct ET<bf16_t> {
    static constexpr int BYTES = 2;
    struct frag { u32x4 v; };
    static __device__ __forceinline__ frag lds_frag(const char *p)
    {
        frag f;
        f.v = *reinterpret_cast<const u32x4 *>(p);
        return f;
    }
    static __device__ __forceinline__ frag lds_frag2(const char *lo, const char *) { return lds_frag(lo); }
    static __device__ __forceinline__ void mma(f32x16 &acc, const frag &a, const frag &b)
    {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a.v),
                                                      __builtin_bit_cast(bf16x8_t, b.v), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f32(bf16_t v)
    {
        return __uint_as_float(((uint32_t)v) << 16);
    }
    static __device__ __forceinline__ bf16_t from_f32(float f)
    {
        __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
        return __builtin_bit_cast(uint16_t, b);
    }
};

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi)
{
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{lo, hi}, b2));   // one v_cvt_pk_bf16_f32 (RNE)
}
// two floats -> packed fp16 pair, round-to-nearest-even; callers clamp to +-65504 first (ET<f16_t>::clamp_lo)
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi)
{
    return __builtin_bit_cast(uint32_t, f16x2_t{(_Float16)lo, (_Float16)hi});
}

// fp16 plans (H3D_F16: BASELINE configs[4] runs ResNet-101-DCN in fp16, experiments/ctdet_coco_resdcn101.sh:3): same
// kernels, v_mfma_f32_32x32x16_f16 (the bf16 rate), 3 more mantissa bits per stored activation; results saturate at
// +-65504 instead of overflowing to infinity.
template <> struct ET<f16_t> {
    static constexpr int BYTES = 2;
    struct frag { u32x4 v; };
    static __device__ __forceinline__ frag lds_frag(const char *p)
    {
        frag f;
        f.v = *reinterpret_cast<const u32x4 *>(p);
        return f;
    }
    static __device__ __forceinline__ frag lds_frag2(const char *lo, const char *) { return lds_frag(lo); }
    static __device__ __forceinline__ void mma(f32x16 &acc, const frag &a, const frag &b)
    {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a.v), __builtin_bit_cast(f16x8_t, b.v), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f32(f16_t v) { return (float)__builtin_bit_cast(_Float16, v.bits); }
    static __device__ __forceinline__ f16_t from_f32(float f)
    {
        f16_t o;
        o.bits = __builtin_bit_cast(uint16_t, (_Float16)__builtin_amdgcn_fmed3f(f, -65504.f, 65504.f));
        return o;
    }
};

// "f16x3" plans (H3D_F16X3, round 5): fp32 STORAGE everywhere (activations, residuals, head maps: the layout, the launches and the
// elementwise kernels of the f32 plan), but every contraction runs on the fp16 matrix cores with both fp32 operands split into two
// fp16 terms, x = hi + lo with hi = fp16(x) and lo = fp16(x - hi) (round to nearest even: |x - hi - lo| <= 2^-24 |x| while lo is a
// normal fp16, <= 3e-8 absolute below that), and THREE products per fp32 product -- hi.hi + hi.lo + lo.hi, the dropped lo.lo is
// 2^-22 relative -- accumulated in fp32 by the MFMA: 3 x v_mfma_f32_32x32x16_f16 (96 cycles) per 16-channel step instead of
// 8 x v_mfma_f32_32x32x2_f32 (512 cycles).  This is the parity arithmetic on the 2.5 PFLOP/s matrix cores: the reference computes
// in fp32 throughout (dcn_v2_cuda.cu:59 `using scalar_t = float`; model.py modules are fp32), and "bit-exact top-k indices" only
// survives fp32-level head errors.  A fragment is the SAME 32 bytes per lane as ET<float>'s (8 K elements), holding the 8 hi halves
// in the first 16 bytes and the 8 lo halves in the second: the LDS geometry of every kernel is the f32 one, filters are pre-split
// by the host (engine.PackedWeights), activations are split by the thread that stages them into LDS.
// CONTRACT: |activation| <= 65504 (fp16's range; larger values saturate, finite) -- three orders of magnitude above anything a
// DLA-34 / Hourglass / ResNet feature map holds.
struct x3_t { float v; };          // element type tag: 4 bytes of fp32 in memory
// 4 fp32 (16 raw bytes) -> their 4 hi and 4 lo fp16 terms
// (CLAMP = false: the caller knows |x| <= 65504 already -- a DeformConv sample blended from clamped apron values)
template <bool CLAMP = true>
__device__ __forceinline__ void x3_split4(const u32x4 raw, u32x2 &hi, u32x2 &lo)
{
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const float a = CLAMP ? __builtin_amdgcn_fmed3f(__uint_as_float(raw[2 * p]), -65504.f, 65504.f) : __uint_as_float(raw[2 * p]);
        const float b = CLAMP ? __builtin_amdgcn_fmed3f(__uint_as_float(raw[2 * p + 1]), -65504.f, 65504.f) : __uint_as_float(raw[2 * p + 1]);
        const _Float16 ha = (_Float16)a, hb = (_Float16)b;
        const _Float16 la = (_Float16)(a - (float)ha), lb = (_Float16)(b - (float)hb);
        hi[p] = __builtin_bit_cast(uint32_t, f16x2_t{ha, hb});
        lo[p] = __builtin_bit_cast(uint32_t, f16x2_t{la, lb});
    }
}
// the staging store of one 16-byte vector (4 fp32) of a 32-byte K group at LDS address `grp`: sub = 0 | 1 = which half of the group's
// 8 elements the vector holds.  hi terms land in bytes [8 sub, 8 sub + 8), lo terms 16 bytes further on.
__device__ __forceinline__ void x3_store4(char *grp, int sub, const u32x4 raw)
{
    u32x2 hi, lo;
    x3_split4(raw, hi, lo);
    *reinterpret_cast<u32x2 *>(grp + 8 * sub) = hi;
    *reinterpret_cast<u32x2 *>(grp + 16 + 8 * sub) = lo;
}
template <> struct ET<x3_t> {
    static constexpr int BYTES = 4;
    struct frag { u32x4 hi, lo; };
    static __device__ __forceinline__ frag lds_frag(const char *p)
    {
        frag f;
        f.hi = *reinterpret_cast<const u32x4 *>(p);
        f.lo = *reinterpret_cast<const u32x4 *>(p + 16);
        return f;
    }
    static __device__ __forceinline__ frag lds_frag2(const char *first, const char *second)
    {
        frag f;
        f.hi = *reinterpret_cast<const u32x4 *>(first);
        f.lo = *reinterpret_cast<const u32x4 *>(second);
        return f;
    }
    // 8 fp32 values held in registers (a blended DeformConv sample, a ReLU-ed accumulator slab) -> operand fragment
    template <bool CLAMP = true>
    static __device__ __forceinline__ frag split8(const float (&x)[8])
    {
        frag f;
        u32x2 h0, l0, h1, l1;
        x3_split4<CLAMP>(u32x4{__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])}, h0, l0);
        x3_split4<CLAMP>(u32x4{__float_as_uint(x[4]), __float_as_uint(x[5]), __float_as_uint(x[6]), __float_as_uint(x[7])}, h1, l1);
        f.hi = u32x4{h0[0], h0[1], h1[0], h1[1]};
        f.lo = u32x4{l0[0], l0[1], l1[0], l1[1]};
        return f;
    }
    static __device__ __forceinline__ void mma(f32x16 &acc, const frag &a, const frag &b)
    {
        // small terms first: they meet an accumulator that is not yet large
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a.lo), __builtin_bit_cast(f16x8_t, b.hi), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a.hi), __builtin_bit_cast(f16x8_t, b.lo), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a.hi), __builtin_bit_cast(f16x8_t, b.hi), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f32(x3_t v) { return v.v; }
    static __device__ __forceinline__ x3_t from_f32(float f) { x3_t o; o.v = f; return o; }
};
// storage type of a plan's activations: the element type itself, fp32 for f16x3 plans
template <typename T> struct StoreT { using type = T; };
template <> struct StoreT<x3_t> { using type = float; };

// Per-type pieces of the epilogues: two packed elements <-> floats, and the store clamp.  clamp(v, lo): lo = 0 (ReLU) or
// -inf; fp16 also bounds the value by +-65504 in the same v_med3_f32 (no extra instruction).
template <typename T> struct EP;
template <> struct EP<bf16_t> {
    static __device__ __forceinline__ uint32_t pack2(float a, float b) { return pack_bf16x2(a, b); }
    static __device__ __forceinline__ void unpack2(uint32_t u, float &a, float &b) { a = __uint_as_float(u << 16); b = __uint_as_float(u & 0xffff0000u); }
    static __device__ __forceinline__ float clamp(float v, float lo) { return fmaxf(v, lo); }
};
template <> struct EP<f16_t> {
    static __device__ __forceinline__ uint32_t pack2(float a, float b) { return pack_f16x2(a, b); }
    static __device__ __forceinline__ void unpack2(uint32_t u, float &a, float &b)
    {
        const f16x2_t h = __builtin_bit_cast(f16x2_t, u);
        a = (float)h[0]; b = (float)h[1];
    }
    static __device__ __forceinline__ float clamp(float v, float lo) { return __builtin_amdgcn_fmed3f(v, fmaxf(lo, -65504.f), 65504.f); }
};
template <> struct EP<float> {
    static __device__ __forceinline__ float clamp(float v, float lo) { return fmaxf(v, lo); }
};

// store 4 consecutive channels held as floats
template <typename T> __device__ __forceinline__ void store4(T *p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4<float>(float *p, float a, float b, float c, float d)
{
    f32x4 v = {a, b, c, d};
    *reinterpret_cast<f32x4 *>(p) = v;
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t *p, float a, float b, float c, float d)
{
    u32x2 v = {pack_bf16x2(a, b), pack_bf16x2(c, d)};
    *reinterpret_cast<u32x2 *>(p) = v;
}
template <> __device__ __forceinline__ void store4<f16_t>(f16_t *p, float a, float b, float c, float d)
{
    u32x2 v = {pack_f16x2(__builtin_amdgcn_fmed3f(a, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f)),
               pack_f16x2(__builtin_amdgcn_fmed3f(c, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(d, -65504.f, 65504.f))};
    *reinterpret_cast<u32x2 *>(p) = v;
}
template <typename T> __device__ __forceinline__ void load4(const T *p, float *o);
template <> __device__ __forceinline__ void load4<float>(const float *p, float *o)
{
    f32x4 v = *reinterpret_cast<const f32x4 *>(p);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t *p, float *o)
{
    u32x2 v = *reinterpret_cast<const u32x2 *>(p);
    o[0] = __uint_as_float(v[0] << 16);
    o[1] = __uint_as_float(v[0] & 0xffff0000u);
    o[2] = __uint_as_float(v[1] << 16);
    o[3] = __uint_as_float(v[1] & 0xffff0000u);
}

template <> __device__ __forceinline__ void load4<f16_t>(const f16_t *p, float *o)
{
    u32x2 v = *reinterpret_cast<const u32x2 *>(p);
    EP<f16_t>::unpack2(v[0], o[0], o[1]);
    EP<f16_t>::unpack2(v[1], o[2], o[3]);
}

// Cooperative global -> LDS staging of TOTAL 16-byte vectors by NTHREADS threads: ALL loads are
// issued before the first LDS store, so the vector-memory latency is paid once per stage instead
// of once per vector (a plain `for (i = tid; ...) lds[i] = g[i]` loop serialises on s_waitcnt).
template <int TOTAL, int NTHREADS, int MAXB = 16, typename LD, typename ST>
__device__ __forceinline__ void stage_vectors(int tid, LD ld, ST st)
{
    constexpr int N = (TOTAL + NTHREADS - 1) / NTHREADS;   // vectors per thread
    constexpr int NB = (N + MAXB - 1) / MAXB;              // batches (bounds the staging registers)
    constexpr int PER = (N + NB - 1) / NB;
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) {
        u32x4 tmp[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int i = tid + (bt * PER + j) * NTHREADS;
            tmp[j] = u32x4{0u, 0u, 0u, 0u};
            if (bt * PER + j < N && i < TOTAL) tmp[j] = ld(i);
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int i = tid + (bt * PER + j) * NTHREADS;
            if (bt * PER + j < N && i < TOTAL) st(i, tmp[j]);
        }
    }
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Workgroup barrier that leaves vmcnt alone.  __syncthreads() is release fence + s_barrier + acquire fence, and the release
// waits for EVERY outstanding LDS-DMA of the wave (hipcc emits s_waitcnt vmcnt(0) in front of the s_barrier: a DMA is an
// LDS write), which collapses a DMA ring of three or more slots to one stage of distance whatever the counted s_waitcnt
// vmcnt(N) in front of it says.  Here the caller states its own vector-memory wait; lgkmcnt(0) retires this wave's LDS
// reads and writes, and the empty asm statements keep the compiler from moving memory operations across the barrier.
__device__ __forceinline__ void h3d_barrier_keep_vmcnt()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// XCD-aware tile order.  Workgroup ids are dealt round robin to the 8 XCDs (id % 8), each with its own 4 MiB L2, so
// with the plain order the four neighbours of a tile -- which share its halo / apron -- sit on four other dies and the
// shared pixels are fetched into several L2s.  With mode 1 die x works on the contiguous id range [x * nb/8, (x+1) * nb/8).
// (mode from the launcher: env H3D_XCD, read once; only when nb is a multiple of 8)
int h3d_xcd_mode();
__device__ __forceinline__ int h3d_tile_id(int bid, int nb, int mode)
{
    if (mode == 0 || (nb & 7)) return bid;
    return (bid & 7) * (nb >> 3) + (bid >> 3);
}

// 16-byte NHWC channel vectors <-> fp32 (elementwise kernels, csrc/dcn4.hip's fused up-sample prologue)
template <typename T>
__device__ __forceinline__ void unpack16(const u32x4 &v, float *o)
{
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __uint_as_float(v[i]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) EP<T>::unpack2(v[i], o[2 * i], o[2 * i + 1]);
    }
}
template <typename T>
__device__ __forceinline__ u32x4 pack16(const float *o)
{
    u32x4 v;
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __float_as_uint(o[i]);
    } else if constexpr (std::is_same_v<T, f16_t>) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            v[i] = pack_f16x2(__builtin_amdgcn_fmed3f(o[2 * i], -65504.f, 65504.f), __builtin_amdgcn_fmed3f(o[2 * i + 1], -65504.f, 65504.f));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = pack_bf16x2(o[2 * i], o[2 * i + 1]);
    }
    return v;
}

// 8 floats -> 8 fp16 (clamped to the fp16 range): the input format of csrc/dcn4.hip
__device__ __forceinline__ u32x4 pack16_f16(const float *o)
{
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        v[i] = __builtin_bit_cast(uint32_t, h2{(_Float16)__builtin_amdgcn_fmed3f(o[2 * i], -65504.f, 65504.f),
                                               (_Float16)__builtin_amdgcn_fmed3f(o[2 * i + 1], -65504.f, 65504.f)});
    return v;
}


// value of lane (l ^ 32): gfx950's v_permlane32_swap exchanges the upper half of one register with the lower half of another in the
// vector ALU (one instruction + a select) -- `__shfl_xor(x, 32)` compiles to a ds_bpermute_b32, an LDS-crossbar round trip
__device__ __forceinline__ uint32_t h3d_xor32(uint32_t x)
{
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);      // r[0]: upper lanes hold the lower lanes' x; r[1]: lower lanes hold the upper lanes' x
    return (threadIdx.x & 32) ? r[0] : r[1];
}
