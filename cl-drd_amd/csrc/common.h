// Shared device helpers for the cldrd gfx950 (CDNA4 / MI355X) kernels.  gfx950 only: 64-lane
// wavefronts, MFMA 16x16x32 / 32x32x16 bf16, LDS-DMA (global_load_lds), ds_read_b64_tr_b16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;   // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;    // MFMA A/B fragment (8 bf16 = 4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
// Streaming accesses for data touched once in a long while (the backward tape: written by the forward, read ~10 ms later): the
// non-temporal hint keeps them from evicting what the NEXT kernel reads out of the 256-MB Infinity Cache.  CLDRD_TAPE_NT=0: A/B builds.
#ifndef CLDRD_TAPE_NT
#define CLDRD_TAPE_NT 1
#endif
__device__ __forceinline__ uint4 ld16_stream(const void* p) {
#if CLDRD_TAPE_NT
    const u32x4 t = __builtin_nontemporal_load((const u32x4*)p);
    return make_uint4(t.x, t.y, t.z, t.w);
#else
    return *(const uint4*)p;
#endif
}
__device__ __forceinline__ void st16_stream(void* p, const uint4& v) {
#if CLDRD_TAPE_NT
    const u32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, (u32x4*)p);
#else
    *(uint4*)p = v;
#endif
}

#define CLDRD_WAVE 64

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// LDS-DMA of 16 bytes per lane with a SCALAR base and a 32-bit per-lane byte offset (the SADDR form of global_load_lds_dwordx4): lane l's
// 16 bytes at sbase + voff land at LDS byte address lds_addr + 16 l.  __builtin_amdgcn_global_load_lds takes a 64-bit per-lane pointer and hipcc
// builds it with one v_lshl_add_u64 per piece (seven to eight per wave and K tile in the GEMM loops, in front of instructions whose issue is
// already the K loop's largest non-MFMA cost); here the address arithmetic stays on the scalar unit.  `lds_addr` must be wave-uniform.
__device__ __forceinline__ void lds_dma16(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory", "m0");
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even f32 -> bf16 via the hardware cast (keeps NaN a NaN, MI355X_MICROARCH correctness table)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}

// two floats -> one dword of bf16 (lo in bits 0..15): ONE v_cvt_pk_bf16_f32.  (f2bf(lo) | f2bf(hi) << 16 compiled to two converts, a
// shift and an or: 4 VALU per pair in epilogues that are VALU-bound.)
typedef __bf16 cldrd_bf16v2 __attribute__((ext_vector_type(2)));
typedef float cldrd_f32v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const cldrd_f32v2 f = {lo, hi};
    const cldrd_bf16v2 b = __builtin_convertvector(f, cldrd_bf16v2);
    return __builtin_bit_cast(uint32_t, b);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf-GELU and its derivative (HF GELUActivation == F.gelu).  erf by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, far below the 16-bit output rounding): one v_rcp + one v_exp + FMAs instead of libm's erff
// (~40 VALU ops), which made the GELU epilogues VALU-bound (100 M activations per FFN GEMM).
// Round 5: the same polynomial in the erfc form, which is what GELU needs - with z = |x| / sqrt 2, t = 1 / (1 + p z):
//     q = Phi(-|x|) = erfc(z) / 2 = (t (a1/2 + t (a2/2 + ...))) exp(-z^2),     gelu(x) = x Phi(x) = max(x, 0) - |x| q
// (x > 0: x (1 - q) = x - x q; x <= 0: x q = -|x| q).  No erf -> copysign -> 0.5 (1 + .) chain, the 1/2 sits in the coefficients and
// exp(-z^2) is 2^(-(c |x|)^2) with c = sqrt(log2(e) / 2): 15 VALU per element (two of them transcendental) instead of 19 for the value
// alone, same accuracy (the identical A-S polynomial; in the NEGATIVE tail even the relative error is A-S's, where the erf form cancelled).
// The derivative Phi(x) + x phi(x) shares q and the exponential: Phi(x) = x > 0 ? 1 - q : q, phi(x) = exp(-x^2 / 2) / sqrt(2 pi).
__device__ __forceinline__ void gelu_q(float x, float& q, float& e) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f), 0.5f * 1.421413741f), 0.5f * -0.284496736f), 0.5f * 0.254829592f);
    const float w = ax * 0.84932180028801904f;          // sqrt(log2(e) / 2): exp(-x^2 / 2) = 2^(-w^2)
    e = __builtin_amdgcn_exp2f(-w * w);
    q = poly * e;
}
__device__ __forceinline__ float gelu_f(float x) {
    float q, e;
    gelu_q(x, q, e);
    return fmaf(-fabsf(x), q, fmaxf(x, 0.f));
}
// value and derivative together (the training forward of FFN1: one rcp and one exp serve both)
__device__ __forceinline__ void gelu_value_grad(float x, float& h, float& dh) {
    float q, e;
    gelu_q(x, q, e);
    h = fmaf(-fabsf(x), q, fmaxf(x, 0.f));
    const float cdf = x > 0.f ? 1.0f - q : q;
    dh = fmaf(x * 0.39894228040143268f, e, cdf);
}
__device__ __forceinline__ float gelu_grad_f(float x) { float h, d; gelu_value_grad(x, h, d); return d; }
// The same, TWO elements per call on the packed fp32 operations of gfx950 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per
// lane and instruction at the rate of one): the polynomial, the products and the final FMA cost half an instruction per element; |x|, max, the
// select and the two transcendentals stay per element.  Same operations in the same order as the scalar forms: bit-identical results.
__device__ __forceinline__ cldrd_f32v2 fma2(cldrd_f32v2 a, cldrd_f32v2 b, cldrd_f32v2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ cldrd_f32v2 splat2(float c) { return (cldrd_f32v2){c, c}; }
__device__ __forceinline__ void add2(float& a0, float& a1, float b0, float b1) {       // one v_pk_add_f32
    const cldrd_f32v2 r = (cldrd_f32v2){a0, a1} + (cldrd_f32v2){b0, b1};
    a0 = r.x; a1 = r.y;
}
__device__ __forceinline__ void mul2(float& a0, float& a1, float b0, float b1) {       // one v_pk_mul_f32
    const cldrd_f32v2 r = (cldrd_f32v2){a0, a1} * (cldrd_f32v2){b0, b1};
    a0 = r.x; a1 = r.y;
}
__device__ __forceinline__ void gelu_q2(cldrd_f32v2 ax, cldrd_f32v2& q, cldrd_f32v2& e) {
    const cldrd_f32v2 d = fma2(splat2(0.3275911f * 0.70710678118654752f), ax, splat2(1.0f));
    const cldrd_f32v2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const cldrd_f32v2 poly = t * fma2(t, fma2(t, fma2(t, fma2(t, splat2(0.5f * 1.061405429f), splat2(0.5f * -1.453152027f)), splat2(0.5f * 1.421413741f)),
                                               splat2(0.5f * -0.284496736f)), splat2(0.5f * 0.254829592f));
    const cldrd_f32v2 w = ax * splat2(0.84932180028801904f);
    const cldrd_f32v2 nw2 = -w * w;
    e = (cldrd_f32v2){__builtin_amdgcn_exp2f(nw2.x), __builtin_amdgcn_exp2f(nw2.y)};
    q = poly * e;
}
__device__ __forceinline__ void gelu_f2(float& x0, float& x1) {
    const cldrd_f32v2 ax = {fabsf(x0), fabsf(x1)};
    cldrd_f32v2 q, e;
    gelu_q2(ax, q, e);
    const cldrd_f32v2 h = fma2(-ax, q, (cldrd_f32v2){fmaxf(x0, 0.f), fmaxf(x1, 0.f)});
    x0 = h.x; x1 = h.y;
}
__device__ __forceinline__ void gelu_value_grad2(float& x0, float& x1, float& d0, float& d1) {
    const cldrd_f32v2 x = {x0, x1}, ax = {fabsf(x0), fabsf(x1)};
    cldrd_f32v2 q, e;
    gelu_q2(ax, q, e);
    const cldrd_f32v2 h = fma2(-ax, q, (cldrd_f32v2){fmaxf(x0, 0.f), fmaxf(x1, 0.f)});
    const cldrd_f32v2 omq = splat2(1.0f) - q;
    const cldrd_f32v2 cdf = {x0 > 0.f ? omq.x : q.x, x1 > 0.f ? omq.y : q.y};
    const cldrd_f32v2 dh = fma2(x * splat2(0.39894228040143268f), e, cdf);
    x0 = h.x; x1 = h.y; d0 = dh.x; d1 = dh.y;
}

// Counter-based dropout, regenerated (never stored) wherever a mask is needed: forward, backward, and the numpy mirror in
// oracle/dropout_ref.py.  An element is addressed by (row, col) of the tensor the mask applies to:
//     rowkey = mix32(row + (mix32(lo32(seed)) ^ hi32(seed) * 0x9E3779B9))  once per row (the seed part is wave-uniform)
//     h      = mix32(rowkey ^ (col >> 1))                                   one hash per PAIR of columns
//     keep   = ((col & 1) ? h >> 16 : h & 0xFFFF) >= thresh16,  thresh16 = round(p * 65536)
// mix32 is the "lowbias32" finaliser: two 32-bit multiplies.  (The first version hashed a flattened 64-bit index with
// splitmix64: two 64-bit multiplies per ELEMENT are ~16 quarter-rate v_mul_*_u32; dropout cost 1.1 ms of a 14.5 ms step.)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t drop_rowkey(uint64_t seed, uint32_t row) {
    return mix32(row + (mix32((uint32_t)seed) ^ ((uint32_t)(seed >> 32) * 0x9E3779B9u)));
}
__device__ __forceinline__ uint32_t drop_pair(uint32_t rowkey, uint32_t col) { return mix32(rowkey ^ (col >> 1)); }
__device__ __forceinline__ bool drop_keep_lo(uint32_t h, uint32_t thresh16) { return (h & 0xFFFFu) >= thresh16; }
__device__ __forceinline__ bool drop_keep_hi(uint32_t h, uint32_t thresh16) { return (h >> 16) >= thresh16; }
__device__ __forceinline__ bool dropout_keep(uint32_t rowkey, uint32_t col, uint32_t thresh16) {
    const uint32_t h = drop_pair(rowkey, col);
    return ((col & 1u) ? (h >> 16) : (h & 0xFFFFu)) >= thresh16;
}
static inline uint32_t dropout_thresh16(float p) { return (uint32_t)(p * 65536.0f + 0.5f); }

// A dropout seed as a kernel argument: the value the caller passed plus, when a seed base is installed (cldrd_set_seed_base), a 64-bit word
// read from DEVICE memory at run time.  A training step captured into a HIP graph replays its kernel arguments unchanged; the part of
// the seed that must change from step to step therefore lives in device memory (the trainer updates it with one tiny launch in front of
// each replay) and the captured arguments are only offsets (layer, call site).  No base installed: the value itself, as before.
struct SeedArg {
    unsigned long long val;
    const unsigned long long* base;
    __device__ __forceinline__ uint64_t get() const { return base ? val + *base : val; }
};
extern thread_local const unsigned long long* g_cldrd_seed_base;      // capi.hip; host side
extern thread_local const float* g_cldrd_optim_hyper;                 // capi.hip: device float[2] = {lr, step size} or null
extern thread_local const float* g_cldrd_loss_scale;                  // capi.hip: device float[72] = {S, 1 / S, good steps, skipped steps, headroom h, 3 unused, 64 scratch} or null
extern thread_local int g_cldrd_loss_scale_interval;                  // finite steps in a row after which S doubles (cldrd_set_loss_scale)
float* cldrd_norm_sink_take(int n);      // capi.hip: n slots of the clip-norm sink for one launch (null: no sink / full), see cldrd_set_norm_sink
void cldrd_norm_sink_miss(void);         // a gradient-producing launch that cannot contribute marks the sink incomplete
static inline SeedArg seed_arg(unsigned long long s) { return SeedArg{s, g_cldrd_seed_base}; }

// XCD-aware bijective block remap (cdna_hip_programming.md section 5, T1): blocks b and b+8 share an XCD, so
// give each XCD a contiguous range of logical tiles (neighbouring tiles share operand panels in that XCD's L2).
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

int cldrd_set_error(const char* msg);

// The PRODUCT library reads no environment variable: an inherited variable must not change what a run computes, or how.  Tuning and
// ablation knobs of the experiments (tools/) exist only in the development build (tools/build_dev.py: -DCLDRD_DEV_BUILD ->
// libcldrd_hip_dev.so, selected with CLDRD_LIB=...); in the product build CLDRD_DEV_INT(name, default) IS the default, at compile time.
#ifdef CLDRD_DEV_BUILD
int cldrd_dev_int(const char* name, int dflt);
#define CLDRD_DEV_INT(name, dflt) cldrd_dev_int(name, dflt)
#else
#define CLDRD_DEV_INT(name, dflt) (dflt)
#endif
// Kernel choices that tests flip IN PROCESS go through cldrd_set_tuning (capi.hip), never through the environment.
extern int g_cldrd_tune_splitk;        // 0: heuristic, 1: never split K, n > 1: n splits (small-M NT GEMM)
extern int g_cldrd_tune_attn_fwd2;     // 1: persistent attention forward where it applies, 0: one item per workgroup
extern int g_cldrd_tune_attn_bwd2;     // the same for the backward
extern int g_cldrd_tune_nt64;          // 1: one-launch 64 x 64 kernel for small-M, K <= 1024 NT GEMMs, 0: the 128 x 128 kernel (split-K + finish)
#define CLDRD_CHECK(cond, msg) do { if (!(cond)) return cldrd_set_error(msg); } while (0)
#define CLDRD_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return cldrd_set_error(hipGetErrorString(e_)); } while (0)
