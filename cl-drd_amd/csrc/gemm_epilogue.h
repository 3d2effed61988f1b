// Shared by the NT GEMM kernels: argument block and the fused epilogue.
#pragma once
#include "common.h"

struct GemmNtArgs {
    const bf16_t* A; const bf16_t* B; void* C;
    int M, N, K, lda, ldb, ldc;
    const float* bias;            // [N] or null
    const void* residual;         // [M, ldr] or null, added last; bf16, or fp32 when res_f32 (the fp32 residual stream)
    int ldr;
    int res_f32 = 0;
    // fp32 residual given as a LayerNorm still to be applied: residual[m][n] holds the pre-LN sum s and the value added is
    // (s - ln_mean[m]) * ln_rstd[m] * ln_gamma[n] + ln_beta[n] - the LayerNorm kernel then need not write its fp32 output copy at all
    // (100 MB per LayerNorm at cfg2; same bytes read here).  All four or none; only with res_f32.
    const float* ln_mean = nullptr; const float* ln_rstd = nullptr; const float* ln_gamma = nullptr; const float* ln_beta = nullptr;
    bf16_t* preact;               // [M, ldc] or null: (alpha*acc + bias) before the activation
    const bf16_t* gelu_pre;       // [M, ldc] or null: multiply by gelu'(gelu_pre)
    int act;                      // bit 0: erf-GELU; bit 1 (derivative form): `preact` receives gelu'(pre-activation) instead of the
                                  // pre-activation, and `gelu_pre` already holds that derivative (the epilogue multiplies, nothing else)
    float alpha;
    uint32_t drop_thresh;         // 0 = no dropout
    float drop_scale;
    uint64_t seed;
    const unsigned long long* seed_base = nullptr;      // SeedArg of common.h: added to `seed` on the device when set
    int out_f32;
    // EPI_FILTER (top-k scan): keep C[m][n] >= thr[m] as candidate (n, score) of query m
    const float* thr; int* counts; int* cand_rows; float* cand_scores; int cap;
    int in_f16 = 0;               // the operands (and a 16-bit C) are fp16, not bf16: top-k scan, high-precision forward flavours
    int c_bf16 = 0;               // with in_f16: a 16-bit C is bf16 all the same (fp16 operands, bf16 result: the QKV projection, whose
                                  // consumers - attention forward and backward - are bf16 kernels)
    bf16_t* c_copy = nullptr;     // with in_f16 and a 16-bit C: a bf16 copy of C [M, ldc] (the backward's MFMAs read bf16: the tape of an
                                  // fp16-operand forward GEMM), or null
    int ksplit = 1;               // small-M kernel: K range split over ksplit workgroups per tile, fp32 partials in `slabs` ([ksplit][M][N]),
    float* slabs = nullptr;       //   summed in a fixed order and finished (epilogue) by splitk_finish_kernel
    int gn = 0;                   // ring kernel: N tiles are walked in groups of gn inside an XCD's range (0: row-major)
    int stagger = 1;              // ring kernel: waves 4..7 issue their LDS-DMA one k-step after waves 0..3 (0: A/B runs)
    int asym = 1;                 // ring kernel, two whole slots: a third A slot, A requested two K tiles ahead (0: the round-2 schedule)
    int early1 = 1;               // ring kernel, two LDS slots: K tile 1 is requested together with K tile 0 at tile start (0: after tile 0 landed)
    int phase_units = 0;          // ring kernel: half of the first round's workgroups start this many s_memtime units late (0: off)
    int tape_f16 = 0;             // with in_f16 (the all-fp16 training mode): `preact` is written and `gelu_pre` is read as fp16, not bf16
};

// compile-time epilogue flavours (each GEMM kernel is instantiated per flavour so the 16x-unrolled epilogue carries
// only the code it needs); EPI_GENERIC reads every switch from GemmNtArgs at run time.
enum : int {
    EPI_BIAS = 1, EPI_PREACT = 2, EPI_GELU = 4, EPI_GELUGRAD = 8, EPI_DROPOUT = 16, EPI_RESIDUAL = 32, EPI_F32 = 64, EPI_FILTER = 128,
    EPI_RES32 = 256,              // the residual operand is fp32 (only with EPI_RESIDUAL)
    EPI_SPLITK = 4096,            // small-M kernel: write raw fp32 partial sums of this workgroup's K range to GemmNtArgs::slabs (no epilogue)
    EPI_RESLN = 2048,             // the fp32 residual is a pre-LN sum: apply the LayerNorm (ln_* of GemmNtArgs) on the fly (with EPI_RES32)
    EPI_DGELU = 1024,             // derivative form of the saved activation input (act bit 1): with EPI_PREACT / EPI_GELUGRAD
    EPI_F16IN = 512,              // A, B (and a 16-bit C) hold fp16, not bf16: the top-k scan over the fp16 index shadow (with EPI_FILTER)
                                  // and the high-precision forward of the query tower (small-M kernel; not with preact / gelu_pre)
    EPI_GENERIC = 1 << 20
};

typedef _Float16 gemm_f16x8 __attribute__((ext_vector_type(8)));
template <int EPI>
__device__ __forceinline__ f32x4 gemm_mfma(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (EPI != EPI_GENERIC && (EPI & EPI_F16IN) != 0)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(gemm_f16x8, a), __builtin_bit_cast(gemm_f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <int EPI> struct EpiFlags {
    const bool bias, preact, gelu, gelugrad, dropout, residual, f32, res32, dgelu, resln;
    __device__ __forceinline__ explicit EpiFlags(const GemmNtArgs& p)
        : bias(EPI == EPI_GENERIC ? p.bias != nullptr : (EPI & EPI_BIAS) != 0),
          preact(EPI == EPI_GENERIC ? p.preact != nullptr : (EPI & EPI_PREACT) != 0),
          gelu(EPI == EPI_GENERIC ? (p.act & 1) != 0 : (EPI & EPI_GELU) != 0),
          gelugrad(EPI == EPI_GENERIC ? p.gelu_pre != nullptr : (EPI & EPI_GELUGRAD) != 0),
          dropout(EPI == EPI_GENERIC ? p.drop_thresh != 0 : (EPI & EPI_DROPOUT) != 0),
          residual(EPI == EPI_GENERIC ? p.residual != nullptr : (EPI & EPI_RESIDUAL) != 0),
          f32(EPI == EPI_GENERIC ? p.out_f32 != 0 : (EPI & EPI_F32) != 0),
          res32(EPI == EPI_GENERIC ? (p.residual != nullptr && p.res_f32 != 0) : (EPI & EPI_RES32) != 0),
          dgelu(EPI == EPI_GENERIC ? (p.act & 2) != 0 : (EPI & EPI_DGELU) != 0),
          resln(EPI == EPI_GENERIC ? (p.residual != nullptr && p.res_f32 != 0 && p.ln_mean != nullptr) : (EPI & EPI_RESLN) != 0) {}
};

static inline int epi_flavour(const GemmNtArgs& a) {
    return (a.in_f16 ? EPI_F16IN : 0) | (a.bias ? EPI_BIAS : 0) | (a.preact ? EPI_PREACT : 0) | ((a.act & 1) ? EPI_GELU : 0) | (a.gelu_pre ? EPI_GELUGRAD : 0) |
           (((a.act & 2) && (a.preact || a.gelu_pre)) ? EPI_DGELU : 0) |
           (a.drop_thresh ? EPI_DROPOUT : 0) | (a.residual ? EPI_RESIDUAL : 0) | (a.out_f32 ? EPI_F32 : 0) |
           ((a.residual && a.res_f32) ? EPI_RES32 : 0) | ((a.residual && a.res_f32 && a.ln_mean) ? EPI_RESLN : 0);
}

__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xFFFF0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xFFFF0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xFFFF0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xFFFF0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
    uint4 o;
    o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
    return o;
}
// two floats -> one dword of fp16 (RNE): ONE v_cvt_pk_f16_f32 on gfx950 (two scalar converts + an or otherwise: 3 VALU per pair)
typedef _Float16 cldrd_f16v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2half(float lo, float hi) {
    const cldrd_f32v2 f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, cldrd_f16v2));
}
__device__ __forceinline__ void unpack8h(const uint4& u, float (&f)[8]) {
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[j] & 0xFFFFu));
        f[2 * j + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[j] >> 16));
    }
}
__device__ __forceinline__ uint4 pack8h(const float (&v)[8]) {
    uint4 o;
    o.x = pack2half(v[0], v[1]); o.y = pack2half(v[2], v[3]); o.z = pack2half(v[4], v[5]); o.w = pack2half(v[6], v[7]);
    return o;
}

// Fused epilogue for 8 consecutive output columns (m, n..n+7); v[] = raw accumulators, bias8/res/gp already loaded.
// order: alpha*acc + bias -> (store preact, or gelu'(preact) in the derivative form) -> GELU -> * gelu'(gelu_pre) (derivative form:
// * gelu_pre) -> dropout -> + residual -> store
template <int EPI>
__device__ __forceinline__ void gemm_nt_apply8(const GemmNtArgs& p, const EpiFlags<EPI>& fl, float (&v)[8], int m, int n,
                                               const float (&bias8)[8], const uint4& res, const uint4& res_hi, const uint4& gp,
                                               unsigned long long seed_eff, bool ok = true) {
    // `ok` guards the STORES only: loads and arithmetic of a lane outside the tile run on clamped addresses / dead values, so that the epilogue
    // is straight-line code and the compiler can count its vmcnt waits (see gemm_nt_epilogue)
    if (p.alpha != 1.0f) {          // scalar test: every encoder GEMM has alpha = 1 (one VALU per element saved in VALU-bound epilogues)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= p.alpha;
    }
    if (fl.bias) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) add2(v[j], v[j + 1], bias8[j], bias8[j + 1]);
    }
    const size_t crow = (size_t)m * p.ldc + n;
    if (fl.preact && fl.gelu && fl.dgelu) {
        // forward of a GELU layer with a tape: the backward only ever needs gelu'(x), so that is what is saved (one exp and one rcp
        // serve both the value and the derivative here; the data-gradient GEMM's epilogue then multiplies instead of evaluating
        // erf and exp again - that epilogue was VALU-bound: 12 us of a 31-us tile round with the MFMA pipe idle)
        float dv[8];
#pragma unroll
        for (int j = 0; j < 8; j += 2) gelu_value_grad2(v[j], v[j + 1], dv[j], dv[j + 1]);
        { const uint4 o = p.tape_f16 ? pack8h(dv) : pack8(dv); if (ok) st16_stream(p.preact + crow, o); }
    } else {
        if (fl.preact) {
            if (fl.dgelu) {
                float dv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) dv[j] = gelu_grad_f(v[j]);
                { const uint4 o = p.tape_f16 ? pack8h(dv) : pack8(dv); if (ok) st16_stream(p.preact + crow, o); }
            } else {
                { const uint4 o = p.tape_f16 ? pack8h(v) : pack8(v); if (ok) st16_stream(p.preact + crow, o); }
            }
        }
        if (fl.gelu) {
#pragma unroll
            for (int j = 0; j < 8; j += 2) gelu_f2(v[j], v[j + 1]);
        }
    }
    if (fl.gelugrad) {
        float g[8];
        if (p.tape_f16) unpack8h(gp, g); else unpack8(gp, g);
        if (fl.dgelu) {
#pragma unroll
            for (int j = 0; j < 8; j += 2) mul2(v[j], v[j + 1], g[j], g[j + 1]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= gelu_grad_f(g[j]);
        }
    }
    if (fl.dropout) {
        const uint32_t rk = drop_rowkey(seed_eff, (uint32_t)m);      // n is a multiple of 8: four column pairs
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const uint32_t h = drop_pair(rk, (uint32_t)(n + j));
            v[j] = drop_keep_lo(h, p.drop_thresh) ? v[j] * p.drop_scale : 0.f;
            v[j + 1] = drop_keep_hi(h, p.drop_thresh) ? v[j + 1] * p.drop_scale : 0.f;
        }
    }
    if (fl.residual) {
        float r[8];
        if (fl.res32) {         // fp32 residual stream: res = columns n..n+3, res_hi = n+4..n+7 (raw float bits)
            r[0] = __uint_as_float(res.x); r[1] = __uint_as_float(res.y); r[2] = __uint_as_float(res.z); r[3] = __uint_as_float(res.w);
            r[4] = __uint_as_float(res_hi.x); r[5] = __uint_as_float(res_hi.y); r[6] = __uint_as_float(res_hi.z); r[7] = __uint_as_float(res_hi.w);
        } else {
            unpack8(res, r);
        }
#pragma unroll
        for (int j = 0; j < 8; j += 2) add2(v[j], v[j + 1], r[j], r[j + 1]);
    }
    if (fl.f32) {
        float* C = (float*)p.C + crow;
        if (ok) {
            *(float4*)C = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)(C + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    } else {
        const bool h16 = (EPI == EPI_GENERIC ? p.in_f16 != 0 : (EPI & EPI_F16IN) != 0) && !p.c_bf16;
        if (h16) {
            const uint4 o = pack8h(v);
            if (ok) *(uint4*)((bf16_t*)p.C + crow) = o;
            if (p.c_copy) { const uint4 o2 = pack8(v); if (ok) st16_stream(p.c_copy + crow, o2); }       // the tape copy: read ~10 ms later by the weight-gradient launch
        } else {
            const uint4 o = pack8(v);
            if (ok) *(uint4*)((bf16_t*)p.C + crow) = o;
        }
    }
}

// Epilogue for a wave that owns MT x NT MFMA-16x16 tiles computed with SWAPPED operands (D'[n][m]): lane holds
// C[m = row0 + mt*16 + (lane & 15)][n = col0 + nt*16 + 4*(lane >> 4) + j], j = 0..3.
//
// Stores in the accumulator layout are 32-B row fragments and every residual / gelu_pre load would be a dependent
// 8-B access: that tail ran at ~1.5 TB/s and cost more than the K loop on the K = 768 shapes.  So the fp32
// accumulators take one trip through a wave-private LDS patch, 32 rows at a time (row stride padded by 4 floats:
// conflict-free ds_write_b128), and come back row-major, 8 columns per lane: every global access of the epilogue is
// a 16-B-per-lane access of full row segments.  The residual / gelu_pre operands of chunk c+1 are requested before
// chunk c is processed, so their latency hides under the LDS trip and the math of the previous chunk.
// `patch` = this wave's LDS scratch, 32 * (NT*16 + 4) floats; the caller has already synchronised the workgroup
// after the last fragment read of the K loop (the patch aliases the staging buffers).
template <int MT, int NT, int EPI>
__device__ __forceinline__ void gemm_nt_epilogue(const GemmNtArgs& p, f32x4 (&acc)[MT][NT], int row0, int col0, int lane,
                                                 float* patch, unsigned long long* dbg = nullptr) {
    constexpr int WCOLS = NT * 16, RS = WCOLS + 4, LPR = WCOLS / 8, RPP = 64 / LPR, NPASS = (32 + RPP - 1) / RPP;
    const EpiFlags<EPI> fl(p);
    const int frow = lane & 15, fq = lane >> 4;
    const int rr = lane / LPR, rc = (lane % LPR) * 8;
    const int n = col0 + rc;
    const bool lane_ok = rr < RPP && n < p.N;       // N % 8 == 0 (checked by the launcher): a started group is complete
    // Round 4: the epilogue is STRAIGHT-LINE code.  Until then its loads sat in exec-masked `if (lane inside the tile)` regions; at the join
    // behind such a region the compiler's wait-count pass must assume the path on which nothing was issued after the load, i.e. every later
    // use waits with `s_waitcnt vmcnt(0)` - and on CDNA4 vmcnt counts STORES too, so each of the 16 passes of a wave began by waiting for the
    // previous pass's stores to be acknowledged by L2 (~1 us under load; the per-column constants below were "pending" in every pass because a
    // pass skipped by an empty exec mask does not wait for them).  tools/epi_ablate.py: 15-25 us per tile for the fused flavours against 3-4 us
    // for the plain one.  Now every load is unconditional on a clamped address, the arithmetic runs on all lanes, only the stores are
    // predicated (no branch around a single store), and the per-tile constants are consumed once, here, by an empty asm: one wait, in
    // dominating code.
    const int nc = n < p.N ? n : 0;
    float bias8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bias8[j] = 0.f;
    if (fl.bias) {
        const float4 b0 = *(const float4*)(p.bias + nc), b1 = *(const float4*)(p.bias + nc + 4);
        bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w;
        bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
    }
    float lng[8], lnb[8];
    if (fl.resln) {
        const float4 g0 = *(const float4*)(p.ln_gamma + nc), g1 = *(const float4*)(p.ln_gamma + nc + 4);
        const float4 b0 = *(const float4*)(p.ln_beta + nc), b1 = *(const float4*)(p.ln_beta + nc + 4);
        lng[0] = g0.x; lng[1] = g0.y; lng[2] = g0.z; lng[3] = g0.w; lng[4] = g1.x; lng[5] = g1.y; lng[6] = g1.z; lng[7] = g1.w;
        lnb[0] = b0.x; lnb[1] = b0.y; lnb[2] = b0.z; lnb[3] = b0.w; lnb[4] = b1.x; lnb[5] = b1.y; lnb[6] = b1.z; lnb[7] = b1.w;
    }
    // The effective dropout seed, read ONCE: until round 4 `*p.seed_base` (a global load: the step's seed lives in device memory for the graph
    // replay) sat inside gemm_nt_apply8, behind the chunk loop's compiler barriers - reloaded in each of the 16 passes of a wave, every time
    // followed by `s_waitcnt vmcnt(0)`, which also waits for the previous pass's STORES and the next chunk's prefetches: ~1.2 us x 16 per tile
    // (tools/epi_ablate.py: the dropout + residual flavours spent 24 us per tile in the epilogue, 20 of them here).
    unsigned long long seed_eff = 0;
    if (fl.dropout) seed_eff = p.seed_base ? p.seed + *p.seed_base : p.seed;
    if (fl.bias) asm volatile("" ::"v"(bias8[0]), "v"(bias8[1]), "v"(bias8[2]), "v"(bias8[3]), "v"(bias8[4]), "v"(bias8[5]), "v"(bias8[6]), "v"(bias8[7]));
    if (fl.resln) {
        asm volatile("" ::"v"(lng[0]), "v"(lng[1]), "v"(lng[2]), "v"(lng[3]), "v"(lng[4]), "v"(lng[5]), "v"(lng[6]), "v"(lng[7]));
        asm volatile("" ::"v"(lnb[0]), "v"(lnb[1]), "v"(lnb[2]), "v"(lnb[3]), "v"(lnb[4]), "v"(lnb[5]), "v"(lnb[6]), "v"(lnb[7]));
    }
    if (fl.dropout) asm volatile("" ::"v"((uint32_t)seed_eff), "v"((uint32_t)(seed_eff >> 32)));
    uint4 res[NPASS], resh[NPASS], gp[NPASS];
    float lmu[NPASS], lrs[NPASS];
    const int mlast = p.M - 1;
    // operands of pass `pass` of chunk `mh` (residual, LayerNorm statistics, gelu' tape): requested right behind the same pass of chunk mh - 1,
    // into the registers that pass has just consumed - a whole chunk of LDS trip, arithmetic and stores ahead of their use, without a second
    // register set
    auto prefetch_pass = [&](int mh, int pass) {
        const int r = pass * RPP + rr;
        const int mr = row0 + mh * 32 + (r < 32 ? r : 31);
        const int m = mr < mlast ? mr : mlast;          // clamped: rows past the tile / the matrix read a valid row nobody stores
        if (fl.residual) {
            if (fl.res32) {
                const uint4* rp = (const uint4*)((const float*)p.residual + (size_t)m * p.ldr + nc);
                res[pass] = ld16_stream(rp);              // the fp32 stream's last reader before the backward
                resh[pass] = ld16_stream(rp + 1);
                if (fl.resln) { lmu[pass] = p.ln_mean[m]; lrs[pass] = p.ln_rstd[m]; }
            } else {
                res[pass] = *(const uint4*)((const bf16_t*)p.residual + (size_t)m * p.ldr + nc);
            }
        }
        if (fl.gelugrad) gp[pass] = ld16_stream(p.gelu_pre + (size_t)m * p.ldc + nc);
    };
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        res[pass] = make_uint4(0, 0, 0, 0);
        resh[pass] = make_uint4(0, 0, 0, 0);
        gp[pass] = make_uint4(0, 0, 0, 0);
        lmu[pass] = 0.f; lrs[pass] = 0.f;
        if (fl.residual || fl.gelugrad) prefetch_pass(0, pass);
    }
    // The LDS executes one wave's instructions in order, so inside this wave-private patch a read issued after a write sees it and a write
    // issued after a read cannot overtake it: no s_waitcnt between them (until round 3 there were two lgkmcnt(0) per chunk, i.e. two LDS
    // round trips in front of every chunk's stores).  Chunk mh+1 is written right behind the reads of chunk mh, so its trip overlaps the
    // arithmetic and the stores of chunk mh; the compiler waits for the read RESULTS where they are used.
    auto write_patch = [&](int mh) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                *(f32x4*)(patch + (t * 16 + frow) * RS + nt * 16 + fq * 4) = acc[2 * mh + t][nt];
    };
    write_patch(0);
#pragma unroll
    for (int mh = 0; mh < MT / 2; ++mh) {
        asm volatile("" ::: "memory");
        float v[NPASS][8];
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int r = pass * RPP + rr;
            const float* src = patch + (r < 32 ? r : 0) * RS + rc;
            const f32x4 lo = *(const f32x4*)src, hi = *(const f32x4*)(src + 4);
            v[pass][0] = lo[0]; v[pass][1] = lo[1]; v[pass][2] = lo[2]; v[pass][3] = lo[3];
            v[pass][4] = hi[0]; v[pass][5] = hi[1]; v[pass][6] = hi[2]; v[pass][7] = hi[3];
        }
        asm volatile("" ::: "memory");
        if (mh + 1 < MT / 2) write_patch(mh + 1);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            uint4 cres = res[pass], cresh = resh[pass];
            const uint4 cgp = gp[pass];
            if (fl.resln) {       // the residual is LayerNorm(pre-LN sum): same expression as ln_fwd_kernel's fp32 output
                const float mu = lmu[pass], rs = lrs[pass];
                uint32_t* lo = (uint32_t*)&cres;
                uint32_t* hi = (uint32_t*)&cresh;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lo[j] = __float_as_uint((__uint_as_float(lo[j]) - mu) * rs * lng[j] + lnb[j]);
                    hi[j] = __float_as_uint((__uint_as_float(hi[j]) - mu) * rs * lng[4 + j] + lnb[4 + j]);
                }
            }
            if ((fl.residual || fl.gelugrad) && mh + 1 < MT / 2) prefetch_pass(mh + 1, pass);
            const int r = pass * RPP + rr;
            const int m = row0 + mh * 32 + r;
            gemm_nt_apply8<EPI>(p, fl, v[pass], m, nc, bias8, cres, cresh, cgp, seed_eff, lane_ok && r < 32 && m < p.M);
        }
#ifdef CLDRD_DEV_BUILD
        if (dbg) { const unsigned long long t = __builtin_readcyclecounter(); if (lane == 0) dbg[mh] = t; }      // tools/epi_stamps.py
#endif
    }
}

// The fp32-out flavours (bias [+ dropout] + fp32 residual [LayerNorm on the fly] -> fp32 sum; plain fp32 out): FOUR columns per lane.
// With eight, a lane's 32 bytes of output went out as two 16-B stores at a 32-B lane stride: every store instruction wrote one half of each
// 32-B sector, and so did the residual loads - tools/micro/store_rate.hip: 174 cycles per store instruction per CU and 3.4 TB/s chip-wide for
// that pattern against 86-110 cycles and 5.7-7.3 TB/s for instructions that write whole row segments.  Here consecutive lanes own consecutive
// 16-B pieces of a row: one load, one LDS read and one store per lane and pass, all of them contiguous row segments.
template <int MT, int NT, int EPI>
__device__ __forceinline__ void gemm_nt_epilogue_f32(const GemmNtArgs& p, f32x4 (&acc)[MT][NT], int row0, int col0, int lane, float* patch,
                                                     unsigned long long* dbg = nullptr) {
    constexpr int WCOLS = NT * 16, RS = WCOLS + 4, LPR = WCOLS / 4, RPP = 64 / LPR, NPASS = (32 + RPP - 1) / RPP;
    const EpiFlags<EPI> fl(p);
    const int frow = lane & 15, fq = lane >> 4;
    const int rr = lane / LPR, rc = (lane % LPR) * 4;
    const int n = col0 + rc;
    const bool lane_ok = rr < RPP && n < p.N;
    const int nc = n < p.N ? n : 0;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f), lng = bias4, lnb = bias4;
    if (fl.bias) bias4 = *(const float4*)(p.bias + nc);
    if (fl.resln) { lng = *(const float4*)(p.ln_gamma + nc); lnb = *(const float4*)(p.ln_beta + nc); }
    unsigned long long seed_eff = 0;
    if (fl.dropout) seed_eff = p.seed_base ? p.seed + *p.seed_base : p.seed;
    if (fl.bias) asm volatile("" ::"v"(bias4.x), "v"(bias4.y), "v"(bias4.z), "v"(bias4.w));
    if (fl.resln) asm volatile("" ::"v"(lng.x), "v"(lng.y), "v"(lng.z), "v"(lng.w), "v"(lnb.x), "v"(lnb.y), "v"(lnb.z), "v"(lnb.w));
    if (fl.dropout) asm volatile("" ::"v"((uint32_t)seed_eff), "v"((uint32_t)(seed_eff >> 32)));
    uint4 res[NPASS];
    float lmu[NPASS], lrs[NPASS];
    const int mlast = p.M - 1;
    auto prefetch_pass = [&](int mh, int pass) {
        const int r = pass * RPP + rr;
        const int mr = row0 + mh * 32 + (r < 32 ? r : 31);
        const int m = mr < mlast ? mr : mlast;
        if (fl.residual) {
            res[pass] = ld16_stream((const float*)p.residual + (size_t)m * p.ldr + nc);
            if (fl.resln) { lmu[pass] = p.ln_mean[m]; lrs[pass] = p.ln_rstd[m]; }
        }
    };
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        res[pass] = make_uint4(0, 0, 0, 0);
        lmu[pass] = 0.f; lrs[pass] = 0.f;
        if (fl.residual) prefetch_pass(0, pass);
    }
    auto write_patch = [&](int mh) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                *(f32x4*)(patch + (t * 16 + frow) * RS + nt * 16 + fq * 4) = acc[2 * mh + t][nt];
    };
    write_patch(0);
#pragma unroll
    for (int mh = 0; mh < MT / 2; ++mh) {
        asm volatile("" ::: "memory");
        f32x4 v[NPASS];
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int r = pass * RPP + rr;
            v[pass] = *(const f32x4*)(patch + (r < 32 ? r : 0) * RS + rc);
        }
        asm volatile("" ::: "memory");
        if (mh + 1 < MT / 2) write_patch(mh + 1);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int r = pass * RPP + rr;
            const int m = row0 + mh * 32 + r;
            float x[4] = {v[pass][0], v[pass][1], v[pass][2], v[pass][3]};
            if (p.alpha != 1.0f) {
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] *= p.alpha;
            }
            if (fl.bias) { x[0] += bias4.x; x[1] += bias4.y; x[2] += bias4.z; x[3] += bias4.w; }
            if (fl.dropout) {       // the same (row, column pair) hash as gemm_nt_apply8: nc is a multiple of 4 -> two column pairs
                const uint32_t rk = drop_rowkey(seed_eff, (uint32_t)m);
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const uint32_t h = drop_pair(rk, (uint32_t)(nc + j));
                    x[j] = drop_keep_lo(h, p.drop_thresh) ? x[j] * p.drop_scale : 0.f;
                    x[j + 1] = drop_keep_hi(h, p.drop_thresh) ? x[j + 1] * p.drop_scale : 0.f;
                }
            }
            if (fl.residual) {
                const uint4 cr = res[pass];
                float rv[4] = {__uint_as_float(cr.x), __uint_as_float(cr.y), __uint_as_float(cr.z), __uint_as_float(cr.w)};
                if (fl.resln) {     // the residual is LayerNorm(pre-LN sum): same expression as ln_fwd_kernel's fp32 output
                    const float mu = lmu[pass], rs = lrs[pass];
                    rv[0] = (rv[0] - mu) * rs * lng.x + lnb.x; rv[1] = (rv[1] - mu) * rs * lng.y + lnb.y;
                    rv[2] = (rv[2] - mu) * rs * lng.z + lnb.z; rv[3] = (rv[3] - mu) * rs * lng.w + lnb.w;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] += rv[j];
            }
            if (fl.residual && mh + 1 < MT / 2) prefetch_pass(mh + 1, pass);
            if (lane_ok && r < 32 && m < p.M) *(float4*)((float*)p.C + (size_t)m * p.ldc + nc) = make_float4(x[0], x[1], x[2], x[3]);
        }
#ifdef CLDRD_DEV_BUILD
        if (dbg) { const unsigned long long t = __builtin_readcyclecounter(); if (lane == 0) dbg[mh] = t; }      // tools/epi_stamps.py
#endif
    }
}

// compile-time test: does this flavour take the four-columns-per-lane fp32 epilogue?
template <int EPI>
constexpr bool epi_is_f32_only() {
    return EPI != EPI_GENERIC && (EPI & EPI_F32) != 0 && (EPI & (EPI_PREACT | EPI_GELU | EPI_GELUGRAD | EPI_FILTER | EPI_SPLITK)) == 0 &&
           ((EPI & EPI_RESIDUAL) == 0 || (EPI & EPI_RES32) != 0);
}

// Top-k scan epilogue (no store of C): lane holds C[m = query][n = 4 consecutive index rows]; anything at or above the
// query's threshold is appended to that query's candidate list.  Hits are ~0.1 % of the elements, so the atomics are rare.
template <int MT, int NT>
__device__ __forceinline__ void gemm_nt_filter_epilogue(const GemmNtArgs& p, f32x4 (&acc)[MT][NT], int row0, int col0, int lane) {
    const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = row0 + mt * 16 + frow;
        if (m >= p.M) continue;
        const float t = p.thr[m];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = col0 + nt * 16 + fq * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = acc[mt][nt][j] * p.alpha;
                if (n + j < p.N && v >= t) {
                    const int pos = atomicAdd(p.counts + m, 1);
                    if (pos < p.cap) {
                        p.cand_rows[(size_t)m * p.cap + pos] = n + j;
                        p.cand_scores[(size_t)m * p.cap + pos] = v;
                    }
                }
            }
        }
    }
}

// Same filter with the roles swapped (ring kernel: rows m = index rows, columns n = queries): thr / counts are per column.
template <int MT, int NT>
__device__ __forceinline__ void gemm_nt_filter_epilogue_cols(const GemmNtArgs& p, f32x4 (&acc)[MT][NT], int row0, int col0, int lane) {
    const int frow = lane & 15, fq = lane >> 4;
    float t[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = col0 + nt * 16 + fq * 4 + j;
            t[nt][j] = n < p.N ? p.thr[n] : __builtin_inff();
        }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = row0 + mt * 16 + frow;
        if (m >= p.M) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = acc[mt][nt][j] * p.alpha;
                if (v >= t[nt][j]) {
                    const int n = col0 + nt * 16 + fq * 4 + j;
                    const int pos = atomicAdd(p.counts + n, 1);
                    if (pos < p.cap) {
                        p.cand_rows[(size_t)n * p.cap + pos] = m;
                        p.cand_scores[(size_t)n * p.cap + pos] = v;
                    }
                }
            }
    }
}
