// Shared by the NT GEMM kernels: argument block and the fused epilogue.
#pragma once
#include "common.h"

struct GemmNtArgs {
    const bf16_t* A; const bf16_t* B; void* C;
    int M, N, K, lda, ldb, ldc;
    const float* bias;            // [N] or null
    const bf16_t* residual;       // [M, ldr] or null, added last
    int ldr;
    bf16_t* preact;               // [M, ldc] or null: (alpha*acc + bias) before the activation
    const bf16_t* gelu_pre;       // [M, ldc] or null: multiply by gelu'(gelu_pre)
    int act;                      // 0 none, 1 erf-GELU
    float alpha;
    uint32_t drop_thresh;         // 0 = no dropout
    float drop_scale;
    uint64_t seed;
    int out_f32;
};


// Fused epilogue for 4 consecutive output columns (m, n..n+3), values v[] = raw accumulators.
// order: alpha*acc + bias -> (store preact) -> GELU -> * gelu'(gelu_pre) -> dropout -> + residual -> store
__device__ __forceinline__ void gemm_nt_apply4(const GemmNtArgs& p, float (&v)[4], int m, int n) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] *= p.alpha;
    const bool full = (n + 3 < p.N);
    if (p.bias) {
        if (full) {
            const float4 b = *(const float4*)(p.bias + n);
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        } else {
            for (int j = 0; j < 4; ++j) if (n + j < p.N) v[j] += p.bias[n + j];
        }
    }
    const size_t crow = (size_t)m * p.ldc + n;
    if (p.preact) {
        if (full) {
            uint2 o; o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
            *(uint2*)(p.preact + crow) = o;
        } else {
            for (int j = 0; j < 4; ++j) if (n + j < p.N) p.preact[crow + j] = f2bf(v[j]);
        }
    }
    if (p.act == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = gelu_f(v[j]);
    }
    if (p.gelu_pre) {
        if (full) {
            const uint2 g = *(const uint2*)(p.gelu_pre + crow);
            v[0] *= gelu_grad_f(__uint_as_float(g.x << 16)); v[1] *= gelu_grad_f(__uint_as_float(g.x & 0xFFFF0000u));
            v[2] *= gelu_grad_f(__uint_as_float(g.y << 16)); v[3] *= gelu_grad_f(__uint_as_float(g.y & 0xFFFF0000u));
        } else {
            for (int j = 0; j < 4; ++j) if (n + j < p.N) v[j] *= gelu_grad_f(bf2f(p.gelu_pre[crow + j]));
        }
    }
    if (p.drop_thresh) {
        const uint64_t e = (uint64_t)m * (uint64_t)p.N + (uint64_t)n;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = dropout_keep(p.seed, e + j, p.drop_thresh) ? v[j] * p.drop_scale : 0.f;
    }
    if (p.residual) {
        const size_t rrow = (size_t)m * p.ldr + n;
        if (full) {
            const uint2 r = *(const uint2*)(p.residual + rrow);
            v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xFFFF0000u);
            v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xFFFF0000u);
        } else {
            for (int j = 0; j < 4; ++j) if (n + j < p.N) v[j] += bf2f(p.residual[rrow + j]);
        }
    }
    if (p.out_f32) {
        float* C = (float*)p.C;
        if (full) *(float4*)(C + crow) = make_float4(v[0], v[1], v[2], v[3]);
        else for (int j = 0; j < 4; ++j) if (n + j < p.N) C[crow + j] = v[j];
    } else {
        bf16_t* C = (bf16_t*)p.C;
        if (full) {
            uint2 o; o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
            *(uint2*)(C + crow) = o;
        } else {
            for (int j = 0; j < 4; ++j) if (n + j < p.N) C[crow + j] = f2bf(v[j]);
        }
    }
}

// Epilogue for a wave that owns MT x NT MFMA-16x16 tiles computed with SWAPPED operands (D'[n][m]): lane holds
// C[m = row0 + mt*16 + (lane & 15)][n = col0 + nt*16 + 4*(lane >> 4) + j], j = 0..3.
//
// The accumulator layout would store 32-B row fragments (16 rows per instruction): the store tail ran at ~1.5 TB/s
// and cost more than the K loop on the K = 768 shapes.  So the fp32 accumulators take one trip through a
// wave-private LDS patch (32 rows at a time, row stride padded by 4 floats: conflict-free ds_write_b128) and come
// back row-major, 4 columns per lane: every global access of the epilogue (stores, residual / gelu_pre loads)
// is then a full 128-B (NT=4) or 96-B (NT=3) row segment per 16 / 12 lanes.
// `patch` = this wave's LDS scratch, 32 * (NT*16 + 4) floats; the caller has already synchronised the workgroup
// after the last fragment read of the K loop (the patch aliases the staging buffers).
template <int MT, int NT>
__device__ __forceinline__ void gemm_nt_epilogue(const GemmNtArgs& p, f32x4 (&acc)[MT][NT], int row0, int col0, int lane,
                                                 float* patch) {
    constexpr int WCOLS = NT * 16, RS = WCOLS + 4, LPR = WCOLS / 4, RPP = 64 / LPR;
    const int frow = lane & 15, fq = lane >> 4;
    const int rr = lane / LPR, rc = (lane % LPR) * 4;
#pragma unroll
    for (int mh = 0; mh < MT / 2; ++mh) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                *(f32x4*)(patch + (t * 16 + frow) * RS + nt * 16 + fq * 4) = acc[2 * mh + t][nt];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
        for (int pass = 0; pass < (32 + RPP - 1) / RPP; ++pass) {
            const int r = pass * RPP + rr;
            const int m = row0 + mh * 32 + r, n = col0 + rc;
            if (rr < RPP && r < 32 && m < p.M && n < p.N) {
                const f32x4 t4 = *(const f32x4*)(patch + r * RS + rc);
                float v[4] = {t4[0], t4[1], t4[2], t4[3]};
                gemm_nt_apply4(p, v, m, n);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}
