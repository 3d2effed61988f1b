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


// Epilogue for a wave that owns MT x NT MFMA-16x16 tiles computed with SWAPPED operands (D'[n][m]): lane holds
// C[m = row0 + mt*16 + (lane & 15)][n = col0 + nt*16 + 4*(lane >> 4) + j], j = 0..3.
// order: alpha*acc + bias -> (store preact) -> GELU -> * gelu'(gelu_pre) -> dropout -> + residual -> store
template <int MT, int NT>
__device__ __forceinline__ void gemm_nt_epilogue(const GemmNtArgs& p, f32x4 (&acc)[MT][NT], int row0, int col0, int lane) {
    const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = row0 + mt * 16 + frow;
        if (m >= p.M) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = col0 + nt * 16 + fq * 4;
            if (n >= p.N) continue;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[mt][nt][j] * p.alpha;
            const bool full = (n + 3 < p.N);
            if (p.bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (n + j < p.N) v[j] += p.bias[n + j];
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
#pragma unroll
                for (int j = 0; j < 4; ++j) if (n + j < p.N) v[j] *= gelu_grad_f(bf2f(p.gelu_pre[crow + j]));
            }
            if (p.drop_thresh) {
                const uint64_t e = (uint64_t)m * (uint64_t)p.N + (uint64_t)n;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = dropout_keep(p.seed, e + j, p.drop_thresh) ? v[j] * p.drop_scale : 0.f;
            }
            if (p.residual) {
                const size_t rrow = (size_t)m * p.ldr + n;
#pragma unroll
                for (int j = 0; j < 4; ++j) if (n + j < p.N) v[j] += bf2f(p.residual[rrow + j]);
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
    }
}
