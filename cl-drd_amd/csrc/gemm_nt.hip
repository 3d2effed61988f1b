// C[M,N] = epilogue( alpha * A[M,K] . B[N,K]^T )   bf16 operands, fp32 accumulate on MFMA.
//
// This is the Linear layer of the encoder (reference call sites models/nway_dual_encoder.py:52,56,64 ->
// HF q_lin/k_lin/v_lin/out_lin/ffn.lin1/ffn.lin2) in "NT" form: both operands are
// K-contiguous (activations [tokens, in], weights [out, in]).  The same kernel serves the data-gradient
// GEMMs with the transposed bf16 weight shadow as B.
//
// Structure (v1): 128x128x64 tiles, 256 threads = 2x2 waves of 64x64, MFMA 16x16x32 bf16.  Operands go
// HBM -> LDS by LDS-DMA (global_load_lds, 16 B/lane, no VGPR round trip) into two 32 KiB stages; the
// LDS image is lane-linear per 1 KiB piece (8 rows x 128 B), XOR-swizzled on the SOURCE address
// (chunk ^= row & 7) and on the ds_read_b128 address, which makes the fragment reads bank-conflict free.
// One barrier per K tile: loads of tile t+1 are issued right after the barrier and fly under the MFMAs
// of tile t.  Fused epilogue: bias, erf-GELU (optionally saving the pre-activation), GELU' multiply
// (data gradient through the activation), dropout, residual add; bf16 or fp32 store.
#include "common.h"
#include "gemm_epilogue.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;        // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // A + B

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmNtArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntn = (p.N + BN - 1) / BN;
    constexpr bool SPLITK = EPI != EPI_GENERIC && (EPI & EPI_SPLITK) != 0;
    const int ntiles = SPLITK ? (int)gridDim.x / p.ksplit : (int)gridDim.x;
    const int split = SPLITK ? (int)blockIdx.x / ntiles : 0;
    const int tile = SPLITK ? (int)blockIdx.x % ntiles : xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int wm = wid >> 1, wn = wid & 1;

    // ---- LDS-DMA staging: wave w owns pieces 4w..4w+3 of each operand tile (piece = 8 rows x 128 B) ----
    const int prow = lane >> 3;                       // row inside the piece == (row & 7)
    const int chunk = (lane & 7) ^ prow;              // source chunk for LDS slot (lane & 7): swizzle on the source
    const bf16_t* ga[4];
    const bf16_t* gb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wid * 4 + i) * 8 + prow;
        const int ra = min(m0 + r, p.M - 1), rb = min(n0 + r, p.N - 1);
        ga[i] = p.A + (size_t)ra * p.lda + chunk * 8;
        gb[i] = p.B + (size_t)rb * p.ldb + chunk * 8;
    }
    auto stage = [&](int s, int k0) {
        char* base = smem + s * STAGE_BYTES + wid * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(ga[i] + k0), LDS_PTR(base + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gb[i] + k0), LDS_PTR(base + TILE_BYTES + i * 1024), 16, 0, 0);
        }
    };

    // ---- fragment addressing: lane reads row (lane & 15) of a 16-row tile, 16-B chunk 4*s + (lane >> 4) ----
    const int frow = lane & 15, fq = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int slot = (4 * s + fq) ^ (frow & 7);
        a_off[s] = (wm * 64 + frow) * 128 + slot * 16;
        b_off[s] = TILE_BYTES + (wn * 64 + frow) * 128 + slot * 16;
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // K range of this workgroup (split-K: a contiguous share of the K tiles; the others: everything)
    const int nk_all = p.K / BK;
    const int kper = SPLITK ? (nk_all + p.ksplit - 1) / p.ksplit : nk_all;
    const int kbeg = split * kper;
    const int nk = min(nk_all, kbeg + kper) - kbeg;
    if (nk > 0) stage(0, kbeg * BK);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage((kt + 1) & 1, (kbeg + kt + 1) * BK);
        const char* sb = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[t] = *(const bf16x8*)(sb + a_off[s] + t * 16 * 128);
                bfr[t] = *(const bf16x8*)(sb + b_off[s] + t * 16 * 128);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    // operands swapped on purpose: D'[n][m], so a lane's 4 accumulators are 4 consecutive n
                    acc[mt][nt] = gemm_mfma<EPI>(bfr[nt], af[mt], acc[mt][nt]);
        }
    }

    if constexpr (SPLITK) {
        // raw partial sums of this K range, accumulator layout: lane holds 4 consecutive columns of row (lane & 15) of each 16 x 16 tile
        float* out = p.slabs + (size_t)split * p.M * p.N;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = m0 + wm * 64 + mt * 16 + (lane & 15);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int n = n0 + wn * 64 + nt * 16 + 4 * (lane >> 4);
                if (m < p.M && n < p.N) *(f32x4*)(out + (size_t)m * p.N + n) = acc[mt][nt];
            }
        }
    } else if constexpr ((EPI & EPI_FILTER) != 0 && EPI != EPI_GENERIC) {
        gemm_nt_filter_epilogue<4, 4>(p, acc, m0 + wm * 64, n0 + wn * 64, lane);
    } else {
        __syncthreads();      // all fragment reads of the last K tile are done: the staging buffers become epilogue scratch
        gemm_nt_epilogue<4, 4, EPI>(p, acc, m0 + wm * 64, n0 + wn * 64, lane, (float*)smem + wid * (32 * 68));
    }
}

// Round 5: small-M problems with a short K range (K <= 1024: five of the eight GEMMs of a transformer layer) in ONE launch.  M = 256 rows
// make 12 tiles of 128 x 128 at N = 768: the kernel above either leaves 95 % of the chip idle or splits K over more workgroups and needs a second
// launch to add the partial sums (10 + 5 us for 0.3 GFLOP).  Here the tile is 64 x 64 (48 workgroups at 256 x 768, 192 at 256 x 3072) and the
// K loop is latency-hiding instead of wide: eight 16-KiB stages of LDS, SEVEN K tiles requested before the first one is consumed, one barrier per
// K tile (tile kt + 7 is requested into the slot tile kt - 1 was read from, right behind the barrier that ends those reads).  A workgroup
// pulls its whole 64 x K + 64 x K operand strip (192 KiB at K = 768) in about one memory latency plus the CU's L2 -> LDS streaming time.
// Same LDS image, swizzle, MFMA order along K and epilogue as the 128 x 128 kernel: the result is bit-identical to its one-pass form.
constexpr int SB = 64;
constexpr int S_TILE = SB * BK * 2;            // 8 KiB per operand tile
constexpr int S_STAGE = 2 * S_TILE;
constexpr int S_NST = 8;                       // stages (128 KiB of LDS: one workgroup per CU - the grid has at most one per CU anyway)
constexpr int S_AHEAD = S_NST - 1;             // K tiles in flight

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool F16, bool SPLITK>
__global__ __launch_bounds__(256, 1) void gemm_nt64_kernel(GemmNtArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntn = (p.N + SB - 1) / SB;
    const int ntiles = SPLITK ? (int)gridDim.x / p.ksplit : (int)gridDim.x;
    const int split = SPLITK ? (int)blockIdx.x / ntiles : 0;
    const int tile = SPLITK ? (int)blockIdx.x % ntiles : xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / ntn) * SB, n0 = (tile % ntn) * SB;
    const int wm = wid >> 1, wn = wid & 1;

    // LDS-DMA staging: wave w owns pieces 2w, 2w + 1 of each operand tile (piece = 8 rows x 128 B)
    const int prow = lane >> 3;
    const int chunk = (lane & 7) ^ prow;
    const bf16_t* ga[2];
    const bf16_t* gb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wid * 2 + i) * 8 + prow;
        const int ra = min(m0 + r, p.M - 1), rb = min(n0 + r, p.N - 1);
        ga[i] = p.A + (size_t)ra * p.lda + chunk * 8;
        gb[i] = p.B + (size_t)rb * p.ldb + chunk * 8;
    }
    auto stage = [&](int s, int k0) {
        char* base = smem + s * S_STAGE + wid * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(ga[i] + k0), LDS_PTR(base + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gb[i] + k0), LDS_PTR(base + S_TILE + i * 1024), 16, 0, 0);
        }
    };
    const int frow = lane & 15, fq = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int slot = (4 * s + fq) ^ (frow & 7);
        a_off[s] = (wm * 32 + frow) * 128 + slot * 16;
        b_off[s] = S_TILE + (wn * 32 + frow) * 128 + slot * 16;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // K range of this workgroup (SPLITK: a contiguous share of the K tiles, as in the 128 x 128 kernel)
    const int nk_all = p.K / BK;
    const int kper = SPLITK ? (nk_all + p.ksplit - 1) / p.ksplit : nk_all;
    const int kbeg = split * kper;
    const int nk = max(0, min(nk_all, kbeg + kper) - kbeg);
    for (int t = 0; t < S_AHEAD && t < nk; ++t) stage(t, (kbeg + t) * BK);
    for (int kt = 0; kt < nk; ++kt) {
        // tiles kt + 1 .. min(nk - 1, kt + 6) may stay in flight (4 requests per wave and tile; they complete in order)
        switch (min(nk - 1, kt + S_AHEAD - 1) - kt) {
            case 0: wait_vmcnt<0>(); break;
            case 1: wait_vmcnt<4>(); break;
            case 2: wait_vmcnt<8>(); break;
            case 3: wait_vmcnt<12>(); break;
            case 4: wait_vmcnt<16>(); break;
            case 5: wait_vmcnt<20>(); break;
            default: wait_vmcnt<24>(); break;
        }
        __syncthreads();
        if (kt + S_AHEAD < nk) stage((kt + S_AHEAD) % S_NST, (kbeg + kt + S_AHEAD) * BK);
        const char* sb = smem + (kt % S_NST) * S_STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = *(const bf16x8*)(sb + a_off[s] + t * 16 * 128);
                bfr[t] = *(const bf16x8*)(sb + b_off[s] + t * 16 * 128);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = gemm_mfma<F16 ? EPI_F16IN : 0>(bfr[nt], af[mt], acc[mt][nt]);
        }
    }
    if constexpr (SPLITK) {
        // raw partial sums of this K range for splitk_finish_kernel (accumulator layout: 4 consecutive columns of one row per lane and tile)
        float* out = p.slabs + (size_t)split * p.M * p.N;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = m0 + wm * 32 + mt * 16 + (lane & 15);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int n = n0 + wn * 32 + nt * 16 + 4 * (lane >> 4);
                if (m < p.M && n < p.N) *(f32x4*)(out + (size_t)m * p.N + n) = acc[mt][nt];
            }
        }
    } else {
        __syncthreads();
        gemm_nt_epilogue<2, 2, EPI_GENERIC>(p, acc, m0 + wm * 32, n0 + wn * 32, lane, (float*)smem + wid * (32 * 36));
    }
}

// Does the 64 x 64 kernel take this shape, and in how many K splits?  0: no; 1: one launch (a grid that fits the chip in one round, a K range its
// prefetch depth covers: a workgroup has at most 112 KiB in flight, i.e. ~50 GB/s - a longer K range needs more workgroups, not more time);
// n > 1: n K ranges of at least 8 K tiles + splitk_finish_kernel, where that still fits one round (K = 2304 / 3072 at 256 x 768: 4-5 ranges
// and 4-5 slabs to add instead of the 128 x 128 kernel's 12).
static int nt64_splits(int M, int N, int K) {
    if (g_cldrd_tune_nt64 == 0 || g_cldrd_tune_splitk != 0) return 0;          // a forced split count means the 128 x 128 kernel (tests)
    if (K % BK != 0 || M >= 1024 || N % 8 != 0) return 0;
    const long long tiles = (long long)((M + SB - 1) / SB) * ((N + SB - 1) / SB);
    if (tiles > 256) return 0;
    if (K <= 1024) return 1;
    const int nk = K / BK;
    int ks = (int)(256 / tiles);
    if (ks > nk / 8) ks = nk / 8;
    return ks >= 4 ? ks : 0;
}

// Split-K, second half: every thread owns 8 consecutive columns of a row, sums the ksplit partials in a fixed order and runs the
// same fused epilogue (run-time flags) as the one-pass kernels.  F16OUT: a 16-bit C is fp16 (the fp16 format of the forward kernels).
template <bool F16OUT>
__global__ __launch_bounds__(256) void splitk_finish_kernel(GemmNtArgs p) {
    const int n8 = p.N / 8;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)p.M * n8) return;
    const int m = (int)(idx / n8), n = (int)(idx % n8) * 8;
    const EpiFlags<EPI_GENERIC> fl(p);
    float v[8];
    {
        const float* s0 = p.slabs + (size_t)m * p.N + n;
        const f32x4 a = *(const f32x4*)s0, b = *(const f32x4*)(s0 + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        for (int k = 1; k < p.ksplit; ++k) {
            const float* sk = s0 + (size_t)k * p.M * p.N;
            const f32x4 c = *(const f32x4*)sk, d = *(const f32x4*)(sk + 4);
            v[0] += c[0]; v[1] += c[1]; v[2] += c[2]; v[3] += c[3]; v[4] += d[0]; v[5] += d[1]; v[6] += d[2]; v[7] += d[3];
        }
    }
    float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (fl.bias) {
        const float4 b0 = *(const float4*)(p.bias + n), b1 = *(const float4*)(p.bias + n + 4);
        bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
    }
    uint4 res = make_uint4(0, 0, 0, 0), resh = make_uint4(0, 0, 0, 0), gp = make_uint4(0, 0, 0, 0);
    if (fl.residual) {
        if (fl.res32) {
            const uint4* rp = (const uint4*)((const float*)p.residual + (size_t)m * p.ldr + n);
            res = rp[0];
            resh = rp[1];
            if (fl.resln) {       // residual = LayerNorm(pre-LN sum): the expression of ln_fwd_kernel / gemm_nt_epilogue
                const float mu = p.ln_mean[m], rs = p.ln_rstd[m];
                uint32_t* lo = (uint32_t*)&res;
                uint32_t* hi = (uint32_t*)&resh;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lo[j] = __float_as_uint((__uint_as_float(lo[j]) - mu) * rs * p.ln_gamma[n + j] + p.ln_beta[n + j]);
                    hi[j] = __float_as_uint((__uint_as_float(hi[j]) - mu) * rs * p.ln_gamma[n + 4 + j] + p.ln_beta[n + 4 + j]);
                }
            }
        } else {
            res = *(const uint4*)((const bf16_t*)p.residual + (size_t)m * p.ldr + n);
        }
    }
    if (fl.gelugrad) gp = *(const uint4*)(p.gelu_pre + (size_t)m * p.ldc + n);
    gemm_nt_apply8<EPI_GENERIC>(p, fl, v, m, n, bias8, res, resh, gp, fl.dropout ? (p.seed_base ? p.seed + *p.seed_base : p.seed) : 0ull);
}

// K splits of the small-M kernel for a shape (1: one pass).  The 128 x 128 kernel holds two workgroups per CU: aim at ~1.5 per CU with
// at least 4 K tiles each, only where the one-pass grid leaves most of the chip idle.
static int splitk_choice(int M, int N, int K) {
    const int force = g_cldrd_tune_splitk;              // 1: never split, n > 1: n splits (cldrd_set_tuning: tests flip it in-process)
    if (force == 1) return 1;
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN), nk = K / BK;
    if (M >= 1024 || tiles >= 128 || nk < 8) return 1;
    int ks = 384 / tiles;
    if (ks > nk / 4) ks = nk / 4;
    if (force > 1) ks = force < nk ? force : nk;
    return ks < 2 ? 1 : ks;
}

template <int EPI>
int launch_nt(const GemmNtArgs& a, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
        attr_set = true;
    }
    const int nblk = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_nt_kernel<EPI>), dim3(nblk), dim3(256), 2 * STAGE_BYTES, st, a);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

}  // namespace

int cldrd_gemm_nt_ring_dispatch(const GemmNtArgs& a, int force_bn, hipStream_t st);   // gemm_nt_ring.hip
int cldrd_gemm_nt_ring_scan(const GemmNtArgs& a, hipStream_t st);                     // gemm_nt_ring.hip
int cldrd_topk_scan_stream(const void* Q, const void* P, int nq, long long rows, int d, const float* thr, int* counts, int* cand_rows,
                           float* cand_scores, int cap, int f16, hipStream_t st);   // topk.hip

extern "C" int cldrd_gemm_nt16_ws(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                     const float* bias, const void* residual, int ldr, void* preact, const void* gelu_pre,
                                     int act, float alpha, float dropout_p, unsigned long long seed, int out_f32, int res_f32,
                                     int io_f16, const float* ln_mean, const float* ln_rstd, const float* ln_gamma,
                                     const float* ln_beta, void* c_copy_bf16, float* workspace, size_t workspace_bytes, void* stream);
extern "C" int cldrd_gemm_nt16_ln(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                     const float* bias, const void* residual, int ldr, void* preact, const void* gelu_pre,
                                     int act, float alpha, float dropout_p, unsigned long long seed, int out_f32, int res_f32,
                                     int io_f16, const float* ln_mean, const float* ln_rstd, const float* ln_gamma,
                                     const float* ln_beta, void* stream);

// Bytes of workspace with which cldrd_gemm_nt16_ws splits the K range of this shape over several workgroups (0: it does not)
extern "C" size_t cldrd_gemm_nt_splitk_workspace(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || K % BK != 0 || N % 8 != 0) return 0;
    if (const int k64 = nt64_splits(M, N, K)) return k64 > 1 ? (size_t)k64 * M * N * sizeof(float) : 0;
    const int ks = splitk_choice(M, N, K);
    return ks > 1 ? (size_t)ks * M * N * sizeof(float) : 0;
}

extern "C" int cldrd_gemm_nt16(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                  const float* bias, const void* residual, int ldr, void* preact, const void* gelu_pre,
                                  int act, float alpha, float dropout_p, unsigned long long seed, int out_f32, int res_f32,
                                  int io_f16, void* stream) {
    return cldrd_gemm_nt16_ln(A, B, C, M, N, K, lda, ldb, ldc, bias, residual, ldr, preact, gelu_pre, act, alpha, dropout_p, seed,
                                 out_f32, res_f32, io_f16, nullptr, nullptr, nullptr, nullptr, stream);
}

// The same with the fp32 residual given as a LayerNorm still to be applied (GemmNtArgs::ln_*): all four pointers or none.
extern "C" int cldrd_gemm_nt16_ln(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                     const float* bias, const void* residual, int ldr, void* preact, const void* gelu_pre,
                                     int act, float alpha, float dropout_p, unsigned long long seed, int out_f32, int res_f32,
                                     int io_f16, const float* ln_mean, const float* ln_rstd, const float* ln_gamma,
                                     const float* ln_beta, void* stream) {
    return cldrd_gemm_nt16_ws(A, B, C, M, N, K, lda, ldb, ldc, bias, residual, ldr, preact, gelu_pre, act, alpha, dropout_p, seed, out_f32,
                                 res_f32, io_f16, ln_mean, ln_rstd, ln_gamma, ln_beta, nullptr, nullptr, 0, stream);
}

// The same with a workspace: small-M problems whose one-pass grid would leave most CUs idle (the CLS-only last layer, the query
// tower: M = 256 rows, K up to 3072 -> 12 workgroups walking 48 K tiles each) are split along K over cldrd_gemm_nt_splitk_workspace()
// bytes of fp32 partials and finished by a second launch (fixed summation order; same epilogue).  workspace = null: one pass.
extern "C" int cldrd_gemm_nt16_ws(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                     const float* bias, const void* residual, int ldr, void* preact, const void* gelu_pre,
                                     int act, float alpha, float dropout_p, unsigned long long seed, int out_f32, int res_f32,
                                     int io_f16, const float* ln_mean, const float* ln_rstd, const float* ln_gamma,
                                     const float* ln_beta, void* c_copy_bf16, float* workspace, size_t workspace_bytes, void* stream) {
    CLDRD_CHECK(io_f16 == 0 || io_f16 == 1 || io_f16 == 3 || io_f16 == 5,
                "gemm_nt: io_f16 is 0 (bf16), 1 (fp16 operands and 16-bit C), 3 (fp16 operands, bf16 C) or 5 (1 + fp16 preact / gelu_pre: the all-fp16 training mode)");
    CLDRD_CHECK(c_copy_bf16 == nullptr || (io_f16 == 1 && !out_f32 && (uintptr_t)c_copy_bf16 % 16 == 0),
                "gemm_nt: the bf16 copy of C goes with fp16 operands and an fp16 C");
    {
        const int nln = (ln_mean != nullptr) + (ln_rstd != nullptr) + (ln_gamma != nullptr) + (ln_beta != nullptr);
        CLDRD_CHECK(nln == 0 || nln == 4, "gemm_nt: ln_mean / ln_rstd / ln_gamma / ln_beta go together");
        CLDRD_CHECK(nln == 0 || (residual != nullptr && res_f32 != 0), "gemm_nt: the LayerNorm-on-the-fly residual needs an fp32 residual");
        CLDRD_CHECK(nln == 0 || (((uintptr_t)ln_gamma % 16 == 0) && ((uintptr_t)ln_beta % 16 == 0)), "gemm_nt: ln_gamma / ln_beta must be 16-byte aligned");
    }
    CLDRD_CHECK(M > 0 && N > 0 && K > 0, "gemm_nt: empty problem");
    CLDRD_CHECK(K % 32 == 0, "gemm_nt: K must be a multiple of 32");
    CLDRD_CHECK(lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && N % 8 == 0 && (residual == nullptr || ldr % 8 == 0),
                "gemm_nt: N, lda, ldb, ldc, ldr must be multiples of 8");
    CLDRD_CHECK(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0) && ((uintptr_t)residual % 16 == 0),
                "gemm_nt: operands must be 16-byte aligned");
    CLDRD_CHECK(dropout_p >= 0.f && dropout_p < 1.f, "gemm_nt: dropout_p out of range");
    CLDRD_CHECK(act >= 0 && act <= 3, "gemm_nt: act must be 0..3 (bit 0 erf-GELU, bit 1 derivative form of preact / gelu_pre)");
    GemmNtArgs a;
    a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.bias = bias; a.residual = residual; a.ldr = ldr; a.res_f32 = (residual != nullptr && res_f32) ? 1 : 0;
    a.ln_mean = ln_mean; a.ln_rstd = ln_rstd; a.ln_gamma = ln_gamma; a.ln_beta = ln_beta;
    a.preact = (bf16_t*)preact; a.gelu_pre = (const bf16_t*)gelu_pre;
    a.act = act; a.alpha = alpha;
    a.drop_thresh = dropout_p > 0.f ? dropout_thresh16(dropout_p) : 0u;
    a.drop_scale = 1.0f / (1.0f - dropout_p);
    a.seed = seed; a.seed_base = g_cldrd_seed_base; a.out_f32 = out_f32;
    a.thr = nullptr; a.counts = nullptr; a.cand_rows = nullptr; a.cand_scores = nullptr; a.cap = 0;
    a.in_f16 = io_f16 ? 1 : 0;
    a.c_bf16 = (io_f16 & 2) ? 1 : 0;
    a.tape_f16 = (io_f16 & 4) ? 1 : 0;
    a.c_copy = (bf16_t*)c_copy_bf16;
    if (const int k64 = (io_f16 && gelu_pre && !(io_f16 & 4)) ? 0 : nt64_splits(M, N, K);
        k64 == 1 || (k64 > 1 && workspace != nullptr && workspace_bytes >= (size_t)k64 * M * N * sizeof(float) && (uintptr_t)workspace % 16 == 0)) {
        a.ksplit = k64;
        a.slabs = workspace;
        const int nblk = ((M + SB - 1) / SB) * ((N + SB - 1) / SB) * k64;
        hipStream_t st = (hipStream_t)stream;
        static bool set64 = false;
        if (!set64) {
            (void)hipFuncSetAttribute((const void*)gemm_nt64_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, S_NST * S_STAGE);
            (void)hipFuncSetAttribute((const void*)gemm_nt64_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, S_NST * S_STAGE);
            (void)hipFuncSetAttribute((const void*)gemm_nt64_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, S_NST * S_STAGE);
            (void)hipFuncSetAttribute((const void*)gemm_nt64_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, S_NST * S_STAGE);
            set64 = true;
        }
        if (k64 == 1) {
            if (io_f16) hipLaunchKernelGGL((gemm_nt64_kernel<true, false>), dim3(nblk), dim3(256), S_NST * S_STAGE, st, a);
            else hipLaunchKernelGGL((gemm_nt64_kernel<false, false>), dim3(nblk), dim3(256), S_NST * S_STAGE, st, a);
            CLDRD_LAUNCH_CHECK();
            return 0;
        }
        if (io_f16) hipLaunchKernelGGL((gemm_nt64_kernel<true, true>), dim3(nblk), dim3(256), S_NST * S_STAGE, st, a);
        else hipLaunchKernelGGL((gemm_nt64_kernel<false, true>), dim3(nblk), dim3(256), S_NST * S_STAGE, st, a);
        CLDRD_LAUNCH_CHECK();
        const size_t nthr = (size_t)M * (N / 8);
        if (io_f16) hipLaunchKernelGGL(splitk_finish_kernel<true>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(splitk_finish_kernel<false>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, a);
        CLDRD_LAUNCH_CHECK();
        return 0;
    }
    if (workspace != nullptr && K % BK == 0 && N % 8 == 0 && !(io_f16 && gelu_pre && !(io_f16 & 4))) {
        const int ks = splitk_choice(M, N, K);
        if (ks > 1 && workspace_bytes >= (size_t)ks * M * N * sizeof(float) && (uintptr_t)workspace % 16 == 0) {
            a.ksplit = ks;
            a.slabs = workspace;
            const int nblk = ((M + BM - 1) / BM) * ((N + BN - 1) / BN) * ks;
            hipStream_t st = (hipStream_t)stream;
            if (io_f16) {
                static bool set16 = false;
                if (!set16) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI_SPLITK | EPI_F16IN>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES); set16 = true; }
                hipLaunchKernelGGL((gemm_nt_kernel<EPI_SPLITK | EPI_F16IN>), dim3(nblk), dim3(256), 2 * STAGE_BYTES, st, a);
            } else {
                static bool set = false;
                if (!set) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI_SPLITK>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES); set = true; }
                hipLaunchKernelGGL((gemm_nt_kernel<EPI_SPLITK>), dim3(nblk), dim3(256), 2 * STAGE_BYTES, st, a);
            }
            CLDRD_LAUNCH_CHECK();
            const size_t nthr = (size_t)M * (N / 8);
            if (io_f16) hipLaunchKernelGGL(splitk_finish_kernel<true>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, a);
            else hipLaunchKernelGGL(splitk_finish_kernel<false>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, a);
            CLDRD_LAUNCH_CHECK();
            return 0;
        }
    }
    if (io_f16) {
        // fp16 operands / 16-bit output (the high-precision forward of the query tower): small-M kernel, forward flavours only
        CLDRD_CHECK(K % BK == 0, "gemm_nt: K must be a multiple of 64");
        CLDRD_CHECK(gelu_pre == nullptr || (io_f16 & 4), "gemm_nt: gelu_pre with fp16 operands needs the fp16 tape format (io_f16 = 5)");
        {
            const int rc = cldrd_gemm_nt_ring_dispatch(a, 0, (hipStream_t)stream);      // large-M FFN forward flavours (gemm_nt_ring16.hip)
            if (rc >= 0) return rc;
        }
        switch (epi_flavour(a)) {
            case EPI_F16IN | EPI_BIAS | EPI_PREACT | EPI_GELU | EPI_DGELU:
                return launch_nt<EPI_F16IN | EPI_BIAS | EPI_PREACT | EPI_GELU | EPI_DGELU>(a, (hipStream_t)stream);
            case EPI_F16IN | EPI_BIAS: return launch_nt<EPI_F16IN | EPI_BIAS>(a, (hipStream_t)stream);
            case EPI_F16IN | EPI_BIAS | EPI_GELU: return launch_nt<EPI_F16IN | EPI_BIAS | EPI_GELU>(a, (hipStream_t)stream);
            case EPI_F16IN | EPI_BIAS | EPI_RESIDUAL | EPI_RES32 | EPI_F32:
                return launch_nt<EPI_F16IN | EPI_BIAS | EPI_RESIDUAL | EPI_RES32 | EPI_F32>(a, (hipStream_t)stream);
            case EPI_F16IN | EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL | EPI_RES32 | EPI_F32:
                return launch_nt<EPI_F16IN | EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL | EPI_RES32 | EPI_F32>(a, (hipStream_t)stream);
            case EPI_F16IN | EPI_BIAS | EPI_RESIDUAL | EPI_RES32 | EPI_F32 | EPI_RESLN:
                return launch_nt<EPI_F16IN | EPI_BIAS | EPI_RESIDUAL | EPI_RES32 | EPI_F32 | EPI_RESLN>(a, (hipStream_t)stream);
            case EPI_F16IN | EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL | EPI_RES32 | EPI_F32 | EPI_RESLN:
                return launch_nt<EPI_F16IN | EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL | EPI_RES32 | EPI_F32 | EPI_RESLN>(a, (hipStream_t)stream);
            // round 4, the all-fp16 training mode: backward flavours of the small-M kernel (CLS-only last layer, query tower)
            case EPI_F16IN: return launch_nt<EPI_F16IN>(a, (hipStream_t)stream);
            case EPI_F16IN | EPI_GELUGRAD | EPI_DGELU: return launch_nt<EPI_F16IN | EPI_GELUGRAD | EPI_DGELU>(a, (hipStream_t)stream);
            case EPI_F16IN | EPI_F32: return launch_nt<EPI_F16IN | EPI_F32>(a, (hipStream_t)stream);
            case EPI_F16IN | EPI_RESIDUAL | EPI_RES32 | EPI_F32: return launch_nt<EPI_F16IN | EPI_RESIDUAL | EPI_RES32 | EPI_F32>(a, (hipStream_t)stream);
            default: return cldrd_set_error("gemm_nt: this epilogue combination is not built for the fp16 format");
        }
    }
    // large-M shapes go to the 256-row ring kernel; development build: CLDRD_GEMM_TILE=128|192|256 forces a variant
    const int force = CLDRD_DEV_INT("CLDRD_GEMM_TILE", 0);
    if (force != 128) {
        const int rc = cldrd_gemm_nt_ring_dispatch(a, force, (hipStream_t)stream);
        if (rc >= 0) return rc;
    }
    CLDRD_CHECK(K % BK == 0, "gemm_nt: K must be a multiple of 64 for M < 1024 or N not a multiple of 192/256");
    switch (epi_flavour(a)) {
        case 0: return launch_nt<0>(a, (hipStream_t)stream);
        case EPI_BIAS: return launch_nt<EPI_BIAS>(a, (hipStream_t)stream);
        case EPI_BIAS | EPI_PREACT | EPI_GELU: return launch_nt<EPI_BIAS | EPI_PREACT | EPI_GELU>(a, (hipStream_t)stream);
        case EPI_BIAS | EPI_PREACT | EPI_GELU | EPI_DGELU: return launch_nt<EPI_BIAS | EPI_PREACT | EPI_GELU | EPI_DGELU>(a, (hipStream_t)stream);
        case EPI_GELUGRAD | EPI_DGELU: return launch_nt<EPI_GELUGRAD | EPI_DGELU>(a, (hipStream_t)stream);
        case EPI_BIAS | EPI_RESIDUAL: return launch_nt<EPI_BIAS | EPI_RESIDUAL>(a, (hipStream_t)stream);
        case EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL: return launch_nt<EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL>(a, (hipStream_t)stream);
        case EPI_GELUGRAD: return launch_nt<EPI_GELUGRAD>(a, (hipStream_t)stream);
        case EPI_RESIDUAL: return launch_nt<EPI_RESIDUAL>(a, (hipStream_t)stream);
        case EPI_F32: return launch_nt<EPI_F32>(a, (hipStream_t)stream);
        case EPI_RESIDUAL | EPI_RES32 | EPI_F32: return launch_nt<EPI_RESIDUAL | EPI_RES32 | EPI_F32>(a, (hipStream_t)stream);     // fp32 gradient stream
        case EPI_BIAS | EPI_RESIDUAL | EPI_RES32 | EPI_F32: return launch_nt<EPI_BIAS | EPI_RESIDUAL | EPI_RES32 | EPI_F32>(a, (hipStream_t)stream);
        case EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL | EPI_RES32 | EPI_F32:
            return launch_nt<EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL | EPI_RES32 | EPI_F32>(a, (hipStream_t)stream);
        case EPI_BIAS | EPI_RESIDUAL | EPI_RES32 | EPI_F32 | EPI_RESLN:
            return launch_nt<EPI_BIAS | EPI_RESIDUAL | EPI_RES32 | EPI_F32 | EPI_RESLN>(a, (hipStream_t)stream);
        case EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL | EPI_RES32 | EPI_F32 | EPI_RESLN:
            return launch_nt<EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL | EPI_RES32 | EPI_F32 | EPI_RESLN>(a, (hipStream_t)stream);
        default: return launch_nt<EPI_GENERIC>(a, (hipStream_t)stream);
    }
}


// Top-k scan over one index shard: scores = Q[nq,d] . P[rows,d]^T on bf16 MFMA; every (query, row) with score >= thr[query]
// is appended to the query's candidate list (counts must be zeroed by the caller; counts[q] may exceed cap = overflow).
static int scan_filter_impl(const void* Q, const void* P, int nq, long long rows, int d, const float* thr, int* counts,
                            int* cand_rows, float* cand_scores, int cap, int f16, void* stream, bool tiled) {
    CLDRD_CHECK(nq > 0 && rows > 0 && rows < 2147483647LL && d % BK == 0 && cap > 0, "topk_scan_filter: bad arguments");
    CLDRD_CHECK(((uintptr_t)Q % 16 == 0) && ((uintptr_t)P % 16 == 0), "topk_scan_filter: operands must be 16-byte aligned");
    GemmNtArgs a;
    a.A = (const bf16_t*)Q; a.B = (const bf16_t*)P; a.C = nullptr;
    a.M = nq; a.N = (int)rows; a.K = d; a.lda = d; a.ldb = d; a.ldc = 0;
    a.bias = nullptr; a.residual = nullptr; a.ldr = 0; a.preact = nullptr; a.gelu_pre = nullptr; a.act = 0; a.alpha = 1.0f;
    a.drop_thresh = 0; a.drop_scale = 1.0f; a.seed = 0; a.out_f32 = 0;
    a.thr = thr; a.counts = counts; a.cand_rows = cand_rows; a.cand_scores = cand_scores; a.cap = cap;
    a.in_f16 = f16 ? 1 : 0;
    if (!tiled && !CLDRD_DEV_INT("CLDRD_SCAN_GEMM", 0)) {          // development build: 1 forces the tiled-GEMM scan (A/B experiments)
        const int rc = cldrd_topk_scan_stream(Q, P, nq, rows, d, thr, counts, cand_rows, cand_scores, cap, f16, (hipStream_t)stream);
        if (rc >= 0) return rc;
    }
    if (rows >= 4096 && nq <= 128 && CLDRD_DEV_INT("CLDRD_GEMM_TILE", 0) != 128 && (double)rows * d * 2.0 < 4.0e9) {
        // large shard: index rows are the M dimension of the 256-row ring kernel, the (<= 128) queries its N tile
        a.A = (const bf16_t*)P; a.B = (const bf16_t*)Q; a.M = (int)rows; a.N = nq;
        return cldrd_gemm_nt_ring_scan(a, (hipStream_t)stream);
    }
    return f16 ? launch_nt<EPI_FILTER | EPI_F16IN>(a, (hipStream_t)stream) : launch_nt<EPI_FILTER>(a, (hipStream_t)stream);
}

extern "C" int cldrd_topk_scan_filter(const void* Q, const void* P, int nq, long long rows, int d, const float* thr, int* counts,
                                      int* cand_rows, float* cand_scores, int cap, int f16, void* stream) {
    return scan_filter_impl(Q, P, nq, rows, d, thr, counts, cand_rows, cand_scores, cap, f16, stream, false);
}

// Same contract, always through the tiled kernels (hits go straight to the global lists: never sets counts[nq]).
extern "C" int cldrd_topk_scan_filter_tiled(const void* Q, const void* P, int nq, long long rows, int d, const float* thr,
                                            int* counts, int* cand_rows, float* cand_scores, int cap, int f16, void* stream) {
    return scan_filter_impl(Q, P, nq, rows, d, thr, counts, cand_rows, cand_scores, cap, f16, stream, true);
}
