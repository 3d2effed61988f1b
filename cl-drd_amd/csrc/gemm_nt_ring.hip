// NT GEMM, large-M variant: C[M,N] = epilogue(alpha * A[M,K] . B[N,K]^T), bf16 in, fp32 accumulate.
//
// Why a second kernel: the 128x128 tile of gemm_nt.hip needs 64 B/clk/CU of L2->LDS traffic at full MFMA rate
// and exposes the load latency once per K tile; both cap it near 16 % of peak on the encoder shapes
// (profiles/r01_kernel_stats_v1.csv).  Here:
//   * 256 x BN output tile (BN = 256 or 192), 512 threads = 2 x 4 waves of 128 x BN/4: 32-37 B/clk/CU;
//   * K is consumed in 32-deep slices through a 4-slot LDS ring filled by LDS-DMA (global_load_lds, 16 B/lane);
//     three slices stay in flight ACROSS the per-slice barrier: counted s_waitcnt vmcnt(N) + raw s_barrier,
//     never __syncthreads (which would drain the DMA queue) - cdna_hip_programming.md section 5 T3/T4;
//   * 64-B LDS rows (one 16x16x32 k-step); the 16-B chunk index is XOR-swizzled with perm[(row>>2)&3],
//     perm = {0,2,3,1}, on the DMA source address and on the ds_read_b128 address: every 16-lane read group
//     touches 16 distinct 16-B slots of the 256-B bank row (conflict-free);
//   * BN = 192 exists because N = 768 / 2304 / 3072 with M = 32768 then give 512 / 1536 / 2048 tiles:
//     whole multiples of the 256 CUs at one workgroup per CU.
#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BM = 256, BK = 32, NSLOT = 4;
constexpr int A_BYTES = BM * BK * 2;     // 16 KiB

__device__ __forceinline__ int swz4(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }   // {0,2,3,1}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

template <int BN>
__global__ __launch_bounds__(512, 2) void gemm_nt_ring_kernel(GemmNtArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int B_BYTES = BN * BK * 2;
    constexpr int SLOT = A_BYTES + B_BYTES;
    constexpr int NT = BN / 64;              // 16-col MFMA tiles per wave
    constexpr int WN = BN / 4;               // wave tile width
    constexpr int BPIECES = BN / 16;         // 1-KiB pieces in the B slice
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntn = p.N / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int wm = wid >> 2, wn = wid & 3;

    // ---- LDS-DMA: piece = 16 rows x 64 B; lane -> row (lane >> 2), LDS chunk (lane & 3), source chunk swizzled ----
    const int prow = lane >> 2;
    const int schunk = (lane & 3) ^ swz4(prow);
    const bool has_b = (2 * wid < BPIECES);
    const bf16_t* ga[2];
    const bf16_t* gb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (2 * wid + i) * 16 + prow;
        ga[i] = p.A + (size_t)min(m0 + r, p.M - 1) * p.lda + schunk * 8;
        gb[i] = p.B + (size_t)min(n0 + r, p.N - 1) * p.ldb + schunk * 8;
    }
    auto stage = [&](int slot, int kt) {
        char* base = smem + slot * SLOT + wid * 2048;
        const int k0 = kt * BK;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(ga[0] + k0), LDS_PTR(base), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(ga[1] + k0), LDS_PTR(base + 1024), 16, 0, 0);
        if (has_b) {
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gb[0] + k0), LDS_PTR(base + A_BYTES), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gb[1] + k0), LDS_PTR(base + A_BYTES + 1024), 16, 0, 0);
        }
    };

    // ---- fragment addressing: row (lane & 15) of a 16-row tile, chunk (lane >> 4) ^ swz ----
    const int frow = lane & 15;
    const int fchunk = ((lane >> 4) ^ swz4(frow)) * 16;
    const int a_off = (wm * 128 + frow) * 64 + fchunk;
    const int b_off = A_BYTES + (wn * WN + frow) * 64 + fchunk;

    f32x4 acc[8][NT];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    const int npro = nk < 3 ? nk : 3;
    for (int s = 0; s < npro; ++s) stage(s, s);

    for (int kt = 0; kt < nk; ++kt) {
        // slices kt+1, kt+2 (if they exist) may stay in flight; per-wave DMA count per slice: 4 (or 2 without B pieces)
        const int ahead = min(2, nk - 1 - kt);
        if (has_b) {
            if (ahead == 2) wait_vmcnt<8>(); else if (ahead == 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
        } else {
            if (ahead == 2) wait_vmcnt<4>(); else if (ahead == 1) wait_vmcnt<2>(); else wait_vmcnt<0>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // slice kt visible to all; everyone is done reading slot (kt-1)&3
        asm volatile("" ::: "memory");
        if (kt + 3 < nk) stage((kt + 3) & 3, kt + 3);
        const char* sb = smem + (kt & 3) * SLOT;
        bf16x8 af[8], bfr[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) bfr[t] = *(const bf16x8*)(sb + b_off + t * 16 * 64);
#pragma unroll
        for (int t = 0; t < 8; ++t) af[t] = *(const bf16x8*)(sb + a_off + t * 16 * 64);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt], af[mt], acc[mt][nt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();     // last slice fully consumed by every wave: the ring becomes epilogue scratch
    asm volatile("" ::: "memory");
    gemm_nt_epilogue<8, NT>(p, acc, m0 + wm * 128, n0 + wn * WN, lane, (float*)smem + wid * (32 * (WN + 4)));
}

template <int BN>
int launch_ring(const GemmNtArgs& a, hipStream_t st) {
    constexpr int lds = NSLOT * (A_BYTES + BN * BK * 2);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<BN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int nblk = ((a.M + BM - 1) / BM) * (a.N / BN);
    hipLaunchKernelGGL(gemm_nt_ring_kernel<BN>, dim3(nblk), dim3(512), lds, st, a);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// Returns -1 if this variant does not apply (caller falls back to the 128x128 kernel), else the launch status.
int cldrd_gemm_nt_ring_dispatch(const GemmNtArgs& a, int force_bn, hipStream_t st) {
    if (a.K % BK != 0) return -1;
    int bn = force_bn;
    if (bn == 0) {
        if (a.M < 1024) return -1;
        const long tiles_m = (a.M + BM - 1) / BM;
        const bool ok256 = a.N % 256 == 0, ok192 = a.N % 192 == 0;
        if (!ok256 && !ok192) return -1;
        if (ok256 && ok192) {
            // prefer the tile count that fills whole rounds of 256 CUs; tie -> the larger tile
            const long t256 = tiles_m * (a.N / 256), t192 = tiles_m * (a.N / 192);
            const double e256 = (double)t256 / (double)(((t256 + 255) / 256) * 256), e192 = (double)t192 / (double)(((t192 + 255) / 256) * 256);
            bn = (e192 > e256 + 0.05) ? 192 : 256;
        } else {
            bn = ok256 ? 256 : 192;
        }
    }
    if (bn == 256 && a.N % 256 == 0) return launch_ring<256>(a, st);
    if (bn == 192 && a.N % 192 == 0) return launch_ring<192>(a, st);
    return -1;
}
