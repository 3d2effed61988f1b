// Weight gradient: dW[N1,N2] (+)= sum_m A[m,N1] * B[m,N2]      (A = dY, B = X; bf16 in, fp32 out)
//
// The reduction index m (tokens) is the ROW index of both operands, so neither is K-contiguous: the
// MFMA fragments are fetched with gfx950's transposing LDS read (ds_read_b64_tr_b16) from row-major
// [64 tokens][128 cols] tiles that arrive by LDS-DMA.  The 16-B slot index is XOR-swizzled with
// f(m) = 2*((m&3) | ((m>>3)&1)<<2) (on the DMA source address and on the read address), which makes
// every transposed read touch all 64 banks exactly once.  M = tokens is huge and N1 x N2 small, so the
// token range is split over `splits` blocks per tile; each writes an fp32 slab and a second pass sums
// the slabs in a fixed order (bitwise reproducible, no float atomics).
//
// Contract: A and B must have ceil(M/64)*64 rows allocated and rows >= M must be zero (the host
// allocates activation / gradient buffers that way and no kernel writes past row M).
#include "common.h"

namespace {

constexpr int BT = 128;          // output tile is BT x BT
constexpr int BK = 64;           // tokens per LDS stage
constexpr int TILE_BYTES = BK * BT * 2;       // 16 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;

__device__ __forceinline__ int swz(int m) { return 2 * ((m & 3) | (((m >> 3) & 1) << 2)); }

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                          float* __restrict__ slabs, int M, int N1, int N2,
                                                          int lda, int ldb, int splits, int ksteps_per_split) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nt2 = N2 / BT;
    const int ntiles = (N1 / BT) * nt2;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int split = id / ntiles, tile = id % ntiles;      // all tiles of one split are neighbours: they share A/B rows
    const int c1 = (tile / nt2) * BT, c2 = (tile % nt2) * BT;
    const int ktotal = (M + BK - 1) / BK;
    const int kbeg = split * ksteps_per_split;
    const int kend = min(ktotal, kbeg + ksteps_per_split);
    const int wm = wid >> 1, wn = wid & 1;

    // LDS-DMA: piece = 4 token rows x 256 B; wave w owns pieces 4w..4w+3 -> rows 16w + 4j + (lane >> 4)
    const bf16_t* ga[4];
    const bf16_t* gb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 16 * wid + 4 * j + (lane >> 4);
        const int chunk = (lane & 15) ^ swz(r);
        ga[j] = A + (size_t)r * lda + c1 + chunk * 8;
        gb[j] = B + (size_t)r * ldb + c2 + chunk * 8;
    }
    auto stage = [&](int s, int kt) {
        char* base = smem + s * STAGE_BYTES + wid * 4096;
        const size_t ra = (size_t)kt * BK * lda, rb = (size_t)kt * BK * ldb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(ga[j] + ra), LDS_PTR(base + j * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gb[j] + rb), LDS_PTR(base + TILE_BYTES + j * 1024), 16, 0, 0);
        }
    };

    // transposed-read addressing: 16-lane group g reads token rows 8g+q (+4), lane (4q+pp) supplies cols 4pp..4pp+3
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    auto tr_off = [&](int base_col, int m, int tile_base) {
        return tile_base + m * 256 + (((base_col >> 3) ^ swz(m)) * 16) + (base_col & 7) * 2;
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (kbeg < kend) {
        stage(0, kbeg);
        for (int kt = kbeg; kt < kend; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < kend) stage((kt - kbeg + 1) & 1, kt + 1);
            const char* sb = smem + ((kt - kbeg) & 1) * STAGE_BYTES;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 af[4], bfr[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    bf16x4 lo, hi;
                    const int m0 = 32 * s + 8 * g + q, m1 = m0 + 4;
                    const int ca = wm * 64 + t * 16 + 4 * pp, cb = wn * 64 + t * 16 + 4 * pp;
                    lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(sb + tr_off(ca, m0, 0)));
                    hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(sb + tr_off(ca, m1, 0)));
                    af[t] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(sb + tr_off(cb, m0, TILE_BYTES)));
                    hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(sb + tr_off(cb, m1, TILE_BYTES)));
                    bfr[t] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int t1 = 0; t1 < 4; ++t1)
#pragma unroll
                    for (int t2 = 0; t2 < 4; ++t2)
                        // swapped operands: D'[n2][n1] so the lane's 4 accumulators run along n2 (contiguous in dW rows)
                        acc[t1][t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[t2], af[t1], acc[t1][t2], 0, 0, 0);
            }
        }
    }
    float* slab = slabs + (size_t)split * N1 * N2;
#pragma unroll
    for (int t1 = 0; t1 < 4; ++t1) {
        const int n1 = c1 + wm * 64 + t1 * 16 + (lane & 15);
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) {
            const int n2 = c2 + wn * 64 + t2 * 16 + 4 * (lane >> 4);
            *(float4*)(slab + (size_t)n1 * N2 + n2) = make_float4(acc[t1][t2][0], acc[t1][t2][1], acc[t1][t2][2], acc[t1][t2][3]);
        }
    }
}

// out[i] (+)= sum_s slabs[s][i], fixed order.
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ out, size_t n4, int splits,
                                    size_t slab_stride4, int accumulate) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 s = ((const float4*)slabs)[i];
        for (int k = 1; k < splits; ++k) {
            const float4 t = ((const float4*)slabs)[i + k * slab_stride4];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        if (accumulate) {
            const float4 o = ((float4*)out)[i];
            s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        }
        ((float4*)out)[i] = s;
    }
}

}  // namespace

extern "C" int cldrd_wgrad_splits(int M, int N1, int N2) {
    const int tiles = (N1 / BT) * (N2 / BT);
    const int ktotal = (M + BK - 1) / BK;
    int splits = (512 + tiles - 1) / tiles;          // ~2 blocks per CU
    if (splits > ktotal) splits = ktotal;
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    return splits;
}

extern "C" int cldrd_wgrad_bf16(const void* A, const void* B, float* dW, int M, int N1, int N2, int lda, int ldb,
                                float* workspace, size_t workspace_bytes, int accumulate, void* stream) {
    CLDRD_CHECK(M > 0, "wgrad: empty problem");
    CLDRD_CHECK(N1 % BT == 0 && N2 % BT == 0, "wgrad: N1 and N2 must be multiples of 128");
    CLDRD_CHECK(lda % 8 == 0 && ldb % 8 == 0, "wgrad: lda/ldb must be multiples of 8");
    CLDRD_CHECK(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)dW % 16 == 0) && ((uintptr_t)workspace % 16 == 0),
                "wgrad: operands must be 16-byte aligned");
    const int splits = cldrd_wgrad_splits(M, N1, N2);
    CLDRD_CHECK(workspace_bytes >= (size_t)splits * N1 * N2 * sizeof(float), "wgrad: workspace too small");
    const int ktotal = (M + BK - 1) / BK;
    const int kps = (ktotal + splits - 1) / splits;
    const int tiles = (N1 / BT) * (N2 / BT);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles * splits), dim3(256), 2 * STAGE_BYTES, (hipStream_t)stream,
                       (const bf16_t*)A, (const bf16_t*)B, workspace, M, N1, N2, lda, ldb, splits, kps);
    CLDRD_LAUNCH_CHECK();
    const size_t n4 = (size_t)N1 * N2 / 4;
    const int rb = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(rb), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, dW, n4,
                       splits, n4, accumulate);
    CLDRD_LAUNCH_CHECK();
    return 0;
}
