// Weight gradient: dW[N1,N2] (+)= sum_m A[m,N1] * B[m,N2]   (A = dY, B = X; bf16 in, fp32 out), optionally with the
// bias gradient db[N1] (+)= sum_m A[m,N1] folded in.
//
// The reduction index m (tokens) is the ROW index of both operands, so neither is K-contiguous: the MFMA fragments are
// fetched with gfx950's transposing LDS read (ds_read_b64_tr_b16) from row-major [64 tokens][cols] tiles that arrive
// by LDS-DMA.  The 16-B slot index is XOR-swizzled with f(m) = 2*((m&3) | ((m>>3)&1)<<2) (on the DMA source address
// and on the read address): every transposed read then touches all 64 banks exactly once.
//
// Tile 256 (n1) x 128 (n2), 512 threads = 4 x 2 waves of 64 x 64, MFMA 16x16x32; 64-token K tiles in two LDS slots.
// Same software pipeline as gemm_nt_ring.hip: per 32-token k-step the A fragments are refilled in place as soon as
// their last MFMA has issued, the B fragments are double-buffered, reads are threaded between the MFMAs
// (sched_group_barrier), one raw s_barrier per K tile, the DMA of K tile t+2 is issued half-way through K tile t.
// M = tokens is huge and N1 x N2 small, so the token range is split over `splits` workgroups per tile; each writes an
// fp32 slab and a second pass sums the slabs in a fixed order (bitwise reproducible, no float atomics).
// The bias gradient costs one extra MFMA per A fragment in the n2 == 0 tiles: B = all-ones.
//
// Contract: A and B must have ceil(M/64)*64 rows allocated and rows >= M must be zero (the host allocates
// activation / gradient buffers that way and no kernel writes past row M).
#include <type_traits>

#include "common.h"

namespace {

constexpr int T2 = 128;                // output tile is (64*WM) x 128, WM = 4 (8 waves) or 2 (4 waves, small models)
constexpr int BK = 64;                 // tokens per LDS slot
constexpr int B_BYTES = BK * T2 * 2;   // 16 KiB

__device__ __forceinline__ int swz(int m) { return 2 * ((m & 3) | (((m >> 3) & 1) << 2)); }

// One MFMA operand = two transposing reads (token rows m and m + 4: same swizzle, fixed byte distance HI).  They are issued
// by hand: hipcc models the ds_read_tr builtin as an LDS access that may alias the LDS-DMA in flight and puts
// s_waitcnt vmcnt(0) in front of it, which drains the two K tiles being prefetched on every k-step (the kernel then runs at
// DMA latency, not at MFMA rate).  The price is that the lgkmcnt waits are ours too, see kstep().
struct Frag { bf16x4 lo, hi; };
template <int OFF, int HI>
__device__ __forceinline__ void tr_issue(Frag& f, uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                 : "=&v"(f.lo), "=&v"(f.hi) : "v"(addr), "n"(OFF), "n"(OFF + HI) : "memory");
}
__device__ __forceinline__ bf16x8 frag8(const Frag& f) { return (bf16x8){f.lo[0], f.lo[1], f.lo[2], f.lo[3], f.hi[0], f.hi[1], f.hi[2], f.hi[3]}; }

template <int WM, bool BIAS>
__global__ __launch_bounds__(128 * WM, 2) void gemm_tn_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                          float* __restrict__ slabs, int M, int N1, int N2, int lda, int ldb,
                                                          int splits, int ksteps_per_split, size_t slab_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T1 = 64 * WM, NW = 2 * WM;
    constexpr int A_BYTES = BK * T1 * 2, SLOT = A_BYTES + B_BYTES;
    constexpr int LPR_A = T1 / 8, RA = 64 / LPR_A;      // lanes per A row, A rows per 1-KiB piece
    constexpr int BPW = 16 / NW;                        // B pieces per wave
    constexpr int NSLOT = 3;                            // K tiles kt+1 and kt+2 stay in flight while kt is consumed
    constexpr int G = 4 + BPW;                          // LDS-DMA instructions per wave per K tile
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nt2 = N2 / T2;
    const int ntiles = (N1 / T1) * nt2;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int split = id / ntiles, tile = id % ntiles;      // the tiles of one split are neighbours: they share A/B rows
    const int c1 = (tile / nt2) * T1, c2 = (tile % nt2) * T2;
    const int ktotal = (M + BK - 1) / BK;
    const int kbeg = split * ksteps_per_split;
    const int nk = min(ktotal, kbeg + ksteps_per_split) - kbeg;
    const int wm = wid >> 1, wn = wid & 1;
    const bool do_bias = BIAS && c2 == 0 && wn == 0;

    // ---- LDS-DMA.  A: piece = RA token rows x (T1*2) B, wave w owns pieces 4w..4w+3.
    //                B: piece = 4 token rows x 256 B, wave w owns pieces BPW*w .. BPW*w + BPW-1.
    uint32_t oa[4], ob[BPW];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = RA * (4 * wid + i) + lane / LPR_A;
        oa[i] = (uint32_t)r * (uint32_t)(lda * 2) + (uint32_t)(c1 * 2) + (uint32_t)(((lane % LPR_A) ^ swz(r)) * 16);
    }
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int r = 4 * (BPW * wid + i) + (lane >> 4);
        ob[i] = (uint32_t)r * (uint32_t)(ldb * 2) + (uint32_t)(c2 * 2) + (uint32_t)(((lane & 15) ^ swz(r)) * 16);
    }
    auto stage = [&](int slot, int kt) {
        char* base = smem + slot * SLOT;
        const char* pa = (const char*)A + (size_t)(kbeg + kt) * BK * lda * 2;
        const char* pb = (const char*)B + (size_t)(kbeg + kt) * BK * ldb * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(pa + oa[i]), LDS_PTR(base + (4 * wid + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BPW; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(pb + ob[i]), LDS_PTR(base + A_BYTES + (BPW * wid + i) * 1024), 16, 0, 0);
    };

    // ---- transposed-read addressing: 16-lane group g reads token rows 8g+q (+4); lane (4q+pp) supplies cols 4pp..4pp+3.
    // Token row m = 32 ks + 8 g + 4 r + q: swz(m) depends on q and g only, so one LDS offset per column tile t covers every
    // k-step (ks: + 32 rows) and both reads (r: + 4 rows) through the instruction's immediate offset.
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int m0 = 8 * g + q;
    uint32_t ra[4], rb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ca = wm * 64 + t * 16 + 4 * pp, cb = wn * 64 + t * 16 + 4 * pp;
        ra[t] = (uint32_t)(uintptr_t)LDS_PTR(smem) + m0 * (T1 * 2) + (((ca >> 3) ^ swz(m0)) * 16) + (ca & 7) * 2;
        rb[t] = (uint32_t)(uintptr_t)LDS_PTR(smem) + A_BYTES + m0 * 256 + (((cb >> 3) ^ swz(m0)) * 16) + (cb & 7) * 2;
    }

    f32x4 acc[4][4];
    f32x4 accb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const short one = (short)0x3F80;     // bf16 1.0
    const bf16x8 ones = (bf16x8){one, one, one, one, one, one, one, one};

    Frag af[4], b0[4], b1[4];
    auto mfma_row = [&](int t1, Frag (&bc)[4]) {
        const bf16x8 a = frag8(af[t1]);
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2)
            // swapped operands: D'[n2][n1] so the lane's 4 accumulators run along n2 (contiguous in dW rows)
            acc[t1][t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag8(bc[t2]), a, acc[t1][t2], 0, 0, 0);
        if (do_bias) accb[t1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, a, accb[t1], 0, 0, 0);
    };
    // LDS returns in issue order.  Issue order per k-step: A0, B'0..B'3, A1, A2, A3 (2 reads each; B' = next k-step's B).
    // Row t1 >= 1 needs A[t1] of the previous k-step: 14 younger reads may stay outstanding; row 0 needs A0 and B: 6.
    auto wait_row0 = [&](Frag (&bc)[4]) {
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(af[0].lo), "+v"(af[0].hi), "+v"(bc[0].lo), "+v"(bc[0].hi), "+v"(bc[1].lo), "+v"(bc[1].hi),
                     "+v"(bc[2].lo), "+v"(bc[2].hi), "+v"(bc[3].lo), "+v"(bc[3].hi));
    };
    // one 32-token k-step over (af, bc); the next fragments come from slot offset noff, k-step NKS of that slot
    auto kstep = [&](Frag (&bc)[4], Frag (&bn)[4], uint32_t noff, auto nks_tag, bool sync, int slot, int kt) {
        constexpr int NKS = decltype(nks_tag)::value;
        constexpr int OA = NKS * 32 * T1 * 2, OB = NKS * 32 * 256;
        wait_row0(bc);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(0, bc);
        __builtin_amdgcn_sched_barrier(0);
        if (sync) {
            // K tile kt+1 must have landed; K tile kt+2 (issued one tile ago) may stay in flight
            if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 3 < nk) stage(slot, kt + 3);          // slot kt % 3: every wave holds tile kt's fragments in registers
            __builtin_amdgcn_sched_barrier(0);
        }
        tr_issue<OA, 4 * T1 * 2>(af[0], ra[0] + noff);
#pragma unroll
        for (int t = 0; t < 4; ++t) tr_issue<OB, 4 * 256>(bn[t], rb[t] + noff);
#pragma unroll
        for (int t1 = 1; t1 < 4; ++t1) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(14)" : "+v"(af[t1].lo), "+v"(af[t1].hi));
            __builtin_amdgcn_sched_barrier(0);
            mfma_row(t1, bc);
            __builtin_amdgcn_sched_barrier(0);
            tr_issue<OA, 4 * T1 * 2>(af[t1], ra[t1] + noff);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using KS0 = std::integral_constant<int, 0>;
    using KS1 = std::integral_constant<int, 1>;

    if (nk > 0) {
        stage(0, 0);
        if (nk > 1) stage(1, 1);
        if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (nk > 2) stage(2, 2);
        tr_issue<0, 4 * T1 * 2>(af[0], ra[0]);
#pragma unroll
        for (int t = 0; t < 4; ++t) tr_issue<0, 4 * 256>(b0[t], rb[t]);
#pragma unroll
        for (int t = 1; t < 4; ++t) tr_issue<0, 4 * T1 * 2>(af[t], ra[t]);
        int cs = 0;                                        // slot of K tile kt
        for (int kt = 0; kt + 1 < nk; ++kt) {
            const int ns = cs == NSLOT - 1 ? 0 : cs + 1;
            kstep(b0, b1, (uint32_t)(cs * SLOT), KS1{}, false, cs, kt);
            kstep(b1, b0, (uint32_t)(ns * SLOT), KS0{}, true, cs, kt);
            cs = ns;
        }
        kstep(b0, b1, (uint32_t)(cs * SLOT), KS1{}, false, 0, nk);
        wait_row0(b1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(0, b1);
#pragma unroll
        for (int t1 = 1; t1 < 4; ++t1) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[t1].lo), "+v"(af[t1].hi));
            mfma_row(t1, b1);
        }
    }

    float* slab = slabs + (size_t)split * slab_stride;
#pragma unroll
    for (int t1 = 0; t1 < 4; ++t1) {
        const int n1 = c1 + wm * 64 + t1 * 16 + (lane & 15);
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) {
            const int n2 = c2 + wn * 64 + t2 * 16 + 4 * (lane >> 4);
            *(float4*)(slab + (size_t)n1 * N2 + n2) = make_float4(acc[t1][t2][0], acc[t1][t2][1], acc[t1][t2][2], acc[t1][t2][3]);
        }
        if (do_bias && lane < 16) slab[(size_t)N1 * N2 + n1] = accb[t1][0];       // every n2 row of D' holds the same column sums
    }
}

// out[i] (+)= sum_s slabs[s][i], fixed order; the tail [n_main, n_main + n_bias) goes to out_bias.
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ out, float* __restrict__ out_bias,
                                    size_t n_main4, size_t n_all4, int splits, size_t slab_stride4, int accumulate) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_all4; i += stride) {
        float4 s = ((const float4*)slabs)[i];
        for (int k = 1; k < splits; ++k) {
            const float4 t = ((const float4*)slabs)[i + k * slab_stride4];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        float4* dst = i < n_main4 ? (float4*)out + i : (float4*)out_bias + (i - n_main4);
        if (accumulate) {
            const float4 o = *dst;
            s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        }
        *dst = s;
    }
}

}  // namespace

static inline int wgrad_t1(int N1) { return N1 % 256 == 0 ? 256 : 128; }

extern "C" int cldrd_wgrad_splits(int M, int N1, int N2) {
    const int tiles = (N1 / wgrad_t1(N1)) * (N2 / T2);
    const int ktotal = (M + BK - 1) / BK;
    if (tiles <= 0) return 1;
    int splits = 256 / tiles;                        // at most one workgroup per CU: a second partial round would double the time
    if (splits > ktotal) splits = ktotal;
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    return splits;
}

template <int WM, bool BIAS>
static int launch_tn(const void* A, const void* B, float* ws, int M, int N1, int N2, int lda, int ldb, int splits, int kps,
                     size_t slab_stride, hipStream_t st) {
    constexpr int lds = 3 * (BK * 64 * WM * 2 + B_BYTES);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<WM, BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int tiles = (N1 / (64 * WM)) * (N2 / T2);
    hipLaunchKernelGGL((gemm_tn_kernel<WM, BIAS>), dim3(tiles * splits), dim3(128 * WM), lds, st, (const bf16_t*)A, (const bf16_t*)B, ws,
                       M, N1, N2, lda, ldb, splits, kps, slab_stride);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

// workspace floats needed: splits * (N1*N2 + N1)
extern "C" int cldrd_wgrad_bf16(const void* A, const void* B, float* dW, float* dbias, int M, int N1, int N2, int lda, int ldb,
                                float* workspace, size_t workspace_bytes, int accumulate, void* stream) {
    CLDRD_CHECK(M > 0, "wgrad: empty problem");
    CLDRD_CHECK(N1 % 128 == 0 && N2 % T2 == 0, "wgrad: N1 and N2 must be multiples of 128");
    CLDRD_CHECK(lda % 8 == 0 && ldb % 8 == 0, "wgrad: lda/ldb must be multiples of 8");
    CLDRD_CHECK(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)dW % 16 == 0) && ((uintptr_t)workspace % 16 == 0),
                "wgrad: operands must be 16-byte aligned");
    CLDRD_CHECK(dbias == nullptr || (uintptr_t)dbias % 16 == 0, "wgrad: dbias must be 16-byte aligned");
    CLDRD_CHECK((double)lda * 2.0 * 72.0 < 4.0e9 && (double)ldb * 2.0 * 72.0 < 4.0e9, "wgrad: row pitch too large");
    const int splits = cldrd_wgrad_splits(M, N1, N2);
    const size_t slab_stride = (size_t)N1 * N2 + (size_t)N1;
    CLDRD_CHECK(workspace_bytes >= (size_t)splits * slab_stride * sizeof(float), "wgrad: workspace too small");
    const int ktotal = (M + BK - 1) / BK;
    const int kps = (ktotal + splits - 1) / splits;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (wgrad_t1(N1) == 256)
        rc = dbias ? launch_tn<4, true>(A, B, workspace, M, N1, N2, lda, ldb, splits, kps, slab_stride, st)
                   : launch_tn<4, false>(A, B, workspace, M, N1, N2, lda, ldb, splits, kps, slab_stride, st);
    else
        rc = dbias ? launch_tn<2, true>(A, B, workspace, M, N1, N2, lda, ldb, splits, kps, slab_stride, st)
                   : launch_tn<2, false>(A, B, workspace, M, N1, N2, lda, ldb, splits, kps, slab_stride, st);
    if (rc) return rc;
    const size_t n_main4 = (size_t)N1 * N2 / 4, n_all4 = n_main4 + (dbias ? (size_t)N1 / 4 : 0);
    const int rb = (int)((n_all4 + 255) / 256 < 2048 ? (n_all4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(rb), dim3(256), 0, st, (const float*)workspace, dW, dbias, n_main4,
                       n_all4, splits, slab_stride / 4, accumulate);
    CLDRD_LAUNCH_CHECK();
    return 0;
}
