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

constexpr int BK = 64;                 // tokens per LDS slot

__device__ __forceinline__ int swz(int m) { return 2 * ((m & 3) | (((m >> 3) & 1) << 2)); }
// 16-byte chunk c of token row m -> position inside the row.  Rows of 256 / 512 bytes: c ^ swz(m).  Rows of 384 bytes (24 chunks,
// the 192-column B tile): chunks 0..15 as before, chunks 16..23 permute among themselves with 2 ((m>>1)&1 | ((m>>3)&1)<<1); odd
// rows start 128 bytes into a 256-byte bank window, and tools/lds_bank_sim.py shows every transposed read still touches each of
// the 64 banks once.  The map is an involution, so the DMA applies it on the source side.
template <int ROWB>
__device__ __forceinline__ int chunk_pos(int c, int m) {
    if (ROWB == 384 && c >= 16) return 16 + ((c - 16) ^ (2 * (((m >> 1) & 1) | (((m >> 3) & 1) << 1))));
    return c ^ swz(m);
}

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

template <int T1, int T2, int NW, bool BIAS>
__global__ __launch_bounds__(64 * NW, 2) void gemm_tn_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                         float* __restrict__ slabs, int M, int N1, int N2, int lda, int ldb,
                                                         int splits, int ksteps_per_split, size_t slab_stride, int stagger) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NBF = T2 == 192 ? 3 : 4;              // B fragments (16 n2 columns each) per wave
    constexpr int WGN = T2 / (16 * NBF), WGM = NW / WGN;        // wave grid; a wave owns (16 FA) x (16 NBF) of the tile
    constexpr int FA = T1 / WGM / 16;                   // A fragments (16 n1 columns each) per wave
    constexpr int A_BYTES = BK * T1 * 2, B_BYTES = BK * T2 * 2, SLOT = A_BYTES + B_BYTES;
    constexpr int LPR_A = T1 / 8, RA = 64 / LPR_A;      // lanes per A row, A rows per 1-KiB piece
    constexpr int APW = A_BYTES / 1024 / NW, BPW = B_BYTES / 1024 / NW;     // pieces per wave
    constexpr int NSLOT = 3 * SLOT <= 160 * 1024 ? 3 : 2;                   // 3: K tiles kt+1 and kt+2 stay in flight
    constexpr int G = APW + BPW;                        // LDS-DMA instructions per wave per K tile
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
    const int wm = wid / WGN, wn = wid % WGN;
    const bool do_bias = BIAS && c2 == 0 && wn == 0;

    // ---- LDS-DMA.  A piece = RA token rows x (T1*2) B; a B piece = 1 KiB of the row-major B image; wave w owns pieces APW*w.. / BPW*w..
    uint32_t oa[APW], ob[BPW];
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int r = RA * (APW * wid + i) + lane / LPR_A;
        oa[i] = (uint32_t)r * (uint32_t)(lda * 2) + (uint32_t)(c1 * 2) + (uint32_t)(((lane % LPR_A) ^ swz(r)) * 16);
    }
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int byte = (BPW * wid + i) * 1024 + lane * 16;            // position in the LDS image of the B tile (row-major, T2*2-byte rows)
        const int r = byte / (T2 * 2), pos = (byte % (T2 * 2)) / 16;
        ob[i] = (uint32_t)r * (uint32_t)(ldb * 2) + (uint32_t)(c2 * 2) + (uint32_t)(chunk_pos<T2 * 2>(pos, r) * 16);
    }
    auto stage = [&](int slot, int kt) {
        char* base = smem + slot * SLOT;
        const char* pa = (const char*)A + (size_t)(kbeg + kt) * BK * lda * 2;
        const char* pb = (const char*)B + (size_t)(kbeg + kt) * BK * ldb * 2;
#pragma unroll
        for (int i = 0; i < APW; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(pa + oa[i]), LDS_PTR(base + (APW * wid + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BPW; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(pb + ob[i]), LDS_PTR(base + A_BYTES + (BPW * wid + i) * 1024), 16, 0, 0);
    };

    // ---- transposed-read addressing: 16-lane group g reads token rows 8g+q (+4); lane (4q+pp) supplies cols 4pp..4pp+3.
    // Token row m = 32 ks + 8 g + 4 r + q: swz(m) depends on q and g only, so one LDS offset per column tile t covers every
    // k-step (ks: + 32 rows) and both reads (r: + 4 rows) through the instruction's immediate offset.
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int m0 = 8 * g + q;
    uint32_t ra[FA], rb[NBF];
#pragma unroll
    for (int t = 0; t < FA; ++t) {
        const int ca = wm * (16 * FA) + t * 16 + 4 * pp;
        ra[t] = (uint32_t)(uintptr_t)LDS_PTR(smem) + m0 * (T1 * 2) + (((ca >> 3) ^ swz(m0)) * 16) + (ca & 7) * 2;
    }
#pragma unroll
    for (int t = 0; t < NBF; ++t) {
        const int cb = wn * (16 * NBF) + t * 16 + 4 * pp;
        rb[t] = (uint32_t)(uintptr_t)LDS_PTR(smem) + A_BYTES + m0 * (T2 * 2) + chunk_pos<T2 * 2>(cb >> 3, m0) * 16 + (cb & 7) * 2;
    }

    f32x4 acc[FA][NBF];
    f32x4 accb[BIAS ? FA : 1];
#pragma unroll
    for (int i = 0; i < FA; ++i) {
        if (BIAS) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NBF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const short one = (short)0x3F80;     // bf16 1.0
    const bf16x8 ones = (bf16x8){one, one, one, one, one, one, one, one};

    Frag af[FA], b0[NBF], b1[NBF];
    auto mfma_row = [&](int t1, Frag (&bc)[NBF]) {
        const bf16x8 a = frag8(af[t1]);
#pragma unroll
        for (int t2 = 0; t2 < NBF; ++t2)
            // swapped operands: D'[n2][n1] so the lane's 4 accumulators run along n2 (contiguous in dW rows)
            acc[t1][t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag8(bc[t2]), a, acc[t1][t2], 0, 0, 0);
        if (BIAS && do_bias) accb[BIAS ? t1 : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, a, accb[BIAS ? t1 : 0], 0, 0, 0);
    };
    // LDS returns in issue order.  Issue order per k-step: A0, B'0..B'3, A1, .., A(FA-1) (2 reads each; B' = next k-step's B).
    // Row 0 needs A0 and B, the 2 (FA-1) reads of A1.. are younger.  Row t1 >= 1 needs A[t1] of the previous k-step:
    // 2 FA + 2 NBF - 2 younger reads (the lgkmcnt field stops at 15: for FA = 8 a few older refills are waited for too).
    constexpr int W0 = 2 * (FA - 1), WR = 2 * FA + 2 * NBF - 2 < 15 ? 2 * FA + 2 * NBF - 2 : 15;
    auto wait_row0 = [&](Frag (&bc)[NBF]) {
        if constexpr (NBF == 4)
            asm volatile("s_waitcnt lgkmcnt(%10)" : "+v"(af[0].lo), "+v"(af[0].hi), "+v"(bc[0].lo), "+v"(bc[0].hi), "+v"(bc[1].lo),
                         "+v"(bc[1].hi), "+v"(bc[2].lo), "+v"(bc[2].hi), "+v"(bc[3].lo), "+v"(bc[3].hi) : "n"(W0));
        else
            asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(af[0].lo), "+v"(af[0].hi), "+v"(bc[0].lo), "+v"(bc[0].hi), "+v"(bc[1].lo),
                         "+v"(bc[1].hi), "+v"(bc[2].lo), "+v"(bc[2].hi) : "n"(W0));
    };
    // waves w and w + NW/2 share a SIMD: the second half issues its LDS-DMA one k-step later, so the two do not stall the
    // MFMA pipe at the same time (measured on the NT kernel: +2..9 %)
    const bool late_wave = NW == 8 && wid >= 4 && stagger;
    // one 32-token k-step over (af, bc); the next fragments come from slot offset noff, k-step NKS of that slot
    auto kstep = [&](Frag (&bc)[NBF], Frag (&bn)[NBF], uint32_t noff, auto nks_tag, bool sync, int slot, int kt) {
        constexpr int NKS = decltype(nks_tag)::value;
        constexpr int OA = NKS * 32 * T1 * 2, OB = NKS * 32 * T2 * 2;
        wait_row0(bc);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(0, bc);
        __builtin_amdgcn_sched_barrier(0);
        if (sync) {
            // K tile kt+1 must have landed; with three slots K tile kt+2 (issued one tile ago) may stay in flight
            if (NSLOT == 3 && kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (!late_wave && kt + NSLOT < nk) stage(slot, kt + NSLOT);  // slot of K tile kt: every wave holds its fragments in registers
            __builtin_amdgcn_sched_barrier(0);
        } else if (late_wave && slot >= 0 && kt + NSLOT < nk) {
            stage(slot, kt + NSLOT);                       // postponed from the previous K tile's barrier (see gemm_nt_ring.hip)
            __builtin_amdgcn_sched_barrier(0);
        }
        tr_issue<OA, 4 * T1 * 2>(af[0], ra[0] + noff);
#pragma unroll
        for (int t = 0; t < NBF; ++t) tr_issue<OB, 4 * T2 * 2>(bn[t], rb[t] + noff);
#pragma unroll
        for (int t1 = 1; t1 < FA; ++t1) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(af[t1].lo), "+v"(af[t1].hi) : "n"(WR));
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
        if (NSLOT == 3 && nk > 1) {
            stage(1, 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (NSLOT == 3) { if (nk > 2) stage(2, 2); } else { if (nk > 1) stage(1, 1); }
        tr_issue<0, 4 * T1 * 2>(af[0], ra[0]);
#pragma unroll
        for (int t = 0; t < NBF; ++t) tr_issue<0, 4 * T2 * 2>(b0[t], rb[t]);
#pragma unroll
        for (int t = 1; t < FA; ++t) tr_issue<0, 4 * T1 * 2>(af[t], ra[t]);
        int cs = 0, ps = -1;                               // slots of K tiles kt and kt-1
        for (int kt = 0; kt + 1 < nk; ++kt) {
            const int ns = cs == NSLOT - 1 ? 0 : cs + 1;
            kstep(b0, b1, (uint32_t)(cs * SLOT), KS1{}, false, ps, kt - 1);
            kstep(b1, b0, (uint32_t)(ns * SLOT), KS0{}, true, cs, kt);
            ps = cs;
            cs = ns;
        }
        kstep(b0, b1, (uint32_t)(cs * SLOT), KS1{}, false, ps, nk - 2);
        wait_row0(b1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(0, b1);
#pragma unroll
        for (int t1 = 1; t1 < FA; ++t1) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[t1].lo), "+v"(af[t1].hi));
            mfma_row(t1, b1);
        }
    }

    float* slab = slabs + (size_t)split * slab_stride;
#pragma unroll
    for (int t1 = 0; t1 < FA; ++t1) {
        const int n1 = c1 + wm * (16 * FA) + t1 * 16 + (lane & 15);
#pragma unroll
        for (int t2 = 0; t2 < NBF; ++t2) {
            const int n2 = c2 + wn * (16 * NBF) + t2 * 16 + 4 * (lane >> 4);
            *(float4*)(slab + (size_t)n1 * N2 + n2) = make_float4(acc[t1][t2][0], acc[t1][t2][1], acc[t1][t2][2], acc[t1][t2][3]);
        }
        if (BIAS && do_bias && lane < 16) slab[(size_t)N1 * N2 + n1] = accb[BIAS ? t1 : 0][0];       // every n2 row of D' holds the same column sums
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

// tile choice: 256 x 256 (8 waves of 128 x 64) when both extents allow, else 256 x 128 (8 waves of 64 x 64), else 128 x 128 (4 waves)
struct WgradTile { int t1, t2; };
static inline WgradTile wgrad_tile(int N1, int N2) {
    static int force = -1;
    if (force < 0) { const char* e = getenv("CLDRD_WGRAD_TILE"); force = e ? atoi(e) : 0; }
    // measured (T = 32768): the 256 x 256 tile needs all 256 VGPRs, spills, and has room for two LDS slots only:
    // 430-480 TF/s against 690-810 TF/s for 256 x 128.  It stays instantiable for experiments (CLDRD_WGRAD_TILE=256).
    if (N1 % 256 == 0 && N2 % 256 == 0 && force == 256) return {256, 256};
    // 256 x 192 (8 waves of 128 x 48): 7 LDS-DMA pieces per 48 MFMAs per wave instead of 6 per 32, the ratio of the NT kernel's BN = 192
    // (measured at T = 32768: 2304 x 768 +6 %, 3072 x 768 +3 %, 768 x 3072 +10 %; 768 x 768 -7 %: with 12 tiles the 21 slabs
    // per tile make the reduction the larger part, so small outputs stay on 256 x 128)
    if (N1 % 256 == 0 && N2 % 192 == 0 && ((force != 128 && ((N1 / 256) * (N2 / 192) >= 24 || force == 192)) || N2 % 128 != 0)) return {256, 192};
    if (N1 % 256 == 0) return {256, 128};
    return {128, 128};
}

// Token-range splits per output tile.  One workgroup occupies a CU (LDS), so tiles * splits workgroups run in
// ceil(tiles * splits / 256) rounds of ceil(ktotal / splits) K tiles each, plus a per-workgroup cost (pipeline fill, slab
// write) of about OVERHEAD K-tile times.  Pick the split count with the smallest modelled time: e.g. 72 tiles -> 7 splits
// (504 workgroups = 1.97 rounds) beats 3 splits (216 workgroups, 40 idle CUs); 270 workgroups would be the worst case.
extern "C" int cldrd_wgrad_splits(int M, int N1, int N2) {
    const WgradTile t = wgrad_tile(N1, N2);
    const int tiles = (N1 / t.t1) * (N2 / t.t2);
    const int ktotal = (M + BK - 1) / BK;
    if (tiles <= 0) return 1;
    static int overhead = -1;
    if (overhead < 0) { const char* e = getenv("CLDRD_WGRAD_OVERHEAD"); overhead = e ? atoi(e) : 6; }
    int best = 1;
    long best_cost = -1;
    for (int sp = 1; sp <= 64 && sp <= ktotal; ++sp) {
        const long rounds = ((long)tiles * sp + 255) / 256;
        const long cost = rounds * ((ktotal + sp - 1) / sp + overhead);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = sp; }
    }
    return best;
}

static inline int wgrad_stagger() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("CLDRD_GEMM_STAGGER"); v = e ? atoi(e) : 1; }
    return v;
}

template <int T1, int T2, int NW, bool BIAS>
static int launch_tn(const void* A, const void* B, float* ws, int M, int N1, int N2, int lda, int ldb, int splits, int kps,
                     size_t slab_stride, hipStream_t st) {
    constexpr int slot = BK * (T1 + T2) * 2;
    constexpr int lds = (3 * slot <= 160 * 1024 ? 3 : 2) * slot;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<T1, T2, NW, BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int tiles = (N1 / T1) * (N2 / T2);
    hipLaunchKernelGGL((gemm_tn_kernel<T1, T2, NW, BIAS>), dim3(tiles * splits), dim3(64 * NW), lds, st, (const bf16_t*)A, (const bf16_t*)B, ws,
                       M, N1, N2, lda, ldb, splits, kps, slab_stride, wgrad_stagger());
    CLDRD_LAUNCH_CHECK();
    return 0;
}

// workspace floats needed: splits * (N1*N2 + N1)
extern "C" int cldrd_wgrad_bf16(const void* A, const void* B, float* dW, float* dbias, int M, int N1, int N2, int lda, int ldb,
                                float* workspace, size_t workspace_bytes, int accumulate, void* stream) {
    CLDRD_CHECK(M > 0, "wgrad: empty problem");
    CLDRD_CHECK(N1 % 128 == 0 && (N2 % 128 == 0 || (N1 % 256 == 0 && N2 % 192 == 0)), "wgrad: N1 and N2 must be multiples of 128 (or 256 x 192)");
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
    const WgradTile t = wgrad_tile(N1, N2);
    int rc;
#define CLDRD_TN(T1_, T2_, NW_) (dbias ? launch_tn<T1_, T2_, NW_, true>(A, B, workspace, M, N1, N2, lda, ldb, splits, kps, slab_stride, st) \
                                       : launch_tn<T1_, T2_, NW_, false>(A, B, workspace, M, N1, N2, lda, ldb, splits, kps, slab_stride, st))
    if (t.t1 == 256 && t.t2 == 256) rc = CLDRD_TN(256, 256, 8);
    else if (t.t1 == 256 && t.t2 == 192) rc = CLDRD_TN(256, 192, 8);
    else if (t.t1 == 256) rc = CLDRD_TN(256, 128, 8);
    else rc = CLDRD_TN(128, 128, 4);
#undef CLDRD_TN
    if (rc) return rc;
    const size_t n_main4 = (size_t)N1 * N2 / 4, n_all4 = n_main4 + (dbias ? (size_t)N1 / 4 : 0);
    const int rb = (int)((n_all4 + 255) / 256 < 2048 ? (n_all4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(rb), dim3(256), 0, st, (const float*)workspace, dW, dbias, n_main4,
                       n_all4, splits, slab_stride / 4, accumulate);
    CLDRD_LAUNCH_CHECK();
    return 0;
}
