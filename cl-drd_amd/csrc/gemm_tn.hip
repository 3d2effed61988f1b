// Weight gradient: dW[N1,N2] (+)= sum_m A[m,N1] * B[m,N2]   (A = dY, B = X; bf16 in, fp32 out), optionally with the
// bias gradient db[N1] (+)= sum_m A[m,N1] folded in.
//
// The reduction index m (tokens) is the ROW index of both operands, so neither is K-contiguous: the MFMA fragments are
// fetched with gfx950's transposing LDS read (ds_read_b64_tr_b16) from row-major [64 tokens][cols] tiles that arrive
// by LDS-DMA.  The 16-B slot index is XOR-swizzled with f(m) = 2*((m&3) | ((m>>3)&1)<<2) (on the DMA source address
// and on the read address): every transposed read then touches all 64 banks exactly once.
//
// (Round 3: the NT kernel's third A slot / separate operand rings were tried here too and removed again - nothing for the step, and the
// restructured K loop read 7.4 instead of 5.9 GB per passage-tower group: profiles/r03_microbench.txt.)
// Tile 256 (n1) x 128 (n2), 512 threads = 4 x 2 waves of 64 x 64, MFMA 16x16x32; 64-token K tiles in two LDS slots.
// Same software pipeline as gemm_nt_ring.hip: per 32-token k-step the A fragments are refilled in place as soon as
// their last MFMA has issued, the B fragments are double-buffered, reads are threaded between the MFMAs
// (sched_group_barrier), one raw s_barrier per K tile, the DMA of K tile t+2 is issued half-way through K tile t.
// M = tokens is huge and N1 x N2 small, so the token range is split over `splits` workgroups per tile; each writes an
// fp32 slab and a second pass sums the slabs in a fixed order (bitwise reproducible, no float atomics).
// The bias gradient costs one extra MFMA per A fragment in the n2 == 0 tiles: B = all-ones.
//
// Rows >= M of the last 64-token K tile are fetched from a zero page instead of the operands (the LDS-DMA source address is per
// lane), so the operands need neither padding nor a zeroed tail.
//
// GROUPED launch (cldrd_wgrad_group): the weight gradients are not on the critical path of the backward (only the data gradients
// are), so the trainer defers them and hands all of a tower's problems - 4 per layer - to ONE launch.  Work item = (problem, token
// split, output tile), ordered tile-fastest so that the workgroups running at the same time on one XCD compute neighbouring tiles
// of the same problem at the same token position (shared A / B panels in that XCD's L2).  With 700+ tiles in a group no token split
// is needed any more (744 tiles = 2.9 rounds of 256 CUs for DistilBERT at cfg2): every workgroup sweeps ALL tokens of its tile and
// writes dW once - no fp32 slabs through HBM, no reduction launches (round 1: 25 GEMM + 25 reduction launches and ~1 GB of slab
// traffic per step).  Token splits + slabs remain for small groups (few tiles), chosen by the same cost model as before.
#include <type_traits>

#include "common.h"

namespace {

constexpr int BK = 64;                 // tokens per LDS slot
constexpr int MAXP = 32;               // problems per launch (kernel-argument block: 32 x 72 B)

struct TnProblem {
    const bf16_t* A; const bf16_t* B; float* dW; float* dbias;      // dbias may be null
    int M, N1, N2, lda, ldb;
    int tiles, nt2;                    // (N1 / T1) * (N2 / T2), N2 / T2
    int nt1, n1_fast;                  // N1 / T1; tile order inside a split: n2 fastest (0) or n1 fastest (1), see cldrd_wgrad_group
    int first;                         // first work item of this problem; items of a problem: split-major, tile-minor
    long long slab_off;                // splits > 1: float offset of this problem's slabs in the workspace
};
struct TnGroupArgs {
    int n, splits, accumulate, stagger;
    float* slabs;
    const float* inv_scale;            // device scalar or null: every output (dW, dbias) is multiplied by it - the operands of the all-fp16
                                       // training mode carry the loss scale, parameter gradients do not (capi.hip: cldrd_set_loss_scale)
    float* sq_out;                     // slab reduction only: [gridDim.y][gridDim.x] sums of squares of what each workgroup wrote, or null
    TnProblem p[MAXP];
};

__device__ __attribute__((aligned(4096))) uint4 g_zero_page[256];     // 4 KiB of zeros: DMA source of token rows >= M

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
// One LDS-DMA piece (1 KiB per wave), saddr form: LDS destination = m0 + lane*16, source = uniform 64-bit base + per-lane 32-bit offset
// (hipcc turned `base + kt*stride + zext(off)` into two 64-bit VALU adds per piece and kept every offset as a register pair: 245 -> 225
// VGPRs for the 256 x 192 tile, +1..2 %).  Tried on the way and dropped: the bias gradient by v_dot2c_f32_bf16 instead of the all-ones
// MFMA (frees 24 VGPRs, but the launch ran 15 % slower even without a bias problem in it) and a 256 x 256 tile (spills only outside
// the K loop, 1018 TF/s at 4096^3 against 877 for 256 x 128, but no better than 256 x 192 on the encoder shapes once the tile counts
// are rounded to whole rounds of CUs).
__device__ __forceinline__ void tn_dma16(uint32_t lds_addr, uint32_t voff, const void* sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory", "m0");
}
// Round 4: the two transposing reads of a fragment are the BUILTIN again, and every LDS-DMA of the kernel is inline asm instead (the
// partial-tile path included).  Rounds 1-3 had it the other way round - reads in asm because hipcc, seeing a global_load_lds builtin
// anywhere in the loop, put s_waitcnt vmcnt(0) in front of every LDS read - and paid for it: two 64-bit asm outputs are two unrelated
// register pairs, so hipcc copied every fragment into a 4-register tuple before its MFMA (56 v_mov per 48 MFMAs in the K loop;
// PMC: 1.66 VALU instructions per MFMA in a kernel that has no arithmetic) and the hand-counted lgkmcnt waits had to live with the
// 4-bit counter.  With no VMEM instruction visible to it the compiler inserts exact lgkmcnt waits for its own reads and allocates lo / hi
// as one tuple (no copies); the vmcnt waits of the DMA ring stay ours.
struct Frag { bf16x4 lo, hi; };
template <int OFF, int HI>
__device__ __forceinline__ void tr_issue(Frag& f, uint32_t addr) {
    typedef __attribute__((address_space(3))) bf16x4* lds4;
    f.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(uintptr_t)(addr + OFF));
    f.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(uintptr_t)(addr + OFF + HI));
}
__device__ __forceinline__ bf16x8 frag8(const Frag& f) { return __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7); }
// one LDS-DMA piece with a per-lane 64-bit source address (partial last K tile: rows >= M come from the zero page)
__device__ __forceinline__ void tn_dma16_addr(uint32_t lds_addr, const void* src) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src) : "memory", "m0");
}

// ABL != 0 (development build only, tools/tn_ablate.py; WRONG results): 1 no LDS-DMA inside the K loop, 2 no fragment reads inside the
// K loop, 3 no barrier and no vmcnt wait inside the K loop.
typedef _Float16 tn_f16x8 __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ f32x4 tn_mfma(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(tn_f16x8, a), __builtin_bit_cast(tn_f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// F16: the operands are fp16 (round 4: the all-fp16 training mode), same MFMA rate, same tiles
template <int T1, int T2, int NW, int ABL = 0, bool F16 = false>
__global__ __launch_bounds__(64 * NW, 2) void gemm_tn_kernel(TnGroupArgs ga) {
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
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    int pi = 0;
    while (pi + 1 < ga.n && id >= ga.p[pi + 1].first) ++pi;               // wave-uniform: scalar loads from the argument block
    const TnProblem& P = ga.p[pi];
    const bf16_t* __restrict__ A = P.A;
    const bf16_t* __restrict__ B = P.B;
    const int M = P.M, N1 = P.N1, N2 = P.N2, lda = P.lda, ldb = P.ldb;
    const int nt2 = P.nt2, ntiles = P.tiles;
    const int local = id - P.first;
    const int split = local / ntiles, tile = local % ntiles;      // the tiles of one split are neighbours: they share A/B rows
    const int c1 = (P.n1_fast ? tile % P.nt1 : tile / nt2) * T1, c2 = (P.n1_fast ? tile / P.nt1 : tile % nt2) * T2;
    const int ktotal = (M + BK - 1) / BK;
    const int ksteps_per_split = (ktotal + ga.splits - 1) / ga.splits;
    const int kbeg = split * ksteps_per_split;
    const int nk = min(ktotal, kbeg + ksteps_per_split) - kbeg;
    const int wm = wid / WGN, wn = wid % WGN;
    const bool do_bias = P.dbias != nullptr && c2 == 0 && wn == 0;
    const int stagger = ga.stagger;
    const int mtail = M & (BK - 1);                    // valid rows of the last K tile (0: it is complete)

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
        if (ABL == 1 && kt >= NSLOT) return;
        const char* pa = (const char*)A + (size_t)(kbeg + kt) * BK * lda * 2;
        const char* pb = (const char*)B + (size_t)(kbeg + kt) * BK * ldb * 2;
        if (mtail != 0 && kbeg + kt == ktotal - 1) {
            // last, partial K tile: token rows >= M come from the zero page (once per workgroup at most; rows recomputed here
            // instead of being kept in registers)
            const char* zp = (const char*)g_zero_page + lane * 16;
            const uint32_t lb = (uint32_t)(uintptr_t)LDS_PTR(smem) + slot * SLOT;
#pragma unroll
            for (int i = 0; i < APW; ++i) {
                const int r = RA * (APW * wid + i) + lane / LPR_A;
                tn_dma16_addr(lb + (APW * wid + i) * 1024, r < mtail ? pa + oa[i] : zp);
            }
#pragma unroll
            for (int i = 0; i < BPW; ++i) {
                const int r = ((BPW * wid + i) * 1024 + lane * 16) / (T2 * 2);
                tn_dma16_addr(lb + A_BYTES + (BPW * wid + i) * 1024, r < mtail ? pb + ob[i] : zp);
            }
            return;
        }
        const uint32_t lbase = (uint32_t)(uintptr_t)LDS_PTR(smem) + slot * SLOT;
#pragma unroll
        for (int i = 0; i < APW; ++i) tn_dma16(lbase + (APW * wid + i) * 1024, oa[i], pa);
#pragma unroll
        for (int i = 0; i < BPW; ++i) tn_dma16(lbase + A_BYTES + (BPW * wid + i) * 1024, ob[i], pb);
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
    f32x4 accb[FA];
#pragma unroll
    for (int i = 0; i < FA; ++i) {
        accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NBF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const short one = F16 ? (short)0x3C00 : (short)0x3F80;     // 1.0 in the operand format
    const bf16x8 ones = (bf16x8){one, one, one, one, one, one, one, one};

    // waves w and w + NW/2 share a SIMD: the second half issues its LDS-DMA one k-step later, so the two do not stall the
    // MFMA pipe at the same time (measured on the NT kernel: +2..9 %)
    const bool late_wave = NW == 8 && wid >= 4 && stagger;
    using KS0 = std::integral_constant<int, 0>;
    using KS1 = std::integral_constant<int, 1>;

    // The whole K sweep, instantiated twice: with and without the bias-gradient MFMA (B = all-ones) per A fragment.  Which one a wave
    // runs is decided ONCE (wave-uniform), not per fragment row: until round 3 every row of the K loop carried a v_cndmask / v_cmp /
    // s_cbranch and three v_mov rebuilding the all-ones operand.
    auto sweep = [&](auto bias_tag) {
        constexpr bool WB = decltype(bias_tag)::value;
        Frag af[FA], b0[NBF], b1[NBF];
        auto tr_issue_ = [&](auto off_tag, auto hi_tag, Frag& f, uint32_t addr) {
            if constexpr (ABL == 2) { asm volatile("" : "+v"(f.lo), "+v"(f.hi)); }
            else tr_issue<decltype(off_tag)::value, decltype(hi_tag)::value>(f, addr);
        };
        auto mfma_row = [&](int t1, Frag (&bc)[NBF]) {
            const bf16x8 a = frag8(af[t1]);
#pragma unroll
            for (int t2 = 0; t2 < NBF; ++t2)
                // swapped operands: D'[n2][n1] so the lane's 4 accumulators run along n2 (contiguous in dW rows)
                acc[t1][t2] = tn_mfma<F16>(frag8(bc[t2]), a, acc[t1][t2]);
            if constexpr (WB) accb[t1] = tn_mfma<F16>(ones, a, accb[t1]);
        };
        // K tile kt+1 must have landed (with three slots K tile kt+2, issued one tile ago, may stay in flight) and every read of this K
        // tile's slot must be back before the slot is handed to the DMA: lgkmcnt(0) covers the compiler's reads, which are all issued
        // above this point (sched_barrier).
        auto sync_and_recycle = [&](int slot, int kt) {
            if (ABL == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else if (NSLOT == 3 && kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (ABL != 3) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (!late_wave && kt + NSLOT < nk) stage(slot, kt + NSLOT);  // slot of K tile kt: every wave holds its fragments in registers
            __builtin_amdgcn_sched_barrier(0);
        };
        // one 32-token k-step over (af, bc), FA < 8 (the 128 x 128 tile); the next fragments come from slot offset noff, k-step NKS of that slot.
        // Reads are issued right behind the MFMA row that last used their destination; the compiler waits (exact lgkmcnt) where a fragment is used.
        auto kstep = [&](Frag (&bc)[NBF], Frag (&bn)[NBF], uint32_t noff, auto nks_tag, bool sync, int slot, int kt) {
            constexpr int NKS = decltype(nks_tag)::value;
            constexpr int OA = NKS * 32 * T1 * 2, OB = NKS * 32 * T2 * 2;
            mfma_row(0, bc);
            __builtin_amdgcn_sched_barrier(0);
            if (sync) {
                sync_and_recycle(slot, kt);
            } else if (late_wave && slot >= 0 && kt + NSLOT < nk) {
                stage(slot, kt + NSLOT);                       // postponed from the previous K tile's barrier (see gemm_nt_ring.hip)
                __builtin_amdgcn_sched_barrier(0);
            }
            tr_issue_(std::integral_constant<int, OA>{}, std::integral_constant<int, 4 * T1 * 2>{}, af[0], ra[0] + noff);
#pragma unroll
            for (int t = 0; t < NBF; ++t) tr_issue_(std::integral_constant<int, OB>{}, std::integral_constant<int, 4 * T2 * 2>{}, bn[t], rb[t] + noff);
#pragma unroll
            for (int t1 = 1; t1 < FA; ++t1) {
                __builtin_amdgcn_sched_barrier(0);
                mfma_row(t1, bc);
                __builtin_amdgcn_sched_barrier(0);
                tr_issue_(std::integral_constant<int, OA>{}, std::integral_constant<int, 4 * T1 * 2>{}, af[t1], ra[t1] + noff);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // FA = 8 (the 256-row tiles): A fragments are refilled FOUR rows after their last use instead of at once, so that every request is
        // four MFMA rows old when its fragment is used: fragment f of k-step s+1 is requested after row f + 4 (rows 0..3 request fragments
        // 4..7 of the SAME k-step, rows 4..7 fragments 0..3 and B of the next one).  A k-step starts with A0..A3 and B in registers only;
        // the slot of K tile kt is free (barrier + DMA of K tile kt+2) after row 4 of its second k-step; the very last k-step still issues
        // its (unused) requests, which the compiler retires before the registers are reused.
        // one 32-token k-step; current fragments at slot offset coff / k-step CKS, the next k-step's at noff / NKS
        auto kstep8 = [&](Frag (&bc)[NBF], Frag (&bn)[NBF], uint32_t coff, auto cks_tag, uint32_t noff, auto nks_tag, bool sync, int slot, int kt) {
            constexpr int CKS = decltype(cks_tag)::value, NKS = decltype(nks_tag)::value;
            constexpr int COA = CKS * 32 * T1 * 2, NOA = NKS * 32 * T1 * 2, NOB = NKS * 32 * T2 * 2;
            mfma_row(0, bc);
            __builtin_amdgcn_sched_barrier(0);
            if (!sync && late_wave && slot >= 0 && kt + NSLOT < nk) {
                stage(slot, kt + NSLOT);                       // postponed from the previous K tile's barrier (see gemm_nt_ring.hip)
                __builtin_amdgcn_sched_barrier(0);
            }
            tr_issue_(std::integral_constant<int, COA>{}, std::integral_constant<int, 4 * T1 * 2>{}, af[4], ra[4] + coff);
#pragma unroll
            for (int t1 = 1; t1 < 4; ++t1) {
                __builtin_amdgcn_sched_barrier(0);
                mfma_row(t1, bc);
                __builtin_amdgcn_sched_barrier(0);
                tr_issue_(std::integral_constant<int, COA>{}, std::integral_constant<int, 4 * T1 * 2>{}, af[t1 + 4], ra[t1 + 4] + coff);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_row(4, bc);
            __builtin_amdgcn_sched_barrier(0);
            if (sync) sync_and_recycle(slot, kt);
#pragma unroll
            for (int t = 0; t < NBF; ++t) tr_issue_(std::integral_constant<int, NOB>{}, std::integral_constant<int, 4 * T2 * 2>{}, bn[t], rb[t] + noff);
            tr_issue_(std::integral_constant<int, NOA>{}, std::integral_constant<int, 4 * T1 * 2>{}, af[0], ra[0] + noff);
#pragma unroll
            for (int t1 = 5; t1 < 8; ++t1) {
                __builtin_amdgcn_sched_barrier(0);
                mfma_row(t1, bc);
                __builtin_amdgcn_sched_barrier(0);
                tr_issue_(std::integral_constant<int, NOA>{}, std::integral_constant<int, 4 * T1 * 2>{}, af[t1 - 4], ra[t1 - 4] + noff);
            }
            __builtin_amdgcn_sched_barrier(0);
        };

        if constexpr (FA == 8) {
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
#pragma unroll
            for (int t = 0; t < NBF; ++t) tr_issue<0, 4 * T2 * 2>(b0[t], rb[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) tr_issue<0, 4 * T1 * 2>(af[t], ra[t]);
            int cs = 0, ps = -1;                               // slots of K tiles kt and kt-1
            for (int kt = 0; kt + 1 < nk; ++kt) {
                const int ns = cs == NSLOT - 1 ? 0 : cs + 1;
                kstep8(b0, b1, (uint32_t)(cs * SLOT), KS0{}, (uint32_t)(cs * SLOT), KS1{}, false, ps, kt - 1);
                kstep8(b1, b0, (uint32_t)(cs * SLOT), KS1{}, (uint32_t)(ns * SLOT), KS0{}, true, cs, kt);
                ps = cs;
                cs = ns;
            }
            kstep8(b0, b1, (uint32_t)(cs * SLOT), KS0{}, (uint32_t)(cs * SLOT), KS1{}, false, ps, nk - 2);
            kstep8(b1, b0, (uint32_t)(cs * SLOT), KS1{}, (uint32_t)(cs * SLOT), KS0{}, false, -1, nk);      // its next-k-step requests read stale LDS: unused
        } else {
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
#pragma unroll
            for (int t1 = 0; t1 < FA; ++t1) mfma_row(t1, b1);
        }
    };
    if (nk > 0) {
        if (do_bias) sweep(std::true_type{});
        else sweep(std::false_type{});
    }

    // ---- output.  One split: the tile is complete, write (or add to) dW / dbias directly.  Several: this split's slab.
    const bool direct = ga.splits == 1;
    float* outW = direct ? P.dW : ga.slabs + P.slab_off + (size_t)split * ((size_t)N1 * N2 + (size_t)N1);
    float* outB = direct ? P.dbias : outW + (size_t)N1 * N2;
    const bool add = direct && ga.accumulate != 0;
    const float osc = (direct && ga.inv_scale) ? *ga.inv_scale : 1.0f;      // slabs stay raw: the reduction applies the factor
#pragma unroll
    for (int t1 = 0; t1 < FA; ++t1) {
        const int n1 = c1 + wm * (16 * FA) + t1 * 16 + (lane & 15);
#pragma unroll
        for (int t2 = 0; t2 < NBF; ++t2) {
            const int n2 = c2 + wn * (16 * NBF) + t2 * 16 + 4 * (lane >> 4);
            float4 v = make_float4(acc[t1][t2][0] * osc, acc[t1][t2][1] * osc, acc[t1][t2][2] * osc, acc[t1][t2][3] * osc);
            float4* dst = (float4*)(outW + (size_t)n1 * N2 + n2);
            if (add) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *dst = v;
        }
        if (do_bias && lane < 16) {       // every n2 row of D' holds the same column sums
            float b = accb[t1][0] * osc;
            if (add) b += outB[n1];
            outB[n1] = b;
        }
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

// the same for every problem of a group in one launch: blockIdx.y = problem
__global__ __launch_bounds__(256) void reduce_slabs_group_kernel(TnGroupArgs ga) {
    const TnProblem& P = ga.p[blockIdx.y];
    const size_t n_main4 = (size_t)P.N1 * P.N2 / 4, n_all4 = n_main4 + (P.dbias ? (size_t)P.N1 / 4 : 0);
    const size_t stride4 = ((size_t)P.N1 * P.N2 + (size_t)P.N1) / 4;
    const float4* slabs = (const float4*)(ga.slabs + P.slab_off);
    const size_t step = (size_t)gridDim.x * blockDim.x;
    const float osc = ga.inv_scale ? *ga.inv_scale : 1.0f;
    float sq = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_all4; i += step) {
        float4 s = slabs[i];
        for (int k = 1; k < ga.splits; ++k) {
            const float4 t = slabs[i + k * stride4];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        s.x *= osc; s.y *= osc; s.z *= osc; s.w *= osc;
        float4* dst = i < n_main4 ? (float4*)P.dW + i : (float4*)P.dbias + (i - n_main4);
        if (ga.accumulate) {
            const float4 o = *dst;
            s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        }
        *dst = s;
        sq += s.x * s.x + s.y * s.y + s.z * s.z + s.w * s.w;
    }
    if (ga.sq_out) {       // the clip norm's share of this workgroup (capi.hip: cldrd_set_norm_sink): fixed order, no atomics
        __shared__ float red[4];
        sq = wave_sum(sq);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) ga.sq_out[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------------
struct WgradTile { int t1, t2; };

// One tile shape per launch: 256 x 192 (8 waves of 128 x 48: 7 LDS-DMA pieces per 48 MFMAs per wave, the ratio of the NT kernel's
// BN = 192) when every problem allows it, else 256 x 128 (8 waves of 64 x 64), else 128 x 128 (4 waves).  The 256 x 256 tile of
// round 1 (all 256 VGPRs, spills, two LDS slots: 430-480 TF/s against 690-810) came back in round 4 without the spills (below).
// CLDRD_WGRAD_TILE=128|192|256 forces one (development build).
static WgradTile wgrad_tile_group(const int* N1, const int* N2, int n) {
    const int force = CLDRD_DEV_INT("CLDRD_WGRAD_TILE", 0);
    bool ok192 = true, ok128w = true;
    for (int i = 0; i < n; ++i) {
        ok192 = ok192 && N1[i] % 256 == 0 && N2[i] % 192 == 0;
        ok128w = ok128w && N1[i] % 256 == 0 && N2[i] % 128 == 0;
    }
    bool ok256 = true;
    for (int i = 0; i < n; ++i) ok256 = ok256 && N1[i] % 256 == 0 && N2[i] % 256 == 0;
    // Round 4: 256 x 256 (8 waves of 128 x 64; 256 VGPRs, no spills with the rewritten K sweep, two 64-KiB LDS slots) for GROUPS: 8 LDS-DMA
    // pieces per 64 MFMAs per wave instead of 7 per 48, and the passage tower's three-layer group becomes 324 tiles instead of 432.  Alone per
    // problem it is +3..7 % on the FFN shapes, -2 % on QKV, -14 % at 4096^3 (tools/tn_ablate.py); in the step 11.70 -> 11.44 and 11.67 ->
    // 11.48 ms (profiles/r04_microbench.txt).  Single problems keep the measured per-shape choice below.
    if (ok256 && force != 192 && force != 128 && (force == 256 || n > 1)) return {256, 256};
    if (ok192 && force != 128 && (force == 192 || !ok128w || n > 1)) return {256, 192};
    if (ok192 && force != 128) {
        // single problem: small outputs stay on 256 x 128 (measured at T = 32768: 768 x 768 -7 % on 256 x 192, the others +3..10 %)
        if ((N1[0] / 256) * (N2[0] / 192) >= 24) return {256, 192};
    }
    if (ok128w) return {256, 128};
    return {128, 128};
}

// Token splits per output tile.  One workgroup occupies a CU (LDS).  With `sp` splits an item of problem p sweeps
// ceil(ktotal_p / sp) K tiles plus a fixed cost (pipeline fill, output write) of about OVERHEAD K-tile times; the items are handed
// out in order, so the launch takes about (total work / 256 CUs) rounded up to whole largest items.  Splits > 1 add the fp32 slabs
// written and read back (bytes / ~5 TB/s, in units of the ~1.5 us a K tile takes) and the reduction launches.  Smallest modelled
// time wins; ties go to fewer splits.  (Weighting every problem by its OWN token count matters: the 120 tiles of the CLS-only last
// layer sweep 4 K tiles, the 744 of the full layers 512 - counting them alike chose 2 splits and paid 0.3 GB of slabs per step.)
static int wgrad_splits_group(const int* M, const int* N1, const int* N2, int n, WgradTile t) {
    int ktotal = 1;
    double out_bytes = 0.0;
    long tiles_all = 0;
    for (int i = 0; i < n; ++i) {
        ktotal = ktotal > (M[i] + BK - 1) / BK ? ktotal : (M[i] + BK - 1) / BK;
        out_bytes += 4.0 * N1[i] * N2[i];
        tiles_all += (long)(N1[i] / t.t1) * (N2[i] / t.t2);
    }
    if (tiles_all <= 0) return 1;
    const int overhead = 6;
    const int force = CLDRD_DEV_INT("CLDRD_WGRAD_SPLITS", 0);
    if (force > 0) return force < ktotal ? force : ktotal;
    const int force_big = CLDRD_DEV_INT("CLDRD_WGRAD_SPLITS_BIG", 0);          // long sweeps only (the passage tower's group)
    if (force_big > 0 && ktotal >= 256) return force_big;
    // Workgroups of one XCD share A / B panels through its 4-MiB L2 only while they sweep the same token range at about the same
    // time; nothing synchronises them, so over a long sweep they drift apart and every one of them streams its operands from HBM
    // (measured at cfg2: 512 K tiles per item, no split: 3.2 ms for the passage tower's group; 2 splits of 256: 2.9 ms + 0.1 ms of
    // slabs).  Items are therefore capped at MAXK K tiles.
    const int maxk = CLDRD_DEV_INT("CLDRD_WGRAD_MAXK", 256) < 1 ? 256 : CLDRD_DEV_INT("CLDRD_WGRAD_MAXK", 256);
    const int sp_min = (ktotal + maxk - 1) / maxk;
    int best = sp_min;
    double best_cost = -1.0;
    for (int sp = sp_min; sp <= 64 && sp <= ktotal; ++sp) {
        double work = 0.0;
        for (int i = 0; i < n; ++i) {
            const long tiles = (long)(N1[i] / t.t1) * (N2[i] / t.t2);
            const int kt = (M[i] + BK - 1) / BK;
            work += (double)tiles * sp * ((kt + sp - 1) / sp + overhead);
        }
        const double item = (ktotal + sp - 1) / sp + overhead;
        double cost = item * (double)(long)((work / 256.0 + item - 1e-9) / item);          // whole largest items
        if (cost < item) cost = item;
        if (sp > 1) cost += 2.0 * sp * out_bytes / 5.0e12 / 1.5e-6 + 4.0;
        if (best_cost < 0 || cost < best_cost - 1e-9) { best_cost = cost; best = sp; }
    }
    return best;
}

extern "C" int cldrd_wgrad_splits(int M, int N1, int N2) {
    if (N1 % 128 != 0 || !(N2 % 128 == 0 || (N1 % 256 == 0 && N2 % 192 == 0))) return 1;
    return wgrad_splits_group(&M, &N1, &N2, 1, wgrad_tile_group(&N1, &N2, 1));
}

static inline int wgrad_stagger() { return CLDRD_DEV_INT("CLDRD_WGRAD_STAGGER", 1); }

template <int T1, int T2, int NW, int ABL = 0, bool F16 = false>
static int launch_tn_group(const TnGroupArgs& g, int items, hipStream_t st) {
    constexpr int slot = BK * (T1 + T2) * 2;
    constexpr int lds = (3 * slot <= 160 * 1024 ? 3 : 2) * slot;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<T1, T2, NW, ABL, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_tn_kernel<T1, T2, NW, ABL, F16>), dim3(items), dim3(64 * NW), lds, st, g);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

// floats of workspace a group needs (0 when no token split is used)
extern "C" size_t cldrd_wgrad_group_workspace(const int* M, const int* N1, const int* N2, int n) {
    if (n <= 0) return 0;
    const WgradTile t = wgrad_tile_group(N1, N2, n);
    const int splits = wgrad_splits_group(M, N1, N2, n, t);
    if (splits == 1) return 0;
    size_t tot = 0;
    for (int i = 0; i < n; ++i) tot += (size_t)splits * ((size_t)N1[i] * N2[i] + (size_t)N1[i]);
    return tot;
}

// dW[i][N1[i], N2[i]] (+)= A[i][:M[i]]^T . B[i][:M[i]] and dbias[i][N1[i]] (+)= column sums of A[i] (dbias[i] may be null) for n
// problems in one launch (more than 32: several launches).  The pointer / shape arrays live on the HOST.
extern "C" int cldrd_wgrad_group(const void* const* A, const void* const* B, float* const* dW, float* const* dbias, const int* M,
                                 const int* N1, const int* N2, const int* lda, const int* ldb, int n, float* workspace,
                                 size_t workspace_bytes, int accumulate, void* stream) {
    CLDRD_CHECK(n > 0, "wgrad_group: no problems");
    for (int i = 0; i < n; ++i) {
        CLDRD_CHECK(M[i] > 0, "wgrad: empty problem");
        CLDRD_CHECK(N1[i] % 128 == 0 && (N2[i] % 128 == 0 || (N1[i] % 256 == 0 && N2[i] % 192 == 0)),
                    "wgrad: N1 and N2 must be multiples of 128 (or 256 x 192)");
        CLDRD_CHECK(lda[i] % 8 == 0 && ldb[i] % 8 == 0, "wgrad: lda/ldb must be multiples of 8");
        CLDRD_CHECK(((uintptr_t)A[i] % 16 == 0) && ((uintptr_t)B[i] % 16 == 0) && ((uintptr_t)dW[i] % 16 == 0), "wgrad: operands must be 16-byte aligned");
        CLDRD_CHECK(dbias[i] == nullptr || (uintptr_t)dbias[i] % 16 == 0, "wgrad: dbias must be 16-byte aligned");
        CLDRD_CHECK((double)lda[i] * 2.0 * 72.0 < 4.0e9 && (double)ldb[i] * 2.0 * 72.0 < 4.0e9, "wgrad: row pitch too large");
    }
    hipStream_t st = (hipStream_t)stream;
    const WgradTile t = wgrad_tile_group(N1, N2, n);
    bool any128 = false;
    for (int i = 0; i < n; ++i) any128 = any128 || N2[i] % 128 != 0;
    CLDRD_CHECK(!(t.t2 != 192 && any128), "wgrad_group: problems with N2 % 128 != 0 need every problem to fit the 256 x 192 tile");
    const int splits = wgrad_splits_group(M, N1, N2, n, t);
    if (splits > 1) {
        size_t need = 0;
        for (int i = 0; i < n; ++i) need += (size_t)splits * ((size_t)N1[i] * N2[i] + (size_t)N1[i]);
        CLDRD_CHECK(workspace != nullptr && ((uintptr_t)workspace % 16 == 0) && workspace_bytes >= need * sizeof(float), "wgrad: workspace too small");
    }
    const int env_order = CLDRD_DEV_INT("CLDRD_WGRAD_ORDER", -1);       // tile order per problem (below); development build: 0 / 1 force n2- / n1-fastest
    size_t slab_off = 0;
    for (int lo = 0; lo < n; lo += MAXP) {
        const int m = n - lo < MAXP ? n - lo : MAXP;
        TnGroupArgs g;
        g.n = m; g.splits = splits; g.accumulate = accumulate & 1; g.stagger = wgrad_stagger(); g.slabs = workspace;
        g.inv_scale = g_cldrd_loss_scale ? g_cldrd_loss_scale + 1 : nullptr;
        g.sq_out = nullptr;
        int items = 0;
        for (int i = 0; i < m; ++i) {
            TnProblem& P = g.p[i];
            const int j = lo + i;
            P.A = (const bf16_t*)A[j]; P.B = (const bf16_t*)B[j]; P.dW = dW[j]; P.dbias = dbias[j];
            P.M = M[j]; P.N1 = N1[j]; P.N2 = N2[j]; P.lda = lda[j]; P.ldb = ldb[j];
            P.nt2 = N2[j] / t.t2; P.nt1 = N1[j] / t.t1; P.tiles = P.nt1 * P.nt2;
            // The 32 workgroups that run side by side on one XCD take consecutive tiles of a problem and share operand panels through its
            // L2.  With n2 fastest they cover (32 / nt2) A panels x all nt2 B panels: fine while nt2 is small (QKV, FFN1: nt2 = 4).  FFN2 has
            // 3 n1 tiles x 16 n2 tiles: n2 fastest covers 2 of the 3 rows, so every panel of h (the 200-MB operand) is fetched by 2 concurrent
            // tiles now and again by the third row later; n1 fastest puts all 3 tiles of an h panel side by side: h streams from HBM once.
            P.n1_fast = (env_order == 1 || (env_order < 0 && P.nt2 > P.nt1 && P.nt1 <= 32)) ? 1 : 0;
            P.first = items; P.slab_off = (long long)slab_off;
            items += P.tiles * splits;
            if (splits > 1) slab_off += (size_t)splits * ((size_t)N1[j] * N2[j] + (size_t)N1[j]);
        }
        int rc;
#ifdef CLDRD_DEV_BUILD                                 // timing-only ablations (WRONG results): development build only
        const int abl = cldrd_dev_int("CLDRD_TN_ABLATE", 0);
        if (abl && t.t1 == 256 && t.t2 == 192) {
            rc = abl == 1 ? launch_tn_group<256, 192, 8, 1>(g, items, st) : abl == 2 ? launch_tn_group<256, 192, 8, 2>(g, items, st)
                                                                                      : launch_tn_group<256, 192, 8, 3>(g, items, st);
        } else if (abl && t.t1 == 256) {
            rc = abl == 1 ? launch_tn_group<256, 128, 8, 1>(g, items, st) : abl == 2 ? launch_tn_group<256, 128, 8, 2>(g, items, st)
                                                                                      : launch_tn_group<256, 128, 8, 3>(g, items, st);
        } else
#endif
        if (t.t2 == 256) {
            rc = (accumulate & 2) ? launch_tn_group<256, 256, 8, 0, true>(g, items, st) : launch_tn_group<256, 256, 8>(g, items, st);
        } else if (accumulate & 2) {             // bit 1 of `accumulate`: fp16 operands
            if (t.t1 == 256 && t.t2 == 192) rc = launch_tn_group<256, 192, 8, 0, true>(g, items, st);
            else if (t.t1 == 256) rc = launch_tn_group<256, 128, 8, 0, true>(g, items, st);
            else rc = launch_tn_group<128, 128, 4, 0, true>(g, items, st);
        } else if (t.t1 == 256 && t.t2 == 192) rc = launch_tn_group<256, 192, 8>(g, items, st);
        else if (t.t1 == 256) rc = launch_tn_group<256, 128, 8>(g, items, st);
        else rc = launch_tn_group<128, 128, 4>(g, items, st);
        if (rc) return rc;
        if (splits == 1) cldrd_norm_sink_miss();      // tiles written directly by the GEMM: no slab reduction to take the sums of squares from
        if (splits > 1) {
            size_t biggest = 0;
            for (int i = 0; i < m; ++i) biggest = biggest > (size_t)g.p[i].N1 * g.p[i].N2 / 4 ? biggest : (size_t)g.p[i].N1 * g.p[i].N2 / 4;
            const int rb = (int)((biggest + 255) / 256 < 256 ? (biggest + 255) / 256 : 256);
            g.sq_out = cldrd_norm_sink_take(rb * m);       // clip-norm partial sums of what this reduction writes, when a sink is set
            hipLaunchKernelGGL(reduce_slabs_group_kernel, dim3(rb, m), dim3(256), 0, st, g);       // one reduction launch for the group
            CLDRD_LAUNCH_CHECK();
        }
    }
    return 0;
}

// the single-problem form (a group of one).  workspace floats needed: cldrd_wgrad_splits(M, N1, N2) * (N1*N2 + N1)
extern "C" int cldrd_wgrad16(const void* A, const void* B, float* dW, float* dbias, int M, int N1, int N2, int lda, int ldb,
                                float* workspace, size_t workspace_bytes, int accumulate, void* stream) {
    return cldrd_wgrad_group(&A, &B, &dW, &dbias, &M, &N1, &N2, &lda, &ldb, 1, workspace, workspace_bytes, accumulate, stream);
}
