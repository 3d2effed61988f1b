// NT GEMM, large-M variant: C[M,N] = epilogue(alpha * A[M,K] . B[N,K]^T), bf16 in, fp32 accumulate.
//
// Why a second kernel: counters on the 128x128 kernel (profiles/r01_gemm_pmc.md) show ~43 % MFMA busy with the
// L2 -> LDS path at ~27 B/clk/CU, i.e. bound by operand traffic: a 128x128 tile moves 32 KiB per 2.1 MFLOP.  Here:
//   * 256 x BN output tile (BN = 256 or 192), 512 threads = 2 x 4 waves of 128 x BN/4: 1.75-2x the flops per byte;
//   * 64-deep K tiles in two LDS slots, every DMA row a full 128-B line (32-deep slices fetched half lines and were no
//     faster than the small tile); operands arrive by LDS-DMA (global_load_lds, 16 B/lane), XOR-swizzled on the source
//     address (chunk ^= row & 7) and on the ds_read_b128 address: conflict-free fragment reads;
//   * software pipeline at 32-deep k-step granularity: while the MFMAs of one k-step run, the ds_reads of the next
//     k-step's fragments are threaded between them (pinned with sched_group_barrier) into a second register set,
//     so a slot is free for the DMA of K tile t+2 half-way through K tile t;  ONE raw s_barrier per 64-deep K tile,
//     placed after the first MFMAs of k-step 1 so the DMA has a whole K-tile period to land;
//   * BN = 192 exists because N = 768 / 2304 / 3072 with M = 32768 then give 512 / 1536 / 2048 tiles:
//     whole multiples of the 256 CUs at one workgroup per CU.
#pragma once
#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BM = 256, BK = 64;

constexpr int A_BYTES = BM * BK * 2;     // 32 KiB

// LDS of one workgroup: [NA A slots | NB B slots] (see the kernel): three whole K tiles where they fit (BN = 128), else two plus a third A slot
template <int BN>
constexpr int ring_lds_bytes() {
    constexpr int b = BN * BK * 2;
    constexpr int nslot = (3 * (A_BYTES + b) <= 160 * 1024) ? 3 : 2;
    constexpr int na = (nslot == 2 && 3 * A_BYTES + 2 * b <= 160 * 1024) ? 3 : nslot;
    return na * A_BYTES + nslot * b;
}

// ABL != 0: timing experiments only (tools/gemm_ablate.py; results are wrong): 1 no s_barrier, 2 no LDS-DMA inside the K loop,
// 3 neither (and no vmcnt waits), 4 no fragment reads inside the K loop, 6 DMA issued but never waited for, 7 no epilogue (one
// conditional store keeps the accumulators alive), 8 epilogue of every tile written to the output's first tile (no HBM write stream).
// Measured at M = 32768, N = 768, K = 3072 (us): full 144 | 1: 139-144 | 2: 118 | 3: 112 | 4: 145 | 6: 140 -> the fragment reads are
// free, barrier + waits cost ~4 %, the ISSUE of the LDS-DMA pieces ~16 % (spreading them one per MFMA row was worse: 157).
// Tried in round 3 and dropped (profiles/r03_microbench.txt): the same wave tile in 128-row workgroups of 4 waves with 80 KiB of LDS, two of
// them per CU, so that one runs its K loop while the other is in its epilogue (21 % of this kernel's time at K = 768).  Each then stages
// its own B tile (40 instead of 28 KiB of LDS-DMA per 128 rows and K tile) and that costs more than the overlap returns: -14 % (QKV) to
// -20 % (FFN1) at K = 768, -4..-8 % at K = 2304 / 3072, -2.7 % on the training step.
#ifdef CLDRD_DEV_BUILD
// ABL == 10 (tools/epi_stamps.py): wave 0 of the first 1024 workgroups leaves cycle-counter stamps: [0] kernel entry, [1] first K tile landed,
// [2] K loop done, [3] epilogue barrier passed, [4..7] after each 32-row chunk of its epilogue
__device__ unsigned long long g_ring_stamps[1024 * 8];
#endif
template <int BN, int EPI, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt_ring_kernel(GemmNtArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int B_BYTES = BN * BK * 2;
    constexpr int NT = BN / 64;              // 16-col MFMA tiles per wave
    constexpr int WN = BN / 4;               // wave tile width
    constexpr int BPW = BN / 64;             // 1-KiB B pieces per wave (8 rows x 128 B each): BN/8 pieces over 8 waves
    constexpr int NSLOT = (3 * (A_BYTES + BN * BK * 2) <= 160 * 1024) ? 3 : 2;     // BN = 128: two K tiles in flight
    constexpr int G = 4 + BPW;               // LDS-DMA instructions per wave per K tile
    // Round 3: separate rings for the two operands.  LDS = [NA A slots | NB B slots].  Where only two whole K tiles fit (BN = 192 / 256:
    // every encoder shape) the A operand still gets a THIRD slot (3 x 32 KiB + 2 x 24 / 32 KiB = 144 / 160 KiB): A is the activation matrix
    // that streams in from HBM / the Infinity Cache (1.5-2 us per request under load, more than the 1.15 us of MFMA work per K tile), B the
    // weight panel that lives in L2.  A is then requested TWO K tiles ahead (p.asym; 0 = two slots each, the round-2 schedule).
    constexpr int NB_ = NSLOT;
    constexpr int NA_MAX = (NSLOT == 2 && 3 * A_BYTES + 2 * B_BYTES <= 160 * 1024) ? 3 : NSLOT;
    const int na_ = (NA_MAX > NSLOT && p.asym != 0) ? NA_MAX : NSLOT;        // A slots in use (uniform)
    constexpr int BRING = NA_MAX * A_BYTES;  // byte offset of the B ring
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef CLDRD_DEV_BUILD
    unsigned long long* stamps = (ABL == 10 && blockIdx.x < 1024 && wid == 0) ? g_ring_stamps + blockIdx.x * 8 : nullptr;
    auto stamp = [&](int k) { if (ABL == 10) { const unsigned long long t = __builtin_readcyclecounter(); if (stamps && lane == 0) stamps[k] = t; } };
#else
    auto stamp = [&](int) {};
    unsigned long long* stamps = nullptr;
#endif
    stamp(0);
    if (p.phase_units > 0 && blockIdx.x < 256 && ((blockIdx.x >> 3) & 1)) {
        // first round of tiles only (later rounds inherit their CU's phase): every second workgroup of an XCD starts `phase_units` of the
        // s_memtime counter late (1 unit = 0.064 us), so that half of the chip is in its K loop while the other half is in its epilogue
        const unsigned long long t0 = __builtin_readcyclecounter();
        while ((long long)(__builtin_readcyclecounter() - t0) < (long long)p.phase_units) __builtin_amdgcn_s_sleep(8);
    }
    const int ntn = (p.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    // Tile order inside an XCD's contiguous range.  Row-major (all N tiles of an M panel, then the next panel) streams the
    // whole B operand through the 4-MiB L2 once per round of tiles: for N = 3072, K = 768 that is 4.7 MB of weights + the
    // A panels + the output stream, and B was re-fetched ~6x per XCD (PMC: 293 MB read for 55 MB of operands).  So the N
    // tiles are walked in groups of `gn` (<= ~2 MB of B): chunk of `mc` M panels (= one XCD's share) x group x panel x tile.
    int mt_, nt_;
    if (p.gn > 0 && p.gn < ntn) {
        const int ntm = (p.M + BM - 1) / BM;
        const int mc = (ntm + 7) / 8;
        const int c = tile / (mc * ntn), r = tile % (mc * ntn);
        const int mrows = min(mc, ntm - c * mc);
        const int g = r / (mrows * p.gn);
        const int r2 = r - g * mrows * p.gn;
        const int gw = min(p.gn, ntn - g * p.gn);
        mt_ = c * mc + r2 / gw;
        nt_ = g * p.gn + r2 % gw;
    } else {
        mt_ = tile / ntn;
        nt_ = tile % ntn;
    }
    const int m0 = mt_ * BM, n0 = nt_ * BN;
    const int wm = wid >> 2, wn = wid & 3;
    const int nk_ = p.K / BK;

    // ---- LDS-DMA: piece = 8 rows x 128 B; lane -> row (lane >> 3), LDS chunk (lane & 7), source chunk swizzled ----
    // 32-bit byte offsets from the operand base (the launcher checks the operands are < 4 GiB)
    const int prow = lane >> 3;
    const int schunk = (lane & 7) ^ prow;
    uint32_t oa[4], ob[BPW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        oa[i] = (uint32_t)min(m0 + (4 * wid + i) * 8 + prow, p.M - 1) * (uint32_t)(p.lda * 2) + schunk * 16;
#pragma unroll
    for (int i = 0; i < BPW; ++i)
        ob[i] = (uint32_t)min(n0 + (BPW * wid + i) * 8 + prow, p.N - 1) * (uint32_t)(p.ldb * 2) + schunk * 16;
    const uint32_t lds0 = lds_addr_of(smem);
    auto stageA = [&](int slot, int kt) {
        const uint32_t base = lds0 + slot * A_BYTES + 4 * wid * 1024;
        const char* pa = (const char*)p.A + kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(pa, oa[i], base + i * 1024);
    };
    auto stageB = [&](int slot, int kt) {
        const uint32_t base = lds0 + BRING + slot * B_BYTES + BPW * wid * 1024;
        const char* pb = (const char*)p.B + kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < BPW; ++i) lds_dma16(pb, ob[i], base + i * 1024);
    };
    // what a slot recycle at the barrier of K tile kt issues: B of K tile kt + NB into tile kt's B slot, then A of K tile kt + na into its
    // A slot - B first, so that the wait for K tile kt+1 (all of B(kt+1), A(kt+1)) can leave exactly the youngest requests, A(kt+2), in flight
    auto recycle = [&](int slotA, int slotB, int kt) {
        if (kt + NB_ < nk_) stageB(slotB, kt + NB_);
        if (kt + na_ < nk_) stageA(slotA, kt + na_);
    };

    // ---- fragment addressing: row (lane & 15) of a 16-row tile; 16-B chunk 4*ks + (lane >> 4), XOR (row & 7) ----
    const int frow = lane & 15;
    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ch = ((4 * ks + (lane >> 4)) ^ (frow & 7)) * 16;
        a_off[ks] = (wm * 128 + frow) * 128 + ch;
        b_off[ks] = BRING + (wn * WN + frow) * 128 + ch;
    }

    f32x4 acc[8][NT];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    bf16x8 af[8], b0[NT], b1[NT];
    auto mfma_row = [&](int mt, bf16x8 (&bc)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = gemm_mfma<EPI>(bc[nt], af[mt], acc[mt][nt]);
    };
    // One 32-deep k-step.  A fragments are refilled IN PLACE for the next k-step as soon as their last MFMA has issued
    // (the refill of af[mt] has 6*NT MFMAs to land); only the B fragments are double-buffered (bc -> bn).
    //   MFMA rows 0,1 | [sync] | reads af[0], af[1], bn[*] | (MFMA row mt, read af[mt]) for mt = 2..7
    // `sync` (k-step 1 only): K tile kt+1 has landed and everybody has finished with this slot -> recycle it.
    // The DMA issue of a K tile costs a wave about as many issue cycles as its 32 MFMAs (8 pieces x 100-185 cycles,
    // MI355X_MICROARCH.md), and the two waves of a SIMD (w and w + 4) leave the barrier together: issued at the same point
    // they leave the MFMA pipe idle for that long.  Waves 4..7 therefore postpone their pieces to the following k-step
    // (`late`), so one wave of each SIMD feeds the MFMA pipe while the other one issues: +2..9 % on the encoder shapes
    // (a whole k-step later is too late for two LDS slots: -10 %).
    const bool late_wave = wid >= 4 && p.stagger != 0;
    auto kstep = [&](bf16x8 (&bc)[NT], bf16x8 (&bn)[NT], const char* na, const char* nb, bool sync, int slot, int slotB, int kt) {
        mfma_row(0, bc);
        mfma_row(1, bc);
        __builtin_amdgcn_sched_barrier(0);
        if (sync) {
            // K tile kt+1 must have landed; what may stay in flight are the requests for later K tiles, which are the YOUNGEST ones of this
            // wave (recycle() order): with three slots each K tile kt+2 (G pieces), with the third A slot A(kt+2) (4 pieces)
            if (ABL == 3 || ABL == 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else if (NSLOT == 3 && kt + 2 < nk_) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G) : "memory");
            else if (NSLOT == 2 && na_ == 3 && kt + 2 < nk_) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (ABL != 1 && ABL != 3) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (ABL != 2 && ABL != 3 && !late_wave) recycle(slot, slotB, kt);      // slots of K tile kt: its fragments are in registers everywhere
            __builtin_amdgcn_sched_barrier(0);
        } else if (ABL != 2 && ABL != 3 && late_wave && slot >= 0) {
            recycle(slot, slotB, kt);                       // slots of the previous K tile, freed at its barrier
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ABL == 4) {
#pragma unroll
            for (int t = 0; t < NT; ++t) bn[t] = bc[t];
#pragma unroll
            for (int mt = 2; mt < 8; ++mt) mfma_row(mt, bc);
            __builtin_amdgcn_sched_barrier(0);
            return;
        }
        af[0] = *(const bf16x8*)(na);
        af[1] = *(const bf16x8*)(na + 16 * 128);
#pragma unroll
        for (int t = 0; t < NT; ++t) bn[t] = *(const bf16x8*)(nb + t * 16 * 128);
#pragma unroll
        for (int mt = 2; mt < 8; ++mt) {
            mfma_row(mt, bc);
            af[mt] = *(const bf16x8*)(na + mt * 16 * 128);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2 + NT, 0);
#pragma unroll
        for (int mt = 2; mt < 8; ++mt) {
            __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto klast = [&](bf16x8 (&bc)[NT]) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) mfma_row(mt, bc);
    };

    // Prologue: every slot is free at tile start, so K tiles 0 and 1 (and A of K tile 2 with the third A slot) are requested back to back and
    // only tile 0 is waited for.  (Until round 3 the two-slot instances requested tile 1 only after tile 0 had landed: its whole HBM /
    // Infinity-Cache latency sat in front of the first slot recycle of every tile; p.early1 = 0 restores that for A/B runs.)
    // Request order = what the waits count on: A0 B0 | B1 A1 | A2.
    stageA(0, 0);
    stageB(0, 0);
    const bool early1 = nk_ > 1 && (NSLOT == 3 || p.early1 != 0 || na_ == 3);
    const bool early2 = early1 && na_ == 3 && NSLOT == 2 && nk_ > 2;
    if (early1) {
        stageB(1, 1);
        stageA(1, 1);
        if (early2) {
            stageA(2, 2);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G + 4) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
        }
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stamp(1);
    if (NSLOT == 3) { if (nk_ > 2) { stageB(2, 2); stageA(2, 2); } } else { if (nk_ > 1 && !early1) { stageB(1, 1); stageA(1, 1); } }
#pragma unroll
    for (int t = 0; t < NT; ++t) b0[t] = *(const bf16x8*)(smem + b_off[0] + t * 16 * 128);
#pragma unroll
    for (int t = 0; t < 8; ++t) af[t] = *(const bf16x8*)(smem + a_off[0] + t * 16 * 128);

    int csA = 0, csB = 0, psA = -1, psB = -1;                                   // A / B slots of K tiles kt and kt-1
    for (int kt = 0; kt + 1 < nk_; ++kt) {
        const int nsA = csA == na_ - 1 ? 0 : csA + 1, nsB = csB == NB_ - 1 ? 0 : csB + 1;
        const char* curA = smem + csA * A_BYTES;
        const char* curB = smem + csB * B_BYTES;
        const char* nxtA = smem + nsA * A_BYTES;
        const char* nxtB = smem + nsB * B_BYTES;
        kstep(b0, b1, curA + a_off[1], curB + b_off[1], false, psA, psB, kt - 1);   // k-step 0; prefetch k-step 1 of this tile's slots
        kstep(b1, b0, nxtA + a_off[0], nxtB + b_off[0], true, csA, csB, kt);        // k-step 1; prefetch k-step 0 of K tile kt+1
        psA = csA; psB = csB;
        csA = nsA; csB = nsB;
    }
    {   // last K tile: nothing left to recycle (a postponed issue of K tile nk-2 would be for K tiles >= nk)
        kstep(b0, b1, smem + csA * A_BYTES + a_off[1], smem + csB * B_BYTES + b_off[1], false, -1, -1, nk_);
        klast(b1);
    }

    if constexpr ((EPI & EPI_FILTER) != 0 && EPI != EPI_GENERIC) {
        gemm_nt_filter_epilogue_cols<8, NT>(p, acc, m0 + wm * 128, n0 + wn * WN, lane);
    } else {
        stamp(2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();     // last K tile fully consumed by every wave: the slots become epilogue scratch
        asm volatile("" ::: "memory");
        stamp(3);
        if constexpr (ABL == 10) {
            if constexpr (epi_is_f32_only<EPI>()) gemm_nt_epilogue_f32<8, NT, EPI>(p, acc, m0 + wm * 128, n0 + wn * WN, lane, (float*)smem + wid * (32 * (WN + 4)), stamps ? stamps + 4 : nullptr);
            else gemm_nt_epilogue<8, NT, EPI>(p, acc, m0 + wm * 128, n0 + wn * WN, lane, (float*)smem + wid * (32 * (WN + 4)), stamps ? stamps + 4 : nullptr);
            return;
        }
        if constexpr (ABL == 7) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
            if (s == 123.456f) ((bf16_t*)p.C)[threadIdx.x] = 0;
            return;
        }
        if constexpr (ABL == 8) {
            gemm_nt_epilogue<8, NT, EPI>(p, acc, wm * 128, wn * WN, lane, (float*)smem + wid * (32 * (WN + 4)));
            return;
        }
        if constexpr (ABL == 11) {      // every CU stores to (and reads the epilogue operands of) a place of its own: tile index folded onto the first 256 -> no HBM streaming beyond the first round, no shared lines
            const int tf = tile & 255;
            gemm_nt_epilogue<8, NT, EPI>(p, acc, (tf & 127) * BM + wm * 128, (tf >> 7) * BN + wn * WN, lane, (float*)smem + wid * (32 * (WN + 4)));
            return;
        }
        if constexpr (ABL == 9) {       // the tile's own columns, M panel folded onto the first four: a cache-resident footprint without same-line conflicts
            gemm_nt_epilogue<8, NT, EPI>(p, acc, (mt_ & 3) * BM + wm * 128, n0 + wn * WN, lane, (float*)smem + wid * (32 * (WN + 4)));
            return;
        }
        if constexpr (epi_is_f32_only<EPI>()) gemm_nt_epilogue_f32<8, NT, EPI>(p, acc, m0 + wm * 128, n0 + wn * WN, lane, (float*)smem + wid * (32 * (WN + 4)));
        else gemm_nt_epilogue<8, NT, EPI>(p, acc, m0 + wm * 128, n0 + wn * WN, lane, (float*)smem + wid * (32 * (WN + 4)));
    }
}

#ifdef CLDRD_DEV_BUILD
extern "C" unsigned long long g_dev_stamps_host[1024 * 8];      // capi.hip (dev build): the last stamped launch's stamps, read with cldrd_dev_stamps()
#endif
template <int BN, int ABL>
int launch_ring_abl(const GemmNtArgs& a, hipStream_t st) {
    constexpr int lds = ring_lds_bytes<BN>();
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<BN, 0, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int nblk = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_nt_ring_kernel<BN, 0, ABL>), dim3(nblk), dim3(512), lds, st, a);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

template <int BN, int EPI>
int launch_ring_epi(const GemmNtArgs& a, hipStream_t st) {
#ifdef CLDRD_DEV_BUILD                                 // timing-only ablations (WRONG results): development build only, never in the product library
    if (EPI == 0 && BN == 192) {                       // ablations exist for the plain BN = 192 instance only
        switch (cldrd_dev_int("CLDRD_GEMM_ABLATE", 0)) {
            case 1: return launch_ring_abl<192, 1>(a, st);
            case 2: return launch_ring_abl<192, 2>(a, st);
            case 3: return launch_ring_abl<192, 3>(a, st);
            case 4: return launch_ring_abl<192, 4>(a, st);
            case 6: return launch_ring_abl<192, 6>(a, st);
            case 7: return launch_ring_abl<192, 7>(a, st);
            case 8: return launch_ring_abl<192, 8>(a, st);
            default: break;
        }
    }
#endif
    constexpr int lds = ring_lds_bytes<BN>();
#ifdef CLDRD_DEV_BUILD
    if constexpr (EPI != 0 && (EPI & EPI_FILTER) == 0 && EPI != EPI_GENERIC) {      // the epilogue ablations (7: none, 8: no HBM streams) for every fused flavour: tools/epi_ablate.py
        const int abl = cldrd_dev_int("CLDRD_GEMM_ABLATE_EPI", 0);
        if (abl == 10) {
            static bool attr10 = false;
            if (!attr10) { (void)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<BN, EPI, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr10 = true; }
            const int nb = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
            hipLaunchKernelGGL((gemm_nt_ring_kernel<BN, EPI, 10>), dim3(nb), dim3(512), lds, st, a);
            CLDRD_LAUNCH_CHECK();
            return hipMemcpyFromSymbolAsync(g_dev_stamps_host, HIP_SYMBOL(g_ring_stamps), sizeof(unsigned long long) * 1024 * 8, 0, hipMemcpyDeviceToHost, st) == hipSuccess ? 0 : 1;
        }
        if (abl == 11) {
            static bool attr11 = false;
            if (!attr11) { (void)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<BN, EPI, 11>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr11 = true; }
            const int nb = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
            hipLaunchKernelGGL((gemm_nt_ring_kernel<BN, EPI, 11>), dim3(nb), dim3(512), lds, st, a);
            CLDRD_LAUNCH_CHECK();
            return 0;
        }
        if (abl == 7 || abl == 8 || abl == 9) {
            static bool attr7 = false;
            if (!attr7) {
                (void)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<BN, EPI, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                (void)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<BN, EPI, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                (void)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<BN, EPI, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                attr7 = true;
            }
            const int nb = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
            if (abl == 7) hipLaunchKernelGGL((gemm_nt_ring_kernel<BN, EPI, 7>), dim3(nb), dim3(512), lds, st, a);
            else if (abl == 9) hipLaunchKernelGGL((gemm_nt_ring_kernel<BN, EPI, 9>), dim3(nb), dim3(512), lds, st, a);
            else hipLaunchKernelGGL((gemm_nt_ring_kernel<BN, EPI, 8>), dim3(nb), dim3(512), lds, st, a);
            CLDRD_LAUNCH_CHECK();
            return 0;
        }
    }
#endif
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<BN, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int nblk = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_nt_ring_kernel<BN, EPI>), dim3(nblk), dim3(512), lds, st, a);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

}  // namespace
