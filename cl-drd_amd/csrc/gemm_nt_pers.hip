// NT GEMM, large-M PERSISTENT variant of gemm_nt_ring.hip: C[M,N] = epilogue(alpha * A[M,K] . B[N,K]^T), bf16 in, fp32 accumulate.
//
// Why: on the encoder's K = 768 shapes a 256 x 256 tile has only 12 K tiles, and the ring kernel spent 17 % (QKV) to 52 % (GELU
// backward) of a tile's time outside the K loop (profiles/r02_microbench.txt: rounds of 23-41 us for a ~19-us K loop): the first
// K tile's DMA (64 KB at the ~30-40 GB/s one CU takes in) with the MFMA pipe idle, an epilogue that went through an LDS patch
// aliasing the staging slots (so nothing of the next tile could be in flight), and a workgroup that cannot retire before its
// stores have drained - with one workgroup per CU, all of it serial.  Here:
//   * one workgroup per CU walks its tiles (same XCD-aware order as the ring kernel: virtual block id = blockIdx + round * grid);
//   * the epilogue never touches LDS: values stay in the MFMA accumulator layout (lane = row, 4 consecutive columns), where an
//     fp32 access is 16 rows x 64 contiguous bytes per instruction as it stands; 16-bit operands / results are exchanged between
//     the 16-lane rows of a wave with v_permlane16_swap so that a lane holds 8 consecutive columns (again 64 B per row);
//   * so the DMA of the next tile's first two K tiles is issued BEFORE the epilogue (after the barrier that ends the last
//     fragment reads) and lands under it;
//   * stores are not waited for: the counted vmcnt waits of the next tile's first two K tiles leave exactly this epilogue's
//     stores outstanding (vmcnt counts loads, stores and LDS-DMA together, in issue order), so they drain under the next K loop.
// K loop: unchanged from the ring kernel (64-deep K tiles, two LDS slots, swizzled LDS-DMA, one s_barrier per K tile).
#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BM = 256, BK = 64;
constexpr int A_BYTES = BM * BK * 2;     // 32 KiB

typedef unsigned pers_u32x2 __attribute__((ext_vector_type(2)));
// v_permlane16_swap: rows (16 lanes) 1 and 3 of `d` trade places with rows 0 and 2 of `s`.  An involution on the pair.
__device__ __forceinline__ void swap16(uint32_t& d, uint32_t& s) {
    const pers_u32x2 r = __builtin_amdgcn_permlane16_swap(d, s, false, false);
    d = r[0]; s = r[1];
}

// One LDS-DMA piece (1 KiB per wave): LDS destination = wave-uniform m0 + lane*16, source = uniform 64-bit base + per-lane 32-bit
// offset.  Written as asm so that the address stays in the saddr form (hipcc turned `base + kt*128 + zext(off)` into two 64-bit
// VALU adds per piece and kept every offset as a register PAIR), and so that hipcc's own waitcnt bookkeeping does not see these
// (its waits for the epilogue's loads can then only be stricter than needed, never laxer: unknown older operations).
__device__ __forceinline__ void dma16(uint32_t lds_addr, uint32_t voff, const void* sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}
// Makes a value opaque to the optimiser at this point: address arithmetic derived from it cannot be hoisted out of the tile loop
// (loop-invariant per-lane terms of the epilogue's addresses cost ~30 VGPRs across the K loop and spilled the 256-wide tile).
__device__ __forceinline__ int opaque_v(int x) { asm volatile("" : "+v"(x)); return x; }

// 16-bit traffic of a wave tile of 8 x NT accumulator tiles (16 x 16 each) is organised in PAIRS of accumulator tiles (A, B):
// after the swap lane (row r, quad fq) holds 8 consecutive columns of tile (fq & 1 ? B : A), half (fq >> 1), row r.
// NT even: (mt, 2s) with (mt, 2s+1), one row tile per group.  NT odd (3): two row tiles per group, the last column tile of the
// first is paired with the last column tile of the second.
template <int NT> struct PairMap {
    static constexpr int MTG = (NT % 2) ? 2 : 1;
    static constexpr int NG = 8 / MTG;
    static constexpr int PPG = (NT % 2) ? NT : NT / 2;
    static constexpr int H = NT / 2;
    __host__ __device__ static constexpr int tA(int s) { return (NT % 2) ? (s < 2 * H ? s / H : 0) : 0; }     // row tile inside the group
    __host__ __device__ static constexpr int nA(int s) { return (NT % 2) ? (s < 2 * H ? 2 * (s % H) : NT - 1) : 2 * s; }
    __host__ __device__ static constexpr int tB(int s) { return (NT % 2) ? (s < 2 * H ? s / H : 1) : 0; }
    __host__ __device__ static constexpr int nB(int s) { return (NT % 2) ? (s < 2 * H ? 2 * (s % H) + 1 : NT - 1) : 2 * s + 1; }
};

template <int NT, int EPI> struct EpiCount {
    using PM = PairMap<NT>;
    static constexpr int out = (EPI & EPI_F32) ? 8 * NT : PM::NG * PM::PPG;
    static constexpr int pre = (EPI & EPI_PREACT) ? PM::NG * PM::PPG : 0;
    static constexpr int stores = out + pre;             // store instructions per wave per FULL tile
};

// Epilogue straight from the accumulators (operands swapped in the MFMA: lane holds C[row0 + mt*16 + (lane & 15)][col0 + nt*16 +
// 4*(lane >> 4) + j], j = 0..3).  FULL: every row of the tile is inside M (no predication: the store count is exact).
template <int NT, int EPI, bool FULL>
__device__ __forceinline__ void pers_epilogue(const GemmNtArgs& p, f32x4 (&acc)[8][NT], int row0, int col0, int lane_in) {
    const int lane = opaque_v(lane_in);
    using PM = PairMap<NT>;
    constexpr int MTG = PM::MTG, NG = PM::NG, PPG = PM::PPG;
    constexpr bool BIAS = (EPI & EPI_BIAS) != 0, PREACT = (EPI & EPI_PREACT) != 0, GELU = (EPI & EPI_GELU) != 0,
                   GGRAD = (EPI & EPI_GELUGRAD) != 0, DROP = (EPI & EPI_DROPOUT) != 0, RES = (EPI & EPI_RESIDUAL) != 0,
                   F32 = (EPI & EPI_F32) != 0, RES32 = (EPI & EPI_RES32) != 0, DGELU = (EPI & EPI_DGELU) != 0;
    constexpr bool RES16 = RES && !RES32;
    constexpr bool IN16 = GGRAD || RES16;
    static_assert(!(GGRAD && RES16), "one 16-bit input stream per epilogue");
    constexpr int NBUF = (NT % 2 || RES32) ? 3 : 4;      // input groups in flight (incl. the one being consumed)
    const int frow = lane & 15, fq = lane >> 4, pick = fq & 1, half = fq >> 1;
    const int r0 = row0 + frow;
    const int c32 = col0 + 4 * fq;                       // fp32 / accumulator layout: + nt*16
    const int c16 = col0 + 8 * half;                     // 16-bit layout: + nt_pick*16
    f32x4 bias4[NT];
    if constexpr (BIAS) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bias4[nt] = *(const f32x4*)(p.bias + c32 + nt * 16);
    }
    const bf16_t* src16 = GGRAD ? p.gelu_pre : (const bf16_t*)p.residual;
    const int ld16 = GGRAD ? p.ldc : p.ldr;
    uint4 in16[NBUF][PPG];
    f32x4 in32[NBUF][MTG][NT];
    auto issue = [&](int g, int b) {
        if constexpr (RES32) {
#pragma unroll
            for (int t = 0; t < MTG; ++t) {
                const int m = r0 + (g * MTG + t) * 16;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (FULL || m < p.M) in32[b][t][nt] = *(const f32x4*)((const float*)p.residual + (size_t)m * p.ldr + c32 + nt * 16);
                    else in32[b][t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        if constexpr (IN16) {
#pragma unroll
            for (int s = 0; s < PPG; ++s) {
                const int m = r0 + (g * MTG + (pick ? PM::tB(s) : PM::tA(s))) * 16;
                const int n = c16 + (pick ? PM::nB(s) : PM::nA(s)) * 16;
                if (FULL || m < p.M) in16[b][s] = *(const uint4*)(src16 + (size_t)m * ld16 + n);
                else in16[b][s] = make_uint4(0, 0, 0, 0);
            }
        }
    };
    // pack a group's values to bf16, exchange rows, one 16-byte store per pair
    auto store16 = [&](bf16_t* dst, int g, float (&v)[MTG][NT][4]) {
#pragma unroll
        for (int s = 0; s < PPG; ++s) {
            const float (&a)[4] = v[PM::tA(s)][PM::nA(s)];
            const float (&b)[4] = v[PM::tB(s)][PM::nB(s)];
            uint32_t ax = pack2bf(a[0], a[1]), ay = pack2bf(a[2], a[3]), bx = pack2bf(b[0], b[1]), by = pack2bf(b[2], b[3]);
            swap16(ax, bx);
            swap16(ay, by);
            const int m = r0 + (g * MTG + (pick ? PM::tB(s) : PM::tA(s))) * 16;
            const int n = c16 + (pick ? PM::nB(s) : PM::nA(s)) * 16;
            if (FULL || m < p.M) *(uint4*)(dst + (size_t)m * p.ldc + n) = make_uint4(ax, ay, bx, by);
        }
    };
    if constexpr (RES32 || IN16) {
#pragma unroll
        for (int g = 0; g < NBUF - 1 && g < NG; ++g) issue(g, g);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if constexpr (RES32 || IN16) {
            if (g + NBUF - 1 < NG) issue(g + NBUF - 1, (g + NBUF - 1) % NBUF);
        }
        const int b = g % NBUF;
        float v[MTG][NT][4];
#pragma unroll
        for (int t = 0; t < MTG; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[t][nt][j] = acc[g * MTG + t][nt][j] * p.alpha;
                    if constexpr (BIAS) v[t][nt][j] += bias4[nt][j];
                }
        if constexpr (PREACT && GELU && DGELU) {         // save gelu'(x) (derivative form, gemm_epilogue.h): one exp + rcp for value and derivative
            float dv[MTG][NT][4];
#pragma unroll
            for (int t = 0; t < MTG; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float c, e;
                        gelu_parts(v[t][nt][j], c, e);
                        dv[t][nt][j] = fmaf(v[t][nt][j] * 0.39894228040143268f, e, c);
                        v[t][nt][j] *= c;
                    }
            store16(p.preact, g, dv);
        } else {
            if constexpr (PREACT) {
                if constexpr (DGELU) {
                    float dv[MTG][NT][4];
#pragma unroll
                    for (int t = 0; t < MTG; ++t)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int j = 0; j < 4; ++j) dv[t][nt][j] = gelu_grad_f(v[t][nt][j]);
                    store16(p.preact, g, dv);
                } else {
                    store16(p.preact, g, v);
                }
            }
            if constexpr (GELU) {
#pragma unroll
                for (int t = 0; t < MTG; ++t)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[t][nt][j] = gelu_f(v[t][nt][j]);
            }
        }
        float w[MTG][NT][4];
        if constexpr (IN16) {           // back from the 8-columns-per-lane layout to the accumulator layout
#pragma unroll
            for (int s = 0; s < PPG; ++s) {
                uint32_t ax = in16[b][s].x, ay = in16[b][s].y, bx = in16[b][s].z, by = in16[b][s].w;
                swap16(ax, bx);
                swap16(ay, by);
                float (&a)[4] = w[PM::tA(s)][PM::nA(s)];
                float (&bb)[4] = w[PM::tB(s)][PM::nB(s)];
                a[0] = __uint_as_float(ax << 16); a[1] = __uint_as_float(ax & 0xFFFF0000u);
                a[2] = __uint_as_float(ay << 16); a[3] = __uint_as_float(ay & 0xFFFF0000u);
                bb[0] = __uint_as_float(bx << 16); bb[1] = __uint_as_float(bx & 0xFFFF0000u);
                bb[2] = __uint_as_float(by << 16); bb[3] = __uint_as_float(by & 0xFFFF0000u);
            }
        }
        if constexpr (GGRAD) {
#pragma unroll
            for (int t = 0; t < MTG; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[t][nt][j] *= DGELU ? w[t][nt][j] : gelu_grad_f(w[t][nt][j]);
        }
        if constexpr (DROP) {
#pragma unroll
            for (int t = 0; t < MTG; ++t) {
                const uint32_t rk = drop_rowkey(p.seed_base ? p.seed + *p.seed_base : p.seed, (uint32_t)(r0 + (g * MTG + t) * 16));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        const uint32_t h = drop_pair(rk, (uint32_t)(c32 + nt * 16 + j));
                        v[t][nt][j] = drop_keep_lo(h, p.drop_thresh) ? v[t][nt][j] * p.drop_scale : 0.f;
                        v[t][nt][j + 1] = drop_keep_hi(h, p.drop_thresh) ? v[t][nt][j + 1] * p.drop_scale : 0.f;
                    }
            }
        }
        if constexpr (RES32) {
#pragma unroll
            for (int t = 0; t < MTG; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[t][nt][j] += in32[b][t][nt][j];
        }
        if constexpr (RES16) {
#pragma unroll
            for (int t = 0; t < MTG; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[t][nt][j] += w[t][nt][j];
        }
        if constexpr (F32) {
#pragma unroll
            for (int t = 0; t < MTG; ++t) {
                const int m = r0 + (g * MTG + t) * 16;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (FULL || m < p.M)
                        *(f32x4*)((float*)p.C + (size_t)m * p.ldc + c32 + nt * 16) = (f32x4){v[t][nt][0], v[t][nt][1], v[t][nt][2], v[t][nt][3]};
            }
        } else {
            store16((bf16_t*)p.C, g, v);
        }
        __builtin_amdgcn_sched_barrier(0);       // one group at a time: bounds the live ranges (the scheduler otherwise interleaves groups up to 256 VGPRs and the allocator spills)
    }
}

template <int BN, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_pers_kernel(GemmNtArgs p, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int B_BYTES = BN * BK * 2;
    constexpr int SLOT = A_BYTES + B_BYTES;
    static_assert(2 * SLOT <= 160 * 1024 && 3 * SLOT > 160 * 1024, "two LDS slots");
    constexpr int NT = BN / 64;              // 16-col MFMA tiles per wave
    constexpr int WN = BN / 4;               // wave tile width
    constexpr int BPW = BN / 64;             // 1-KiB B pieces per wave
    constexpr int G = 4 + BPW;               // LDS-DMA instructions per wave per K tile
    constexpr int NST = EpiCount<NT, EPI>::stores;
    static_assert(NST + G <= 63, "vmcnt is a 6-bit counter");
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntn = p.N / BN;
    const int ntm = (p.M + BM - 1) / BM;
    const int wm = wid >> 2, wn = wid & 3;
    const int nk_ = p.K / BK;                // >= 2 (launcher)
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);

    // ---- LDS-DMA: piece = 8 rows x 128 B; lane -> row (lane >> 3), LDS chunk (lane & 7), source chunk swizzled ----
    uint32_t oa[4], ob[BPW];
    int lm0 = 0, ln0 = 0;                    // tile the DMA offsets currently point at
    auto locate = [&](int vb) {
        const int l = opaque_v(lane);
        const int prow = l >> 3;
        const int schunk = (l & 7) ^ prow;
        // tile order inside an XCD's contiguous range: see gemm_nt_ring.hip (N tiles walked in groups of gn for short K)
        const int tile = xcd_remap(vb, ntiles);
        int mt_, nt_;
        if (p.gn > 0 && p.gn < ntn) {
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
        lm0 = mt_ * BM;
        ln0 = nt_ * BN;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            oa[i] = (uint32_t)min(lm0 + (4 * wid + i) * 8 + prow, p.M - 1) * (uint32_t)(p.lda * 2) + schunk * 16;
#pragma unroll
        for (int i = 0; i < BPW; ++i)
            ob[i] = (uint32_t)(ln0 + (BPW * wid + i) * 8 + prow) * (uint32_t)(p.ldb * 2) + schunk * 16;
    };
    auto stage = [&](int slot, int kt) {
        const uint32_t base = lds0 + slot * SLOT;
        const char* pa = (const char*)p.A + kt * (BK * 2);
        const char* pb = (const char*)p.B + kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(base + (4 * wid + i) * 1024, oa[i], pa);
#pragma unroll
        for (int i = 0; i < BPW; ++i) dma16(base + A_BYTES + (BPW * wid + i) * 1024, ob[i], pb);
    };
    int vb = blockIdx.x;
    bool more = vb + (int)gridDim.x < ntiles;        // a tile follows the one being accumulated
    // The ring never drains: K tile index k2 >= nk of the current tile is K tile k2 - nk of the workgroup's next tile.
    auto stage_ring = [&](int slot, int k2) {
        if (k2 < nk_) {
            stage(slot, k2);
        } else if (more) {
            if (k2 == nk_) locate(vb + gridDim.x);
            stage(slot, k2 - nk_);
        }
    };

    // ---- fragment addressing: row (lane & 15) of a 16-row tile; 16-B chunk 4*ks + (lane >> 4), XOR (row & 7) ----
    const int frow = lane & 15;
    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ch = ((4 * ks + (lane >> 4)) ^ (frow & 7)) * 16;
        a_off[ks] = (wm * 128 + frow) * 128 + ch;
        b_off[ks] = A_BYTES + (wn * WN + frow) * 128 + ch;
    }

    f32x4 acc[8][NT];
    bf16x8 af[8], b0[NT], b1[NT];
    auto mfma_row = [&](int mt, bf16x8 (&bc)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = gemm_mfma<EPI>(bc[nt], af[mt], acc[mt][nt]);
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    auto read_first = [&](const char* s) {       // fragments of k-step 0 of the K tile in slot `s`
#pragma unroll
        for (int t = 0; t < NT; ++t) b0[t] = *(const bf16x8*)(s + b_off[0] + t * 16 * 128);
#pragma unroll
        for (int t = 0; t < 8; ++t) af[t] = *(const bf16x8*)(s + a_off[0] + t * 16 * 128);
    };
    const bool late_wave = wid >= 4 && p.stagger != 0;
    int post_slot = -1, post_k2 = 0;             // waves 4..7: the DMA issue postponed from the last barrier (see gemm_nt_ring.hip)
    // One 32-deep k-step (see gemm_nt_ring.hip).  `first`: the sync of a tile's first K tile, when the previous tile's NST stores are
    // in flight BEHIND the DMA of K tile 1 - leave exactly those outstanding.
    auto kstep = [&](bf16x8 (&bc)[NT], bf16x8 (&bn)[NT], const char* na, const char* nb, bool sync, int slot, int k2, bool first) {
        mfma_row(0, bc);
        mfma_row(1, bc);
        __builtin_amdgcn_sched_barrier(0);
        if (sync) {
            if (first) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (!late_wave) stage_ring(slot, k2);
            else { post_slot = slot; post_k2 = k2; }
            __builtin_amdgcn_sched_barrier(0);
        } else if (late_wave && post_slot >= 0) {
            stage_ring(post_slot, post_k2);
            post_slot = -1;
            __builtin_amdgcn_sched_barrier(0);
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

    locate(vb);
    int em0 = lm0, en0 = ln0;                    // tile being accumulated
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stage(1, 1);
    zero_acc();
    read_first(smem);
    bool pend = false;                           // exactly NST stores of the previous epilogue may still be in flight
    int kt = 0, cs = 0;
    for (;;) {
        const int ns = cs ^ 1;
        const char* cur = smem + cs * SLOT;
        const char* nxt = smem + ns * SLOT;
        kstep(b0, b1, cur + a_off[1], cur + b_off[1], false, -1, 0, false);               // k-step 0; prefetch k-step 1 of this slot
        kstep(b1, b0, nxt + a_off[0], nxt + b_off[0], true, cs, kt + 2, pend && kt == 0); // k-step 1; prefetch k-step 0 of the next K tile
        if (kt == nk_ - 1) {
            // End of a tile.  The ring already holds K tile 0 of the next tile (landed: the barrier above) and K tile 1 is in flight -
            // waves 4..7 issue theirs now, not one k-step later, so that every wave's stores come AFTER its last DMA in issue order.
            if (late_wave && post_slot >= 0) {
                stage_ring(post_slot, post_k2);
                post_slot = -1;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (em0 + BM <= p.M) {
                pers_epilogue<NT, EPI, true>(p, acc, em0 + wm * 128, en0 + wn * WN, lane);
                pend = true;
            } else {
                pers_epilogue<NT, EPI, false>(p, acc, em0 + wm * 128, en0 + wn * WN, lane);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                pend = false;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!more) break;
            vb += gridDim.x;
            more = vb + (int)gridDim.x < ntiles;
            em0 = lm0;
            en0 = ln0;
            kt = 0;
            zero_acc();
            read_first(nxt);                     // (the prefetch inside the last k-step is dead: re-read rather than keep 48 VGPRs live across the epilogue)
        } else {
            ++kt;
        }
        cs = ns;
    }
}

int pers_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n = v;
    }
    return n;
}

template <int BN, int EPI>
int launch_pers_epi(const GemmNtArgs& a, hipStream_t st) {
    constexpr int lds = 2 * (A_BYTES + BN * BK * 2);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_pers_kernel<BN, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int ntiles = ((a.M + BM - 1) / BM) * (a.N / BN);
    const char* pe = getenv("CLDRD_GEMM_PERSIST");          // 2: one tile per workgroup (A/B runs: the register epilogue without the tile walk)
    const int cap = (pe && atoi(pe) == 2) ? ntiles : pers_num_cus();
    const int nblk = ntiles < cap ? ntiles : cap;
    hipLaunchKernelGGL((gemm_nt_pers_kernel<BN, EPI>), dim3(nblk), dim3(512), lds, st, a, ntiles);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

template <int BN>
int launch_pers(const GemmNtArgs& a, hipStream_t st) {
    switch (epi_flavour(a)) {
        case 0: return launch_pers_epi<BN, 0>(a, st);
        case EPI_BIAS: return launch_pers_epi<BN, EPI_BIAS>(a, st);
        case EPI_BIAS | EPI_PREACT | EPI_GELU: return launch_pers_epi<BN, EPI_BIAS | EPI_PREACT | EPI_GELU>(a, st);
        case EPI_BIAS | EPI_PREACT | EPI_GELU | EPI_DGELU: return launch_pers_epi<BN, EPI_BIAS | EPI_PREACT | EPI_GELU | EPI_DGELU>(a, st);
        case EPI_GELUGRAD | EPI_DGELU: return launch_pers_epi<BN, EPI_GELUGRAD | EPI_DGELU>(a, st);
        case EPI_BIAS | EPI_GELU: return launch_pers_epi<BN, EPI_BIAS | EPI_GELU>(a, st);
        case EPI_BIAS | EPI_RESIDUAL: return launch_pers_epi<BN, EPI_BIAS | EPI_RESIDUAL>(a, st);
        case EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL: return launch_pers_epi<BN, EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL>(a, st);
        case EPI_GELUGRAD: return launch_pers_epi<BN, EPI_GELUGRAD>(a, st);
        case EPI_RESIDUAL: return launch_pers_epi<BN, EPI_RESIDUAL>(a, st);
        case EPI_F32: return launch_pers_epi<BN, EPI_F32>(a, st);
        case EPI_BIAS | EPI_RESIDUAL | EPI_RES32 | EPI_F32: return launch_pers_epi<BN, EPI_BIAS | EPI_RESIDUAL | EPI_RES32 | EPI_F32>(a, st);
        case EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL | EPI_RES32 | EPI_F32:
            return launch_pers_epi<BN, EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL | EPI_RES32 | EPI_F32>(a, st);
        default: return -1;                      // other combinations: the ring kernel's generic epilogue
    }
}

}  // namespace

// Returns -1 when this variant does not apply (the caller goes on to the ring kernel), else the launch status.
// `a.gn` / `a.stagger` are set by the caller (cldrd_gemm_nt_ring_dispatch).
int cldrd_gemm_nt_pers_dispatch(const GemmNtArgs& a, int bn, hipStream_t st) {
    if (a.in_f16 || a.K % BK != 0 || a.K < 2 * BK || a.thr != nullptr) return -1;
    if (bn == 256 && a.N % 256 == 0) return launch_pers<256>(a, st);
    if (bn == 192 && a.N % 192 == 0) return launch_pers<192>(a, st);
    return -1;
}
