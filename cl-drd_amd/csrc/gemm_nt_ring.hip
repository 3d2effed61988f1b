// Large-M NT GEMM: dispatch and the bf16 instantiations of gemm_nt_ring_kernel (gemm_nt_ring_kernel.h); the fp16-operand forward
// flavours are instantiated in gemm_nt_ring16.hip (a second translation unit: the two compile side by side).
#include "gemm_nt_ring_kernel.h"

namespace {

template <int BN>
int launch_ring(const GemmNtArgs& a, hipStream_t st) {
    switch (epi_flavour(a)) {
        case 0: return launch_ring_epi<BN, 0>(a, st);
        case EPI_BIAS: return launch_ring_epi<BN, EPI_BIAS>(a, st);
        case EPI_BIAS | EPI_PREACT | EPI_GELU: return launch_ring_epi<BN, EPI_BIAS | EPI_PREACT | EPI_GELU>(a, st);
        case EPI_BIAS | EPI_PREACT | EPI_GELU | EPI_DGELU: return launch_ring_epi<BN, EPI_BIAS | EPI_PREACT | EPI_GELU | EPI_DGELU>(a, st);
        case EPI_GELUGRAD | EPI_DGELU: return launch_ring_epi<BN, EPI_GELUGRAD | EPI_DGELU>(a, st);
        case EPI_BIAS | EPI_GELU: return launch_ring_epi<BN, EPI_BIAS | EPI_GELU>(a, st);
        case EPI_BIAS | EPI_RESIDUAL: return launch_ring_epi<BN, EPI_BIAS | EPI_RESIDUAL>(a, st);
        case EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL: return launch_ring_epi<BN, EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL>(a, st);
        case EPI_GELUGRAD: return launch_ring_epi<BN, EPI_GELUGRAD>(a, st);
        case EPI_RESIDUAL: return launch_ring_epi<BN, EPI_RESIDUAL>(a, st);
        case EPI_F32: return launch_ring_epi<BN, EPI_F32>(a, st);
        // data gradients of the fp32 gradient stream: fp32 residual (the LayerNorm backward's dx) in, fp32 sum out
        case EPI_RESIDUAL | EPI_RES32 | EPI_F32: return launch_ring_epi<BN, EPI_RESIDUAL | EPI_RES32 | EPI_F32>(a, st);
        case EPI_BIAS | EPI_RESIDUAL | EPI_RES32 | EPI_F32: return launch_ring_epi<BN, EPI_BIAS | EPI_RESIDUAL | EPI_RES32 | EPI_F32>(a, st);
        case EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL | EPI_RES32 | EPI_F32:
            return launch_ring_epi<BN, EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL | EPI_RES32 | EPI_F32>(a, st);
        case EPI_BIAS | EPI_RESIDUAL | EPI_RES32 | EPI_F32 | EPI_RESLN:
            return launch_ring_epi<BN, EPI_BIAS | EPI_RESIDUAL | EPI_RES32 | EPI_F32 | EPI_RESLN>(a, st);
        case EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL | EPI_RES32 | EPI_F32 | EPI_RESLN:
            return launch_ring_epi<BN, EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL | EPI_RES32 | EPI_F32 | EPI_RESLN>(a, st);
        default: return launch_ring_epi<BN, EPI_GENERIC>(a, st);
    }
}

}  // namespace

// Top-k scan with the index rows as M (256-row tiles) and up to 128 queries as N.
int cldrd_gemm_nt_ring_scan(const GemmNtArgs& a, hipStream_t st) {
    return a.in_f16 ? launch_ring_epi<128, EPI_FILTER | EPI_F16IN>(a, st) : launch_ring_epi<128, EPI_FILTER>(a, st);
}

int cldrd_gemm_nt_ring16_launch(const GemmNtArgs& a, int bn, hipStream_t st);    // gemm_nt_ring16.hip

// Returns -1 if this variant does not apply (caller falls back to the 128x128 kernel), else the launch status.
int cldrd_gemm_nt_ring_dispatch(const GemmNtArgs& a_in, int force_bn, hipStream_t st) {
    GemmNtArgs a = a_in;
    int gn_force;
    a.stagger = CLDRD_DEV_INT("CLDRD_GEMM_STAGGER", 1);      // settled A/B pairs (profiles/r02, r03_microbench.txt): live in the development build only
    a.early1 = CLDRD_DEV_INT("CLDRD_GEMM_EARLY1", 1);
    a.asym = CLDRD_DEV_INT("CLDRD_GEMM_ASYM", 1);
    gn_force = CLDRD_DEV_INT("CLDRD_GEMM_GN", -1);
    if (a.K % BK != 0) return -1;
    if ((double)a.M * a.lda * 2.0 >= 4.0e9 || (double)a.N * a.ldb * 2.0 >= 4.0e9) return -1;   // 32-bit DMA offsets
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
            bn = (e192 > e256 + 0.01 * CLDRD_DEV_INT("CLDRD_GEMM_BN_MARGIN", 15)) ? 192 : 256;     // measured: N=768 -> 192, N=2304/3072 -> 256 (profiles/r01_gemm_bench.txt)
        } else {
            bn = ok256 ? 256 : 192;
        }
    }
    // N-tile group: about 2 MB of B (bn rows x K) per group
    // (only for short K: with K >= 2048 the tiles of a round sweep K in step, L2 holds the current K slices of every panel, and a
    // group pass would re-read the A panels - PMC: 436 MB instead of 293 MB for N = 768, K = 2304 / 3072)
    a.gn = gn_force >= 0 ? gn_force : (a.K <= 1024 ? (int)(2.0e6 / ((double)bn * a.K * 2.0) + 0.5) : 0);
    if (a.gn < 0) a.gn = 0;
    {   // development build: phase shift of half the first round's workgroups (tools/epi_ablate.py), launches of >= MIN_ROUNDS rounds only
        const int ph = CLDRD_DEV_INT("CLDRD_GEMM_PHASE", 0), minr = CLDRD_DEV_INT("CLDRD_GEMM_PHASE_MIN_ROUNDS", 3);
        const long tiles = (long)((a.M + BM - 1) / BM) * (a.N / bn);
        a.phase_units = (ph > 0 && tiles >= 256L * minr) ? ph : 0;
    }
    // (A persistent tile walk with a register epilogue - gemm_nt_pers.hip of rounds 2-3 - measured equal to this kernel within 1 % on
    // the encoder shapes and in the training step, profiles/r02_microbench.txt, and was removed in round 4.)
    if (a.in_f16) return cldrd_gemm_nt_ring16_launch(a, bn, st);       // fp16 operands: the forward FFN flavours, or -1
    if (bn == 256 && a.N % 256 == 0) return launch_ring<256>(a, st);
    if (bn == 192 && a.N % 192 == 0) return launch_ring<192>(a, st);
    return -1;
}
