// fp16-operand instantiations of gemm_nt_ring_kernel: the forward FFN GEMMs of the encoder (HF ffn.lin1 / ffn.lin2, intermediate.dense /
// output.dense; reference call sites models/nway_dual_encoder.py:52,56,64), its out-projection (round 3: fp16 context from the attention kernel)
// and its QKV projection (fp16 operands, bf16 result).
//
// Why these two GEMMs: of all 16-bit rounding points of a layer the operands of the FFN GEMMs carry the logit drift (CPU emulation on the
// cfg1 golden, DESIGN.md section 2: every operand bf16 0.234, FFN operands fp16 and the rest bf16 0.051, everything fp16 0.024, against
// 0.036 for the reference's own fp16 autocast).  fp16 has the same MFMA rate as bf16 and 8x finer rounding, and forward activations of
// a BERT encoder are far inside its range (the reference itself trains under fp16 autocast, nway_listwise_1.py:334).  The backward's
// MFMAs multiply these activations with bf16 gradients, so the training forward leaves a bf16 copy of h (GemmNtArgs::c_copy) and of the
// LayerNorm output next to the fp16 tensors.
#include "gemm_nt_ring_kernel.h"

namespace {

template <int BN>
int launch_ring16(const GemmNtArgs& a, hipStream_t st) {
    switch (epi_flavour(a)) {
        case EPI_F16IN | EPI_BIAS: return launch_ring_epi<BN, EPI_F16IN | EPI_BIAS>(a, st);      // QKV projection: fp16 operands, bf16 q / k / v (c_bf16)
        case EPI_F16IN | EPI_BIAS | EPI_PREACT | EPI_GELU | EPI_DGELU:
            return launch_ring_epi<BN, EPI_F16IN | EPI_BIAS | EPI_PREACT | EPI_GELU | EPI_DGELU>(a, st);
        case EPI_F16IN | EPI_BIAS | EPI_GELU: return launch_ring_epi<BN, EPI_F16IN | EPI_BIAS | EPI_GELU>(a, st);
        // out-projection of the first layer (its residual is the embedding block's fp32 output: no LayerNorm still to apply)
        case EPI_F16IN | EPI_BIAS | EPI_RESIDUAL | EPI_RES32 | EPI_F32:
            return launch_ring_epi<BN, EPI_F16IN | EPI_BIAS | EPI_RESIDUAL | EPI_RES32 | EPI_F32>(a, st);
        case EPI_F16IN | EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL | EPI_RES32 | EPI_F32:
            return launch_ring_epi<BN, EPI_F16IN | EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL | EPI_RES32 | EPI_F32>(a, st);
        case EPI_F16IN | EPI_BIAS | EPI_RESIDUAL | EPI_RES32 | EPI_F32 | EPI_RESLN:
            return launch_ring_epi<BN, EPI_F16IN | EPI_BIAS | EPI_RESIDUAL | EPI_RES32 | EPI_F32 | EPI_RESLN>(a, st);
        case EPI_F16IN | EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL | EPI_RES32 | EPI_F32 | EPI_RESLN:
            return launch_ring_epi<BN, EPI_F16IN | EPI_BIAS | EPI_DROPOUT | EPI_RESIDUAL | EPI_RES32 | EPI_F32 | EPI_RESLN>(a, st);
        // round 4, the all-fp16 training mode: the backward's data-gradient GEMMs on fp16 operands (gradients carry the loss scale)
        case EPI_F16IN: return launch_ring_epi<BN, EPI_F16IN>(a, st);
        case EPI_F16IN | EPI_GELUGRAD | EPI_DGELU: return launch_ring_epi<BN, EPI_F16IN | EPI_GELUGRAD | EPI_DGELU>(a, st);
        case EPI_F16IN | EPI_F32: return launch_ring_epi<BN, EPI_F16IN | EPI_F32>(a, st);
        default: return -1;                               // not built: the caller falls back to the 128 x 128 kernel (or refuses)
    }
}

}  // namespace

// -1: this (flavour, tile) is not built for fp16 operands
int cldrd_gemm_nt_ring16_launch(const GemmNtArgs& a, int bn, hipStream_t st) {
    if (bn == 256 && a.N % 256 == 0) return launch_ring16<256>(a, st);
    if (bn == 192 && a.N % 192 == 0) return launch_ring16<192>(a, st);
    return -1;
}
