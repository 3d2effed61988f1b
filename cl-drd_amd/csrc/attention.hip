// Fused multi-head self-attention for the BERT/DistilBERT encoder, forward and backward.
//
// Reference call site: HF DistilBertSelfAttention / BertSelfAttention reached from
// models/nway_dual_encoder.py:52,56,64: softmax(Q K^T / sqrt(dh) + key mask) -> dropout -> . V,
// dh = 64, sequences <= 256 tokens.  One workgroup owns one (sequence, head): Q, K, V (and dO) live in LDS
// for the whole kernel, so the L x L score matrix never touches HBM and no cross-workgroup reduction exists.
//
// MFMA 32x32x16 bf16 throughout.  Forward computes S^T = K Q^T so that a lane owns one query column: the
// row softmax is a register-local reduction plus one cross-half shuffle, and the S^T accumulator tile is
// reused directly as the A operand of P.V (cdna_hip_programming.md section 3, "accumulator tile as the next MFMA's
// operand"); V is consumed through ds_read_b64_tr_b16 transposing reads.  Backward recomputes P from the
// saved log-sum-exp in two sweeps: key-on-lane (dK, dV accumulate in registers over all query blocks) and
// query-on-lane (dQ accumulates over all key blocks).  Probabilities and dS are rounded to bf16 only as
// MFMA operands; softmax statistics, LSE and delta are fp32.
#include "common.h"

namespace {

constexpr int RSB = 144;                 // LDS row stride in bytes (64 bf16 + 8 pad): b128 row reads conflict-free
constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

__device__ __forceinline__ int rowmap(int t, int h) { return (t & 3) + 8 * (t >> 2) + 4 * h; }

__device__ __forceinline__ bf16x8 row_frag(const char* tile, int row, int s, int h) {
    return *(const bf16x8*)(tile + row * RSB + (2 * s + h) * 16);
}
// 8 elements k = {row0 + 0..3, row0 + 8 + 0..3} of column (col0 + (lane & 31)) -- the k order of an accumulator tile
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int row0, int col0, int lane) {
    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const char* p = tile + (row0 + qq) * RSB + (col0 + 16 * (g & 1) + 4 * pp) * 2;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)p);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(p + 8 * RSB));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ bf16x8 pack8(const float* v) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (short)f2bf(v[j]);
    return r;
}

// 16-bit activation format of the FORWARD kernels: bf16 (default), or fp16 for the high-precision forward of the query tower
// (encoder.py: 11-bit significand operands, same MFMA rate; the backward always runs on the bf16 tape).
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16_t f2h(float f) { const _Float16 h = (_Float16)f; return __builtin_bit_cast(bf16_t, h); }
__device__ __forceinline__ float h2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
template <bool F16> __device__ __forceinline__ bf16_t f2x(float f) { if constexpr (F16) return f2h(f); else return f2bf(f); }
template <bool F16> __device__ __forceinline__ float x2f(bf16_t v) { if constexpr (F16) return h2f(v); else return bf2f(v); }
template <bool F16>
__device__ __forceinline__ bf16x8 pack8x(const float* v) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (short)f2x<F16>(v[j]);
    return r;
}
template <bool F16>
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <bool F16> __device__ __forceinline__ uint32_t pack2x(float lo, float hi) {
    if constexpr (F16) {
        typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
        const cldrd_f32v2 f = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, h2_t));      // one v_cvt_pk_f16_f32
    } else return pack2bf(lo, hi);
}
template <bool F16> __device__ __forceinline__ uint32_t pack2x_scaled(float lo, float hi, float scale) {      // one v_pk_mul_f32 + one convert
    const cldrd_f32v2 m = (cldrd_f32v2){lo, hi} * splat2(scale);
    return pack2x<F16>(m.x, m.y);
}
// sum of the two products of a dword pair of 16-bit values (delta = rowsum(dO . O))
template <bool F16> __device__ __forceinline__ float dot2x(uint32_t a, uint32_t b) {
    if constexpr (F16) return h2f((bf16_t)(a & 0xFFFFu)) * h2f((bf16_t)(b & 0xFFFFu)) + h2f((bf16_t)(a >> 16)) * h2f((bf16_t)(b >> 16));
    else return __uint_as_float(a << 16) * __uint_as_float(b << 16) + __uint_as_float(a & 0xFFFF0000u) * __uint_as_float(b & 0xFFFF0000u);
}

// dropout by keep BITS: value * (bit `pos` of w ? 1 : 0) as one v_bfe_i32 (0 / all ones) and one v_and
__device__ __forceinline__ float keep_bit(float v, uint32_t w, int pos) {
    return __uint_as_float(__float_as_uint(v) & (uint32_t)__builtin_amdgcn_sbfe((int)w, pos, 1));
}

// Output rows of the 32 x 32 MFMA layout: lanes r and r + 32 hold ADJACENT 8-byte pieces (4 head dimensions each) of the same row, so the
// natural store is 8 bytes per lane - twice the store instructions, and a row-per-lane store is issue-bound per instruction
// (tools/micro/store_rate.hip; cdna_hip_programming.md T21).  One v_permlane32_swap per dword trades the upper half-wave's piece of
// column group u against the lower half-wave's piece of group u + 1: afterwards a lane of the lower half holds the 16 contiguous bytes of
// group u, a lane of the upper half those of group u + 1 -> one 16-byte store at column 8 (u + h).  Executed by all lanes (outside `if (row < L)`).
__device__ __forceinline__ uint4 widen_pair(const uint2& g0, const uint2& g1) {
    typedef unsigned attn_u32x2 __attribute__((ext_vector_type(2)));
    const attn_u32x2 rx = __builtin_amdgcn_permlane32_swap(g0.x, g1.x, false, false);
    const attn_u32x2 ry = __builtin_amdgcn_permlane32_swap(g0.y, g1.y, false, false);
    return make_uint4(rx[0], ry[0], rx[1], ry[1]);
}

// Packed ("varlen") batches: with `cu` given, the rows of sequence `seq` are rows cu[seq] .. cu[seq + 1] of qkv / ctx / dctx / dqkv (pack.hip's
// layout: Tp = sum of the lengths rows, no padding rows anywhere) instead of rows seq * L .. seq * L + L of the padded layout; keys and queries
// >= the sequence's length arrive in LDS as zero rows and are masked, exactly what the padded layout holds after cldrd_unpack_rows16 - so the two
// layouts give the same bits - and are neither read nor written.  LSE, keep bits and probabilities stay [nseq, H, L(, ..)] (small).
// Blocks of 32 keys / queries that lie entirely beyond a sequence's length are SKIPPED (nb = ceil(len / 32) live blocks): a masked key's
// probability is exactly 0 and a zero query row's output is never stored, so every skipped MFMA would have added +0.0f - same bits, less work
// (a packed MS MARCO batch at max_length 256 holds 37 % real tokens: 3 of 8 blocks per side).  The LSE of a skipped (all-zero) query row has the
// closed form log(len): scores 0, maximum 0, denominator = the number of unmasked keys, exactly what the arithmetic gives.
// A block loop that can be LEFT EARLY ends in a branch right behind an MFMA, and the code at the branch target reads that MFMA's accumulator.
// hipcc (ROCm 7.2) counts the wait states between an XDL write and a VALU read of its result along the fall-through path only: in
// attn_fwd_full_kernel<4> with two live blocks the exit edge reached `v_accvgpr_read a63 / a62` four cycles behind the MFMA that writes
// a[48:63] - stale scores for keys 58, 59, 62, 63 of every sequence of 59 .. 64 tokens (tools/wgrad_attn_fuzz.py found it; the keys are masked
// in shorter sequences, which is why nothing else noticed).  mfma_drain() ends every block iteration of such a loop, in front of the next
// iteration's exit test: 24 wait states (more than the longest XDL result latency) on BOTH ways out of the branch, pinned against the schedulers;
// they pass under the last MFMA of the block, which occupies the pipe for longer than that.
__device__ __forceinline__ void mfma_drain() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

#if defined(CLDRD_DEV_BUILD) && defined(CLDRD_NO_BLOCK_SKIP)
#define CLDRD_LIVE_BLOCKS(len, NKB) (NKB)                 // development build: compute every block (bisecting)
#else
#define CLDRD_LIVE_BLOCKS(len, NKB) (((len) + 31) >> 5)
#endif
// A launch may cover a LIST of sequences (seq_list, device int32; null: all of them in order): item i / H is then the list position and
// seq_list[i / H] the sequence - how a packed batch with L > 128 sends its sequences of at most 128 tokens (most of an MS MARCO batch) to the
// L <= 128 kernels (persistent, double-buffered) and only the long ones to the streaming / one-item kernels: `L` stays the stride of LSE and of
// the dropout row keys, the kernel's tile height comes from its NKB.
__device__ __forceinline__ int seq_of(const int* __restrict__ seq_list, int pos) { return seq_list ? seq_list[pos] : pos; }
struct SeqRows { int row0, len; };
__device__ __forceinline__ SeqRows seq_rows(const int* __restrict__ cu, int seq, int L) {
    if (cu) { const int c0 = cu[seq]; return {c0, cu[seq + 1] - c0}; }
    return {seq * L, L};
}

// copy a [L, 64] bf16 head slice (row stride ld elements) into an LDS tile of Lp rows, zero-filling rows >= L
__device__ __forceinline__ void load_tile(char* tile, const bf16_t* src, int ld, int L, int Lp) {
    for (int idx = threadIdx.x; idx < Lp * 8; idx += blockDim.x) {
        const int row = idx >> 3, ch = idx & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < L) v = *(const uint4*)(src + (size_t)row * ld + ch * 8);
        *(uint4*)(tile + row * RSB + ch * 16) = v;
    }
}

// DROP is a template parameter: as a run-time test hipcc branched on it per element (16 branches per MFMA tile, with the
// accumulators re-read from AGPRs on both sides), which made these kernels VALU-bound.
//
// Forward with all the scores of a query block in registers (16 NKB accumulators): for L <= 128 this is ~7 % faster than the
// streaming form below (16 score MFMAs back to back, one softmax pass, 16 P.V MFMAs) and is what the train step uses.
template <int NKB, bool DROP, bool F16 = false>
__global__ __launch_bounds__(256) void attn_fwd_full_kernel(const bf16_t* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                        bf16_t* __restrict__ ctx, float* __restrict__ lse, int L, int H,
                                                        float scale, uint32_t drop_thresh, float drop_scale, SeedArg seed_a,
                                                        bf16_t* __restrict__ ctx16, const int* __restrict__ cu, const int* __restrict__ seq_list) {
    const uint64_t seed = seed_a.get();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int Lp = 32 * NKB;
    char* sQ = smem;
    char* sK = sQ + Lp * RSB;
    char* sV = sK + Lp * RSB;
    float* sBias = (float*)(sV + Lp * RSB);
    const int seq = seq_of(seq_list, blockIdx.x / H), hd = blockIdx.x % H;
    const int dm = H * 64, ld = 3 * dm;
    const SeqRows sr = seq_rows(cu, seq, L);
    const int len = sr.len;
    const bf16_t* base = qkv + (size_t)sr.row0 * ld + hd * 64;
    load_tile(sQ, base, ld, len, Lp);
    load_tile(sK, base + dm, ld, len, Lp);
    load_tile(sV, base + 2 * dm, ld, len, Lp);
    for (int k = threadIdx.x; k < Lp; k += blockDim.x)
        sBias[k] = (k < len && (!mask || mask[(size_t)seq * L + k] != 0)) ? 0.f : NEG_BIG;
    __syncthreads();

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nb = CLDRD_LIVE_BLOCKS(len, NKB);          // live blocks (see seq_rows)
    for (int qb = wid; qb < NKB; qb += 4) {
        if (qb >= nb) {       // a query block beyond the sequence: nothing to store but the LSE of its zero rows
            const int q = qb * 32 + r;
            if (h == 0 && q < L && lse) lse[((size_t)seq * H + hd) * L + q] = (0.f + __log2f((float)len)) * LN2;
            continue;
        }
        bf16x8 qf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = row_frag(sQ, qb * 32 + r, s, h);
        f32x16 S[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb >= nb) break;
            S[kb] = (f32x16){0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s)
                S[kb] = mfma32<F16>(row_frag(sK, kb * 32 + r, s, h), qf[s], S[kb]);
            mfma_drain();
        }
        // S[kb][t] = <K[key], Q[q]> with key = 32 kb + rowmap(t, h), q = 32 qb + r
        // softmax in the log2 domain (v_exp_f32 is 2^x): scores * scale * log2(e) + bias, bias read 4 keys at a time
        // (keys rowmap(4u .. 4u+3, h) = 8u + 4h + 0..3 are consecutive)
        const float scale2 = scale * LOG2E;
        float mx = NEG_BIG * 4.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (kb >= nb) break;
                const float4 b4 = *(const float4*)(sBias + kb * 32 + 8 * u + 4 * h);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int j = 0; j < 4; j += 2) {      // two keys per packed fp32 instruction (v_pk_fma_f32: same IEEE results as two v_fma_f32)
                    const cldrd_f32v2 v = fma2((cldrd_f32v2){S[kb][4 * u + j], S[kb][4 * u + j + 1]}, splat2(scale2), (cldrd_f32v2){bb[j], bb[j + 1]});
                    S[kb][4 * u + j] = v.x; S[kb][4 * u + j + 1] = v.y;
                    mx = fmaxf(mx, fmaxf(v.x, v.y));
                }
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                if (kb >= nb) break;
                const cldrd_f32v2 d = (cldrd_f32v2){S[kb][t], S[kb][t + 1]} - splat2(mx);
                const float e0 = __builtin_amdgcn_exp2f(d.x), e1 = __builtin_amdgcn_exp2f(d.y);
                S[kb][t] = e0; S[kb][t + 1] = e1;
                sum += e0;
                sum += e1;
            }
        sum += __shfl_xor(sum, 32, 64);
        const int q = qb * 32 + r;
        if (h == 0 && q < L && lse) lse[((size_t)seq * H + hd) * L + q] = (mx + __log2f(sum)) * LN2;
        const float inv = 1.0f / sum;
        // dropout mask element (row, col) = ((seq*H + hd)*L + q, key); keys rowmap(t, h), t even / odd, are a column pair
        const uint32_t rk = drop_rowkey(seed, (uint32_t)((seq * H + hd) * L + q));
        f32x16 O[2] = {(f32x16){0.f}, (f32x16){0.f}};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb >= nb) break;
            float pv[16];
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                cldrd_f32v2 p = (cldrd_f32v2){S[kb][t], S[kb][t + 1]} * splat2(inv);
                if (DROP) {
                    const uint32_t hh = drop_pair(rk, (uint32_t)(kb * 32 + rowmap(t, h)));
                    p = p * splat2(drop_scale);
                    p.x = drop_keep_lo(hh, drop_thresh) ? p.x : 0.f;
                    p.y = drop_keep_hi(hh, drop_thresh) ? p.y : 0.f;
                }
                pv[t] = p.x; pv[t + 1] = p.y;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pa = pack8x<F16>(pv + 8 * s2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)      // operands swapped: O^T[d][q], so that a lane (= query) holds consecutive head dimensions
                    O[dt] = mfma32<F16>(tr_frag(sV, kb * 32 + 16 * s2 + 4 * h, dt * 32, lane), pa, O[dt]);
            }
            mfma_drain();
        }
        // O[dt][t] = ctx^T[d = 32 dt + rowmap(t, h)][q = 32 qb + r]: registers 4u .. 4u+3 are 4 consecutive head dimensions of the lane's query row;
        // one v_permlane32_swap pair makes them 16 contiguous bytes (widen_pair): 4 (+ 4 for the fp16 copy) 16-byte stores per lane instead of
        // 32 (+ 32) two-byte ones.  (Round 2 tried the transposed form with 8-byte stores and found it 8 % slower; round 5, with the widened
        // stores: profiles/r05_microbench.txt section 6.)
        {
            const size_t orow = ((size_t)sr.row0 + q) * dm + hd * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                    uint2 o[2], o16[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int b = 4 * (u + k);
                        o[k].x = pack2x<F16>(O[dt][b], O[dt][b + 1]);
                        o[k].y = pack2x<F16>(O[dt][b + 2], O[dt][b + 3]);
                        o16[k].x = (uint32_t)f2x<true>(O[dt][b]) | ((uint32_t)f2x<true>(O[dt][b + 1]) << 16);
                        o16[k].y = (uint32_t)f2x<true>(O[dt][b + 2]) | ((uint32_t)f2x<true>(O[dt][b + 3]) << 16);
                    }
                    if (ctx) {       // ctx: the kernel's own format (may be absent); ctx16: an fp16 copy of a bf16 pass (out-projection operand)
                        const uint4 w = widen_pair(o[0], o[1]);
                        if (q < len) *(uint4*)(ctx + orow + dt * 32 + 8 * (u + h)) = w;
                    }
                    if (ctx16) {
                        const uint4 w = widen_pair(o16[0], o16[1]);
                        if (q < len) *(uint4*)(ctx16 + orow + dt * 32 + 8 * (u + h)) = w;
                    }
                }
        }
    }
}


// Forward for L <= 128 and many (sequence, head) items: one persistent 8-wave workgroup per CU.  Waves 0..3 compute (the body of
// attn_fwd_full_kernel, one query block each), waves 4..7 only move data: they issue the global loads of the NEXT item's Q, K, V into
// registers, wait, and write them to the second LDS buffer - so an item's HBM latency sits under the previous item's arithmetic instead
// of in front of its own (the kernel above does load -> barrier -> compute per workgroup and overlaps only across the two workgroups
// a CU can hold).  One __syncthreads per item.
template <int NKB, bool DROP, bool F16 = false>
__global__ __launch_bounds__(512) void attn_fwd2_kernel(const bf16_t* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                         bf16_t* __restrict__ ctx, float* __restrict__ lse, int L, int H,
                                                         float scale, uint32_t drop_thresh, float drop_scale, SeedArg seed_a, int nitems,
                                                         uint32_t* __restrict__ bits_out, bf16_t* __restrict__ ctx16, const int* __restrict__ cu, const int* __restrict__ seq_list) {
    const uint64_t seed = seed_a.get();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int Lp = 32 * NKB;
    constexpr int TILE = Lp * RSB;
    constexpr int NBITS = DROP ? Lp * NKB : 0;          // dropout keep bits of an item: dword [kb][q] = keys 32 kb .. 32 kb + 31 of query q
    constexpr int BUF = 3 * TILE + (Lp + NBITS) * (int)sizeof(float);
    constexpr int NCH = NKB;                            // 16-byte chunks per loader thread and tile: Lp * 8 / 256
    const int tid = threadIdx.x, wid = tid >> 6;
    const bool loader = wid >= 4;
    const int rw = wid & 3;
    const int dm = H * 64, ld = 3 * dm;
    auto opaque = [](int x) { asm volatile("" : "+v"(x)); return x; };      // see attn_bwd2_kernel
    int tb = tid & 255;
    uint4 pq[NCH], pk[NCH], pv[NCH];
    float p_bias = 0.f;
    auto issue = [&](int item) {
        const int seq = seq_of(seq_list, item / H), hd = item % H;
        const SeqRows nr = seq_rows(cu, seq, L);
        const bf16_t* base = qkv + (size_t)nr.row0 * ld + hd * 64;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int idx = tb + 256 * j, row = idx >> 3, ch = idx & 7;
            pq[j] = pk[j] = pv[j] = make_uint4(0, 0, 0, 0);
            if (row < nr.len) {
                const bf16_t* rp = base + (size_t)row * ld + ch * 8;
                pq[j] = *(const uint4*)rp;
                pk[j] = *(const uint4*)(rp + dm);
                pv[j] = *(const uint4*)(rp + 2 * dm);
            }
        }
        if (tb < Lp) p_bias = (tb < nr.len && (!mask || mask[(size_t)seq * L + tb] != 0)) ? 0.f : NEG_BIG;
    };
    auto commit = [&](int b) {
        char* base = smem + b * BUF;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int idx = tb + 256 * j, row = idx >> 3, ch = idx & 7;
            const int off = row * RSB + ch * 16;
            *(uint4*)(base + off) = pq[j];
            *(uint4*)(base + TILE + off) = pk[j];
            *(uint4*)(base + 2 * TILE + off) = pv[j];
        }
        if (tb < Lp) ((float*)(base + 3 * TILE))[tb] = p_bias;
    };
    // The loader waves have time to spare: they also evaluate the dropout hash of the next item (one mix32 per key PAIR, common.h) and
    // leave the keep decisions as bits in LDS - and in `bits_out` for the backward, which then hashes nothing.  With the hash in the
    // compute waves (one per SIMD here) dropout cost 23 us of 70 per layer.
    auto make_bits = [&](int b, int item) {
        if constexpr (DROP) {
            const int seq = seq_of(seq_list, item / H), hd = item % H;
            uint32_t* sb = (uint32_t*)(smem + b * BUF + 3 * TILE) + Lp;
            for (int idx = tb; idx < NBITS; idx += 256) {
                const int q = idx % Lp, kb = idx / Lp;
                const uint32_t rk = drop_rowkey(seed, (uint32_t)((seq * H + hd) * L + q));
                uint32_t w = 0;
#pragma unroll
                for (int pp = 0; pp < 16; ++pp) {
                    const uint32_t hh = drop_pair(rk, (uint32_t)(kb * 32 + 2 * pp));
                    w |= (drop_keep_lo(hh, drop_thresh) ? 1u : 0u) << (2 * pp);
                    w |= (drop_keep_hi(hh, drop_thresh) ? 1u : 0u) << (2 * pp + 1);
                }
                sb[idx] = w;
                if (bits_out) bits_out[(size_t)item * NBITS + idx] = w;
            }
        }
    };
    int item = blockIdx.x;
    if (loader && item < nitems) { issue(item); commit(0); make_bits(0, item); }
    __syncthreads();
    int cur = 0;
    for (; item < nitems; item += gridDim.x) {
        const int next = item + gridDim.x;
        const char* sQ = smem + cur * BUF;
        const char* sK = sQ + TILE;
        const char* sV = sK + TILE;
        const float* sBias = (const float*)(sV + TILE);
        const uint32_t* sBits = (const uint32_t*)(sBias + Lp);
        const int seq = seq_of(seq_list, item / H), hd = item % H;
        const SeqRows sr = seq_rows(cu, seq, L);
        const int len = sr.len, nb = CLDRD_LIVE_BLOCKS(len, NKB);
        const int t_ = opaque(tid);
        const int lane = t_ & 63, r = lane & 31, h = lane >> 5;
        tb = t_ & 255;
        if (loader) {
            if (next < nitems) { issue(next); commit(cur ^ 1); make_bits(cur ^ 1, next); }
        } else if (rw < NKB && rw >= nb) {      // a query block beyond the sequence: nothing to store but the LSE of its zero rows
            const int q = rw * 32 + r;
            if (h == 0 && q < L && lse) lse[((size_t)seq * H + hd) * L + q] = (0.f + __log2f((float)len)) * LN2;
        } else if (rw < nb) {
            const int qb = rw;
        bf16x8 qf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = row_frag(sQ, qb * 32 + r, s, h);
        f32x16 S[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb >= nb) break;
            S[kb] = (f32x16){0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s)
                S[kb] = mfma32<F16>(row_frag(sK, kb * 32 + r, s, h), qf[s], S[kb]);
            mfma_drain();
        }
        // S[kb][t] = <K[key], Q[q]> with key = 32 kb + rowmap(t, h), q = 32 qb + r
        // softmax in the log2 domain (v_exp_f32 is 2^x): scores * scale * log2(e) + bias, bias read 4 keys at a time
        // (keys rowmap(4u .. 4u+3, h) = 8u + 4h + 0..3 are consecutive)
        const float scale2 = scale * LOG2E;
        float mx = NEG_BIG * 4.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (kb >= nb) break;
                const float4 b4 = *(const float4*)(sBias + kb * 32 + 8 * u + 4 * h);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int j = 0; j < 4; j += 2) {      // two keys per packed fp32 instruction (v_pk_fma_f32: same IEEE results as two v_fma_f32)
                    const cldrd_f32v2 v = fma2((cldrd_f32v2){S[kb][4 * u + j], S[kb][4 * u + j + 1]}, splat2(scale2), (cldrd_f32v2){bb[j], bb[j + 1]});
                    S[kb][4 * u + j] = v.x; S[kb][4 * u + j + 1] = v.y;
                    mx = fmaxf(mx, fmaxf(v.x, v.y));
                }
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                if (kb >= nb) break;
                const cldrd_f32v2 d = (cldrd_f32v2){S[kb][t], S[kb][t + 1]} - splat2(mx);
                const float e0 = __builtin_amdgcn_exp2f(d.x), e1 = __builtin_amdgcn_exp2f(d.y);
                S[kb][t] = e0; S[kb][t + 1] = e1;
                sum += e0;
                sum += e1;
            }
        sum += __shfl_xor(sum, 32, 64);
        const int q = qb * 32 + r;
        if (h == 0 && q < L && lse) lse[((size_t)seq * H + hd) * L + q] = (mx + __log2f(sum)) * LN2;
        const float inv = 1.0f / sum;
        // dropout mask element (row, col) = ((seq*H + hd)*L + q, key); keys rowmap(t, h), t even / odd, are a column pair
        const uint32_t rk = drop_rowkey(seed, (uint32_t)((seq * H + hd) * L + q));
        f32x16 O[2] = {(f32x16){0.f}, (f32x16){0.f}};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb >= nb) break;
            float pv[16];
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                cldrd_f32v2 p = (cldrd_f32v2){S[kb][t], S[kb][t + 1]} * splat2(inv);
                if (DROP) {       // keys rowmap(t, h), rowmap(t, h) + 1 of block kb: two adjacent bits of the loader's dword
                    const uint32_t w = sBits[kb * Lp + q] >> (4 * h);
                    p = p * splat2(drop_scale);
                    p.x = keep_bit(p.x, w, rowmap(t, 0));
                    p.y = keep_bit(p.y, w, rowmap(t, 0) + 1);
                }
                pv[t] = p.x; pv[t + 1] = p.y;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pa = pack8x<F16>(pv + 8 * s2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)      // operands swapped: O^T[d][q], so that a lane (= query) holds consecutive head dimensions
                    O[dt] = mfma32<F16>(tr_frag(sV, kb * 32 + 16 * s2 + 4 * h, dt * 32, lane), pa, O[dt]);
            }
            mfma_drain();
        }
        // O[dt][t] = ctx^T[d = 32 dt + rowmap(t, h)][q = 32 qb + r]: registers 4u .. 4u+3 are 4 consecutive head dimensions of the lane's query row;
        // one v_permlane32_swap pair makes them 16 contiguous bytes (widen_pair): 4 (+ 4 for the fp16 copy) 16-byte stores per lane instead of
        // 32 (+ 32) two-byte ones.  (Round 2 tried the transposed form with 8-byte stores and found it 8 % slower; round 5, with the widened
        // stores: profiles/r05_microbench.txt section 6.)
        {
            const size_t orow = ((size_t)sr.row0 + q) * dm + hd * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                    uint2 o[2], o16[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int b = 4 * (u + k);
                        o[k].x = pack2x<F16>(O[dt][b], O[dt][b + 1]);
                        o[k].y = pack2x<F16>(O[dt][b + 2], O[dt][b + 3]);
                        o16[k].x = (uint32_t)f2x<true>(O[dt][b]) | ((uint32_t)f2x<true>(O[dt][b + 1]) << 16);
                        o16[k].y = (uint32_t)f2x<true>(O[dt][b + 2]) | ((uint32_t)f2x<true>(O[dt][b + 3]) << 16);
                    }
                    if (ctx) {       // ctx: the kernel's own format (may be absent); ctx16: an fp16 copy of a bf16 pass (out-projection operand)
                        const uint4 w = widen_pair(o[0], o[1]);
                        if (q < len) *(uint4*)(ctx + orow + dt * 32 + 8 * (u + h)) = w;
                    }
                    if (ctx16) {
                        const uint4 w = widen_pair(o16[0], o16[1]);
                        if (q < len) *(uint4*)(ctx16 + orow + dt * 32 + 8 * (u + h)) = w;
                    }
                }
        }
            }
        __syncthreads();          // the other buffer is complete, and nobody reads this one any more
        cur ^= 1;
    }
}

// Forward, streaming over the key blocks with a running maximum (one pass, "flash" form): per 32-key block
//   S^T = K_kb . Q^T (lane = query column)  ->  m' = max(m, colmax)  ->  O^T *= 2^(m - m'),  l = l 2^(m - m') + sum p,
//   p = 2^(s - m')  ->  O^T += V_kb^T . (keep . p)         (the S^T tile is the B operand as it is; V through ds_read_tr)
// and at the end ctx = O^T / l (x 1/(1-p_drop)), LSE = m + log2 l.  Everything per query is a per-lane scalar because the
// output is accumulated TRANSPOSED (lane = query column, registers = head dimension): no cross-lane traffic but one
// shuffle per block, and 16 + 32 accumulator registers whatever L is (the first version kept all of S in registers:
// 16 NKB of them, which spilled from L = 192 up and ran 2x slower at L = 256).
template <int NKB, bool DROP, bool F16 = false>
__global__ __launch_bounds__(NKB > 4 ? 512 : 256) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                        bf16_t* __restrict__ ctx, float* __restrict__ lse, int L, int H,
                                                        float scale, uint32_t drop_thresh, float drop_scale, SeedArg seed_a,
                                                        bf16_t* __restrict__ ctx16, const int* __restrict__ cu, const int* __restrict__ seq_list) {
    const uint64_t seed = seed_a.get();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int Lp = 32 * NKB;
    constexpr int NWAVES = NKB > 4 ? 8 : 4;          // one query / key block per wave up to L = 256 (2 waves per SIMD at one workgroup per CU)
    char* sQ = smem;
    char* sK = sQ + Lp * RSB;
    char* sV = sK + Lp * RSB;
    float* sBias = (float*)(sV + Lp * RSB);
    const int seq = seq_of(seq_list, blockIdx.x / H), hd = blockIdx.x % H;
    const int dm = H * 64, ld = 3 * dm;
    const SeqRows sr = seq_rows(cu, seq, L);
    const int len = sr.len;
    const bf16_t* base = qkv + (size_t)sr.row0 * ld + hd * 64;
    load_tile(sQ, base, ld, len, Lp);
    load_tile(sK, base + dm, ld, len, Lp);
    load_tile(sV, base + 2 * dm, ld, len, Lp);
    for (int k = threadIdx.x; k < Lp; k += blockDim.x)
        sBias[k] = (k < len && (!mask || mask[(size_t)seq * L + k] != 0)) ? 0.f : NEG_BIG;
    __syncthreads();

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const float scale2 = scale * LOG2E;      // scores in the log2 domain (v_exp_f32 is 2^x)
    const int nb = CLDRD_LIVE_BLOCKS(len, NKB);          // live blocks (see seq_rows)
    for (int qb = wid; qb < NKB; qb += NWAVES) {
        const int q = qb * 32 + r;
        if (qb >= nb) {       // a query block beyond the sequence: nothing to store but the LSE of its zero rows
            if (h == 0 && q < L && lse) lse[((size_t)seq * H + hd) * L + q] = (0.f + __log2f((float)len)) * LN2;
            continue;
        }
        bf16x8 qf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = row_frag(sQ, qb * 32 + r, s, h);
        // dropout mask element (row, col) = ((seq*H + hd)*L + q, key); keys rowmap(t, h), t even / odd, are a column pair
        const uint32_t rk = drop_rowkey(seed, (uint32_t)((seq * H + hd) * L + q));
        float m = NEG_BIG * 4.f, lsum = 0.f;                       // lsum: this half's share of the denominator
        f32x16 O[2] = {(f32x16){0.f}, (f32x16){0.f}};              // O[dt][t] = ctx^T[d = 32 dt + rowmap(t, h)][q]
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb >= nb) break;
            f32x16 S = (f32x16){0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s)
                S = mfma32<F16>(row_frag(sK, kb * 32 + r, s, h), qf[s], S);
            // S[t] = <K[key], Q[q]> with key = 32 kb + rowmap(t, h); keys rowmap(4u .. 4u+3, h) = 8u + 4h + 0..3 are consecutive
            float v[16];
            float mloc = NEG_BIG * 4.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 b4 = *(const float4*)(sBias + kb * 32 + 8 * u + 4 * h);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int j = 0; j < 4; j += 2) {      // packed fp32: two keys per instruction
                    const cldrd_f32v2 w = fma2((cldrd_f32v2){S[4 * u + j], S[4 * u + j + 1]}, splat2(scale2), (cldrd_f32v2){bb[j], bb[j + 1]});
                    v[4 * u + j] = w.x; v[4 * u + j + 1] = w.y;
                    mloc = fmaxf(mloc, fmaxf(w.x, w.y));
                }
            }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float mn = fmaxf(m, mloc);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            m = mn;
            float psum = 0.f;
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                const cldrd_f32v2 dv = (cldrd_f32v2){v[t], v[t + 1]} - splat2(mn);
                float p0 = __builtin_amdgcn_exp2f(dv.x), p1 = __builtin_amdgcn_exp2f(dv.y);
                psum += p0 + p1;
                if (DROP) {
                    const uint32_t hh = drop_pair(rk, (uint32_t)(kb * 32 + rowmap(t, h)));
                    p0 = drop_keep_lo(hh, drop_thresh) ? p0 : 0.f;
                    p1 = drop_keep_hi(hh, drop_thresh) ? p1 : 0.f;
                }
                v[t] = p0; v[t + 1] = p1;
            }
            lsum = fmaf(lsum, alpha, psum);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int t = 0; t < 16; t += 2) {
                    const cldrd_f32v2 o2 = (cldrd_f32v2){O[dt][t], O[dt][t + 1]} * splat2(alpha);
                    O[dt][t] = o2.x; O[dt][t + 1] = o2.y;
                }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pb = pack8x<F16>(v + 8 * s2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    O[dt] = mfma32<F16>(tr_frag(sV, kb * 32 + 16 * s2 + 4 * h, dt * 32, lane), pb, O[dt]);
            }
            mfma_drain();
        }
        const float l = lsum + __shfl_xor(lsum, 32, 64);
        if (h == 0 && q < L && lse) lse[((size_t)seq * H + hd) * L + q] = (m + __log2f(l)) * LN2;
        const float inv = (DROP ? drop_scale : 1.0f) / l;
        {
            const size_t orow = ((size_t)sr.row0 + q) * dm + hd * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                    uint2 o[2], o16[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int b = 4 * (u + k);
                        const float v0 = O[dt][b] * inv, v1 = O[dt][b + 1] * inv, v2 = O[dt][b + 2] * inv, v3 = O[dt][b + 3] * inv;
                        o[k].x = pack2x<F16>(v0, v1);
                        o[k].y = pack2x<F16>(v2, v3);
                        o16[k].x = (uint32_t)f2x<true>(v0) | ((uint32_t)f2x<true>(v1) << 16);
                        o16[k].y = (uint32_t)f2x<true>(v2) | ((uint32_t)f2x<true>(v3) << 16);
                    }
                    if (ctx) {
                        const uint4 w = widen_pair(o[0], o[1]);
                        if (q < len) *(uint4*)(ctx + orow + dt * 32 + 8 * (u + h)) = w;
                    }
                    if (ctx16) {       // fp16 copy (out-projection operand)
                        const uint4 w = widen_pair(o16[0], o16[1]);
                        if (q < len) *(uint4*)(ctx16 + orow + dt * 32 + 8 * (u + h)) = w;
                    }
                }
        }
    }
}

// Streaming forward for 128 < L <= 256 and many (sequence, head) items (round 5: cfg5's index encode runs at max_length 256, reference
// retriever/index_text.py:37; cfg4 trains BERT-base at L = 256): ONE persistent 8-wave workgroup per CU walks its items.  The one-item form
// above does load -> barrier -> compute per workgroup, and at L = 256 an item's Q, K, V tiles (110 KB of LDS) leave room for one workgroup
// per CU: nothing overlaps the 147 KB of HBM reads of an item with the arithmetic of another, and every CU loads at the same moment.  Here
//   * Q never goes through LDS: a wave's four Q fragments are 64 bytes per lane straight from global memory (row_frag's layout);
//   * K, V (+ the key bias) are double-buffered in LDS (2 x 74 KB): each thread requests its 16-byte pieces of the NEXT item into registers
//     before it starts on the current item (issue early), and writes them to the other buffer when it is done (write late): an item's HBM
//     latency sits under the previous item's arithmetic; one __syncthreads per item;
//   * the body is attn_fwd_kernel's, instruction for instruction (same results bit for bit: tests flip cldrd_set_tuning("attn_fwd2", 0)).
template <int NKB, bool DROP, bool F16 = false>
__global__ __launch_bounds__(512) void attn_fwd3_kernel(const bf16_t* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                         bf16_t* __restrict__ ctx, float* __restrict__ lse, int L, int H,
                                                         float scale, uint32_t drop_thresh, float drop_scale, SeedArg seed_a, int nitems,
                                                         bf16_t* __restrict__ ctx16, const int* __restrict__ cu, const int* __restrict__ seq_list) {
    const uint64_t seed = seed_a.get();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int Lp = 32 * NKB;
    constexpr int TILE = Lp * RSB;
    constexpr int BUF = 2 * TILE + Lp * (int)sizeof(float);
    constexpr int NCH = (Lp * 8 + 511) / 512;           // 16-byte pieces per thread and tile
    const int dm = H * 64, ld = 3 * dm;
    auto opaque = [](int x) { asm volatile("" : "+v"(x)); return x; };      // see attn_bwd2_kernel: keeps 40 address halves from being hoisted
    int tid = threadIdx.x;
    u32x4 pk[NCH], pv[NCH];          // (first-class vector values: as `uint4` structs hipcc kept the two arrays in scratch memory)
    bf16x8 qn[4];
    long long pm = 0;
    int p_len = 0;                   // row count of the prefetched item (L, or its length in the packed layout)
    // Every load of issue() is UNCONDITIONAL on a clamped row (no exec-masked region, no branch): the compiler can then count its vmcnt waits -
    // the pieces are consumed at commit(), behind this item's ctx stores, and must not wait for those (vmcnt counts stores too).  A row >= L
    // therefore arrives as a copy of row L - 1 instead of zeros: a K row that the key bias masks (score + -1e30 is -1e30 whatever the score),
    // a V row that meets p = 0, a Q row whose output nobody stores - all finite, results unchanged bit for bit.
    const long long* mask_or_any = mask ? (const long long*)mask : (const long long*)qkv;      // null mask: the value loaded is ignored
    auto issue = [&](int item) {
        const int seq = seq_of(seq_list, item / H), hd = item % H;
        const SeqRows nr = seq_rows(cu, seq, L);
        p_len = nr.len;
        const bf16_t* base = qkv + (size_t)nr.row0 * ld + hd * 64;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int idx = tid + 512 * j, row = min(idx >> 3, nr.len - 1), ch = idx & 7;
            const bf16_t* rp = base + (size_t)row * ld + ch * 8;
            pk[j] = *(const u32x4*)(rp + dm);
            pv[j] = *(const u32x4*)(rp + 2 * dm);
        }
        pm = mask_or_any[mask ? (size_t)seq * L + min(tid, L - 1) : (size_t)0];
        // this wave's Q block of the item: row 32 qb + r, 16-byte pieces 2 s + h (row_frag's layout)
        const int lane = tid & 63, r = lane & 31, h = lane >> 5, qb = min(tid >> 6, NKB - 1);
        const int qrow = min(qb * 32 + r, nr.len - 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) qn[s] = *(const bf16x8*)(base + (size_t)qrow * ld + (2 * s + h) * 8);
    };
    auto commit = [&](int b) {
        char* base = smem + b * BUF;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int idx = tid + 512 * j, row = idx >> 3, ch = idx & 7;
            if (NCH * 512 == Lp * 8 || idx < Lp * 8) {
                const int off = row * RSB + ch * 16;
                *(u32x4*)(base + off) = pk[j];
                *(u32x4*)(base + TILE + off) = pv[j];
            }
        }
        if (tid < Lp) ((float*)(base + 2 * TILE))[tid] = (tid < p_len && (!mask || pm != 0)) ? 0.f : NEG_BIG;
    };
    int item = blockIdx.x;
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    if (item < nitems) {
        issue(item);
        commit(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = qn[s];
    }
    __syncthreads();
    int cur = 0;
    const float scale2 = scale * LOG2E;      // scores in the log2 domain (v_exp_f32 is 2^x)
    for (; item < nitems; item += gridDim.x) {
        const int next = item + gridDim.x;
        tid = opaque((int)threadIdx.x);
        // in flight while this item is computed.  UNCONDITIONAL (the last item of a workgroup requests itself again: cache hits, and the
        // commit below lands in the buffer nobody reads any more): with the prefetch registers live across a branch hipcc kept them in scratch
        // memory - a scratch store and an s_waitcnt vmcnt(0) behind every load
        issue(next < nitems ? next : item);
        const char* sK = smem + cur * BUF;
        const char* sV = sK + TILE;
        const float* sBias = (const float*)(sV + TILE);
        const int seq = seq_of(seq_list, item / H), hd = item % H;
        const SeqRows sr = seq_rows(cu, seq, L);
        const int len = sr.len, nb = CLDRD_LIVE_BLOCKS(len, NKB);
        const int lane = tid & 63, wid = tid >> 6;
        const int r = lane & 31, h = lane >> 5;
        if (wid < NKB && wid >= nb) {          // a query block beyond the sequence: nothing to store but the LSE of its zero rows (see seq_rows)
            const int q = wid * 32 + r;
            if (h == 0 && q < L && lse) lse[((size_t)seq * H + hd) * L + q] = (0.f + __log2f((float)len)) * LN2;
        } else if (wid < nb) {
            const int qb = wid;
            const int q = qb * 32 + r;
            // dropout mask element (row, col) = ((seq*H + hd)*L + q, key); keys rowmap(t, h), t even / odd, are a column pair
            const uint32_t rk = drop_rowkey(seed, (uint32_t)((seq * H + hd) * L + q));
            float m = NEG_BIG * 4.f, lsum = 0.f;                       // lsum: this half's share of the denominator
            f32x16 O[2] = {(f32x16){0.f}, (f32x16){0.f}};              // O[dt][t] = ctx^T[d = 32 dt + rowmap(t, h)][q]
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                if (kb >= nb) break;
                f32x16 S = (f32x16){0.f};
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    S = mfma32<F16>(row_frag(sK, kb * 32 + r, s, h), qf[s], S);
                float v[16];
                float mloc = NEG_BIG * 4.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 b4 = *(const float4*)(sBias + kb * 32 + 8 * u + 4 * h);
                    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {      // packed fp32: two keys per instruction
                        const cldrd_f32v2 w = fma2((cldrd_f32v2){S[4 * u + j], S[4 * u + j + 1]}, splat2(scale2), (cldrd_f32v2){bb[j], bb[j + 1]});
                        v[4 * u + j] = w.x; v[4 * u + j + 1] = w.y;
                        mloc = fmaxf(mloc, fmaxf(w.x, w.y));
                    }
                }
                mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
                const float mn = fmaxf(m, mloc);
                const float alpha = __builtin_amdgcn_exp2f(m - mn);
                m = mn;
                float psum = 0.f;
#pragma unroll
                for (int t = 0; t < 16; t += 2) {
                    const cldrd_f32v2 dv = (cldrd_f32v2){v[t], v[t + 1]} - splat2(mn);
                    float p0 = __builtin_amdgcn_exp2f(dv.x), p1 = __builtin_amdgcn_exp2f(dv.y);
                    psum += p0 + p1;
                    if (DROP) {
                        const uint32_t hh = drop_pair(rk, (uint32_t)(kb * 32 + rowmap(t, h)));
                        p0 = drop_keep_lo(hh, drop_thresh) ? p0 : 0.f;
                        p1 = drop_keep_hi(hh, drop_thresh) ? p1 : 0.f;
                    }
                    v[t] = p0; v[t + 1] = p1;
                }
                lsum = fmaf(lsum, alpha, psum);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int t = 0; t < 16; t += 2) {
                        const cldrd_f32v2 o2 = (cldrd_f32v2){O[dt][t], O[dt][t + 1]} * splat2(alpha);
                        O[dt][t] = o2.x; O[dt][t + 1] = o2.y;
                    }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 pb = pack8x<F16>(v + 8 * s2);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        O[dt] = mfma32<F16>(tr_frag(sV, kb * 32 + 16 * s2 + 4 * h, dt * 32, lane), pb, O[dt]);
                }
                mfma_drain();
            }
            const float l = lsum + __shfl_xor(lsum, 32, 64);
            if (h == 0 && q < L && lse) lse[((size_t)seq * H + hd) * L + q] = (m + __log2f(l)) * LN2;
            const float inv = (DROP ? drop_scale : 1.0f) / l;
            const size_t orow = ((size_t)sr.row0 + q) * dm + hd * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                    uint2 o[2], o16[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int b = 4 * (u + k);
                        const float v0 = O[dt][b] * inv, v1 = O[dt][b + 1] * inv, v2 = O[dt][b + 2] * inv, v3 = O[dt][b + 3] * inv;
                        o[k].x = pack2x<F16>(v0, v1);
                        o[k].y = pack2x<F16>(v2, v3);
                        o16[k].x = (uint32_t)f2x<true>(v0) | ((uint32_t)f2x<true>(v1) << 16);
                        o16[k].y = (uint32_t)f2x<true>(v2) | ((uint32_t)f2x<true>(v3) << 16);
                    }
                    if (ctx) {
                        const uint4 w = widen_pair(o[0], o[1]);
                        if (q < len) *(uint4*)(ctx + orow + dt * 32 + 8 * (u + h)) = w;
                    }
                    if (ctx16) {       // fp16 copy (out-projection operand)
                        const uint4 w = widen_pair(o16[0], o16[1]);
                        if (q < len) *(uint4*)(ctx16 + orow + dt * 32 + 8 * (u + h)) = w;
                    }
                }
        }
        commit(cur ^ 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = qn[s];
        __syncthreads();          // the other buffer is complete, and nobody reads this one any more
        cur ^= 1;
    }
}

// VARLEN (a packed batch, launch_bwd_d): the two block loops are ROLLED and run over the live blocks only - a sequence of 74 tokens at L = 256 does
// 3 x 3 block pairs instead of 8 x 8 (measured on packed cfg4, BERT-base L = 256, 29 % real tokens: this kernel was 16 % of the step with whole waves
// skipped only).  Same per-block arithmetic in the same order: the same bits.  The padded instantiation keeps its unrolled loops.
template <int NKB, bool DROP, bool F16 = false, bool VARLEN = false>
__global__ __launch_bounds__(NKB > 4 ? 512 : 256) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                        const bf16_t* __restrict__ ctx, const bf16_t* __restrict__ dctx,
                                                        const float* __restrict__ lse, bf16_t* __restrict__ dqkv, int L, int H,
                                                        float scale, uint32_t drop_thresh, float drop_scale, SeedArg seed_a,
                                                        const int* __restrict__ cu, const int* __restrict__ seq_list) {
    const uint64_t seed = seed_a.get();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int Lp = 32 * NKB;
    constexpr int NWAVES = NKB > 4 ? 8 : 4;          // one query / key block per wave up to L = 256 (2 waves per SIMD at one workgroup per CU)
    char* sQ = smem;
    char* sK = sQ + Lp * RSB;
    char* sV = sK + Lp * RSB;
    char* sdO = sV + Lp * RSB;
    float* sBias = (float*)(sdO + Lp * RSB);
    float* sLse = sBias + Lp;
    float* sDelta = sLse + Lp;
    uint32_t* sRk = (uint32_t*)(sDelta + Lp);          // dropout row key of every query row (common.h)
    const int seq = seq_of(seq_list, blockIdx.x / H), hd = blockIdx.x % H;
    const int dm = H * 64, ld = 3 * dm;
    const SeqRows sr = seq_rows(cu, seq, L);
    const int len = sr.len;
    const bf16_t* base = qkv + (size_t)sr.row0 * ld + hd * 64;
    load_tile(sQ, base, ld, len, Lp);
    load_tile(sK, base + dm, ld, len, Lp);
    load_tile(sV, base + 2 * dm, ld, len, Lp);
    {   // dO tile + delta[q] = sum_d dO[q][d] * O[q][d]
        const bf16_t* dob = dctx + (size_t)sr.row0 * dm + hd * 64;
        const bf16_t* ob = ctx + (size_t)sr.row0 * dm + hd * 64;
        for (int idx = threadIdx.x; idx < Lp * 8; idx += blockDim.x) {
            const int row = idx >> 3, ch = idx & 7;
            uint4 v = make_uint4(0, 0, 0, 0), o = make_uint4(0, 0, 0, 0);
            if (row < len) {
                v = *(const uint4*)(dob + (size_t)row * dm + ch * 8);
                o = *(const uint4*)(ob + (size_t)row * dm + ch * 8);
            }
            *(uint4*)(sdO + row * RSB + ch * 16) = v;
            const uint32_t vv[4] = {v.x, v.y, v.z, v.w}, oo[4] = {o.x, o.y, o.z, o.w};
            float dsum = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) dsum += dot2x<F16>(vv[j], oo[j]);
            dsum += __shfl_xor(dsum, 1, 64); dsum += __shfl_xor(dsum, 2, 64); dsum += __shfl_xor(dsum, 4, 64);
            if (ch == 0) sDelta[row] = dsum;
        }
    }
    for (int k = threadIdx.x; k < Lp; k += blockDim.x) {
        sBias[k] = (k < len && (!mask || mask[(size_t)seq * L + k] != 0)) ? 0.f : NEG_BIG;
        sLse[k] = k < L ? lse[((size_t)seq * H + hd) * L + k] * LOG2E : 1.0e30f;      // log2 domain; rows >= L: P = 2^-inf = 0
        sRk[k] = drop_rowkey(seed, (uint32_t)((seq * H + hd) * L + k));
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;

    const float scale2 = scale * LOG2E;      // scores, mask bias and LSE live in the log2 domain (v_exp_f32 is 2^x)

    const int nb = CLDRD_LIVE_BLOCKS(len, NKB);          // live blocks (see seq_rows): whole waves are skipped; inside a live wave the padded
                                             // instantiation's block loops stay unrolled over all NKB blocks (an early exit from an unrolled
                                             // loop cost 40-90 VGPRs and spilled from NKB = 5 up), the VARLEN one's are rolled and stop at nb
    const int nbl = VARLEN ? nb : NKB;
    // ---------------- sweep A: key on lane; dK, dV for 32 keys accumulate over all query blocks ----------------
    for (int kb = wid; kb < NKB; kb += NWAVES) {
        if (kb >= nb) continue;
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) { kf[s] = row_frag(sK, kb * 32 + r, s, h); vf[s] = row_frag(sV, kb * 32 + r, s, h); }
        const int key = kb * 32 + r;
        const float bias_k = sBias[key];
        f32x16 dK[2] = {(f32x16){0.f}, (f32x16){0.f}}, dV[2] = {(f32x16){0.f}, (f32x16){0.f}};
#pragma unroll (VARLEN ? 1 : NKB)
        for (int qb = 0; qb < nbl; ++qb) {
            f32x16 S = (f32x16){0.f}, dP = (f32x16){0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                S = mfma32<F16>(row_frag(sQ, qb * 32 + r, s, h), kf[s], S);
                dP = mfma32<F16>(row_frag(sdO, qb * 32 + r, s, h), vf[s], dP);
            }
            // S[t], dP[t]: query q = 32 qb + rowmap(t, h), key = 32 kb + r
            float pd[16], ds[16];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // queries rowmap(4u .. 4u+3, h) are consecutive: their LSE / delta / dropout row keys come as 16-byte LDS reads
                const int q4 = qb * 32 + 8 * u + 4 * h;
                const float4 l4 = *(const float4*)(sLse + q4), d4 = *(const float4*)(sDelta + q4);
                const float ll[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
                uint4 k4 = make_uint4(0, 0, 0, 0);
                if (DROP) k4 = *(const uint4*)(sRk + q4);
                const uint32_t kk[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
                for (int j = 0; j < 4; j += 2) {      // two queries per packed fp32 instruction (the same IEEE operations as the scalar form)
                    const int t = 4 * u + j;
                    const cldrd_f32v2 a2 = fma2((cldrd_f32v2){S[t], S[t + 1]}, splat2(scale2), splat2(bias_k)) - (cldrd_f32v2){ll[j], ll[j + 1]};
                    const cldrd_f32v2 p2 = {__builtin_amdgcn_exp2f(a2.x), __builtin_amdgcn_exp2f(a2.y)};
                    cldrd_f32v2 pd2 = p2, dp2 = {dP[t], dP[t + 1]};
                    if (DROP) {
                        const bool keep0 = dropout_keep(kk[j], (uint32_t)key, drop_thresh), keep1 = dropout_keep(kk[j + 1], (uint32_t)key, drop_thresh);
                        pd2 = p2 * splat2(drop_scale);
                        dp2 = dp2 * splat2(drop_scale);
                        pd2.x = keep0 ? pd2.x : 0.f; pd2.y = keep1 ? pd2.y : 0.f;
                        dp2.x = keep0 ? dp2.x : 0.f; dp2.y = keep1 ? dp2.y : 0.f;
                    }
                    pd[t] = pd2.x; pd[t + 1] = pd2.y;
                    const cldrd_f32v2 ds2 = p2 * (dp2 - (cldrd_f32v2){dd[j], dd[j + 1]});
                    ds[t] = ds2.x; ds[t + 1] = ds2.y;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pb = pack8x<F16>(pd + 8 * s2), sb = pack8x<F16>(ds + 8 * s2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dV[dt] = mfma32<F16>(tr_frag(sdO, qb * 32 + 16 * s2 + 4 * h, dt * 32, lane), pb, dV[dt]);
                    dK[dt] = mfma32<F16>(tr_frag(sQ, qb * 32 + 16 * s2 + 4 * h, dt * 32, lane), sb, dK[dt]);
                }
            }
            if (VARLEN) mfma_drain();
        }
        // dV[dt][t] = dV[key][d = 32 dt + rowmap(t, h)]: regs 4u..4u+3 are 4 consecutive d
        {
            bf16_t* ok = dqkv + ((size_t)sr.row0 + key) * ld + dm + hd * 64;
            bf16_t* ov = ok + dm;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                    uint2 a[2], b[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int t = 4 * (u + k);
                        a[k].x = pack2x_scaled<F16>(dK[dt][t], dK[dt][t + 1], scale);
                        a[k].y = pack2x_scaled<F16>(dK[dt][t + 2], dK[dt][t + 3], scale);
                        b[k].x = pack2x<F16>(dV[dt][t], dV[dt][t + 1]);
                        b[k].y = pack2x<F16>(dV[dt][t + 2], dV[dt][t + 3]);
                    }
                    const uint4 wa = widen_pair(a[0], a[1]), wb = widen_pair(b[0], b[1]);      // 16 contiguous bytes per lane (see widen_pair)
                    const int d0 = dt * 32 + 8 * (u + h);
                    if (key < len) {
                        *(uint4*)(ok + d0) = wa;
                        *(uint4*)(ov + d0) = wb;
                    }
                }
        }
    }

    // ---------------- sweep B: query on lane; dQ for 32 queries accumulates over all key blocks ----------------
    for (int qb = wid; qb < NKB; qb += NWAVES) {
        if (qb >= nb) continue;
        bf16x8 qf[4], dof[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) { qf[s] = row_frag(sQ, qb * 32 + r, s, h); dof[s] = row_frag(sdO, qb * 32 + r, s, h); }
        const int q = qb * 32 + r;
        const float lse_q = sLse[q], delta_q = sDelta[q];
        const uint32_t rk_q = sRk[q];
        f32x16 dQ[2] = {(f32x16){0.f}, (f32x16){0.f}};
#pragma unroll (VARLEN ? 1 : NKB)
        for (int kb = 0; kb < nbl; ++kb) {
            f32x16 ST = (f32x16){0.f}, dPT = (f32x16){0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                ST = mfma32<F16>(row_frag(sK, kb * 32 + r, s, h), qf[s], ST);
                dPT = mfma32<F16>(row_frag(sV, kb * 32 + r, s, h), dof[s], dPT);
            }
            float ds[16];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int key4 = kb * 32 + 8 * u + 4 * h;
                const float4 b4 = *(const float4*)(sBias + key4);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const int t = 4 * u + j;
                    // packed fp32 (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: two keys per instruction, the same IEEE operations as before)
                    const cldrd_f32v2 a2 = fma2((cldrd_f32v2){ST[t], ST[t + 1]}, splat2(scale2), (cldrd_f32v2){bb[j], bb[j + 1]}) - splat2(lse_q);
                    const cldrd_f32v2 p2 = {__builtin_amdgcn_exp2f(a2.x), __builtin_amdgcn_exp2f(a2.y)};
                    cldrd_f32v2 dp2 = {dPT[t], dPT[t + 1]};
                    if (DROP) dp2 = dp2 * splat2(drop_scale);
                    float dp0 = dp2.x, dp1 = dp2.y;
                    if (DROP) {
                        const uint32_t hh = drop_pair(rk_q, (uint32_t)(key4 + j));
                        dp0 = drop_keep_lo(hh, drop_thresh) ? dp0 : 0.f;
                        dp1 = drop_keep_hi(hh, drop_thresh) ? dp1 : 0.f;
                    }
                    const cldrd_f32v2 ds2 = p2 * ((cldrd_f32v2){dp0, dp1} - splat2(delta_q));
                    ds[t] = ds2.x; ds[t + 1] = ds2.y;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 sa = pack8x<F16>(ds + 8 * s2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    // operands swapped: D[d][q], so a lane (= query) ends up with 4 consecutive head dimensions per register quad and dQ
                    // leaves in 8-byte stores (the other order gave 32 two-byte stores per lane)
                    dQ[dt] = mfma32<F16>(tr_frag(sK, kb * 32 + 16 * s2 + 4 * h, dt * 32, lane), sa, dQ[dt]);
            }
            if (VARLEN) mfma_drain();
        }
        {                 // dQ[dt][t] = dQ[q][d = 32 dt + rowmap(t, h)]: regs 4u..4u+3 are 4 consecutive d
            bf16_t* oq = dqkv + ((size_t)sr.row0 + q) * ld + hd * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                    uint2 a[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int t = 4 * (u + k);
                        a[k].x = pack2x_scaled<F16>(dQ[dt][t], dQ[dt][t + 1], scale);
                        a[k].y = pack2x_scaled<F16>(dQ[dt][t + 2], dQ[dt][t + 3], scale);
                    }
                    const uint4 w = widen_pair(a[0], a[1]);
                    if (q < len) *(uint4*)(oq + dt * 32 + 8 * (u + h)) = w;
                }
        }
    }
}

// Backward, persistent two-role form for L <= 128 and many (sequence, head) items.
//
// Measured on the kernel above at cfg2 (3072 items, two 4-wave workgroups per CU): loading an item's tiles takes 50 us of the
// 158 when every CU does it at once (HBM-bound: 252 MB), sweep A 3.2 us and sweep B 4.2 us per item at ONE wave per SIMD - and a
// workgroup does the three one after the other, so a CU overlaps them only across its two resident workgroups (LDS and the VGPRs
// allow no third).  Here one 8-wave workgroup per CU walks its items:
//   * waves 0..3 run sweep A while waves 4..7 run sweep B on the same tiles (two waves per SIMD, different work);
//   * waves 4..7 first issue the global loads of the NEXT item into registers (Q, K, V, dO, O: 80 VGPRs), run their sweep, then
//     write them to the other LDS buffer (with delta, mask bias, LSE, dropout row keys): issue-early / write-late, the HBM latency
//     sits under the sweep;  one __syncthreads per item.
// The sweeps are the code of attn_bwd_kernel with the block loops rolled (the prefetch registers need the room).
template <int NKB, bool DROP, bool BITS, bool F16 = false>
__global__ __launch_bounds__(512) void attn_bwd2_kernel(const bf16_t* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                         const bf16_t* __restrict__ ctx, const bf16_t* __restrict__ dctx,
                                                         const float* __restrict__ lse, bf16_t* __restrict__ dqkv, int L, int H,
                                                         float scale, uint32_t drop_thresh, float drop_scale, SeedArg seed_a, int nitems,
                                                         const uint32_t* __restrict__ drop_bits, const int* __restrict__ cu, const int* __restrict__ seq_list) {
    const uint64_t seed = seed_a.get();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int Lp = 32 * NKB;
    constexpr int TILE = Lp * RSB;
    constexpr int NBITS = BITS ? Lp * NKB : 0;          // the forward's dropout keep bits (attn_fwd2_kernel): dword [kb][q]
    constexpr int NBW = (NBITS + 255) / 256;
    constexpr int BUF = 4 * TILE + (4 * Lp + NBITS) * (int)sizeof(float);
    constexpr int NCH = NKB;                            // 16-byte chunks per thread and tile: Lp * 8 / 256
    const int tid = threadIdx.x, wid = tid >> 6;
    const bool role_b = wid >= 4;
    const int rw = wid & 3;
    const int dm = H * 64, ld = 3 * dm;
    const float scale2 = scale * LOG2E;
    // per-lane indices are re-derived from an OPAQUE copy of the thread id in every iteration: hipcc otherwise hoists the loop-
    // invariant halves of ~40 64-bit load / store addresses out of the item loop and spills them (76 VGPRs of scratch at L = 128)
    auto opaque = [](int x) { asm volatile("" : "+v"(x)); return x; };
    int tb = tid & 255;
    uint4 pq[NCH], pk[NCH], pv[NCH], pdo[NCH], po[NCH];
    // raw prefetched scalars of the next item: consumed in commit(), i.e. BEHIND the sweep.  (Until round 4 issue() compared the mask word
    // and scaled the LSE at once: the compiler had to wait for those two loads right there with vmcnt(0) - the counter is in order - and
    // with them for the twenty tile loads issued just before: the "issue early, write late" prefetch waited for itself.)
    long long p_mask = 1;
    float p_lse = 0.f;
    int p_len = 0;                   // row count of the prefetched item (L, or its length in the packed layout)
    uint32_t pbits[NBW > 0 ? NBW : 1];
    auto issue = [&](int item) {
        const int seq = seq_of(seq_list, item / H), hd = item % H;
        const SeqRows nr = seq_rows(cu, seq, L);
        p_len = nr.len;
        const bf16_t* base = qkv + (size_t)nr.row0 * ld + hd * 64;
        const bf16_t* dob = dctx + (size_t)nr.row0 * dm + hd * 64;
        const bf16_t* ob = ctx + (size_t)nr.row0 * dm + hd * 64;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int idx = tb + 256 * j, row = idx >> 3, ch = idx & 7;
            pq[j] = pk[j] = pv[j] = pdo[j] = po[j] = make_uint4(0, 0, 0, 0);
            if (row < nr.len) {
                const bf16_t* rp = base + (size_t)row * ld + ch * 8;
                pq[j] = ld16_stream(rp);                          // q, k, v, o: the forward's tape, read once (common.h)
                pk[j] = ld16_stream(rp + dm);
                pv[j] = ld16_stream(rp + 2 * dm);
                pdo[j] = *(const uint4*)(dob + (size_t)row * dm + ch * 8);
                po[j] = ld16_stream(ob + (size_t)row * dm + ch * 8);
            }
        }
        if (tb < Lp) {
            const int tc = tb < L ? tb : L - 1;           // clamped: the compare with L happens in commit()
            p_mask = mask ? mask[(size_t)seq * L + tc] : 1;
            p_lse = lse[((size_t)seq * H + hd) * L + tc];
        }
        if constexpr (BITS) {
#pragma unroll
            for (int j = 0; j < NBW; ++j) pbits[j] = tb + 256 * j < NBITS ? drop_bits[(size_t)item * NBITS + tb + 256 * j] : 0u;
        }
    };
    auto commit = [&](int b, int item) {
        const int seq = seq_of(seq_list, item / H), hd = item % H;
        char* base = smem + b * BUF;
        float* fl = (float*)(base + 4 * TILE);
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int idx = tb + 256 * j, row = idx >> 3, ch = idx & 7;
            const int off = row * RSB + ch * 16;
            *(uint4*)(base + off) = pq[j];
            *(uint4*)(base + TILE + off) = pk[j];
            *(uint4*)(base + 2 * TILE + off) = pv[j];
            *(uint4*)(base + 3 * TILE + off) = pdo[j];
            const uint32_t vv[4] = {pdo[j].x, pdo[j].y, pdo[j].z, pdo[j].w}, oo[4] = {po[j].x, po[j].y, po[j].z, po[j].w};
            float dsum = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) dsum += dot2x<F16>(vv[e], oo[e]);
            dsum += __shfl_xor(dsum, 1, 64); dsum += __shfl_xor(dsum, 2, 64); dsum += __shfl_xor(dsum, 4, 64);
            if (ch == 0) fl[2 * Lp + row] = dsum;
        }
        if (tb < Lp) {
            fl[tb] = (tb < p_len && p_mask != 0) ? 0.f : NEG_BIG;
            fl[Lp + tb] = tb < L ? p_lse * LOG2E : 1.0e30f;
            ((uint32_t*)fl)[3 * Lp + tb] = drop_rowkey(seed, (uint32_t)((seq * H + hd) * L + tb));
        }
        if constexpr (BITS) {
#pragma unroll
            for (int j = 0; j < NBW; ++j)
                if (tb + 256 * j < NBITS) ((uint32_t*)fl)[4 * Lp + tb + 256 * j] = pbits[j];
        }
    };
    int item = blockIdx.x;
    if (role_b && item < nitems) { issue(item); commit(0, item); }
    __syncthreads();
    int cur = 0;
    for (; item < nitems; item += gridDim.x) {
        const int next = item + gridDim.x;
        const bool has_next = next < nitems;
        const char* sQ = smem + cur * BUF;
        const char* sK = sQ + TILE;
        const char* sV = sK + TILE;
        const char* sdO = sV + TILE;
        const float* sBias = (const float*)(sdO + TILE);
        const float* sLse = sBias + Lp;
        const float* sDelta = sLse + Lp;
        const uint32_t* sRk = (const uint32_t*)(sDelta + Lp);
        const uint32_t* sBits = sRk + Lp;
        const int seq = seq_of(seq_list, item / H), hd = item % H;
        const SeqRows sr = seq_rows(cu, seq, L);
        const int len = sr.len, nb = CLDRD_LIVE_BLOCKS(len, NKB);
        const int t_ = opaque(tid);
        const int lane = t_ & 63, r = lane & 31, h = lane >> 5;
        tb = t_ & 255;
        if (role_b) {
            if (has_next) issue(next);
            if (rw < nb) {        // ---- sweep B: query on lane; dQ for 32 queries accumulates over all (live) key blocks
                const int qb = rw;
        bf16x8 qf[4], dof[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) { qf[s] = row_frag(sQ, qb * 32 + r, s, h); dof[s] = row_frag(sdO, qb * 32 + r, s, h); }
        const int q = qb * 32 + r;
        const float lse_q = sLse[q], delta_q = sDelta[q];
        const uint32_t rk_q = sRk[q];
        f32x16 dQ[2] = {(f32x16){0.f}, (f32x16){0.f}};
        for (int kb = 0; kb < nb; ++kb) {
            f32x16 ST = (f32x16){0.f}, dPT = (f32x16){0.f};
            uint32_t wbits = 0;
            if constexpr (BITS) wbits = sBits[kb * Lp + q] >> (4 * h);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                ST = mfma32<F16>(row_frag(sK, kb * 32 + r, s, h), qf[s], ST);
                dPT = mfma32<F16>(row_frag(sV, kb * 32 + r, s, h), dof[s], dPT);
            }
            float ds[16];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int key4 = kb * 32 + 8 * u + 4 * h;
                const float4 b4 = *(const float4*)(sBias + key4);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const int t = 4 * u + j;
                    // packed fp32 (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: two keys per instruction, the same IEEE operations as before)
                    const cldrd_f32v2 a2 = fma2((cldrd_f32v2){ST[t], ST[t + 1]}, splat2(scale2), (cldrd_f32v2){bb[j], bb[j + 1]}) - splat2(lse_q);
                    const cldrd_f32v2 p2 = {__builtin_amdgcn_exp2f(a2.x), __builtin_amdgcn_exp2f(a2.y)};
                    cldrd_f32v2 dp2 = {dPT[t], dPT[t + 1]};
                    if (DROP) dp2 = dp2 * splat2(drop_scale);
                    float dp0 = dp2.x, dp1 = dp2.y;
                    if (DROP) {
                        if constexpr (BITS) {       // keys key4 + j, key4 + j + 1: adjacent bits of the forward's dword [kb][q]
                            dp0 = keep_bit(dp0, wbits, 8 * u + j);
                            dp1 = keep_bit(dp1, wbits, 8 * u + j + 1);
                        } else {
                            const uint32_t hh = drop_pair(rk_q, (uint32_t)(key4 + j));
                            dp0 = drop_keep_lo(hh, drop_thresh) ? dp0 : 0.f;
                            dp1 = drop_keep_hi(hh, drop_thresh) ? dp1 : 0.f;
                        }
                    }
                    const cldrd_f32v2 ds2 = p2 * ((cldrd_f32v2){dp0, dp1} - splat2(delta_q));
                    ds[t] = ds2.x; ds[t + 1] = ds2.y;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 sa = pack8x<F16>(ds + 8 * s2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    // operands swapped: D[d][q], so a lane (= query) ends up with 4 consecutive head dimensions per register quad and dQ
                    // leaves in 8-byte stores (the other order gave 32 two-byte stores per lane)
                    dQ[dt] = mfma32<F16>(tr_frag(sK, kb * 32 + 16 * s2 + 4 * h, dt * 32, lane), sa, dQ[dt]);
            }
        }
        {                 // dQ[dt][t] = dQ[q][d = 32 dt + rowmap(t, h)]: regs 4u..4u+3 are 4 consecutive d
            bf16_t* oq = dqkv + ((size_t)sr.row0 + q) * ld + hd * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                    uint2 a[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int t = 4 * (u + k);
                        a[k].x = pack2x_scaled<F16>(dQ[dt][t], dQ[dt][t + 1], scale);
                        a[k].y = pack2x_scaled<F16>(dQ[dt][t + 2], dQ[dt][t + 3], scale);
                    }
                    const uint4 w = widen_pair(a[0], a[1]);
                    if (q < len) *(uint4*)(oq + dt * 32 + 8 * (u + h)) = w;
                }
        }
                }
            if (has_next) commit(cur ^ 1, next);
        } else if (rw < nb) {     // ---- sweep A: key on lane; dK, dV for 32 keys accumulate over all (live) query blocks
            const int kb = rw;
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) { kf[s] = row_frag(sK, kb * 32 + r, s, h); vf[s] = row_frag(sV, kb * 32 + r, s, h); }
        const int key = kb * 32 + r;
        const float bias_k = sBias[key];
        f32x16 dK[2] = {(f32x16){0.f}, (f32x16){0.f}}, dV[2] = {(f32x16){0.f}, (f32x16){0.f}};
        for (int qb = 0; qb < nb; ++qb) {
            f32x16 S = (f32x16){0.f}, dP = (f32x16){0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                S = mfma32<F16>(row_frag(sQ, qb * 32 + r, s, h), kf[s], S);
                dP = mfma32<F16>(row_frag(sdO, qb * 32 + r, s, h), vf[s], dP);
            }
            // S[t], dP[t]: query q = 32 qb + rowmap(t, h), key = 32 kb + r
            float pd[16], ds[16];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // queries rowmap(4u .. 4u+3, h) are consecutive: their LSE / delta / dropout row keys come as 16-byte LDS reads
                const int q4 = qb * 32 + 8 * u + 4 * h;
                const float4 l4 = *(const float4*)(sLse + q4), d4 = *(const float4*)(sDelta + q4);
                const float ll[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
                uint4 k4 = make_uint4(0, 0, 0, 0);
                if (DROP) k4 = BITS ? *(const uint4*)(sBits + kb * Lp + q4) : *(const uint4*)(sRk + q4);      // BITS: dwords [kb][q4 .. q4+3], bit = key
                const uint32_t kk[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
                for (int j = 0; j < 4; j += 2) {      // two queries per packed fp32 instruction (the same IEEE operations as the scalar form)
                    const int t = 4 * u + j;
                    const cldrd_f32v2 a2 = fma2((cldrd_f32v2){S[t], S[t + 1]}, splat2(scale2), splat2(bias_k)) - (cldrd_f32v2){ll[j], ll[j + 1]};
                    const cldrd_f32v2 p2 = {__builtin_amdgcn_exp2f(a2.x), __builtin_amdgcn_exp2f(a2.y)};
                    cldrd_f32v2 pd2 = p2, dp2 = {dP[t], dP[t + 1]};
                    if (DROP) {
                        pd2 = p2 * splat2(drop_scale);
                        dp2 = dp2 * splat2(drop_scale);
                        if constexpr (BITS) {
                            pd2.x = keep_bit(pd2.x, kk[j], r); pd2.y = keep_bit(pd2.y, kk[j + 1], r);
                            dp2.x = keep_bit(dp2.x, kk[j], r); dp2.y = keep_bit(dp2.y, kk[j + 1], r);
                        } else {
                            const bool keep0 = dropout_keep(kk[j], (uint32_t)key, drop_thresh), keep1 = dropout_keep(kk[j + 1], (uint32_t)key, drop_thresh);
                            pd2.x = keep0 ? pd2.x : 0.f; pd2.y = keep1 ? pd2.y : 0.f;
                            dp2.x = keep0 ? dp2.x : 0.f; dp2.y = keep1 ? dp2.y : 0.f;
                        }
                    }
                    pd[t] = pd2.x; pd[t + 1] = pd2.y;
                    const cldrd_f32v2 ds2 = p2 * (dp2 - (cldrd_f32v2){dd[j], dd[j + 1]});
                    ds[t] = ds2.x; ds[t + 1] = ds2.y;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pb = pack8x<F16>(pd + 8 * s2), sb = pack8x<F16>(ds + 8 * s2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dV[dt] = mfma32<F16>(tr_frag(sdO, qb * 32 + 16 * s2 + 4 * h, dt * 32, lane), pb, dV[dt]);
                    dK[dt] = mfma32<F16>(tr_frag(sQ, qb * 32 + 16 * s2 + 4 * h, dt * 32, lane), sb, dK[dt]);
                }
            }
        }
        // dV[dt][t] = dV[key][d = 32 dt + rowmap(t, h)]: regs 4u..4u+3 are 4 consecutive d
        {
            bf16_t* ok = dqkv + ((size_t)sr.row0 + key) * ld + dm + hd * 64;
            bf16_t* ov = ok + dm;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                    uint2 a[2], b[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int t = 4 * (u + k);
                        a[k].x = pack2x_scaled<F16>(dK[dt][t], dK[dt][t + 1], scale);
                        a[k].y = pack2x_scaled<F16>(dK[dt][t + 2], dK[dt][t + 3], scale);
                        b[k].x = pack2x<F16>(dV[dt][t], dV[dt][t + 1]);
                        b[k].y = pack2x<F16>(dV[dt][t + 2], dV[dt][t + 3]);
                    }
                    const uint4 wa = widen_pair(a[0], a[1]), wb = widen_pair(b[0], b[1]);      // 16 contiguous bytes per lane (see widen_pair)
                    const int d0 = dt * 32 + 8 * (u + h);
                    if (key < len) {
                        *(uint4*)(ok + d0) = wa;
                        *(uint4*)(ov + d0) = wb;
                    }
                }
        }
            }
        __syncthreads();          // the other buffer is complete, and nobody reads this one any more
        cur ^= 1;
    }
}

int attn_num_cus();
bool attn_fwd2_enabled(int nseq, int L, int H);

template <int NKB, bool DROP>
int launch_fwd_f16(const void* qkv, const long long* mask, void* ctx, float* lse, int nseq, int L, int H, float scale, float p,
                   unsigned long long seed, hipStream_t st, const int* cu, const int* seq_list) {      // fp16 in, fp16 out: no second copy
    const size_t lds = 3 * 32 * NKB * RSB + 32 * NKB * sizeof(float);
    (void)hipFuncSetAttribute((const void*)attn_fwd_full_kernel<NKB, DROP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((attn_fwd_full_kernel<NKB, DROP, true>), dim3(nseq * H), dim3(256), lds, st, (const bf16_t*)qkv, (const int64_t*)mask,
                       (bf16_t*)ctx, lse, L, H, scale, DROP ? dropout_thresh16(p) : 0u, 1.0f / (1.0f - p), seed_arg(seed), (bf16_t*)nullptr, cu, seq_list);
    CLDRD_LAUNCH_CHECK();
    return 0;
}
template <int NKB>
int launch_fwd_h(const void* qkv, const long long* mask, void* ctx, float* lse, int nseq, int L, int H, float scale, float p,
                 unsigned long long seed, hipStream_t st, const int* cu, const int* seq_list) {
    return p > 0.f && dropout_thresh16(p) > 0 ? launch_fwd_f16<NKB, true>(qkv, mask, ctx, lse, nseq, L, H, scale, p, seed, st, cu, seq_list)
                                              : launch_fwd_f16<NKB, false>(qkv, mask, ctx, lse, nseq, L, H, scale, 0.f, seed, st, cu, seq_list);
}

template <int NKB, bool DROP, bool F16 = false>
int launch_fwd_d(const void* qkv, const long long* mask, void* ctx, float* lse, int nseq, int L, int H, float scale, float p,
                 unsigned long long seed, uint32_t* bits_out, void* ctx16, hipStream_t st, const int* cu, const int* seq_list) {
    const size_t lds = 3 * 32 * NKB * RSB + 32 * NKB * sizeof(float);
    if constexpr (NKB <= 4) {
        // many items: the persistent loader / compute kernel (CLDRD_ATTN_FWD2=0 keeps the one-item-per-workgroup kernel: A/B runs and tests)
        const int nitems = nseq * H, cus = attn_num_cus();
        if (attn_fwd2_enabled(nseq, 32 * NKB, H)) {       // (the tile height, not L: a listed launch of short sequences has L > 128)
            const size_t lds2 = 2 * (lds + (DROP ? 32 * NKB * NKB * sizeof(uint32_t) : 0));
            (void)hipFuncSetAttribute((const void*)attn_fwd2_kernel<NKB, DROP, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            hipLaunchKernelGGL((attn_fwd2_kernel<NKB, DROP, F16>), dim3(cus), dim3(512), lds2, st, (const bf16_t*)qkv, (const int64_t*)mask,
                               (bf16_t*)ctx, lse, L, H, scale, DROP ? dropout_thresh16(p) : 0u, 1.0f / (1.0f - p), seed_arg(seed), nitems,
                               bits_out, (bf16_t*)ctx16, cu, seq_list);
            CLDRD_LAUNCH_CHECK();
            return 0;
        }
        (void)hipFuncSetAttribute((const void*)attn_fwd_full_kernel<NKB, DROP, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((attn_fwd_full_kernel<NKB, DROP, F16>), dim3(nseq * H), dim3(256), lds, st, (const bf16_t*)qkv, (const int64_t*)mask,
                           (bf16_t*)ctx, lse, L, H, scale, DROP ? dropout_thresh16(p) : 0u, 1.0f / (1.0f - p), seed_arg(seed), (bf16_t*)ctx16, cu, seq_list);
    } else {
        // many items at 128 < L <= 256: the persistent streaming kernel (K / V double-buffered, Q from global memory); cldrd_set_tuning("attn_fwd2", 0)
        // keeps the one-item-per-workgroup kernel (tests: the two are bit-identical)
        const int nitems = nseq * H, cus = attn_num_cus();
        if (nitems >= 2 * cus && g_cldrd_tune_attn_fwd2 != 0) {
            const size_t lds3 = 2 * (2 * 32 * NKB * RSB + 32 * NKB * sizeof(float));
            (void)hipFuncSetAttribute((const void*)attn_fwd3_kernel<NKB, DROP, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
            hipLaunchKernelGGL((attn_fwd3_kernel<NKB, DROP, F16>), dim3(cus), dim3(512), lds3, st, (const bf16_t*)qkv, (const int64_t*)mask,
                               (bf16_t*)ctx, lse, L, H, scale, DROP ? dropout_thresh16(p) : 0u, 1.0f / (1.0f - p), seed_arg(seed), nitems,
                               (bf16_t*)ctx16, cu, seq_list);
            CLDRD_LAUNCH_CHECK();
            return 0;
        }
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<NKB, DROP, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((attn_fwd_kernel<NKB, DROP, F16>), dim3(nseq * H), dim3(512), lds, st, (const bf16_t*)qkv, (const int64_t*)mask,
                           (bf16_t*)ctx, lse, L, H, scale, DROP ? dropout_thresh16(p) : 0u, 1.0f / (1.0f - p), seed_arg(seed), (bf16_t*)ctx16, cu, seq_list);
    }
    CLDRD_LAUNCH_CHECK();
    return 0;
}
template <int NKB, bool F16 = false>
int launch_fwd(const void* qkv, const long long* mask, void* ctx, float* lse, int nseq, int L, int H, float scale, float p,
               unsigned long long seed, uint32_t* bits_out, void* ctx16, hipStream_t st, const int* cu, const int* seq_list) {
    return p > 0.f && dropout_thresh16(p) > 0 ? launch_fwd_d<NKB, true, F16>(qkv, mask, ctx, lse, nseq, L, H, scale, p, seed, bits_out, ctx16, st, cu, seq_list)
                                              : launch_fwd_d<NKB, false, F16>(qkv, mask, ctx, lse, nseq, L, H, scale, 0.f, seed, nullptr, ctx16, st, cu, seq_list);
}
int attn_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n = v;
    }
    return n;
}

bool attn_fwd2_enabled(int nseq, int L, int H) {
    return L <= 128 && nseq * H >= 2 * attn_num_cus() && g_cldrd_tune_attn_fwd2 != 0;
}

template <int NKB, bool DROP, bool F16 = false>
int launch_bwd_d(const void* qkv, const long long* mask, const void* ctx, const void* dctx, const float* lse, void* dqkv, int nseq,
                 int L, int H, float scale, float p, unsigned long long seed, const uint32_t* drop_bits, hipStream_t st, const int* cu, const int* seq_list) {
    if constexpr (NKB <= 4) {
        // many items: the persistent two-role kernel (cldrd_set_tuning("attn_bwd2", 0) keeps the one-item-per-workgroup kernel: tests)
        const int nitems = nseq * H, cus = attn_num_cus();
        if (nitems >= 2 * cus && g_cldrd_tune_attn_bwd2 != 0) {
            const size_t lds1 = 4 * 32 * NKB * RSB + 4 * 32 * NKB * sizeof(float);
            if (DROP && drop_bits) {
                const size_t lds2 = 2 * (lds1 + 32 * NKB * NKB * sizeof(uint32_t));
                (void)hipFuncSetAttribute((const void*)attn_bwd2_kernel<NKB, DROP, DROP, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
                hipLaunchKernelGGL((attn_bwd2_kernel<NKB, DROP, DROP, F16>), dim3(cus), dim3(512), lds2, st, (const bf16_t*)qkv, (const int64_t*)mask,
                                   (const bf16_t*)ctx, (const bf16_t*)dctx, lse, (bf16_t*)dqkv, L, H, scale,
                                   DROP ? dropout_thresh16(p) : 0u, 1.0f / (1.0f - p), seed_arg(seed), nitems, drop_bits, cu, seq_list);
            } else {
                (void)hipFuncSetAttribute((const void*)attn_bwd2_kernel<NKB, DROP, false, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds1));
                hipLaunchKernelGGL((attn_bwd2_kernel<NKB, DROP, false, F16>), dim3(cus), dim3(512), 2 * lds1, st, (const bf16_t*)qkv, (const int64_t*)mask,
                                   (const bf16_t*)ctx, (const bf16_t*)dctx, lse, (bf16_t*)dqkv, L, H, scale,
                                   DROP ? dropout_thresh16(p) : 0u, 1.0f / (1.0f - p), seed_arg(seed), nitems, (const uint32_t*)nullptr, cu, seq_list);
            }
            CLDRD_LAUNCH_CHECK();
            return 0;
        }
    }
    const size_t lds = 4 * 32 * NKB * RSB + 4 * 32 * NKB * sizeof(float);
    if (cu != nullptr && NKB > 1) {       // a packed batch: the instantiation whose block loops stop at the sequence's last live block
        (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<NKB, DROP, F16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((attn_bwd_kernel<NKB, DROP, F16, true>), dim3(nseq * H), dim3(NKB > 4 ? 512 : 256), lds, st, (const bf16_t*)qkv,
                           (const int64_t*)mask, (const bf16_t*)ctx, (const bf16_t*)dctx, lse, (bf16_t*)dqkv, L, H, scale,
                           DROP ? dropout_thresh16(p) : 0u, 1.0f / (1.0f - p), seed_arg(seed), cu, seq_list);
        CLDRD_LAUNCH_CHECK();
        return 0;
    }
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<NKB, DROP, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((attn_bwd_kernel<NKB, DROP, F16>), dim3(nseq * H), dim3(NKB > 4 ? 512 : 256), lds, st, (const bf16_t*)qkv, (const int64_t*)mask,
                       (const bf16_t*)ctx, (const bf16_t*)dctx, lse, (bf16_t*)dqkv, L, H, scale,
                       DROP ? dropout_thresh16(p) : 0u, 1.0f / (1.0f - p), seed_arg(seed), cu, seq_list);
    CLDRD_LAUNCH_CHECK();
    return 0;
}
template <int NKB, bool F16 = false>
int launch_bwd(const void* qkv, const long long* mask, const void* ctx, const void* dctx, const float* lse, void* dqkv, int nseq,
               int L, int H, float scale, float p, unsigned long long seed, const uint32_t* drop_bits, hipStream_t st, const int* cu, const int* seq_list) {
    return p > 0.f && dropout_thresh16(p) > 0
               ? launch_bwd_d<NKB, true, F16>(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, scale, p, seed, drop_bits, st, cu, seq_list)
               : launch_bwd_d<NKB, false, F16>(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, scale, 0.f, seed, nullptr, st, cu, seq_list);
}

}  // namespace

// qkv: bf16 [nseq*L, 3*H*64] (Q | K | V, heads contiguous inside each); mask: int64 [nseq, L] (0 = padded key) or null;
// ctx: bf16 [nseq*L, H*64]; lse: fp32 [nseq, H, L] (may be null for inference).
// Does the forward for this shape run the persistent kernel that leaves the dropout keep bits behind?  Returns the number of 32-bit
// words of the bit array (nseq*H items x [key block][query]), 0 when the bits are not produced (the backward then hashes itself).
extern "C" long long cldrd_attention_bits_words(int nseq, int L, int H, float dropout_p) {
    if (!(dropout_p > 0.f && dropout_thresh16(dropout_p) > 0) || nseq <= 0 || L <= 0 || H <= 0 || !attn_fwd2_enabled(nseq, L, H)) return 0;
    const long long nkb = (L + 31) / 32;
    return (long long)nseq * H * 32 * nkb * nkb;
}

extern "C" int cldrd_attention_fwd_bits(const void* qkv, const long long* mask, void* ctx, float* lse, int nseq, int L, int H,
                                        float dropout_p, unsigned long long seed, int io_f16, void* drop_bits_out, void* ctx_f16_copy,
                                        void* stream);
extern "C" int cldrd_attention_fwd(const void* qkv, const long long* mask, void* ctx, float* lse, int nseq, int L, int H,
                                   float dropout_p, unsigned long long seed, int io_f16, void* stream) {
    return cldrd_attention_fwd_bits(qkv, mask, ctx, lse, nseq, L, H, dropout_p, seed, io_f16, nullptr, nullptr, stream);
}
// drop_bits_out (optional, cldrd_attention_bits_words() words): receives the dropout keep bits for cldrd_attention_bwd_bits.
// ctx_f16_copy (optional, bf16 pass only): the same context in fp16 - the operand of an fp16 out-projection GEMM; ctx itself may then be null
// (an evaluation forward keeps no bf16 tape).
static int attention_fwd_impl(const void* qkv, const long long* mask, void* ctx, float* lse, int nseq, int L, int H, float dropout_p,
                              unsigned long long seed, int io_f16, void* drop_bits_out, void* ctx_f16_copy, const int* cu, void* stream,
                              const int* seq_list = nullptr, int n_list = 0, int Ltile = 0);
extern "C" int cldrd_attention_fwd_bits(const void* qkv, const long long* mask, void* ctx, float* lse, int nseq, int L, int H,
                                        float dropout_p, unsigned long long seed, int io_f16, void* drop_bits_out, void* ctx_f16_copy,
                                        void* stream) {
    return attention_fwd_impl(qkv, mask, ctx, lse, nseq, L, H, dropout_p, seed, io_f16, drop_bits_out, ctx_f16_copy, nullptr, stream);
}
// The same on a PACKED batch (pack.hip): qkv [Tp, 3*H*64] and ctx [Tp, H*64] hold the rows cu_rows[m] .. cu_rows[m + 1] of sequence m
// (cu_rows: int32 [nseq + 1] on the device, every length in 1 .. L); keys >= a sequence's length are masked (right padding, what an HF
// tokenizer's attention mask says), no mask tensor is read.  lse [nseq, H, L] and the keep bits keep their padded shapes.  Bit for bit what
// cldrd_unpack_rows16 -> cldrd_attention_fwd_bits -> cldrd_gather_rows give, without the two row moves and without loading padding rows.
extern "C" int cldrd_attention_fwd_varlen(const void* qkv_packed, const int* cu_rows, void* ctx_packed, float* lse, int nseq, int L, int H,
                                          float dropout_p, unsigned long long seed, int io_f16, void* drop_bits_out, void* ctx_f16_copy,
                                          void* stream) {
    CLDRD_CHECK(cu_rows != nullptr, "attention_fwd_varlen: cu_rows is required");
    return attention_fwd_impl(qkv_packed, nullptr, ctx_packed, lse, nseq, L, H, dropout_p, seed, io_f16, drop_bits_out, ctx_f16_copy, cu_rows, stream);
}
// The same for a LIST of the batch's sequences (seq_list: device int32 [n_list], positions in 0 .. nseq - 1, each at most Ltile <= L tokens long):
// the launch runs the kernels of tile height Ltile - a packed batch at L = 256 sends its sequences of at most 128 tokens through the persistent
// L <= 128 kernels and the rest through a second call (round 6).  LSE rows and dropout row keys keep the stride L of the whole batch, so the
// backward of a sequence must be given the same L (any list).  No keep bits are produced for L > 128 (cldrd_attention_bits_words).
extern "C" int cldrd_attention_fwd_varlen_list(const void* qkv_packed, const int* cu_rows, void* ctx_packed, float* lse, int nseq, int L, int H,
                                               float dropout_p, unsigned long long seed, int io_f16, void* drop_bits_out, void* ctx_f16_copy,
                                               const int* seq_list, int n_list, int Ltile, void* stream) {
    CLDRD_CHECK(cu_rows != nullptr && seq_list != nullptr, "attention_fwd_varlen_list: cu_rows and seq_list are required");
    return attention_fwd_impl(qkv_packed, nullptr, ctx_packed, lse, nseq, L, H, dropout_p, seed, io_f16, drop_bits_out, ctx_f16_copy, cu_rows, stream,
                              seq_list, n_list, Ltile);
}
static int attention_fwd_impl(const void* qkv, const long long* mask, void* ctx, float* lse, int nseq_all, int L, int H, float dropout_p,
                              unsigned long long seed, int io_f16, void* drop_bits_out, void* ctx_f16_copy, const int* cu, void* stream,
                              const int* seq_list, int n_list, int Ltile) {
    CLDRD_CHECK(nseq_all > 0 && L > 0 && L <= 256 && H > 0, "attention_fwd: need 0 < L <= 256");
    CLDRD_CHECK(seq_list == nullptr || (cu != nullptr && n_list > 0 && n_list <= nseq_all && Ltile > 0 && Ltile <= L),
                "attention_fwd: a sequence list goes with a packed batch, 0 < n_list <= nseq, 0 < Ltile <= L");
    const int nseq = seq_list ? n_list : nseq_all;          // sequences of THIS launch (items = nseq x H)
    const int Lt = seq_list ? Ltile : L;                    // rows a sequence of this launch can have: the kernels' tile height
    CLDRD_CHECK(ctx != nullptr || ctx_f16_copy != nullptr, "attention_fwd: no output");
    CLDRD_CHECK(!(io_f16 && (ctx_f16_copy != nullptr || ctx == nullptr)), "attention_fwd: the fp16 pass writes ctx only");
    CLDRD_CHECK(io_f16 == 0 || io_f16 == 1 || io_f16 == 5, "attention_fwd: io_f16 is 0 (bf16), 1 (fp16, L <= 128, no keep bits) or 5 (fp16, every kernel of the bf16 path)");
    const float scale = 0.125f;   // 1 / sqrt(64)
    const int nkb = (Lt + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    if (io_f16 == 5) {            // fp16 activations through the whole kernel family (round 4: the all-fp16 training mode)
        switch (nkb) {
            case 1: return launch_fwd<1, true>(qkv, mask, ctx, lse, nseq, L, H, scale, dropout_p, seed, (uint32_t*)drop_bits_out, nullptr, st, cu, seq_list);
            case 2: return launch_fwd<2, true>(qkv, mask, ctx, lse, nseq, L, H, scale, dropout_p, seed, (uint32_t*)drop_bits_out, nullptr, st, cu, seq_list);
            case 3: return launch_fwd<3, true>(qkv, mask, ctx, lse, nseq, L, H, scale, dropout_p, seed, (uint32_t*)drop_bits_out, nullptr, st, cu, seq_list);
            case 4: return launch_fwd<4, true>(qkv, mask, ctx, lse, nseq, L, H, scale, dropout_p, seed, (uint32_t*)drop_bits_out, nullptr, st, cu, seq_list);
            case 5: return launch_fwd<5, true>(qkv, mask, ctx, lse, nseq, L, H, scale, dropout_p, seed, (uint32_t*)drop_bits_out, nullptr, st, cu, seq_list);
            case 6: return launch_fwd<6, true>(qkv, mask, ctx, lse, nseq, L, H, scale, dropout_p, seed, (uint32_t*)drop_bits_out, nullptr, st, cu, seq_list);
            case 7: return launch_fwd<7, true>(qkv, mask, ctx, lse, nseq, L, H, scale, dropout_p, seed, (uint32_t*)drop_bits_out, nullptr, st, cu, seq_list);
            default: return launch_fwd<8, true>(qkv, mask, ctx, lse, nseq, L, H, scale, dropout_p, seed, (uint32_t*)drop_bits_out, nullptr, st, cu, seq_list);
        }
    }
    if (io_f16) {                 // fp16 activations: the all-scores-in-registers kernel only (L <= 128)
        CLDRD_CHECK(Lt <= 128, "attention_fwd: the fp16 forward handles L <= 128");
        switch (nkb) {
            case 1: return launch_fwd_h<1>(qkv, mask, ctx, lse, nseq, L, H, scale, dropout_p, seed, st, cu, seq_list);
            case 2: return launch_fwd_h<2>(qkv, mask, ctx, lse, nseq, L, H, scale, dropout_p, seed, st, cu, seq_list);
            case 3: return launch_fwd_h<3>(qkv, mask, ctx, lse, nseq, L, H, scale, dropout_p, seed, st, cu, seq_list);
            default: return launch_fwd_h<4>(qkv, mask, ctx, lse, nseq, L, H, scale, dropout_p, seed, st, cu, seq_list);
        }
    }
    switch (nkb) {
        case 1: return launch_fwd<1>(qkv, mask, ctx, lse, nseq, L, H, scale, dropout_p, seed, (uint32_t*)drop_bits_out, ctx_f16_copy, st, cu, seq_list);
        case 2: return launch_fwd<2>(qkv, mask, ctx, lse, nseq, L, H, scale, dropout_p, seed, (uint32_t*)drop_bits_out, ctx_f16_copy, st, cu, seq_list);
        case 3: return launch_fwd<3>(qkv, mask, ctx, lse, nseq, L, H, scale, dropout_p, seed, (uint32_t*)drop_bits_out, ctx_f16_copy, st, cu, seq_list);
        case 4: return launch_fwd<4>(qkv, mask, ctx, lse, nseq, L, H, scale, dropout_p, seed, (uint32_t*)drop_bits_out, ctx_f16_copy, st, cu, seq_list);
        case 5: return launch_fwd<5>(qkv, mask, ctx, lse, nseq, L, H, scale, dropout_p, seed, (uint32_t*)drop_bits_out, ctx_f16_copy, st, cu, seq_list);
        case 6: return launch_fwd<6>(qkv, mask, ctx, lse, nseq, L, H, scale, dropout_p, seed, (uint32_t*)drop_bits_out, ctx_f16_copy, st, cu, seq_list);
        case 7: return launch_fwd<7>(qkv, mask, ctx, lse, nseq, L, H, scale, dropout_p, seed, (uint32_t*)drop_bits_out, ctx_f16_copy, st, cu, seq_list);
        default: return launch_fwd<8>(qkv, mask, ctx, lse, nseq, L, H, scale, dropout_p, seed, (uint32_t*)drop_bits_out, ctx_f16_copy, st, cu, seq_list);
    }
}

extern "C" int cldrd_attention_bwd_bits(const void* qkv, const long long* mask, const void* ctx, const void* dctx, const float* lse,
                                        void* dqkv, int nseq, int L, int H, float dropout_p, unsigned long long seed, const void* drop_bits,
                                        void* stream);
extern "C" int cldrd_attention_bwd(const void* qkv, const long long* mask, const void* ctx, const void* dctx, const float* lse,
                                   void* dqkv, int nseq, int L, int H, float dropout_p, unsigned long long seed, void* stream) {
    return cldrd_attention_bwd_bits(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, dropout_p, seed, nullptr, stream);
}
extern "C" int cldrd_attention_bwd_x(const void* qkv, const long long* mask, const void* ctx, const void* dctx, const float* lse,
                                     void* dqkv, int nseq, int L, int H, float dropout_p, unsigned long long seed, const void* drop_bits,
                                     int io_f16, void* stream);
// drop_bits (optional): what cldrd_attention_fwd_bits left for the same (nseq, L, H, dropout_p, seed); null: the mask is re-hashed.
extern "C" int cldrd_attention_bwd_bits(const void* qkv, const long long* mask, const void* ctx, const void* dctx, const float* lse,
                                        void* dqkv, int nseq, int L, int H, float dropout_p, unsigned long long seed, const void* drop_bits,
                                        void* stream) {
    return cldrd_attention_bwd_x(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, dropout_p, seed, drop_bits, 0, stream);
}
// io_f16 != 0: q / k / v, ctx, dctx and dqkv are fp16 (the all-fp16 training mode: gradients carry the loss scale)
static int attention_bwd_impl(const void* qkv, const long long* mask, const void* ctx, const void* dctx, const float* lse, void* dqkv, int nseq,
                              int L, int H, float dropout_p, unsigned long long seed, const void* drop_bits, int io_f16, const int* cu, void* stream,
                              const int* seq_list = nullptr, int n_list = 0, int Ltile = 0);
extern "C" int cldrd_attention_bwd_x(const void* qkv, const long long* mask, const void* ctx, const void* dctx, const float* lse,
                                     void* dqkv, int nseq, int L, int H, float dropout_p, unsigned long long seed, const void* drop_bits,
                                     int io_f16, void* stream) {
    return attention_bwd_impl(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, dropout_p, seed, drop_bits, io_f16, nullptr, stream);
}
// The backward on a PACKED batch (see cldrd_attention_fwd_varlen): qkv, ctx, dctx, dqkv are [Tp, .]; rows of dqkv that do not exist in the
// packed layout (padding) are not written - in the padded layout they receive zeros.
extern "C" int cldrd_attention_bwd_varlen(const void* qkv_packed, const int* cu_rows, const void* ctx_packed, const void* dctx_packed,
                                          const float* lse, void* dqkv_packed, int nseq, int L, int H, float dropout_p, unsigned long long seed,
                                          const void* drop_bits, int io_f16, void* stream) {
    CLDRD_CHECK(cu_rows != nullptr, "attention_bwd_varlen: cu_rows is required");
    return attention_bwd_impl(qkv_packed, nullptr, ctx_packed, dctx_packed, lse, dqkv_packed, nseq, L, H, dropout_p, seed, drop_bits, io_f16, cu_rows, stream);
}
extern "C" int cldrd_attention_bwd_varlen_list(const void* qkv_packed, const int* cu_rows, const void* ctx_packed, const void* dctx_packed,
                                               const float* lse, void* dqkv_packed, int nseq, int L, int H, float dropout_p, unsigned long long seed,
                                               const void* drop_bits, int io_f16, const int* seq_list, int n_list, int Ltile, void* stream) {
    CLDRD_CHECK(cu_rows != nullptr && seq_list != nullptr, "attention_bwd_varlen_list: cu_rows and seq_list are required");
    return attention_bwd_impl(qkv_packed, nullptr, ctx_packed, dctx_packed, lse, dqkv_packed, nseq, L, H, dropout_p, seed, drop_bits, io_f16, cu_rows,
                              stream, seq_list, n_list, Ltile);
}
static int attention_bwd_impl(const void* qkv, const long long* mask, const void* ctx, const void* dctx, const float* lse, void* dqkv, int nseq_all,
                              int L, int H, float dropout_p, unsigned long long seed, const void* drop_bits, int io_f16, const int* cu, void* stream,
                              const int* seq_list, int n_list, int Ltile) {
    CLDRD_CHECK(nseq_all > 0 && L > 0 && L <= 256 && H > 0, "attention_bwd: need 0 < L <= 256");
    CLDRD_CHECK(seq_list == nullptr || (cu != nullptr && n_list > 0 && n_list <= nseq_all && Ltile > 0 && Ltile <= L),
                "attention_bwd: a sequence list goes with a packed batch, 0 < n_list <= nseq, 0 < Ltile <= L");
    const int nseq = seq_list ? n_list : nseq_all;          // sequences of THIS launch; Lt: the rows one of them can have (tile height)
    const int Lt = seq_list ? Ltile : L;
    CLDRD_CHECK(lse != nullptr, "attention_bwd: lse is required");
    const float scale = 0.125f;
    const int nkb = (Lt + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    if (io_f16) {
        switch (nkb) {
            case 1: return launch_bwd<1, true>(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, scale, dropout_p, seed, (const uint32_t*)drop_bits, st, cu, seq_list);
            case 2: return launch_bwd<2, true>(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, scale, dropout_p, seed, (const uint32_t*)drop_bits, st, cu, seq_list);
            case 3: return launch_bwd<3, true>(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, scale, dropout_p, seed, (const uint32_t*)drop_bits, st, cu, seq_list);
            case 4: return launch_bwd<4, true>(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, scale, dropout_p, seed, (const uint32_t*)drop_bits, st, cu, seq_list);
            case 5: return launch_bwd<5, true>(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, scale, dropout_p, seed, (const uint32_t*)drop_bits, st, cu, seq_list);
            case 6: return launch_bwd<6, true>(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, scale, dropout_p, seed, (const uint32_t*)drop_bits, st, cu, seq_list);
            case 7: return launch_bwd<7, true>(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, scale, dropout_p, seed, (const uint32_t*)drop_bits, st, cu, seq_list);
            default: return launch_bwd<8, true>(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, scale, dropout_p, seed, (const uint32_t*)drop_bits, st, cu, seq_list);
        }
    }
    switch (nkb) {
        case 1: return launch_bwd<1>(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, scale, dropout_p, seed, (const uint32_t*)drop_bits, st, cu, seq_list);
        case 2: return launch_bwd<2>(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, scale, dropout_p, seed, (const uint32_t*)drop_bits, st, cu, seq_list);
        case 3: return launch_bwd<3>(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, scale, dropout_p, seed, (const uint32_t*)drop_bits, st, cu, seq_list);
        case 4: return launch_bwd<4>(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, scale, dropout_p, seed, (const uint32_t*)drop_bits, st, cu, seq_list);
        case 5: return launch_bwd<5>(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, scale, dropout_p, seed, (const uint32_t*)drop_bits, st, cu, seq_list);
        case 6: return launch_bwd<6>(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, scale, dropout_p, seed, (const uint32_t*)drop_bits, st, cu, seq_list);
        case 7: return launch_bwd<7>(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, scale, dropout_p, seed, (const uint32_t*)drop_bits, st, cu, seq_list);
        default: return launch_bwd<8>(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, scale, dropout_p, seed, (const uint32_t*)drop_bits, st, cu, seq_list);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// CLS-only attention for the LAST encoder layer.  The reference pools `last_hidden_state[:, 0, :]`
// (models/nway_dual_encoder.py:52,56,64), so in the last layer only the query of token 0 matters:
// K and V are still needed for every token, but the score matrix shrinks to one row per (sequence, head) and
// everything after attention runs on one row per sequence.  One wavefront per (sequence, head).
namespace {

template <bool F16>
__global__ __launch_bounds__(64) void attn_cls_fwd_kernel(const bf16_t* __restrict__ qc, const bf16_t* __restrict__ kv,
                                                           const int64_t* __restrict__ mask, bf16_t* __restrict__ ctx,
                                                           float* __restrict__ probs, int L, int H, float scale,
                                                           uint32_t drop_thresh, float drop_scale, SeedArg seed_a, bf16_t* __restrict__ ctx16,
                                                           const int* __restrict__ cu) {
    const uint64_t seed = seed_a.get();
    __shared__ float sp[256];
    __shared__ float sq[64];
    const int seq = blockIdx.x / H, hd = blockIdx.x % H, lane = threadIdx.x;
    const int dm = H * 64;
    sq[lane] = x2f<F16>(qc[(size_t)seq * dm + hd * 64 + lane]);
    __syncthreads();
    const SeqRows sr = seq_rows(cu, seq, L);          // packed K | V rows: see seq_rows
    const int len = sr.len;
    const bf16_t* kb = kv + (size_t)sr.row0 * 2 * dm + hd * 64;
    const bf16_t* vb = kb + dm;
    float sc[4];
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int key = lane + 64 * i;
        float s = -1.0e30f;
        if (key < len && (!mask || mask[(size_t)seq * L + key] != 0)) {
            const bf16_t* kr = kb + (size_t)key * 2 * dm;
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint4 u = *(const uint4*)(kr + c * 8);
                const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    a += x2f<F16>((bf16_t)(w[j] & 0xFFFFu)) * sq[c * 8 + 2 * j] + x2f<F16>((bf16_t)(w[j] >> 16)) * sq[c * 8 + 2 * j + 1];
            }
            s = a * scale;
        }
        sc[i] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { sc[i] = __expf(sc[i] - mx); sum += (lane + 64 * i < L) ? sc[i] : 0.f; }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int key = lane + 64 * i;
        if (key < L) {
            const float p = sc[i] * inv;
            probs[((size_t)seq * H + hd) * L + key] = p;
            float pd = p;
            if (drop_thresh) pd = dropout_keep(drop_rowkey(seed, (uint32_t)((seq * H + hd) * L)), (uint32_t)key, drop_thresh) ? p * drop_scale : 0.f;
            sp[key] = pd;
        }
    }
    __syncthreads();
    // ctx = P V.  Eight lanes share a key row (16 bytes = 8 features each), eight keys per instruction: L / 8 row-block loads instead of
    // the L dependent 2-byte-per-lane loads this loop was until round 3 (24 of the kernel's 34 us at L = 128: pure load latency, on the
    // step's critical path between the last K / V projection and the loss).  The eight key groups are combined through LDS.
    const int c8 = lane & 7, g = lane >> 3;
    float o8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o8[j] = 0.f;
    for (int k0 = 0; k0 < len; k0 += 8) {           // keys >= len: p = 0 and (padded layout) zero rows: + 0.0f, skipped
        const int key = k0 + g;
        if (key < len) {
            const float pk = sp[key];
            const uint4 u = *(const uint4*)(vb + (size_t)key * 2 * dm + c8 * 8);
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o8[2 * j] += pk * x2f<F16>((bf16_t)(w[j] & 0xFFFFu));
                o8[2 * j + 1] += pk * x2f<F16>((bf16_t)(w[j] >> 16));
            }
        }
    }
    __shared__ float so[8][64];
#pragma unroll
    for (int j = 0; j < 8; ++j) so[g][c8 * 8 + j] = o8[j];
    __syncthreads();
    float o = 0.f;
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) o += so[gg][lane];
    if (ctx) ctx[(size_t)seq * dm + hd * 64 + lane] = f2x<F16>(o);
    if (ctx16) ctx16[(size_t)seq * dm + hd * 64 + lane] = f2x<true>(o);       // fp16 copy of a bf16 pass (out-projection operand)
}

template <bool F16>
__global__ __launch_bounds__(64) void attn_cls_bwd_kernel(const bf16_t* __restrict__ qc, const bf16_t* __restrict__ kv,
                                                           const float* __restrict__ probs, const bf16_t* __restrict__ dctx,
                                                           bf16_t* __restrict__ dqc, bf16_t* __restrict__ dkv, int L, int H,
                                                           float scale, uint32_t drop_thresh, float drop_scale, SeedArg seed_a,
                                                           const int* __restrict__ cu) {
    const uint64_t seed = seed_a.get();
    __shared__ float sds[256], spd[256], sdo[64];
    const int seq = blockIdx.x / H, hd = blockIdx.x % H, lane = threadIdx.x;
    const int dm = H * 64;
    sdo[lane] = x2f<F16>(dctx[(size_t)seq * dm + hd * 64 + lane]);
    const float qd = x2f<F16>(qc[(size_t)seq * dm + hd * 64 + lane]);
    __syncthreads();
    const SeqRows sr = seq_rows(cu, seq, L);
    const int len = sr.len;
    const bf16_t* kb = kv + (size_t)sr.row0 * 2 * dm + hd * 64;
    const bf16_t* vb = kb + dm;
    float p[4], dp[4], dot = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int key = lane + 64 * i;
        p[i] = dp[i] = 0.f;
        if (key < len) {
            p[i] = probs[((size_t)seq * H + hd) * L + key];
            const bf16_t* vr = vb + (size_t)key * 2 * dm;
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint4 u = *(const uint4*)(vr + c * 8);
                const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    a += x2f<F16>((bf16_t)(w[j] & 0xFFFFu)) * sdo[c * 8 + 2 * j] + x2f<F16>((bf16_t)(w[j] >> 16)) * sdo[c * 8 + 2 * j + 1];
            }
            float pdv = p[i];
            if (drop_thresh) {
                const bool keep = dropout_keep(drop_rowkey(seed, (uint32_t)((seq * H + hd) * L)), (uint32_t)key, drop_thresh);
                pdv = keep ? p[i] * drop_scale : 0.f;
                a = keep ? a * drop_scale : 0.f;
            }
            dp[i] = a;
            spd[key] = pdv;
            dot += p[i] * a;
        }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int key = lane + 64 * i;
        if (key < len) sds[key] = p[i] * (dp[i] - dot) * scale;
    }
    __syncthreads();
    // dq = sum_key ds[key] K[key]; dK[key] = ds[key] q; dV[key] = p_drop[key] dO.  Eight lanes per key row (16 bytes = 8 features
    // each), eight keys per instruction (see attn_cls_fwd_kernel): L / 8 row-block loads and 2 L / 8 row-block stores instead of L + 2 L
    // two-byte-per-lane accesses; the eight key groups' dq partial sums are combined through LDS.
    __shared__ float sqd[64];
    sqd[lane] = qd;
    __syncthreads();
    bf16_t* dkb = dkv + (size_t)sr.row0 * 2 * dm + hd * 64;
    const int c8 = lane & 7, g = lane >> 3;
    float q8[8], do8[8], dq8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { q8[j] = sqd[c8 * 8 + j]; do8[j] = sdo[c8 * 8 + j]; dq8[j] = 0.f; }
    for (int k0 = 0; k0 < len; k0 += 8) {
        const int key = k0 + g;
        if (key < len) {
            const float ds = sds[key], pd = spd[key];
            const uint4 u = *(const uint4*)(kb + (size_t)key * 2 * dm + c8 * 8);
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
            uint32_t ok[4], ov[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dq8[2 * j] += ds * x2f<F16>((bf16_t)(w[j] & 0xFFFFu));
                dq8[2 * j + 1] += ds * x2f<F16>((bf16_t)(w[j] >> 16));
                ok[j] = (uint32_t)f2x<F16>(ds * q8[2 * j]) | ((uint32_t)f2x<F16>(ds * q8[2 * j + 1]) << 16);
                ov[j] = (uint32_t)f2x<F16>(pd * do8[2 * j]) | ((uint32_t)f2x<F16>(pd * do8[2 * j + 1]) << 16);
            }
            *(uint4*)(dkb + (size_t)key * 2 * dm + c8 * 8) = make_uint4(ok[0], ok[1], ok[2], ok[3]);
            *(uint4*)(dkb + (size_t)key * 2 * dm + dm + c8 * 8) = make_uint4(ov[0], ov[1], ov[2], ov[3]);
        }
    }
    __shared__ float sdq[8][64];
#pragma unroll
    for (int j = 0; j < 8; ++j) sdq[g][c8 * 8 + j] = dq8[j];
    __syncthreads();
    float dq = 0.f;
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) dq += sdq[gg][lane];
    dqc[(size_t)seq * dm + hd * 64 + lane] = f2x<F16>(dq);
}

// dst[m * stride_rows] += src[m]  (bf16 rows of d elements)
template <int FMT>      // 0: bf16 += bf16; 1: fp32 += fp32; 2: fp16 dst += fp32 src (fp32 add, one rounding: the fp16 gradient stream)
__global__ __launch_bounds__(256) void add_rows_strided_kernel(void* __restrict__ dst_v, const void* __restrict__ src_v, int M, int d,
                                                                int stride_rows) {
    const int m = blockIdx.x;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        const size_t o = (size_t)m * stride_rows * d + c;
        if (FMT == 1) ((float*)dst_v)[o] += ((const float*)src_v)[(size_t)m * d + c];
        else if (FMT == 2) ((_Float16*)dst_v)[o] = (_Float16)((float)((const _Float16*)dst_v)[o] + ((const float*)src_v)[(size_t)m * d + c]);
        else ((bf16_t*)dst_v)[o] = f2bf(bf2f(((const bf16_t*)dst_v)[o]) + bf2f(((const bf16_t*)src_v)[(size_t)m * d + c]));
    }
}

}  // namespace

// qc: bf16 [nseq, H*64] (CLS queries); kv: bf16 [nseq*L, 2*H*64] = K | V; ctx: bf16 [nseq, H*64]; probs: fp32 [nseq, H, L]
static int attention_cls_fwd_impl(const void* qc, const void* kv, const long long* mask, void* ctx, float* probs, int nseq, int L, int H,
                                  float dropout_p, unsigned long long seed, int io_f16, void* ctx_f16_copy, const int* cu, void* stream);
extern "C" int cldrd_attention_cls_fwd(const void* qc, const void* kv, const long long* mask, void* ctx, float* probs, int nseq, int L,
                                       int H, float dropout_p, unsigned long long seed, int io_f16, void* ctx_f16_copy, void* stream) {
    return attention_cls_fwd_impl(qc, kv, mask, ctx, probs, nseq, L, H, dropout_p, seed, io_f16, ctx_f16_copy, nullptr, stream);
}
// kv PACKED: [Tp, 2*H*64], rows cu_rows[m] .. cu_rows[m + 1] of sequence m (see cldrd_attention_fwd_varlen); probs stays [nseq, H, L]
extern "C" int cldrd_attention_cls_fwd_varlen(const void* qc, const void* kv_packed, const int* cu_rows, void* ctx, float* probs, int nseq, int L,
                                              int H, float dropout_p, unsigned long long seed, int io_f16, void* ctx_f16_copy, void* stream) {
    CLDRD_CHECK(cu_rows != nullptr, "attention_cls_fwd_varlen: cu_rows is required");
    return attention_cls_fwd_impl(qc, kv_packed, nullptr, ctx, probs, nseq, L, H, dropout_p, seed, io_f16, ctx_f16_copy, cu_rows, stream);
}
static int attention_cls_fwd_impl(const void* qc, const void* kv, const long long* mask, void* ctx, float* probs, int nseq, int L, int H,
                                  float dropout_p, unsigned long long seed, int io_f16, void* ctx_f16_copy, const int* cu, void* stream) {
    CLDRD_CHECK(nseq > 0 && L > 0 && L <= 256 && H > 0 && probs != nullptr, "attention_cls_fwd: need 0 < L <= 256 and a probs buffer");
    CLDRD_CHECK((ctx != nullptr || ctx_f16_copy != nullptr) && !(io_f16 && (ctx_f16_copy != nullptr || ctx == nullptr)), "attention_cls_fwd: outputs");
    const uint32_t th = dropout_p > 0.f ? dropout_thresh16(dropout_p) : 0u;
    if (io_f16)
        hipLaunchKernelGGL(attn_cls_fwd_kernel<true>, dim3(nseq * H), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)qc, (const bf16_t*)kv,
                           (const int64_t*)mask, (bf16_t*)ctx, probs, L, H, 0.125f, th, 1.0f / (1.0f - dropout_p), seed_arg(seed), (bf16_t*)nullptr, cu);
    else
        hipLaunchKernelGGL(attn_cls_fwd_kernel<false>, dim3(nseq * H), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)qc, (const bf16_t*)kv,
                           (const int64_t*)mask, (bf16_t*)ctx, probs, L, H, 0.125f, th, 1.0f / (1.0f - dropout_p), seed_arg(seed), (bf16_t*)ctx_f16_copy, cu);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

// dqc: bf16 [nseq, H*64]; dkv: bf16 [nseq*L, 2*H*64] (every row written)
static int attention_cls_bwd_impl(const void* qc, const void* kv, const float* probs, const void* dctx, void* dqc, void* dkv, int nseq, int L,
                                  int H, float dropout_p, unsigned long long seed, int io_f16, const int* cu, void* stream);
extern "C" int cldrd_attention_cls_bwd_x(const void* qc, const void* kv, const float* probs, const void* dctx, void* dqc, void* dkv,
                                         int nseq, int L, int H, float dropout_p, unsigned long long seed, int io_f16, void* stream) {
    return attention_cls_bwd_impl(qc, kv, probs, dctx, dqc, dkv, nseq, L, H, dropout_p, seed, io_f16, nullptr, stream);
}
// kv and dkv PACKED: [Tp, 2*H*64] (every row of dkv written)
extern "C" int cldrd_attention_cls_bwd_varlen(const void* qc, const void* kv_packed, const int* cu_rows, const float* probs, const void* dctx,
                                              void* dqc, void* dkv_packed, int nseq, int L, int H, float dropout_p, unsigned long long seed,
                                              int io_f16, void* stream) {
    CLDRD_CHECK(cu_rows != nullptr, "attention_cls_bwd_varlen: cu_rows is required");
    return attention_cls_bwd_impl(qc, kv_packed, probs, dctx, dqc, dkv_packed, nseq, L, H, dropout_p, seed, io_f16, cu_rows, stream);
}
static int attention_cls_bwd_impl(const void* qc, const void* kv, const float* probs, const void* dctx, void* dqc, void* dkv, int nseq, int L,
                                  int H, float dropout_p, unsigned long long seed, int io_f16, const int* cu, void* stream) {
    CLDRD_CHECK(nseq > 0 && L > 0 && L <= 256 && H > 0, "attention_cls_bwd: need 0 < L <= 256");
    const uint32_t th = dropout_p > 0.f ? dropout_thresh16(dropout_p) : 0u;
    if (io_f16)
        hipLaunchKernelGGL(attn_cls_bwd_kernel<true>, dim3(nseq * H), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)qc, (const bf16_t*)kv, probs,
                           (const bf16_t*)dctx, (bf16_t*)dqc, (bf16_t*)dkv, L, H, 0.125f, th, 1.0f / (1.0f - dropout_p), seed_arg(seed), cu);
    else
        hipLaunchKernelGGL(attn_cls_bwd_kernel<false>, dim3(nseq * H), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)qc, (const bf16_t*)kv, probs,
                           (const bf16_t*)dctx, (bf16_t*)dqc, (bf16_t*)dkv, L, H, 0.125f, th, 1.0f / (1.0f - dropout_p), seed_arg(seed), cu);
    CLDRD_LAUNCH_CHECK();
    return 0;
}
extern "C" int cldrd_attention_cls_bwd(const void* qc, const void* kv, const float* probs, const void* dctx, void* dqc, void* dkv,
                                       int nseq, int L, int H, float dropout_p, unsigned long long seed, void* stream) {
    return cldrd_attention_cls_bwd_x(qc, kv, probs, dctx, dqc, dkv, nseq, L, H, dropout_p, seed, 0, stream);
}

extern "C" int cldrd_add_rows_strided(void* dst, const void* src, int M, int d, int stride_rows, int f32, void* stream) {
    CLDRD_CHECK(M > 0 && d > 0 && stride_rows > 0, "add_rows_strided: bad shape");
    if (f32 == 1) hipLaunchKernelGGL(add_rows_strided_kernel<1>, dim3(M), dim3(256), 0, (hipStream_t)stream, dst, src, M, d, stride_rows);
    else if (f32 == 2) hipLaunchKernelGGL(add_rows_strided_kernel<2>, dim3(M), dim3(256), 0, (hipStream_t)stream, dst, src, M, d, stride_rows);
    else hipLaunchKernelGGL(add_rows_strided_kernel<0>, dim3(M), dim3(256), 0, (hipStream_t)stream, dst, src, M, d, stride_rows);
    CLDRD_LAUNCH_CHECK();
    return 0;
}
