// Exact inner-product top-k over one index shard (reference retriever/retrieval_utils.py:131-153 -> faiss
// IndexFlatIP.search).
//
// faiss runs an fp32 SGEMM over the whole index per 128-query batch; on MI355X fp32 MFMA is 1/16 of the bf16 rate and
// that scan would be matrix-bound.  Here the scan reads a bf16 shadow of the index (half the HBM bytes, bf16 MFMA) and
// keeps only candidates above a per-query threshold (filter epilogue of the NT GEMM, gemm_nt.hip); the candidates are
// re-scored in exact fp32 from the fp32 rows, sorted (score desc, row asc) and cut to k.  Exactness is PROVEN per
// batch on the host: |scan - exact| <= eps_q = 2^-8 |q| max|p|, so if thr_q <= (k-th exact candidate score) - eps_q no
// row outside the candidate list can belong to the top-k (retriever/retrieval_utils.py docstring); otherwise the host
// lowers the threshold and rescans.
//
// Kernels here: per-query k-th largest of the sample scores (threshold estimate, radix select), fp32 re-score,
// bitonic sort + cut, max row norm.
#include "common.h"
#ifndef CLDRD_SCAN_NT
#define CLDRD_SCAN_NT 1
#endif
#include <stdlib.h>

namespace {

__device__ __forceinline__ uint32_t orderable(float f) {      // monotone float -> uint32
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_orderable(uint32_t o) {
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(u);
}

// One digit of the MSB-first radix select: hist[0..256) holds the counts of the current digit among the values that match the
// prefix so far; pick the largest digit b whose suffix count (digits >= b) reaches `remaining`, append it to the prefix and keep
// the rank inside that bin.  Threads 0..255 own one bin each (thread i -> digit 255 - i, so an inclusive scan in thread order is
// the suffix sum from the top); a single thread walking the bins serially cost ~10 us per digit (256 dependent LDS reads).
// Called by the whole block between two of the caller's barriers; has one barrier inside.
__device__ __forceinline__ void radix_pick_digit(const unsigned int* hist, unsigned int* wsum, unsigned int prefix, int shift,
                                                 unsigned int* sel_prefix, unsigned int* sel_remaining) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned int rem = *sel_remaining;            // read by everybody BEFORE the barrier, written by one thread after it
    unsigned int c = 0, incl = 0;
    if (threadIdx.x < 256) {
        c = hist[255 - threadIdx.x];
        incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[wid] = incl;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        for (int w = 0; w < wid; ++w) incl += wsum[w];
        const unsigned int excl = incl - c;
        const bool here = excl < rem && rem <= incl;
        const bool bottom = threadIdx.x == 255 && incl < rem;       // fewer matching values than the rank asked for: lowest digit
        if (here || bottom) {
            *sel_prefix = prefix | ((unsigned int)(255 - threadIdx.x) << shift);
            *sel_remaining = rem - excl;
        }
    }
}

// thr[q] = kth-largest value of scores[q][0..S) (kth is 1-based, clamped to S).  One block per query, MSB-first radix
// select with 8-bit digits, 256-bin LDS histogram per pass.  NR > 0 (S <= NR * 1024: the threshold sample): the row is read ONCE
// into NR registers per thread (independent loads, one L2 latency) and the four passes run on registers; NR = 0: every pass re-reads
// the row (a dependent-latency loop: ~4 us per pass at S = 16384).
template <int NR>
__global__ __launch_bounds__(1024) void kth_largest_kernel(const float* __restrict__ scores, int ld, int S, int kth,
                                                           float* __restrict__ thr) {
    __shared__ unsigned int hist[256], wsum[4];
    __shared__ unsigned int sel_prefix, sel_remaining;
    const float* row = scores + (size_t)blockIdx.x * ld;
    if (threadIdx.x == 0) { sel_prefix = 0; sel_remaining = (unsigned)min(kth, S); }
    constexpr bool REG = NR > 0;
    uint32_t own[REG ? NR : 1];
    if constexpr (REG) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int i = threadIdx.x + 1024 * j;
            own[j] = orderable(row[i < S ? i : 0]);
        }
    }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (threadIdx.x < 256) hist[threadIdx.x] = 0;
        __syncthreads();
        const unsigned int prefix = sel_prefix;
        const unsigned int pmask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        if constexpr (REG) {
#pragma unroll
            for (int j = 0; j < NR; ++j)
                if ((int)threadIdx.x + 1024 * j < S && (own[j] & pmask) == prefix) atomicAdd(&hist[(own[j] >> shift) & 255u], 1u);
        } else {
            for (int i = threadIdx.x; i < S; i += blockDim.x) {
                const uint32_t o = orderable(row[i]);
                if ((o & pmask) == prefix) atomicAdd(&hist[(o >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
        radix_pick_digit(hist, wsum, prefix, shift, &sel_prefix, &sel_remaining);
        __syncthreads();
    }
    if (threadIdx.x == 0) thr[blockIdx.x] = from_orderable(sel_prefix);
}

// exact re-score: one wave per candidate, fixed summation order (lane-strided partial sums, then a butterfly), accumulated in fp64 and
// rounded to fp32 ONCE - the oracle's (and the retrieval contract's) definition of a score, whatever the data.  (Until round 3 the sums
// were fp32 FMAs: fine for embeddings, but on rows with heavy-tailed norms a low-ranked score of 2.3 next to |q||p| ~ 1e4 came out 1.4e-4
// off, tools/search_fuzz.py.  768 fp64 FMAs per candidate are nothing next to fetching its 3-KiB row.)
__global__ __launch_bounds__(256) void rescore_kernel(const float* __restrict__ q, const float* __restrict__ P, int d,
                                                       const int* __restrict__ counts, const int* __restrict__ cand_rows,
                                                       float* __restrict__ cand_scores, int cap) {
    const int qi = blockIdx.y;
    const int n = min(counts[qi], cap);
    const int lane = threadIdx.x & 63;
    const float* qr = q + (size_t)qi * d;
    for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < n; c += gridDim.x * 4) {
        const float* pr = P + (size_t)cand_rows[(size_t)qi * cap + c] * d;
        double s = 0.0;
        for (int j = lane * 4; j < d; j += 256) {
            const float4 a = *(const float4*)(qr + j);
#if CLDRD_SCAN_NT
            const uint4 bu = ld16_stream(pr + j);              // an index row is read once per pass: keep it out of the caches
            const float4 b = make_float4(__uint_as_float(bu.x), __uint_as_float(bu.y), __uint_as_float(bu.z), __uint_as_float(bu.w));
#else
            const float4 b = *(const float4*)(pr + j);
#endif
            s = fma((double)a.x, (double)b.x, s); s = fma((double)a.y, (double)b.y, s);
            s = fma((double)a.z, (double)b.z, s); s = fma((double)a.w, (double)b.w, s);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) cand_scores[(size_t)qi * cap + c] = (float)s;
    }
}

// per query: sort candidates by (score desc, row asc) in LDS, write the best k (missing: row -1, score -inf).
// key = (~orderable(score)) << 32 | row, ascending.  NP = cap rounded up to a power of two (<= 8192).
__global__ __launch_bounds__(1024) void topk_sort_kernel(const int* __restrict__ counts, const int* __restrict__ cand_rows,
                                                         const float* __restrict__ cand_scores, int cap, int NP, int k,
                                                         float* __restrict__ D, int* __restrict__ I) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const int qi = blockIdx.x;
    const int n = min(counts[qi], cap);
    for (int i = threadIdx.x; i < NP; i += blockDim.x) {
        unsigned long long key = ~0ull;
        if (i < n) {
            const uint32_t o = ~orderable(cand_scores[(size_t)qi * cap + i]);
            key = ((unsigned long long)o << 32) | (uint32_t)cand_rows[(size_t)qi * cap + i];
        }
        keys[i] = key;
    }
    __syncthreads();
    // Bitonic network.  Pair t exchanges keys (t / stride) 2 stride + t % stride and + stride: the 64 pairs of one wave (t = 64 w ..
    // 64 w + 63, then + blockDim.x) stay inside ONE 128-key chunk while stride <= 64, so those steps (56 of the 66 at NP = 2048) only
    // need the wave's own LDS accesses in order - a compiler fence, no workgroup barrier.  The barrier comes back around every step
    // with stride >= 128.
    bool wide_before = true;
    for (int size = 2; size <= NP; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const bool wide = stride >= 128;
            if (wide || wide_before) __syncthreads();
            else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
            wide_before = wide;
            for (int t = threadIdx.x; t < NP / 2; t += blockDim.x) {
                const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1)), hi = lo + stride;      // stride is a power of two
                const bool up = ((lo & size) == 0);
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const unsigned long long key = i < NP ? keys[i] : ~0ull;
        const bool ok = i < n;
        D[(size_t)qi * k + i] = ok ? from_orderable(~(uint32_t)(key >> 32)) : -__builtin_inff();
        I[(size_t)qi * k + i] = ok ? (int)(uint32_t)key : -1;
    }
}

__global__ __launch_bounds__(256) void row_norm_max_kernel(const float* __restrict__ P, size_t rows, int d, unsigned int* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    float best = 0.f;
    for (size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (size_t)gridDim.x * 4) {
        const float* pr = P + r * d;
        float s = 0.f;
        for (int j = lane * 4; j < d; j += 256) {
            const float4 a = *(const float4*)(pr + j);
            s += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
        }
        best = fmaxf(best, wave_sum(s));
    }
    if (lane == 0) atomicMax(out, __float_as_uint(best));       // non-negative floats order like their bit patterns
}

// strided row gather + cast: dst[i] = bf16(src[i * stride]) for the threshold sample
__global__ __launch_bounds__(256) void gather_cast_rows_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n_out,
                                                                size_t stride, int d) {
    const int lane = threadIdx.x & 63;
    for (size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n_out; r += (size_t)gridDim.x * 4) {
        const float* s = src + r * stride * d;
        bf16_t* o = dst + r * d;
        for (int j = lane * 4; j < d; j += 256) {
            const float4 a = *(const float4*)(s + j);
            uint2 u; u.x = pack2bf(a.x, a.y); u.y = pack2bf(a.z, a.w);
            *(uint2*)(o + j) = u;
        }
    }
}



// ---------------------------------------------------------------------------------------------------------------
// fp16 shadow of the index / the queries.  fp16 (11-bit significand) instead of bf16 (8): |scan - exact| shrinks 8x, and with it
// the band of candidates that must be re-scored in fp32 (retrieval_utils.py docstring).  flag <- 1 when a value does not fit
// (|x| > 65504 or NaN): the caller refuses the index / the query instead of scanning infinities.
__device__ __forceinline__ uint32_t pack2h(float lo, float hi) {
    const _Float16 a = (_Float16)lo, b = (_Float16)hi;
    return (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
}
__global__ __launch_bounds__(256) void cast_f16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, size_t n4,
                                                        unsigned int* __restrict__ flag) {
    bool bad = false;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 a = ((const float4*)src)[i];
        bad |= !(fabsf(a.x) <= 65504.f) | !(fabsf(a.y) <= 65504.f) | !(fabsf(a.z) <= 65504.f) | !(fabsf(a.w) <= 65504.f);
        uint2 u; u.x = pack2h(a.x, a.y); u.y = pack2h(a.z, a.w);
        ((uint2*)dst)[i] = u;
    }
    if (bad && flag) atomicOr(flag, 1u);
}

// ---- attaching a shard (FlatIPIndex._attach): the statistics and shadows of the index in three launches over the fp32 rows ----
// (until round 5 these were chunks of at::native kernels: mean, subtract, double-precision row norms, casts, a strided gather)
// 1. column sums in fp64: block b walks rows b, b + grid, ...; thread t owns four adjacent columns (a row is read as one coalesced segment);
//    partial[b][d] doubles, reduced in block order by index_mean_finish_kernel -> mu = mean row (fp32), deterministic.
__global__ __launch_bounds__(256) void index_colsum_kernel(const float* __restrict__ P, size_t rows, int d, double* __restrict__ partial) {
    // thread t owns the 4 columns 4 t .. 4 t + 3 (+ 1024 per further group): one 16-byte load per row and group (the first version read 4 bytes
    // per lane: 2.2 TB/s, profiles/r06_retrieve_summary.txt); d % 4 == 0, d <= 2048
    constexpr int MAXG = 2;
    double acc[MAXG][4];
#pragma unroll
    for (int g = 0; g < MAXG; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[g][j] = 0.0;
    for (size_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const float* pr = P + r * d;
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            const int c = 4 * threadIdx.x + 1024 * g;
            if (c < d) {
                const float4 v = *(const float4*)(pr + c);
                acc[g][0] += (double)v.x; acc[g][1] += (double)v.y; acc[g][2] += (double)v.z; acc[g][3] += (double)v.w;
            }
        }
    }
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
        const int c = 4 * threadIdx.x + 1024 * g;
        if (c < d) {
#pragma unroll
            for (int j = 0; j < 4; ++j) partial[(size_t)blockIdx.x * d + c + j] = acc[g][j];
        }
    }
}
__global__ __launch_bounds__(256) void index_mean_finish_kernel(const double* __restrict__ partial, int nblk, int d, size_t rows, float* __restrict__ mu) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= d) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += partial[(size_t)b * d + j];
    mu[j] = (float)(s / (double)rows);
}
// 2. one pass: c = p - mu (fp32, as the scan's operand is defined), fp16 shadow of c, max_r |c_r|^2 (fp64 sum per row, the maximum as the
//    bit pattern of its fp32 value: non-negative floats order like their bits), the bf16 threshold sample (row r = i * stride for
//    i < s_rows), range flag.  One wave per row, 16-byte loads, 8-byte stores.
__global__ __launch_bounds__(256) void index_center_cast_kernel(const float* __restrict__ P, const float* __restrict__ mu, size_t rows, int d,
                                                                 uint16_t* __restrict__ P16, bf16_t* __restrict__ sample, size_t s_stride,
                                                                 size_t s_rows, unsigned int* __restrict__ cmax_bits, unsigned int* __restrict__ flag) {
    const int lane = threadIdx.x & 63;
    double best = 0.0;
    bool bad = false;
    for (size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (size_t)gridDim.x * 4) {
        const float* pr = P + r * d;
        uint16_t* o = P16 + r * d;
        const bool in_sample = sample != nullptr && r % s_stride == 0 && r / s_stride < s_rows;
        bf16_t* so = in_sample ? sample + (r / s_stride) * d : nullptr;
        double s = 0.0;
        for (int j = lane * 4; j < d; j += 256) {
            const float4 a = *(const float4*)(pr + j);
            const float4 m = *(const float4*)(mu + j);
            const float c0 = a.x - m.x, c1 = a.y - m.y, c2 = a.z - m.z, c3 = a.w - m.w;
            s += (double)c0 * (double)c0 + (double)c1 * (double)c1 + (double)c2 * (double)c2 + (double)c3 * (double)c3;
            bad |= !(fabsf(c0) <= 65504.f) | !(fabsf(c1) <= 65504.f) | !(fabsf(c2) <= 65504.f) | !(fabsf(c3) <= 65504.f);
            uint2 u; u.x = pack2h(c0, c1); u.y = pack2h(c2, c3);
            *(uint2*)(o + j) = u;
            if (in_sample) { uint2 v; v.x = pack2bf(c0, c1); v.y = pack2bf(c2, c3); *(uint2*)(so + j) = v; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        best = s > best ? s : best;
    }
    if (lane == 0 && best > 0.0) atomicMax(cmax_bits, __float_as_uint((float)best));
    if (bad && flag) atomicOr(flag, 1u);
}
// 3. IndexIDMap on the device: row position -> id (ids table, or position + id_offset), -1 stays -1
__global__ __launch_bounds__(256) void map_ids_kernel(const int* __restrict__ I, const long long* __restrict__ table, long long id_offset,
                                                       long long* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int r = I[i];
        out[i] = r < 0 ? -1ll : (table ? table[r] : (long long)r + id_offset);
    }
}

// one block per query: fp16 + bf16 copies, L2 norm (fp32, fixed order), range flag
__global__ __launch_bounds__(256) void prep_queries_kernel(const float* __restrict__ q, uint16_t* __restrict__ qh, bf16_t* __restrict__ qb,
                                                            float* __restrict__ qnorm, int d, unsigned int* __restrict__ flag) {
    __shared__ float red[4];
    const float* row = q + (size_t)blockIdx.x * d;
    float s = 0.f;
    bool bad = false;
    for (int j = threadIdx.x * 4; j < d; j += 1024) {
        const float4 a = *(const float4*)(row + j);
        s += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
        bad |= !(fabsf(a.x) <= 65504.f) | !(fabsf(a.y) <= 65504.f) | !(fabsf(a.z) <= 65504.f) | !(fabsf(a.w) <= 65504.f);
        uint2 u; u.x = pack2h(a.x, a.y); u.y = pack2h(a.z, a.w);
        *(uint2*)(qh + (size_t)blockIdx.x * d + j) = u;
        uint2 v; v.x = pack2bf(a.x, a.y); v.y = pack2bf(a.z, a.w);
        *(uint2*)(qb + (size_t)blockIdx.x * d + j) = v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) qnorm[blockIdx.x] = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    if (bad && flag) atomicOr(flag, 1u);
}

// eps[q] >= |fp16 scan score - fp32 exact score| for every index row (pmax = largest row norm, both operands rounded to fp16):
//   rounding of q and p (RNE, u = 2^-11 each, Cauchy-Schwarz)            |q| pmax (2^-10 + 2^-22)
//   fp32 accumulation: MFMA chain (d adds, 2^-23 each if the adder truncates) + re-score (d fmas, 2^-24 each):  |q| pmax d 2^-22
//   values below the fp16 normal range (flushed or rounded, <= 2^-14)    2^-14 sqrt(d) (|q| + pmax) + d 2^-28
// thr[q] = est[q] - 2 eps[q]: the scan keeps a row when its fp16 score reaches thr (cldrd_topk_select explains the 2).
__global__ void thresholds_kernel(const float* __restrict__ est, const float* __restrict__ qnorm, float pmax, int d, float* __restrict__ thr,
                                  float* __restrict__ eps, int nq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const double qn = qnorm[i], pm = pmax, dd = d;
    double e = qn * pm * (0x1p-10 + 0x1p-22 + dd * 0x1p-22) + 0x1p-14 * sqrt(dd) * (qn + pm) + dd * 0x1p-28;
    const float ef = (float)(e * (1.0 + 1e-6)) + 1e-30f;
    eps[i] = ef;
    if (est) thr[i] = est[i] - 2.0f * ef;
}

// exhaustive mode (rows <= cap): every row is a candidate of every query
__global__ void all_candidates_kernel(int* __restrict__ counts, int* __restrict__ cand_rows, float* __restrict__ cand_scores, int rows, int cap) {
    const int q = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += gridDim.x * blockDim.x) {
        cand_rows[(size_t)q * cap + i] = i;
        cand_scores[(size_t)q * cap + i] = 0.f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[q] = rows;
}

// Per query, on the fp16 scan scores s^ of its candidate list (c = counts[q] entries, all rows with s^ >= thr[q]):
//   t^ = the kk-th largest s^  (the true kk-th largest over the whole shard as soon as c >= kk);
//   keep the candidates with s^ >= t^ - 2 eps.  Why that is enough: a row with s^ < t^ - 2 eps has exact score
//   s <= s^ + eps < t^ - eps, while each of the kk rows with s^ >= t^ has s >= t^ - eps: kk rows beat it strictly.
//   Every row with s^ >= t^ - 2 eps is IN the list when thr <= t^ - 2 eps; that, c >= kk, no overflow and no dropped hit make
//   the query PROVEN: the exact top-kk is inside the kept set, which the next two kernels re-score in fp32 and sort.
// status[q]: 0 proven | 1 fewer than kk candidates | 2 list overflow (c > cap) | 4 hits dropped by the scan | 8 threshold above
// t^ - 2 eps | 16 kept set larger than cap2.  khat[q] = t^ (or -inf).  exhaustive != 0: the list holds every row; keep all.
__global__ __launch_bounds__(1024) void select_compact_kernel(const int* __restrict__ counts, const int* __restrict__ dropped,
                                                               const int* __restrict__ cand_rows, const float* __restrict__ cand_scores, int cap,
                                                               int kk, const float* __restrict__ thr, const float* __restrict__ eps,
                                                               int* __restrict__ rows2, int cap2, int* __restrict__ n2, int* __restrict__ status,
                                                               float* __restrict__ khat, int exhaustive) {
    extern __shared__ __attribute__((aligned(16))) unsigned int sc[];      // cap orderable scores
    __shared__ unsigned int hist[256], wsum[4];
    __shared__ unsigned int sel_prefix, sel_remaining, nkeep;
    const int q = blockIdx.x;
    const int c = counts[q];
    const int n = min(c, cap);
    for (int i = threadIdx.x; i < n; i += blockDim.x) sc[i] = orderable(cand_scores[(size_t)q * cap + i]);
    if (threadIdx.x == 0) { sel_prefix = 0; sel_remaining = (unsigned)kk; nkeep = 0; }
    __syncthreads();
    int st = 0;
    float cut = -__builtin_inff(), t_hat = -__builtin_inff();
    if (!exhaustive) {
        if (c < kk) st |= 1;
        if (c > cap) st |= 2;
        if (*dropped != 0) st |= 4;
        if (n >= kk) {
            for (int pass = 0; pass < 4; ++pass) {
                const int shift = 24 - 8 * pass;
                if (threadIdx.x < 256) hist[threadIdx.x] = 0;
                __syncthreads();
                const unsigned int prefix = sel_prefix;
                const unsigned int pmask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
                for (int i = threadIdx.x; i < n; i += blockDim.x) {
                    const uint32_t o = sc[i];
                    if ((o & pmask) == prefix) atomicAdd(&hist[(o >> shift) & 255u], 1u);
                }
                __syncthreads();
                radix_pick_digit(hist, wsum, prefix, shift, &sel_prefix, &sel_remaining);
                __syncthreads();
            }
            t_hat = from_orderable(sel_prefix);
            // rounded DOWN: with |t^| >> eps the fp32 rounding of t^ - 2 eps (half an ulp of t^) can exceed the 1e-6 relative slack
            // inside eps, and a row with scan score in [t^ - 2 eps, cut) would fall outside the proven band
            cut = from_orderable(orderable((float)((double)t_hat - 2.0 * (double)eps[q])) - 1u);       // the next float below
            if (!(thr[q] <= cut)) st |= 8;
        }
    }
    const unsigned int ocut = orderable(cut);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        if (sc[i] >= ocut) {
            const unsigned int pos = atomicAdd(&nkeep, 1u);
            if (pos < (unsigned)cap2) rows2[(size_t)q * cap2 + pos] = cand_rows[(size_t)q * cap + i];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (nkeep > (unsigned)cap2) st |= 16;
        n2[q] = (int)min(nkeep, (unsigned)cap2);
        status[q] = st;
        khat[q] = t_hat;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Streaming scan: the HBM-bound part of the search.  The 128 queries of a batch stay in REGISTERS (wave w keeps the
// MFMA B fragments of queries 16w..16w+15 for the whole K range: KS*4 VGPRs), the index streams through a 3-slot LDS
// ring in tiles of 32 whole rows = 32*d*2 bytes of CONTIGUOUS memory per tile (full-page DRAM bursts; the tiled GEMM
// fetched 128-B slivers of 256 different rows per step and topped out at ~2.3 TB/s).  Persistent: one workgroup per
// CU walks tiles b, b+grid, ...; two tiles (96 KiB at d = 768) stay in flight per CU behind a counted vmcnt.
// Hits (score >= thr[query], ~0.2 % of the scores) go to small LDS lists, ONE PER WAVE: the list length is a wave-uniform
// register and a hit's slot comes from ballot + mbcnt, so an append is three ds_writes and no LDS atomic (a shared list cost
// one ds_add_rtn round trip per hit row: ~400 cycles per 16 x 16 score block with a hit, 17 % of the pass at ~3000 candidates
// per query).  The lists are flushed to the per-query candidate lists with global atomics only when one of them fills up or
// at the end, so the DMA queue is never drained in the loop.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// NW waves x NQS sets of 16 queries each = NQ queries per pass: <8, 1> = the reference's 128-query batch (2 waves per SIMD, 96 of
// 256 VGPRs hold queries); <4, 4> = 256 queries per pass (one wave per SIMD, 384 of its 512 VGPRs hold queries): the index bytes are
// read once per 256 queries instead of once per 128, and with 4 MFMAs per A fragment the pass is still HBM-bound at d = 768.
// The same MFMA with the query fragment pinned in an AGPR quad.  With 384 query registers per lane hipcc keeps most of them in
// AGPRs and copies each fragment to VGPRs before every use (416 v_accvgpr_read per tile: as much issue time as the MFMAs); MFMA
// can read its B operand from AGPRs directly, which the "a" constraint forces.  Inline asm is opaque to hipcc's hazard
// recognizer: the caller separates the last MFMA of a chain from the first VALU read of its result with mfma_result_fence().
template <bool F16>
__device__ __forceinline__ f32x4 mfma16_bq(bf16x8 a, bf16x8 bq_agpr, f32x4 c) {
    if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(bq_agpr));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(bq_agpr));
    return c;
}
__device__ __forceinline__ void mfma_result_fence() { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory"); }

template <int KS, int ablate, bool F16, int NW = 8, int NQS = 1>     // d = 32 * KS; ablate != 0: timing experiments only (tools/scan_bench.py); F16: fp16 shadow
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 1) void scan_stream_kernel(const bf16_t* __restrict__ P, const bf16_t* __restrict__ Q, int nq,
                                                              long long rows, const float* __restrict__ thr,
                                                              int* __restrict__ counts, int* __restrict__ cand_rows,
                                                              float* __restrict__ cand_scores, int cap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int R = 32, ROWB = KS * 64, TILEB = R * ROWB, PIECES = TILEB / 1024, PPW = PIECES / NW, NSLOT = 3;
    constexpr int NQ = 16 * NQS * NW;          // queries per pass
    constexpr int LCAP = (160 * 1024 - NSLOT * TILEB - 128 - 8 * NQ) / 12 < 4096 ? (160 * 1024 - NSLOT * TILEB - 128 - 8 * NQ) / 12 : 4096;   // LDS hit lists
    constexpr int WCAP = LCAP / NW;            // slots of one wave's list: entries [wid WCAP, (wid + 1) WCAP) of lq / lrow / lscore
    static_assert(PIECES % NW == 0 && NQ <= 256 && NW <= 8, "tile pieces must divide over the waves; the hit list packs the query in 8 bits");
    int* lcount = (int*)(smem + NSLOT * TILEB);   // [0,8) list lengths at the last flush call, [8,32) three rotating snapshots of them
    int* qcnt = lcount + 32;                   // [0,NQ) per-query hit counts, [NQ,2NQ) global bases (flush scratch)
    int* lq = qcnt + 2 * NQ;
    int* lrow = lq + LCAP;
    float* lscore = (float*)(lrow + LCAP);
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long ntiles = (rows + R - 1) / R;

    // ---- this wave's 16 NQS queries as B fragments for every k-step (registers, loaded once): 4 KS NQS registers.
    // Fragment f = s KS + ks lives in an AGPR quad for f < NA (all 256 AGPRs when NQS = 4: sets 0, 1 and two thirds of set 2) and in
    // VGPRs otherwise; one wave per SIMD owns all 512 registers of its lanes then (see mfma16_bq for why the split is by hand).
    constexpr int NA = (NQS > 1 && NW == 4) ? (NQS * KS < 64 ? NQS * KS : 64) : 0, NV = NQS * KS - NA;
    const int qn0 = wid * 16 * NQS + (lane & 15);          // query of set s: qn0 + 16 s
    bf16x8 bqa[NA > 0 ? NA : 1], bqv[NV > 0 ? NV : 1];
    float thr_lane[NQS];
#pragma unroll
    for (int s = 0; s < NQS; ++s) {
        const int qn = qn0 + 16 * s;
        const bf16_t* qp = Q + (size_t)min(qn, nq - 1) * (KS * 32) + 8 * (lane >> 4);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int f = s * KS + ks;
            const bf16x8 v = *(const bf16x8*)(qp + ks * 32);
            if (f < NA) bqa[f < NA ? f : 0] = v; else bqv[f >= NA ? f - NA : 0] = v;
        }
        thr_lane[s] = qn < nq ? thr[qn] : __builtin_inff();
    }
    // wait for the loads here, not inside the tile loop (and consume the thresholds: otherwise hipcc waits vmcnt(0) at their first use)
#pragma unroll
    for (int f = 0; f < NA; ++f) asm volatile("" : "+a"(bqa[f]));
#pragma unroll
    for (int f = 0; f < NV; ++f) asm volatile("" : "+v"(bqv[f]));
#pragma unroll
    for (int s = 0; s < NQS; ++s) asm volatile("" : "+v"(thr_lane[s]));
    const uint32_t lcount_off = (uint32_t)(uintptr_t)LDS_PTR(lcount), lq_off = (uint32_t)(uintptr_t)LDS_PTR(lq);
    static_assert(8 * LCAP < 65536, "ds_write offset field");
    if (threadIdx.x < 32) lcount[threadIdx.x] = 0;
    int wcnt = 0;                                               // hits this wave has appended since the last flush (wave-uniform)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // nothing but LDS-DMA is in flight from here on
    __syncthreads();                                            // the first snapshot read comes BEFORE the first barrier of the tile loop

    // ---- DMA addressing: piece p = wid*PPW + j covers LDS bytes [1024 p, 1024 p + 1024) of the tile image
    uint32_t soff[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int byte = (wid * PPW + j) * 1024 + lane * 16;
        const int row = byte / ROWB, cl = (byte % ROWB) / 16;
        soff[j] = (uint32_t)(row * ROWB + ((cl ^ (row & 15)) * 16));      // XOR on the source side, LDS stays linear
    }
    // saddr form, hand-issued: scalar 64-bit tile base + one 32-bit lane offset per piece (through the builtin every piece cost a
    // 64-bit VALU add and a register pair, and the <8 waves x 32 queries> instance spilled inside the tile loop).  hipcc does not
    // count these in its vmcnt bookkeeping; the waits below are written by hand anyway.
    const long long last = rows * ROWB - 16;                              // clamp the tail tile inside the allocation
    const uint32_t smem_off = (uint32_t)(uintptr_t)LDS_PTR(smem);
    auto stage = [&](int slot, long long tile) {
        const uint32_t dst = smem_off + (uint32_t)(slot * TILEB + wid * PPW * 1024);
        const char* sbase = (const char*)P + tile * (long long)TILEB;
        const long long room = last - tile * (long long)TILEB;
        const uint32_t lim = room < (long long)TILEB ? (uint32_t)room : 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < PPW; ++j)
#if CLDRD_SCAN_NT
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(dst + j * 1024), "v"(min(soff[j], lim)), "s"(sbase) : "memory", "m0");
#else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst + j * 1024), "v"(min(soff[j], lim)), "s"(sbase) : "memory", "m0");
#endif
    };

    uint32_t rd_off[4];
#pragma unroll
    for (int kr = 0; kr < 4; ++kr)
        rd_off[kr] = (lane & 15) * ROWB + ((((kr << 2) ^ (lane & 12)) | ((lane >> 4) ^ (lane & 3))) * 16);
    // Move the LDS hit list to the per-query candidate lists: one global atomic per (block, query) reserves the block's range in
    // that query's list (per-hit atomics on 128 addresses serialise in L2).  Called by the whole workgroup (barriers inside).
    auto flush = [&]() {
        if (ablate == 4) { wcnt = 0; return; }                                                          // timing experiment: hits vanish
        for (int i = threadIdx.x; i < 2 * NQ; i += blockDim.x) qcnt[i] = 0;
        if (lane == 0) lcount[wid] = wcnt;
        __syncthreads();
        auto live = [&](int e) { const int w = e / WCAP; return w < NW && e - w * WCAP < min(lcount[w], WCAP); };
        for (int e = threadIdx.x; e < NW * WCAP; e += blockDim.x)
            if (live(e)) lq[e] |= atomicAdd(qcnt + lq[e], 1) << 8;                                      // rank inside the block
        __syncthreads();
        for (int i = threadIdx.x; i < NQ; i += blockDim.x)
            if (qcnt[i] > 0) qcnt[NQ + i] = ablate == 3 ? 0 : atomicAdd(counts + i, qcnt[i]);         // ablate 3: no global atomics (timing only)
        __syncthreads();
        for (int e = threadIdx.x; e < NW * WCAP; e += blockDim.x) {
            if (!live(e)) continue;
            const int q = lq[e] & 255;
            const int pos = qcnt[NQ + q] + (lq[e] >> 8);
            if (pos < cap) { cand_rows[(size_t)q * cap + pos] = lrow[e]; cand_scores[(size_t)q * cap + pos] = lscore[e]; }
        }
        if (threadIdx.x < NW && lcount[threadIdx.x] > WCAP) atomicAdd(counts + nq, lcount[threadIdx.x] - WCAP);   // counts[nq] = hits dropped (the caller rescans)
        if (threadIdx.x >= 64 && threadIdx.x < 64 + 24) lcount[8 + threadIdx.x - 64] = 0;      // the snapshots in flight describe the lists just emptied
        __syncthreads();
        wcnt = 0;
    };
    const long long t0 = blockIdx.x, step = gridDim.x;
    if (t0 < ntiles) stage(0, t0);
    if (t0 + step < ntiles) stage(1, t0 + step);
    int cs = 0, it = 0;
    for (long long t = t0; t < ntiles; t += step, ++it) {
        // Level of the hit list for the flush decision below.  The decision must be workgroup-uniform (flush() has barriers) but
        // lcount moves as soon as a fast wave emits hits of the current tile, so everybody reads a SNAPSHOT wave 0 took at the end
        // of tile it-2 (three rotating slots; that write is ordered by the barrier of tile it-1, and wave 0 cannot overwrite the slot
        // before the end of tile it+1, i.e. after every wave has passed the next barrier).  The read sits in the SAME asm statement
        // as the wait that lands it: issued separately, hipcc copied the destination register before the wait (stale value in the
        // last tile -> waves disagreeing on the flush -> mismatched barriers; caught by tools/scan_debug.py on 625-tile shards).
        // tile t must have landed; tile t+step may stay in flight.  NO global memory operation other than the DMA may appear
        // inside this loop: hipcc would put s_waitcnt vmcnt(0) next to it and drain the two tiles in flight (measured: ~1 us per hit)
        uint32_t snap = 0;
        const uint32_t snap_addr = lcount_off + 4u * (8u + 8u * (uint32_t)((it + 1) % 3) + (uint32_t)(lane & 7));      // lane l: wave l & 7
        if (t + step < ntiles) asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt vmcnt(%2) lgkmcnt(0)" : "=&v"(snap) : "v"(snap_addr), "n"(PPW) : "memory");
        else asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" : "=&v"(snap) : "v"(snap_addr) : "memory");
        __builtin_amdgcn_s_barrier();                                       // ... for every wave; slot of tile t-step is free
        asm volatile("" ::: "memory");
        // Hit-list level check: `snap` was read at the top of this iteration (see there); the decision is workgroup-uniform.
        if ((ablate == 0 || ablate >= 3) && __builtin_amdgcn_ballot_w64(snap > (uint32_t)(WCAP / 2)) != 0) flush();      // some wave's list is half full (rare)
        const int fs = cs == 0 ? NSLOT - 1 : cs - 1;                        // slot of tile t-step
        const char* sb = smem + cs * TILEB;
        cs = cs == NSLOT - 1 ? 0 : cs + 1;
        if (ablate == 1) {                                                  // timing experiments only: DMA stream alone
            if (t + 2 * step < ntiles) stage(fs, t + 2 * step);
            continue;
        }
        // Two half-row chunks of A fragments are kept in flight ahead of the MFMAs that consume them: with 2 waves per SIMD
        // nothing else hides the LDS latency (measured: reads alone and MFMAs alone both keep up with the DMA stream,
        // read -> wait -> MFMA in one chain does not).  sched_barrier pins the order, hipcc still counts the lgkmcnt waits.
        // chunk = CH k-steps of one 16-row half (smaller chunks when most registers hold queries)
        constexpr int CH = NQS > 1 ? (NW == 8 ? 2 : (KS % 4 == 0 ? 4 : 2)) : (KS % 3 == 0 ? KS / 3 : KS / 2), NCM = KS / CH, NC = 2 * NCM;
        bf16x8 ab[2][CH];
        // chunk position of k-step ks in row r: ((4 ks + (lane >> 4)) ^ (r & 15)) = 16 (ks >> 2) + (((ks & 3) << 2) ^ (r & 12) | (lane >> 4) ^ (r & 3)):
        // four lane-dependent bases, everything else is an immediate offset of the ds_read
        const char* pk[4];
#pragma unroll
        for (int kr = 0; kr < 4; ++kr) pk[kr] = sb + rd_off[kr];
        auto fetch = [&](bf16x8(&a)[CH], int c) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int ks = (c % NCM) * CH + i;
                a[i] = *(const bf16x8*)(pk[ks & 3] + (c / NCM) * 16 * ROWB + (ks >> 2) * 256);
            }
        };
        auto emit = [&](f32x4 acc, int mt, int qn, float thr_q) {        // acc[j] = <P[row], Q[qn]> with row = 32 t + 16 mt + 4 (lane >> 4) + j
            if (ablate == 2) { asm volatile("" ::"v"(acc)); return; }
            const int row0 = (int)(t * R) + mt * 16 + 4 * (lane >> 4);
            // one test for the four rows of every lane first: no hit in the whole 16 x 16 block is the common case.  Everything
            // below branches on ballots only, so the control flow - and with it wcnt - stays wave-uniform.
            if (__builtin_amdgcn_ballot_w64(fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])) >= thr_q) == 0) return;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool hit = acc[j] >= thr_q && row0 + j < (int)rows;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
                if (m == 0) continue;
                const int pos = wcnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (hit && pos < WCAP) {
                    // hand-issued LDS stores: through plain stores hipcc orders these LDS accesses after the LDS-DMA in flight
                    // (s_waitcnt vmcnt(0) per hit), which stalls the stream
                    const uint32_t a = lq_off + 4u * (uint32_t)(wid * WCAP + pos);
                    asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:%4\n\tds_write_b32 %0, %3 offset:%5"
                                 ::"v"(a), "v"(qn), "v"(row0 + j), "v"(acc[j]), "n"(4 * LCAP), "n"(8 * LCAP) : "memory");
                } else if (hit) {
                    // This wave's on-chip list is full (the flush decision lags two tiles; a burst of hits - rows that score high for EVERY
                    // query of the tile, the normal case on real dual-encoder embeddings - can outrun it): the hit goes straight to the
                    // query's global list, as in the tiled kernels.  Slow (a returning global atomic: hipcc drains the LDS-DMA queue for
                    // it), rare, and nothing is ever dropped: until round 4 such hits were counted in counts[nq] and the whole pass was
                    // scanned again by the tiled kernels (every pass of the CLS-like shard: 27 rescans for 28 passes).
                    const int gp = atomicAdd(counts + qn, 1);
                    if (gp < cap) { cand_rows[(size_t)qn * cap + gp] = row0 + j; cand_scores[(size_t)qn * cap + gp] = acc[j]; }
                }
                wcnt = min(wcnt + (int)__builtin_popcountll(m), WCAP);
            }
        };
        fetch(ab[0], 0);
        fetch(ab[1], 1);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 * step < ntiles) stage(fs, t + 2 * step);                 // issued under the latency of the first fragment reads
        f32x4 acc[NQS];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            __builtin_amdgcn_sched_barrier(0);
            if (c % NCM == 0) {
#pragma unroll
                for (int s = 0; s < NQS; ++s) acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < CH; ++i)
#pragma unroll
                for (int s = 0; s < NQS; ++s) {        // NQS independent chains
                    const int f = s * KS + (c % NCM) * CH + i;
                    if (f < NA) {
                        acc[s] = mfma16_bq<F16>(ab[c & 1][i], bqa[f < NA ? f : 0], acc[s]);
                    } else {
                        // a chain that changes from the asm form to the builtin form: give the opaque MFMA's result its wait states
                        if (NA > 0 && f == NA && f % KS != 0) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[s]));
                        acc[s] = mfma16<F16>(ab[c & 1][i], bqv[f >= NA ? f - NA : 0], acc[s]);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
            if (c + 2 < NC) fetch(ab[c & 1], c + 2);
            if (c % NCM == NCM - 1) {
                if constexpr (NA > 0) mfma_result_fence();
#pragma unroll
                for (int s = 0; s < NQS; ++s) emit(acc[s], c / NCM, qn0 + 16 * s, thr_lane[s]);
            }
        }
        if (ablate == 0 || ablate >= 3) {           // snapshot of this wave's list length for the check two tiles on (hand-issued: see emit)
            const uint32_t a = lcount_off + 4u * (8u + 8u * (uint32_t)(it % 3) + (uint32_t)wid);
            if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(wcnt) : "memory");
        }
    }
    // ---- what is left in the list
    __syncthreads();
    flush();
}

template <int KS, int ABL, bool F16, int NW = 8, int NQS = 1>
int launch_scan_stream_abl(const void* Q, const void* P, int nq, long long rows, const float* thr, int* counts, int* cand_rows,
                           float* cand_scores, int cap, hipStream_t st) {
    constexpr int tb = 3 * 32 * KS * 64, NQ = 16 * NQS * NW;
    constexpr int lcap = (160 * 1024 - tb - 128 - 8 * NQ) / 12 < 4096 ? (160 * 1024 - tb - 128 - 8 * NQ) / 12 : 4096;
    constexpr int lds = tb + 128 + 8 * NQ + lcap * 12;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)scan_stream_kernel<KS, ABL, F16, NW, NQS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const long long ntiles = (rows + 31) / 32;
    const int grid = (int)(ntiles < 256 ? ntiles : 256);
    hipLaunchKernelGGL((scan_stream_kernel<KS, ABL, F16, NW, NQS>), dim3(grid), dim3(64 * NW), lds, st, (const bf16_t*)P, (const bf16_t*)Q, nq, rows, thr,
                       counts, cand_rows, cand_scores, cap);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

template <int KS>
int launch_scan_stream(const void* Q, const void* P, int nq, long long rows, const float* thr, int* counts, int* cand_rows,
                       float* cand_scores, int cap, bool f16, hipStream_t st) {
#ifdef CLDRD_DEV_BUILD                                 // timing-only ablations (WRONG results): development build only, never in the product library
    if (KS == 24 && !f16) {                            // ablations exist for the d = 768 bf16 instance only
        switch (cldrd_dev_int("CLDRD_SCAN_ABLATE", 0)) {
            case 1: return launch_scan_stream_abl<24, 1, false>(Q, P, nq, rows, thr, counts, cand_rows, cand_scores, cap, st);
            case 2: return launch_scan_stream_abl<24, 2, false>(Q, P, nq, rows, thr, counts, cand_rows, cand_scores, cap, st);
            default: break;
        }
    }
#endif
    if (nq > 128) {                                    // 129..256 queries: the 8-wave x 32-query instance (fp16 shadow, d = 768 only)
#ifdef CLDRD_DEV_BUILD
        if (KS == 24 && f16) {
            switch (cldrd_dev_int("CLDRD_SCAN_ABLATE", 0)) {
                case 3: return launch_scan_stream_abl<24, 3, true, 8, 2>(Q, P, nq, rows, thr, counts, cand_rows, cand_scores, cap, st);
                case 4: return launch_scan_stream_abl<24, 4, true, 8, 2>(Q, P, nq, rows, thr, counts, cand_rows, cand_scores, cap, st);
                default: break;
            }
        }
#endif
        if (KS == 24 && f16) return launch_scan_stream_abl<24, 0, true, 8, 2>(Q, P, nq, rows, thr, counts, cand_rows, cand_scores, cap, st);
        return -1;
    }
    if (f16) return launch_scan_stream_abl<KS, 0, true>(Q, P, nq, rows, thr, counts, cand_rows, cand_scores, cap, st);
    return launch_scan_stream_abl<KS, 0, false>(Q, P, nq, rows, thr, counts, cand_rows, cand_scores, cap, st);
}

}  // namespace

extern "C" int cldrd_topk_kth_largest(const float* scores, int ld, int nq, int S, int kth, float* thr, void* stream) {
    CLDRD_CHECK(nq > 0 && S > 0 && kth >= 1, "topk_kth_largest: bad arguments");
    if (S <= 8 * 1024) hipLaunchKernelGGL(kth_largest_kernel<8>, dim3(nq), dim3(1024), 0, (hipStream_t)stream, scores, ld, S, kth, thr);
    else if (S <= 32 * 1024) hipLaunchKernelGGL(kth_largest_kernel<32>, dim3(nq), dim3(1024), 0, (hipStream_t)stream, scores, ld, S, kth, thr);
    else hipLaunchKernelGGL(kth_largest_kernel<0>, dim3(nq), dim3(1024), 0, (hipStream_t)stream, scores, ld, S, kth, thr);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

extern "C" int cldrd_topk_rescore(const float* q, const float* P, int d, const int* counts, const int* cand_rows,
                                  float* cand_scores, int nq, int cap, void* stream) {
    CLDRD_CHECK(nq > 0 && d % 4 == 0 && cap > 0, "topk_rescore: bad arguments");
    hipLaunchKernelGGL(rescore_kernel, dim3(64, nq), dim3(256), 0, (hipStream_t)stream, q, P, d, counts, cand_rows, cand_scores, cap);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

extern "C" int cldrd_topk_sort(const int* counts, const int* cand_rows, const float* cand_scores, int nq, int cap, int k, float* D,
                               int* I, void* stream) {
    CLDRD_CHECK(nq > 0 && cap > 0 && cap <= 8192 && k > 0, "topk_sort: need 0 < cap <= 8192");
    int NP = 2;
    while (NP < cap) NP <<= 1;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)topk_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8);
        attr_set = true;
    }
    hipLaunchKernelGGL(topk_sort_kernel, dim3(nq), dim3(1024), (size_t)NP * 8, (hipStream_t)stream, counts, cand_rows, cand_scores, cap,
                       NP, k, D, I);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

// out: device uint32 holding the bit pattern of max_r |P[r]|^2 (zero it first)
extern "C" int cldrd_row_sqnorm_max(const float* P, size_t rows, int d, unsigned int* out, void* stream) {
    CLDRD_CHECK(rows > 0 && d % 4 == 0, "row_sqnorm_max: bad arguments");
    const int nb = (int)((rows + 3) / 4 < 2048 ? (rows + 3) / 4 : 2048);
    hipLaunchKernelGGL(row_norm_max_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, P, rows, d, out);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

extern "C" int cldrd_gather_cast_rows(const float* src, void* dst, size_t n_out, size_t stride, int d, void* stream) {
    CLDRD_CHECK(n_out > 0 && stride > 0 && d % 4 == 0, "gather_cast_rows: bad arguments");
    const int nb = (int)((n_out + 3) / 4 < 2048 ? (n_out + 3) / 4 : 2048);
    hipLaunchKernelGGL(gather_cast_rows_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n_out, stride, d);
    CLDRD_LAUNCH_CHECK();
    return 0;
}


extern "C" int cldrd_cast_f16(const float* src, void* dst, size_t n, unsigned int* flag, void* stream) {
    CLDRD_CHECK(n % 4 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 8 == 0), "cast_f16: n % 4 == 0 and aligned operands");
    if (n == 0) return 0;
    const size_t n4 = n / 4;
    const int nb = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(cast_f16_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, src, (uint16_t*)dst, n4, flag);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

// Attach statistics of an index shard (FlatIPIndex._attach).  cldrd_index_col_mean: mu[d] = mean row of P[rows, d] (fp64 sums, fixed order);
// workspace = cldrd_index_col_mean_workspace(rows, d) bytes on the device.
extern "C" size_t cldrd_index_col_mean_workspace(size_t rows, int d) {
    const size_t nb = rows < 1024 ? rows : 1024;
    return nb * (size_t)d * sizeof(double);
}
extern "C" int cldrd_index_col_mean(const float* P, size_t rows, int d, float* mu, void* workspace, size_t workspace_bytes, void* stream) {
    CLDRD_CHECK(rows > 0 && d > 0 && d <= 2048 && d % 4 == 0 && (uintptr_t)P % 16 == 0, "index_col_mean: rows > 0, 0 < d <= 2048, d % 4 == 0, 16-byte aligned rows");
    CLDRD_CHECK(workspace != nullptr && workspace_bytes >= cldrd_index_col_mean_workspace(rows, d) && (uintptr_t)workspace % 8 == 0, "index_col_mean: workspace too small");
    const int nb = (int)(rows < 1024 ? rows : 1024);
    hipLaunchKernelGGL(index_colsum_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, P, rows, d, (double*)workspace);
    CLDRD_LAUNCH_CHECK();
    hipLaunchKernelGGL(index_mean_finish_kernel, dim3((d + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, nb, d, rows, mu);
    CLDRD_LAUNCH_CHECK();
    return 0;
}
// P16[r] = fp16(P[r] - mu); sample[i] = bf16(P[i * s_stride] - mu) for i < s_rows (sample may be null); *cmax_bits (zero it first) = bit pattern
// of max_r |P[r] - mu|^2 as fp32; *flag |= 1 when a centred value does not fit fp16.
extern "C" int cldrd_index_center_cast(const float* P, const float* mu, size_t rows, int d, void* P16, void* sample_bf16, size_t s_stride,
                                       size_t s_rows, unsigned int* cmax_bits, unsigned int* flag, void* stream) {
    CLDRD_CHECK(rows > 0 && d > 0 && d % 4 == 0 && cmax_bits != nullptr, "index_center_cast: bad arguments");
    CLDRD_CHECK(sample_bf16 == nullptr || s_stride > 0, "index_center_cast: sample stride");
    CLDRD_CHECK((uintptr_t)P % 16 == 0 && (uintptr_t)mu % 16 == 0 && (uintptr_t)P16 % 8 == 0, "index_center_cast: aligned operands");
    const int nb = (int)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
    hipLaunchKernelGGL(index_center_cast_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, P, mu, rows, d, (uint16_t*)P16, (bf16_t*)sample_bf16, s_stride,
                       s_rows, cmax_bits, flag);
    CLDRD_LAUNCH_CHECK();
    return 0;
}
// out[i] = I[i] < 0 ? -1 : (ids ? ids[I[i]] : I[i] + id_offset): faiss IndexIDMap on the device
extern "C" int cldrd_map_ids(const int* I, const long long* ids, long long id_offset, long long* out, size_t n, void* stream) {
    if (n == 0) return 0;
    CLDRD_CHECK(I != nullptr && out != nullptr, "map_ids: null operands");
    const int nb = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(map_ids_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, I, ids, id_offset, out, n);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

extern "C" int cldrd_topk_prep_queries(const float* q, void* qh, void* qb, float* qnorm, int nq, int d, unsigned int* flag, void* stream) {
    CLDRD_CHECK(nq > 0 && d > 0 && d % 4 == 0, "topk_prep_queries: bad arguments");
    hipLaunchKernelGGL(prep_queries_kernel, dim3(nq), dim3(256), 0, (hipStream_t)stream, q, (uint16_t*)qh, (bf16_t*)qb, qnorm, d, flag);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

extern "C" int cldrd_topk_thresholds(const float* est, const float* qnorm, float pmax, int d, float* thr, float* eps, int nq, void* stream) {
    CLDRD_CHECK(nq > 0 && d > 0 && (est == nullptr || thr != nullptr), "topk_thresholds: bad arguments");
    hipLaunchKernelGGL(thresholds_kernel, dim3((nq + 255) / 256), dim3(256), 0, (hipStream_t)stream, est, qnorm, pmax, d, thr, eps, nq);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

static int launch_select(const int* counts, const int* dropped, const int* cand_rows, const float* cand_scores, int nq, int cap, int kk,
                         const float* thr, const float* eps, int* rows2, int cap2, int* n2, int* status, float* khat, int exhaustive,
                         hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)select_compact_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 4);
        attr_set = true;
    }
    hipLaunchKernelGGL(select_compact_kernel, dim3(nq), dim3(1024), (size_t)cap * 4, st, counts, dropped, cand_rows, cand_scores, cap, kk, thr,
                       eps, rows2, cap2, n2, status, khat, exhaustive);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

extern "C" int cldrd_topk_select(const int* counts, const int* cand_rows, const float* cand_scores, int nq, int cap, int kk, const float* thr,
                                 const float* eps, int* rows2, int cap2, int* n2, int* status, float* khat, int exhaustive, void* stream) {
    CLDRD_CHECK(nq > 0 && cap > 0 && cap <= 8192 && cap2 > 0 && kk > 0, "topk_select: need 0 < cap <= 8192");
    return launch_select(counts, counts + nq, cand_rows, cand_scores, nq, cap, kk, thr, eps, rows2, cap2, n2, status, khat, exhaustive,
                         (hipStream_t)stream);
}

int cldrd_topk_scan_stream(const void* Q, const void* P, int nq, long long rows, int d, const float* thr, int* counts, int* cand_rows,
                           float* cand_scores, int cap, int f16, hipStream_t st);
extern "C" int cldrd_topk_scan_filter(const void* Q, const void* P, int nq, long long rows, int d, const float* thr, int* counts,
                                      int* cand_rows, float* cand_scores, int cap, int f16, void* stream);
extern "C" int cldrd_topk_scan_filter_tiled(const void* Q, const void* P, int nq, long long rows, int d, const float* thr, int* counts,
                                            int* cand_rows, float* cand_scores, int cap, int f16, void* stream);

// The whole search of one shard for nq queries (device resident, fp32 + fp16 copies), in batches of 128 as the reference searches
// (retriever/retrieve_top_passages.py:88, retrieval_utils.py:131-153), enqueued back to back on `stream` with no host round trip:
//   scan (fp16 MFMA, HBM-bound) -> select t^ and the 2 eps band -> exact fp32 re-score -> sort + cut -> D, I, status.
// qtile = 128: one pass over the index per reference batch; 256: two batches share a pass (the index bytes are read once per 256
// queries; d = 768 keeps 256 queries in the registers of a CU).  counts: int[nb * (qtile + 1)] zeroed by the caller
// (nb = ceil(nq / qtile); per pass qtile list lengths + 1 dropped-hit counter); cand_rows / cand_scores: [qtile, cap] scratch;
// rows2 / scores2: [qtile, cap2] scratch; n2, status, khat: [nq]; D, I: [nq, k].
// exhaustive: bit 0 (rows <= cap): no scan, every row is re-scored; bit 1: scan with the TILED kernels, whose hits go straight to
// the global candidate lists - slower, but it has no on-chip hit list and so can never drop a hit (status bit 4): the retry form
// for passes whose hit density overflowed the streaming scan's per-wave lists.  The caller reads `status` once at the end and
// redoes the (rare) unproven queries with thresholds of its choice through this same entry point.
extern "C" int cldrd_flatip_search(const float* q32, const void* qh, const float* thr, const float* eps, const void* Ph, const float* P32,
                                   long long rows, int d, int nq, int k, int qtile, int* counts, int* cand_rows, float* cand_scores, int cap,
                                   int* rows2, float* scores2, int cap2, int* n2, int* status, float* khat, float* D, int* I,
                                   int exhaustive, void* stream) {
    CLDRD_CHECK(nq > 0 && rows > 0 && k > 0 && cap > 0 && cap <= 8192 && cap2 > 0 && cap2 <= 8192 && d % 4 == 0, "flatip_search: bad arguments");
    CLDRD_CHECK(qtile == 128 || qtile == 256, "flatip_search: the query tile is 128 (the reference's batch) or 256 (two batches per pass over the index)");
    const int tiled = (exhaustive >> 1) & 1;
    exhaustive &= 1;
    CLDRD_CHECK(!exhaustive || rows <= cap, "flatip_search: exhaustive mode needs rows <= cap");
    hipStream_t st = (hipStream_t)stream;
    const int kk = (int)(k < rows ? k : rows);
    for (int lo = 0, b = 0; lo < nq; lo += qtile, ++b) {
        const int m = nq - lo < qtile ? nq - lo : qtile;
        int* cb = counts + (size_t)b * (qtile + 1);
        int rc;
        if (exhaustive) {
            hipLaunchKernelGGL(all_candidates_kernel, dim3((unsigned)((rows + 255) / 256), m), dim3(256), 0, st, cb, cand_rows, cand_scores, (int)rows, cap);
            CLDRD_LAUNCH_CHECK();
        } else if (tiled) {
            // the tiled kernels take at most 128 queries per call; list lengths of query j of this pass stay at cb[j]
            for (int h = 0; h < m; h += 128) {
                const int mh = m - h < 128 ? m - h : 128;
                rc = cldrd_topk_scan_filter_tiled((const char*)qh + (size_t)(lo + h) * d * 2, Ph, mh, rows, d, thr + lo + h, cb + h,
                                                  cand_rows + (size_t)h * cap, cand_scores + (size_t)h * cap, cap, 1, st);
                if (rc) return rc;
            }
        } else {
            rc = cldrd_topk_scan_filter(qh ? (const char*)qh + (size_t)lo * d * 2 : nullptr, Ph, m, rows, d, thr + lo, cb, cand_rows, cand_scores, cap, 1, st);
            if (rc) return rc;
        }
        rc = launch_select(cb, cb + m, cand_rows, cand_scores, m, cap, kk, thr + lo, eps + lo, rows2, cap2, n2 + lo, status + lo, khat + lo, exhaustive, st);
        if (rc) return rc;
        rc = cldrd_topk_rescore(q32 + (size_t)lo * d, P32, d, n2 + lo, rows2, scores2, m, cap2, st);
        if (rc) return rc;
        rc = cldrd_topk_sort(n2 + lo, rows2, scores2, m, cap2, k, D + (size_t)lo * k, I + (size_t)lo * k, st);
        if (rc) return rc;
    }
    return 0;
}

// Streaming form of cldrd_topk_scan_filter for d in {128, 256, 768} and nq <= 128 (returns -1 when it does not apply).
int cldrd_topk_scan_stream(const void* Q, const void* P, int nq, long long rows, int d, const float* thr, int* counts, int* cand_rows,
                           float* cand_scores, int cap, int f16, hipStream_t st) {
    if (nq > 256 || rows < 64 || rows >= 2147483647LL / 32) return -1;
    switch (d) {
        case 128: return launch_scan_stream<4>(Q, P, nq, rows, thr, counts, cand_rows, cand_scores, cap, f16 != 0, st);
        case 256: return launch_scan_stream<8>(Q, P, nq, rows, thr, counts, cand_rows, cand_scores, cap, f16 != 0, st);
        case 768: return launch_scan_stream<24>(Q, P, nq, rows, thr, counts, cand_rows, cand_scores, cap, f16 != 0, st);
        default: return -1;
    }
}
