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

namespace {

__device__ __forceinline__ uint32_t orderable(float f) {      // monotone float -> uint32
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_orderable(uint32_t o) {
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(u);
}

// thr[q] = kth-largest value of scores[q][0..S) (kth is 1-based, clamped to S).  One block per query, MSB-first radix
// select with 8-bit digits: 4 passes over the row (L2 resident), 256-bin LDS histogram per pass.
__global__ __launch_bounds__(1024) void kth_largest_kernel(const float* __restrict__ scores, int ld, int S, int kth,
                                                           float* __restrict__ thr) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned int sel_prefix, sel_remaining;
    const float* row = scores + (size_t)blockIdx.x * ld;
    if (threadIdx.x == 0) { sel_prefix = 0; sel_remaining = (unsigned)min(kth, S); }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (threadIdx.x < 256) hist[threadIdx.x] = 0;
        __syncthreads();
        const unsigned int prefix = sel_prefix;
        const unsigned int pmask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int i = threadIdx.x; i < S; i += blockDim.x) {
            const uint32_t o = orderable(row[i]);
            if ((o & pmask) == prefix) atomicAdd(&hist[(o >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int rem = sel_remaining, b = 255;
            for (;; --b) {                       // walk from the largest digit down
                const unsigned int c = hist[b];
                if (c >= rem || b == 0) break;
                rem -= c;
            }
            sel_prefix = prefix | (b << shift);
            sel_remaining = rem;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) thr[blockIdx.x] = from_orderable(sel_prefix);
}

// exact fp32 re-score: one wave per candidate, fixed summation order (lane-strided partial sums, then a butterfly)
__global__ __launch_bounds__(256) void rescore_kernel(const float* __restrict__ q, const float* __restrict__ P, int d,
                                                       const int* __restrict__ counts, const int* __restrict__ cand_rows,
                                                       float* __restrict__ cand_scores, int cap) {
    const int qi = blockIdx.y;
    const int n = min(counts[qi], cap);
    const int lane = threadIdx.x & 63;
    const float* qr = q + (size_t)qi * d;
    for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < n; c += gridDim.x * 4) {
        const float* pr = P + (size_t)cand_rows[(size_t)qi * cap + c] * d;
        float s = 0.f;
        for (int j = lane * 4; j < d; j += 256) {
            const float4 a = *(const float4*)(qr + j), b = *(const float4*)(pr + j);
            s = fmaf(a.x, b.x, s); s = fmaf(a.y, b.y, s); s = fmaf(a.z, b.z, s); s = fmaf(a.w, b.w, s);
        }
        s = wave_sum(s);
        if (lane == 0) cand_scores[(size_t)qi * cap + c] = s;
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
    for (int size = 2; size <= NP; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < NP / 2; t += blockDim.x) {
                const int lo = (t / stride) * 2 * stride + (t % stride), hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
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

}  // namespace

extern "C" int cldrd_topk_kth_largest(const float* scores, int ld, int nq, int S, int kth, float* thr, void* stream) {
    CLDRD_CHECK(nq > 0 && S > 0 && kth >= 1, "topk_kth_largest: bad arguments");
    hipLaunchKernelGGL(kth_largest_kernel, dim3(nq), dim3(1024), 0, (hipStream_t)stream, scores, ld, S, kth, thr);
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
