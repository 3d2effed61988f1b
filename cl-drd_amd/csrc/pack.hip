// Variable-length packing of a batch (MI355X-first replacement for the padding the reference inherits from HuggingFace: its tokenizer pads
// every sequence of a batch to the longest one, dataset/sequence_dataset.py:50-51, dataset/nway_dataset.py:103-107, and every Linear /
// LayerNorm of the encoder then runs over the padding too - ~40 % of the rows of an MS MARCO batch).
//
// Packed layout: the tokens of sequence m are rows cu[m] .. cu[m] + len[m] of a [Tp, features] matrix, Tp = sum of the lengths.  GEMMs,
// LayerNorm and the weight gradients work on any row count, so they simply see fewer rows.  Attention reads the same packed rows through cu
// (attention.hip, cldrd_attention_*_varlen; until round 6 it kept a padded [nseq * L, .] layout and unpack_rows16 / gather_rows moved the rows
// there and back around every call).  What is left for these kernels: gathering token ids and CLS rows (rows cu[m]), putting the CLS rows'
// gradients back - and unpack_rows16 as the reference layout the tests compare the packed attention with.
#include "common.h"

namespace {

// dst[(m * L + j), :] = j < len[m] ? src[cu[m] + j, :] : 0     (16-bit elements, w % 8 == 0); one wave per destination row
__global__ __launch_bounds__(256) void unpack_rows16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, const int* __restrict__ cu,
                                                             int nseq, int L, int w) {
    const int lane = threadIdx.x & 63;
    const long long rows = (long long)nseq * L;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
        const int m = (int)(r / L), j = (int)(r % L);
        const int c0 = cu[m], len = cu[m + 1] - c0;
        uint4* d = (uint4*)(dst + (size_t)r * w);
        if (j < len) {
            const uint4* s = (const uint4*)(src + (size_t)(c0 + j) * w);
            for (int c = lane; c < w / 8; c += 64) d[c] = s[c];
        } else {
            for (int c = lane; c < w / 8; c += 64) d[c] = make_uint4(0, 0, 0, 0);
        }
    }
}

// dst[p, :] = src[idx[p], :]   (rows of `bytes` bytes, bytes % 16 == 0); one wave per row
__global__ __launch_bounds__(256) void gather_rows_kernel(const char* __restrict__ src, const int* __restrict__ idx, char* __restrict__ dst, int n, int bytes) {
    const int lane = threadIdx.x & 63;
    for (int p = blockIdx.x * 4 + (threadIdx.x >> 6); p < n; p += gridDim.x * 4) {
        const uint4* s = (const uint4*)(src + (size_t)idx[p] * bytes);
        uint4* d = (uint4*)(dst + (size_t)p * bytes);
        for (int c = lane; c < bytes / 16; c += 64) d[c] = s[c];
    }
}

// dst[p] = src[idx[p]]   (int64 elements: the token ids of the packed rows)
__global__ __launch_bounds__(256) void gather_i64_kernel(const long long* __restrict__ src, const int* __restrict__ idx, long long* __restrict__ dst, int n) {
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) dst[p] = src[idx[p]];
}

// g[idx[r], :] = bf16(dcls[r, :])   (g zeroed by the launcher)
template <int FMT>      // 0 bf16, 1 fp32, 2 fp16
__global__ void scatter_cls_idx_kernel(const float* __restrict__ dcls, void* __restrict__ g, const int* __restrict__ idx, int d) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        if (FMT == 1) ((float*)g)[(size_t)idx[r] * d + c] = dcls[(size_t)r * d + c];
        else if (FMT == 2) ((_Float16*)g)[(size_t)idx[r] * d + c] = (_Float16)dcls[(size_t)r * d + c];
        else ((bf16_t*)g)[(size_t)idx[r] * d + c] = f2bf(dcls[(size_t)r * d + c]);
    }
}

// dst[idx[m], :] += src[m, :]   (bf16 rows, fp32 add, one rounding)
template <int FMT>      // 0: bf16 += bf16; 1: fp32 += fp32; 2: fp16 dst += fp32 src (fp32 add, one rounding: the fp16 gradient stream)
__global__ void add_rows_idx_kernel(void* __restrict__ dst_v, const void* __restrict__ src_v, const int* __restrict__ idx, int d) {
    const int m = blockIdx.x;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        const size_t o = (size_t)idx[m] * d + c;
        if (FMT == 1) ((float*)dst_v)[o] += ((const float*)src_v)[(size_t)m * d + c];
        else if (FMT == 2) ((_Float16*)dst_v)[o] = (_Float16)((float)((const _Float16*)dst_v)[o] + ((const float*)src_v)[(size_t)m * d + c]);
        else ((bf16_t*)dst_v)[o] = f2bf(bf2f(((const bf16_t*)dst_v)[o]) + bf2f(((const bf16_t*)src_v)[(size_t)m * d + c]));
    }
}

}  // namespace

extern "C" int cldrd_unpack_rows16(const void* src_packed, void* dst_padded, const int* cu, int nseq, int L, int w, void* stream) {
    CLDRD_CHECK(nseq > 0 && L > 0 && w > 0 && w % 8 == 0, "unpack_rows16: bad shape (w must be a multiple of 8)");
    const long long rows = (long long)nseq * L;
    const int nb = (int)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
    hipLaunchKernelGGL(unpack_rows16_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src_packed, (bf16_t*)dst_padded, cu, nseq, L, w);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

extern "C" int cldrd_gather_rows(const void* src, const int* idx, void* dst, int n, int row_bytes, void* stream) {
    CLDRD_CHECK(n > 0 && row_bytes > 0 && row_bytes % 16 == 0, "gather_rows: rows must be a multiple of 16 bytes");
    const int nb = (n + 3) / 4 < 4096 ? (n + 3) / 4 : 4096;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const char*)src, idx, (char*)dst, n, row_bytes);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

extern "C" int cldrd_gather_i64(const long long* src, const int* idx, long long* dst, int n, void* stream) {
    CLDRD_CHECK(n > 0, "gather_i64: n must be positive");
    const int nb = (n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048;
    hipLaunchKernelGGL(gather_i64_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, src, idx, dst, n);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

extern "C" int cldrd_scatter_cls_grad_idx(const float* dcls, void* g, int R, int d, const int* idx, int T, int g_f32, void* stream) {
    CLDRD_CHECK(R > 0 && d > 0 && T >= R, "scatter_cls_grad_idx: bad shape");
    if (hipMemsetAsync(g, 0, (size_t)T * d * (g_f32 == 1 ? sizeof(float) : sizeof(bf16_t)), (hipStream_t)stream) != hipSuccess) return cldrd_set_error("scatter_cls_grad_idx: memset failed");
    if (g_f32 == 1) hipLaunchKernelGGL(scatter_cls_idx_kernel<1>, dim3(R), dim3(256), 0, (hipStream_t)stream, dcls, g, idx, d);
    else if (g_f32 == 2) hipLaunchKernelGGL(scatter_cls_idx_kernel<2>, dim3(R), dim3(256), 0, (hipStream_t)stream, dcls, g, idx, d);
    else hipLaunchKernelGGL(scatter_cls_idx_kernel<0>, dim3(R), dim3(256), 0, (hipStream_t)stream, dcls, g, idx, d);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

extern "C" int cldrd_add_rows_idx(void* dst, const void* src, int M, int d, const int* idx, int f32, void* stream) {
    CLDRD_CHECK(M > 0 && d > 0, "add_rows_idx: bad shape");
    if (f32 == 1) hipLaunchKernelGGL(add_rows_idx_kernel<1>, dim3(M), dim3(256), 0, (hipStream_t)stream, dst, src, idx, d);
    else if (f32 == 2) hipLaunchKernelGGL(add_rows_idx_kernel<2>, dim3(M), dim3(256), 0, (hipStream_t)stream, dst, src, idx, d);
    else hipLaunchKernelGGL(add_rows_idx_kernel<0>, dim3(M), dim3(256), 0, (hipStream_t)stream, dst, src, idx, d);
    CLDRD_LAUNCH_CHECK();
    return 0;
}
