// Row kernels of the encoder: embedding + LayerNorm, residual LayerNorm (forward / backward), the
// parameter-gradient reductions that go with them, and column sums for bias gradients.
//
// Reference call sites: HF Embeddings.forward (word + position (+ token type) -> LayerNorm(eps 1e-12) ->
// dropout) and the two post-LN LayerNorms of every transformer block.  All of these
// are HBM-bound: one wavefront owns one token row (d <= 1024), 8-byte bf16 loads per lane, two-pass
// mean/variance in registers, wave-level shuffles for the reductions; LN statistics and all parameter
// gradients are fp32.  Parameter gradients are reduced deterministically: per-block partial rows, then
// one column-sum pass (no float atomics) - except the embedding-table scatter, which uses fp32 atomics.
#include <type_traits>

#include "common.h"

namespace {

constexpr int MAX_IT = 4;          // d <= 4 * 256

struct RowF { float v[MAX_IT][4]; };

__device__ __forceinline__ void load_row_bf16(const bf16_t* row, int d, int lane, RowF& r) {
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < d) {
            const uint2 u = *(const uint2*)(row + c);
            r.v[it][0] = __uint_as_float(u.x << 16); r.v[it][1] = __uint_as_float(u.x & 0xFFFF0000u);
            r.v[it][2] = __uint_as_float(u.y << 16); r.v[it][3] = __uint_as_float(u.y & 0xFFFF0000u);
        } else {
            r.v[it][0] = r.v[it][1] = r.v[it][2] = r.v[it][3] = 0.f;
        }
    }
}
// fp16 rows (round 4, the all-fp16 training mode: activation gradients are fp16 and carry the loss scale)
__device__ __forceinline__ float ln_h2f(uint32_t bits16) { return (float)__builtin_bit_cast(_Float16, (uint16_t)bits16); }
__device__ __forceinline__ void load_row_f16(const bf16_t* row, int d, int lane, RowF& r) {
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < d) {
            const uint2 u = *(const uint2*)(row + c);
            r.v[it][0] = ln_h2f(u.x & 0xFFFFu); r.v[it][1] = ln_h2f(u.x >> 16);
            r.v[it][2] = ln_h2f(u.y & 0xFFFFu); r.v[it][3] = ln_h2f(u.y >> 16);
        } else {
            r.v[it][0] = r.v[it][1] = r.v[it][2] = r.v[it][3] = 0.f;
        }
    }
}
__device__ __forceinline__ void load_row_f32(const float* row, int d, int lane, RowF& r) {
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < d) {
            const float4 u = *(const float4*)(row + c);
            r.v[it][0] = u.x; r.v[it][1] = u.y; r.v[it][2] = u.z; r.v[it][3] = u.w;
        } else {
            r.v[it][0] = r.v[it][1] = r.v[it][2] = r.v[it][3] = 0.f;
        }
    }
}
// the same through the streaming path (common.h: ld16_stream): the backward's read of the pre-LN sums the forward left on the tape
__device__ __forceinline__ void load_row_f32_stream(const float* row, int d, int lane, RowF& r) {
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < d) {
            const uint4 u = ld16_stream(row + c);
            r.v[it][0] = __uint_as_float(u.x); r.v[it][1] = __uint_as_float(u.y); r.v[it][2] = __uint_as_float(u.z); r.v[it][3] = __uint_as_float(u.w);
        } else {
            r.v[it][0] = r.v[it][1] = r.v[it][2] = r.v[it][3] = 0.f;
        }
    }
}
// INTERLEAVED column map for kernels that scatter a row with float atomics: element (it, j) of a lane is column
// it*256 + j*64 + lane, so one wave-instruction touches 64 consecutive floats = 256 contiguous bytes.  With the vector map above
// (4 consecutive columns per lane) an atomic instruction spreads its 64 dwords over 1 KiB - four times the 64-B requests at the
// memory side, where float atomics execute (MI355X_MICROARCH.md, Global float atomics): the embedding backward ran 380 us, not 130.
__device__ __forceinline__ int icol(int it, int j, int lane) { return it * 256 + j * 64 + lane; }
__device__ __forceinline__ void load_row_bf16_i(const bf16_t* row, int d, int lane, RowF& r) {
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int c = icol(it, j, lane); r.v[it][j] = c < d ? bf2f(row[c]) : 0.f; }
}
__device__ __forceinline__ void load_row_f16_i(const bf16_t* row, int d, int lane, RowF& r) {
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int c = icol(it, j, lane); r.v[it][j] = c < d ? ln_h2f(row[c]) : 0.f; }
}
__device__ __forceinline__ void load_row_f32_i(const float* row, int d, int lane, RowF& r) {
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int c = icol(it, j, lane); r.v[it][j] = c < d ? row[c] : 0.f; }
}
// the same with non-temporal stores: a tape copy nobody reads before the backward's weight-gradient launch (~10 ms later) should not
// push the operands of the next GEMM out of the Infinity Cache
__device__ __forceinline__ void store_row_bf16_stream(bf16_t* row, int d, int lane, const RowF& r) {
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < d) {
            typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
            const u32x2_t u = {pack2bf(r.v[it][0], r.v[it][1]), pack2bf(r.v[it][2], r.v[it][3])};
            __builtin_nontemporal_store(u, (u32x2_t*)(row + c));
        }
    }
}
__device__ __forceinline__ void store_row_bf16(bf16_t* row, int d, int lane, const RowF& r) {
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < d) {
            uint2 u; u.x = pack2bf(r.v[it][0], r.v[it][1]); u.y = pack2bf(r.v[it][2], r.v[it][3]);
            *(uint2*)(row + c) = u;
        }
    }
}
// fp16 instead of bf16 (the high-precision forward of the query tower)
__device__ __forceinline__ void store_row_f16(bf16_t* row, int d, int lane, const RowF& r) {
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < d) {
            typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
            const cldrd_f32v2 f0 = {r.v[it][0], r.v[it][1]}, f1 = {r.v[it][2], r.v[it][3]};
            uint2 u;
            u.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(f0, h2_t));
            u.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(f1, h2_t));
            *(uint2*)(row + c) = u;
        }
    }
}
__device__ __forceinline__ void store_row_f32(float* row, int d, int lane, const RowF& r) {
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < d) *(float4*)(row + c) = make_float4(r.v[it][0], r.v[it][1], r.v[it][2], r.v[it][3]);
    }
}
__device__ __forceinline__ void row_stats(const RowF& r, int d, int lane, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) s += (r.v[it][0] + r.v[it][1]) + (r.v[it][2] + r.v[it][3]);
    mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < d) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float t = r.v[it][j] - mean; q += t * t; }
        }
    }
    rstd = rsqrtf(wave_sum(q) / (float)d + eps);
}

// out = LN(x) * gamma + beta;   optional fp32 copy of rows r with r % cls_stride == 0 (the CLS pooling of
// reference models/nway_dual_encoder.py:52,56,64 folded into the last LayerNorm).
// X32: the input (pre-LN residual sum) is fp32 and, when out32 != null, the output is also kept in fp32 next to the bf16 copy the
// GEMMs read (the fp32 residual stream: the reference's autocast keeps LayerNorm inputs / outputs in fp32, nway_listwise_1.py:334).
template <int DC, bool X32>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, bf16_t* __restrict__ out, float* __restrict__ out32,
                                                      float* __restrict__ mean_o, float* __restrict__ rstd_o, int T, int d_rt,
                                                      float eps, float* __restrict__ cls_out, int cls_stride, int out_f16,
                                                      bf16_t* __restrict__ out_copy) {
    const int d = DC ? DC : d_rt;      // compile-time row width: the `c < d` tests and the unused 4th column pass fold away
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    RowF r;
    if (X32) load_row_f32((const float*)x + (size_t)row * d, d, lane, r);
    else load_row_bf16((const bf16_t*)x + (size_t)row * d, d, lane, r);
    float mean, rstd;
    row_stats(r, d, lane, eps, mean, rstd);
    const bool cls = cls_out && (row % cls_stride == 0);
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < d) {
            const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
            r.v[it][0] = (r.v[it][0] - mean) * rstd * g.x + b.x; r.v[it][1] = (r.v[it][1] - mean) * rstd * g.y + b.y;
            r.v[it][2] = (r.v[it][2] - mean) * rstd * g.z + b.z; r.v[it][3] = (r.v[it][3] - mean) * rstd * g.w + b.w;
            if (cls) *(float4*)(cls_out + (size_t)(row / cls_stride) * d + c) = make_float4(r.v[it][0], r.v[it][1], r.v[it][2], r.v[it][3]);
        }
    }
    if (out_f16) store_row_f16(out + (size_t)row * d, d, lane, r);
    else store_row_bf16(out + (size_t)row * d, d, lane, r);
    if (out_copy) store_row_bf16_stream(out_copy + (size_t)row * d, d, lane, r);       // fp16 `out` for the forward GEMM + the bf16 tape copy
    if (X32 && out32) store_row_f32(out32 + (size_t)row * d, d, lane, r);
    if (lane == 0) { if (mean_o) mean_o[row] = mean; if (rstd_o) rstd_o[row] = rstd; }
}

// word + position (+ type) embedding -> LN -> dropout.  Tables are the fp32 master weights (HF keeps the
// embedding lookup and LayerNorm in fp32 under autocast).
template <int DC, bool DROP>
__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ word,
                                                            const float* __restrict__ pos, const float* __restrict__ type0,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            bf16_t* __restrict__ out, float* __restrict__ out32, float* __restrict__ mean_o,
                                                            float* __restrict__ rstd_o, int T, int L, int d_rt, int vocab, float eps,
                                                            uint32_t drop_thresh, float drop_scale, SeedArg seed_a, int out_f16,
                                                            const int* __restrict__ pos_idx, bf16_t* __restrict__ out_copy) {
    const uint64_t seed = seed_a.get();
    const int d = DC ? DC : d_rt;      // compile-time row width: the `c < d` tests and the unused 4th column pass fold away
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    int64_t id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    RowF r, p;
    load_row_f32(word + (size_t)id * d, d, lane, r);
    load_row_f32(pos + (size_t)(pos_idx ? pos_idx[row] : row % L) * d, d, lane, p);      // pos_idx: packed batches (csrc/pack.hip)
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) r.v[it][j] += p.v[it][j];
    if (type0) {
        load_row_f32(type0, d, lane, p);
#pragma unroll
        for (int it = 0; it < MAX_IT; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) r.v[it][j] += p.v[it][j];
    }
    float mean, rstd;
    row_stats(r, d, lane, eps, mean, rstd);
    const uint32_t rk = drop_rowkey(seed, (uint32_t)row);
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < d) {
            const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
            const float gg[4] = {g.x, g.y, g.z, g.w}, bb[4] = {b.x, b.y, b.z, b.w};
            const uint32_t h0 = drop_pair(rk, c), h1 = drop_pair(rk, c + 2);
            const bool keep[4] = {drop_keep_lo(h0, drop_thresh), drop_keep_hi(h0, drop_thresh), drop_keep_lo(h1, drop_thresh), drop_keep_hi(h1, drop_thresh)};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float y = (r.v[it][j] - mean) * rstd * gg[j] + bb[j];
                if (DROP) y = keep[j] ? y * drop_scale : 0.f;
                r.v[it][j] = y;
            }
        }
    }
    if (out_f16) store_row_f16(out + (size_t)row * d, d, lane, r);
    else store_row_bf16(out + (size_t)row * d, d, lane, r);
    if (out_copy) store_row_bf16(out_copy + (size_t)row * d, d, lane, r);
    if (out32) store_row_f32(out32 + (size_t)row * d, d, lane, r);
    if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
}

// Sum three per-column accumulators over the waves of a block and write them to partial[blockIdx.x][3][d].
// Binary tree over the waves through LDS with plain 16-byte stores / loads (fixed order: bitwise reproducible; LDS float
// atomics from 8 waves onto the same 3*d addresses cost more than the row streaming itself).
// LDS: (waves / 2) * 3 * MAX_IT * 64 float4 = 6 KiB per wave pair.
template <bool INTERLEAVED = false>
__device__ __forceinline__ void block_partials(float* smem, RowF& a, RowF& b, RowF& c3, int d, int lane, float* __restrict__ partial) {
    const int w = threadIdx.x >> 6;
    float4* s4 = (float4*)smem;
    RowF* acc[3] = {&a, &b, &c3};
    for (int half = (int)(blockDim.x >> 7); half >= 1; half >>= 1) {
        if (w >= half && w < 2 * half) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int it = 0; it < MAX_IT; ++it)
                    s4[(((w - half) * 3 + k) * MAX_IT + it) * 64 + lane] = make_float4(acc[k]->v[it][0], acc[k]->v[it][1], acc[k]->v[it][2], acc[k]->v[it][3]);
        }
        __syncthreads();
        if (w < half) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int it = 0; it < MAX_IT; ++it) {
                    const float4 t = s4[((w * 3 + k) * MAX_IT + it) * 64 + lane];
                    acc[k]->v[it][0] += t.x; acc[k]->v[it][1] += t.y; acc[k]->v[it][2] += t.z; acc[k]->v[it][3] += t.w;
                }
        }
        __syncthreads();
    }
    if (w == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int it = 0; it < MAX_IT; ++it) {
                if (INTERLEAVED) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = icol(it, j, lane);
                        if (c < d) partial[(size_t)blockIdx.x * 3 * d + k * d + c] = acc[k]->v[it][j];
                    }
                } else {
                    const int c = it * 256 + lane * 4;
                    if (c < d)
                        *(float4*)(partial + (size_t)blockIdx.x * 3 * d + k * d + c) = make_float4(acc[k]->v[it][0], acc[k]->v[it][1], acc[k]->v[it][2], acc[k]->v[it][3]);
                }
            }
    }
}

// LayerNorm backward.  x = the LN input (pre-LN residual sum), dy = gradient of the LN output.
//   dx  = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat))                       -> dx (residual path)
//   dx2 = dropout-masked dx (the branch that went through dropout before the residual add), or null
//   partial[blk] = { sum dy*xhat (dgamma), sum dy (dbeta), sum dx2-or-dx (bias grad of the preceding Linear) }
// G32 (round 3): the gradient STREAM is fp32 - dy is read and dx written as fp32 rows; dx2, the MFMA operand of the next data-gradient
// GEMM, stays bf16 (and is then written without dropout too).  The stream is rounded to 16 bits nowhere between the loss and the embeddings.
// dy_branch (optional, bf16): the output of the data-gradient GEMM of the branch that joins the stream here (FFN / attention), added on
// load: dy = stream + branch.  That GEMM then has a plain bf16 epilogue (no fp32 residual in, no fp32 sum out: 200 MB less per GEMM at
// T = 32768 than adding in its epilogue) and only the branch's own contribution is rounded, once, like any MFMA operand.
// GS: format of the gradient STREAM (dy in, dx out): 0 = bf16 (rounds 1-2), 1 = fp32 (round 3), 2 = fp16 (round 5, the all-fp16 training mode:
// the stream carries the loss scale like every other 16-bit gradient tensor; dy_branch and dx2 are fp16 too).  With GS = 2 the stream tensor dx IS
// the MFMA operand of the next data / weight gradient unless dropout makes the two differ: dx2 may be null, and a LayerNorm backward then moves
// 250 MB at T = 32768 instead of the 400 MB of the fp32 stream (read 50 + 50 + 100, write 50).  Measured on the reference's fp32 gradients
// (tools/grad_cos_report.py -> profiles/r05_grad_cosines.txt, every ranked tensor of cfg1-4): min cosine 0.99990 / 0.99992 / 0.99994 / 0.99992 / 0.99995 with the stream rounded to
// fp16 at exactly these points, against 0.99992 / 0.99994 / 0.99998 / 0.99995 / 0.99996 with the fp32 stream (profiles/r05_grad_cosines.txt).
template <int DC, bool DROP, bool X32, int GS = 0>
__global__ __launch_bounds__(512) void ln_bwd_kernel(const void* __restrict__ dy_v, const void* __restrict__ x,
                                                      const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                      const float* __restrict__ gamma, void* __restrict__ dx_v,
                                                      bf16_t* __restrict__ dx2, float* __restrict__ partial, int T, int d_rt,
                                                      uint32_t drop_thresh, float drop_scale, SeedArg seed_a,
                                                      const bf16_t* __restrict__ dy_branch, int h16, const float* __restrict__ inv_scale) {
    // h16: dy_branch and dx2 are fp16 (the all-fp16 training mode); inv_scale: the per-block parameter-gradient sums leave multiplied by it
    const uint64_t seed = seed_a.get();
    const int d = DC ? DC : d_rt;      // compile-time row width: the `c < d` tests and the unused 4th column pass fold away
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    const int lane = threadIdx.x & 63;
    RowF dg, db, dbias;
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) dg.v[it][j] = db.v[it][j] = dbias.v[it][j] = 0.f;
    const int wpb = blockDim.x >> 6;      // waves per block: 8 -> 16 waves per CU keep enough loads in flight
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < T; row += gridDim.x * wpb) {
        RowF g, xr;
        if (GS != 0) {
            if (GS == 2) load_row_f16((const bf16_t*)dy_v + (size_t)row * d, d, lane, g);
            else load_row_f32_stream((const float*)dy_v + (size_t)row * d, d, lane, g);
            if (dy_branch) {       // dy = stream + the 16-bit output of the branch's data-gradient GEMM, added here instead of in its epilogue
                RowF br;
                if (h16) load_row_f16(dy_branch + (size_t)row * d, d, lane, br);
                else load_row_bf16(dy_branch + (size_t)row * d, d, lane, br);
#pragma unroll
                for (int it = 0; it < MAX_IT; ++it)
#pragma unroll
                    for (int j = 0; j < 4; ++j) g.v[it][j] += br.v[it][j];
            }
        } else {
            load_row_bf16((const bf16_t*)dy_v + (size_t)row * d, d, lane, g);
        }
        if (X32) load_row_f32_stream((const float*)x + (size_t)row * d, d, lane, xr);
        else load_row_bf16((const bf16_t*)x + (size_t)row * d, d, lane, xr);
        const float mean = mean_i[row], rstd = rstd_i[row];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < MAX_IT; ++it) {
            const int c = it * 256 + lane * 4;
            if (c < d) {
                const float4 gm = *(const float4*)(gamma + c);
                const float gg[4] = {gm.x, gm.y, gm.z, gm.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (xr.v[it][j] - mean) * rstd;
                    const float dyv = g.v[it][j];
                    dg.v[it][j] += dyv * xh; db.v[it][j] += dyv;
                    const float t = dyv * gg[j];
                    s1 += t; s2 += t * xh;
                    xr.v[it][j] = xh; g.v[it][j] = t;
                }
            }
        }
        s1 = wave_sum(s1) / (float)d; s2 = wave_sum(s2) / (float)d;
        RowF o2;
        const uint32_t rk = drop_rowkey(seed, (uint32_t)row);
#pragma unroll
        for (int it = 0; it < MAX_IT; ++it) {
            const int c = it * 256 + lane * 4;
            const uint32_t h0 = drop_pair(rk, c), h1 = drop_pair(rk, c + 2);
            const bool keep[4] = {drop_keep_lo(h0, drop_thresh), drop_keep_hi(h0, drop_thresh), drop_keep_lo(h1, drop_thresh), drop_keep_hi(h1, drop_thresh)};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = (c < d) ? rstd * (g.v[it][j] - s1 - xr.v[it][j] * s2) : 0.f;
                g.v[it][j] = v;
                if (DROP && c < d) v = keep[j] ? v * drop_scale : 0.f;
                o2.v[it][j] = v;
                dbias.v[it][j] += v;
            }
        }
        if (GS == 1) store_row_f32((float*)dx_v + (size_t)row * d, d, lane, g);
        else if (GS == 2) store_row_f16((bf16_t*)dx_v + (size_t)row * d, d, lane, g);
        else store_row_bf16((bf16_t*)dx_v + (size_t)row * d, d, lane, g);
        if (dx2) { if (h16) store_row_f16(dx2 + (size_t)row * d, d, lane, o2); else store_row_bf16(dx2 + (size_t)row * d, d, lane, o2); }
    }
    if (inv_scale) {
        const float is = *inv_scale;
#pragma unroll
        for (int it = 0; it < MAX_IT; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) { dg.v[it][j] *= is; db.v[it][j] *= is; dbias.v[it][j] *= is; }
    }
    block_partials(lsm, dg, db, dbias, d, lane, partial);
}

// Embedding backward: dy -> (dropout) -> LN backward (statistics saved, input recomputed from the tables)
// -> scatter-add into the word / position tables (fp32 atomics); LN-parameter and token-type gradients go
// through per-block partials {dgamma, dbeta, dtype}.
template <int DC, bool DROP, int GS = 0>      // GS: format of dy, as in ln_bwd_kernel (0 bf16, 1 fp32, 2 fp16)
__global__ __launch_bounds__(256) void embed_ln_bwd_kernel(const void* __restrict__ dy_v, const int64_t* __restrict__ ids,
                                                            const float* __restrict__ word, const float* __restrict__ pos,
                                                            const float* __restrict__ type0, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                            float* __restrict__ dword, float* __restrict__ dpos,
                                                            float* __restrict__ partial, int T, int L, int d_rt, int vocab,
                                                            uint32_t drop_thresh, float drop_scale, SeedArg seed_a, int pos_uniform,
                                                            const int* __restrict__ pos_idx, const bf16_t* __restrict__ dy_branch, int h16,
                                                            const float* __restrict__ inv_scale) {
    const uint64_t seed = seed_a.get();
    const float is = inv_scale ? *inv_scale : 1.0f;       // the table and LayerNorm-parameter gradients leave without the loss scale
    const int d = DC ? DC : d_rt;      // compile-time row width: the `c < d` tests and the unused 4th column pass fold away
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    const int lane = threadIdx.x & 63;
    // pos_uniform: the row stride of a wave (4 * gridDim.x) is a multiple of L, so every row of this wave sits at the same
    // position: its position-table gradient is summed in registers and leaves with ONE atomic per element (the per-row
    // atomics of nseq sequences onto the same L x d addresses were the bulk of this kernel's time).
    RowF dg, db, dt, dp;
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) dg.v[it][j] = db.v[it][j] = dt.v[it][j] = dp.v[it][j] = 0.f;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < T; row += gridDim.x * 4) {
        int64_t id = ids[row];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const int l = pos_idx ? pos_idx[row] : row % L;
        RowF g, xr, p;
        if (GS != 0) {
            if (GS == 2) load_row_f16_i((const bf16_t*)dy_v + (size_t)row * d, d, lane, g);
            else load_row_f32_i((const float*)dy_v + (size_t)row * d, d, lane, g);
            if (dy_branch) {       // + the bf16 output of layer 0's last data-gradient GEMM (see ln_bwd_kernel)
                RowF br;
                if (h16) load_row_f16_i(dy_branch + (size_t)row * d, d, lane, br);
                else load_row_bf16_i(dy_branch + (size_t)row * d, d, lane, br);
#pragma unroll
                for (int it = 0; it < MAX_IT; ++it)
#pragma unroll
                    for (int j = 0; j < 4; ++j) g.v[it][j] += br.v[it][j];
            }
        } else {
            load_row_bf16_i((const bf16_t*)dy_v + (size_t)row * d, d, lane, g);
        }
        {   // a row whose incoming gradient is exactly zero (padded positions: nothing attends to them) contributes exactly
            // zero to every sum below: skip it, and with it the atomics of all pad tokens onto the one [PAD] table row
            float amax = 0.f;
#pragma unroll
            for (int it = 0; it < MAX_IT; ++it)
#pragma unroll
                for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fabsf(g.v[it][j]));
            if (wave_max(amax) == 0.f) continue;
        }
        load_row_f32_i(word + (size_t)id * d, d, lane, xr);
        load_row_f32_i(pos + (size_t)l * d, d, lane, p);
#pragma unroll
        for (int it = 0; it < MAX_IT; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) xr.v[it][j] += p.v[it][j];
        if (type0) {
            load_row_f32_i(type0, d, lane, p);
#pragma unroll
            for (int it = 0; it < MAX_IT; ++it)
#pragma unroll
                for (int j = 0; j < 4; ++j) xr.v[it][j] += p.v[it][j];
        }
        const float mean = mean_i[row], rstd = rstd_i[row];
        const uint32_t rk = drop_rowkey(seed, (uint32_t)row);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < MAX_IT; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = icol(it, j, lane);
                if (c < d) {
                    float dyv = g.v[it][j];
                    if (DROP) dyv = dropout_keep(rk, (uint32_t)c, drop_thresh) ? dyv * drop_scale : 0.f;
                    const float xh = (xr.v[it][j] - mean) * rstd;
                    dg.v[it][j] += dyv * xh; db.v[it][j] += dyv;
                    const float t = dyv * gamma[c];
                    s1 += t; s2 += t * xh;
                    xr.v[it][j] = xh; g.v[it][j] = t;
                }
            }
        s1 = wave_sum(s1) / (float)d; s2 = wave_sum(s2) / (float)d;
#pragma unroll
        for (int it = 0; it < MAX_IT; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = icol(it, j, lane);
                if (c < d) {
                    const float v = rstd * (g.v[it][j] - s1 - xr.v[it][j] * s2) * is;
                    dt.v[it][j] += v;
                    atomicAdd(dword + (size_t)id * d + c, v);          // 64 lanes x 4 B contiguous per instruction
                    if (pos_uniform) dp.v[it][j] += v;
                    else atomicAdd(dpos + (size_t)l * d + c, v);
                }
            }
    }
    if (pos_uniform) {
        const int l = (blockIdx.x * 4 + (threadIdx.x >> 6)) % L;
#pragma unroll
        for (int it = 0; it < MAX_IT; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = icol(it, j, lane);
                if (c < d) atomicAdd(dpos + (size_t)l * d + c, dp.v[it][j]);
            }
    }
    if (inv_scale) {      // dt (and the table atomics above) already carry 1 / S
#pragma unroll
        for (int it = 0; it < MAX_IT; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) { dg.v[it][j] *= is; db.v[it][j] *= is; }
    }
    block_partials<true>(lsm, dg, db, dt, d, lane, partial);
}

// out[c] (+)= sum_b partial[b][c].  Block = 64 columns x 16 row groups (coalesced 256-B row reads, 16 short load chains
// per column instead of one long one), fixed-order LDS combine.
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ partial, int nblk, int n, float* __restrict__ out0,
                                                                float* __restrict__ out1, float* __restrict__ out2, int seg, int accumulate) {
    constexpr int RG = 16;
    __shared__ float red[RG][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f;
    if (c < n) {
        int b = rg;
        for (; b + RG < nblk; b += 2 * RG) { s0 += partial[(size_t)b * n + c]; s1 += partial[(size_t)(b + RG) * n + c]; }
        if (b < nblk) s0 += partial[(size_t)b * n + c];
    }
    red[rg][threadIdx.x & 63] = s0 + s1;
    __syncthreads();
    if (rg != 0 || c >= n) return;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < RG; i += 4) s += (red[i][threadIdx.x] + red[i + 1][threadIdx.x]) + (red[i + 2][threadIdx.x] + red[i + 3][threadIdx.x]);
    float* out = c < seg ? out0 : (c < 2 * seg ? out1 : out2);
    if (!out) return;
    const int i = c % seg;
    out[i] = accumulate ? out[i] + s : s;
}

// column sums of a bf16 [T, N] matrix: partial[chunk][N]; thread owns 4 columns, block owns 1024 columns x rows_per_blk rows
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, float* __restrict__ partial, int T, int N,
                                                      int ld, int rows_per_blk) {
    const int c = blockIdx.x * 1024 + threadIdx.x * 4;
    if (c >= N) return;
    const int r0 = blockIdx.y * rows_per_blk, r1 = min(T, r0 + rows_per_blk);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int r = r0; r < r1; ++r) {
        const uint2 u = *(const uint2*)(x + (size_t)r * ld + c);
        s0 += __uint_as_float(u.x << 16); s1 += __uint_as_float(u.x & 0xFFFF0000u);
        s2 += __uint_as_float(u.y << 16); s3 += __uint_as_float(u.y & 0xFFFF0000u);
    }
    *(float4*)(partial + (size_t)blockIdx.y * N + c) = make_float4(s0, s1, s2, s3);
}

// g[T, d] = 0 except rows r*stride <- dcls[r] (gradient of the CLS pooling)
template <int FMT>      // 0 bf16, 1 fp32, 2 fp16
__global__ void scatter_cls_kernel(const float* __restrict__ dcls, void* __restrict__ g, int R, int d, int stride) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        if (FMT == 1) ((float*)g)[(size_t)r * stride * d + c] = dcls[(size_t)r * d + c];
        else if (FMT == 2) ((_Float16*)g)[(size_t)r * stride * d + c] = (_Float16)dcls[(size_t)r * d + c];
        else ((bf16_t*)g)[(size_t)r * stride * d + c] = f2bf(dcls[(size_t)r * d + c]);
    }
}

}  // namespace

// run f(integral_constant<int, DC>, bool_constant<DROP>) with DC = d when d is one of the encoder widths, else 0 (run-time d)
template <class F>
static void ln_dispatch(int d, bool drop, F&& f) {
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (d == 768) { if (drop) f(std::integral_constant<int, 768>{}, T_{}); else f(std::integral_constant<int, 768>{}, F_{}); }
    else if (d == 1024) { if (drop) f(std::integral_constant<int, 1024>{}, T_{}); else f(std::integral_constant<int, 1024>{}, F_{}); }
    else { if (drop) f(std::integral_constant<int, 0>{}, T_{}); else f(std::integral_constant<int, 0>{}, F_{}); }
}

static inline int ln_bwd_blocks(int T) {
    const int cap = 512;
    int b = (T + 7) / 8;
    return b < cap ? (b ? b : 1) : cap;
}

extern "C" int cldrd_ln_partial_blocks(int T) { return ln_bwd_blocks(T); }

extern "C" int cldrd_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* out, float* mean, float* rstd,
                                   int T, int d, float eps, float* cls_out, int cls_stride, int x_f32, float* out32, int out_f16,
                                   void* out_bf16_copy, void* stream) {
    CLDRD_CHECK(T > 0 && d > 0 && d <= 1024 && d % 4 == 0, "layernorm_fwd: need 0 < d <= 1024, d % 4 == 0");
    CLDRD_CHECK(out_bf16_copy == nullptr || out_f16, "layernorm_fwd: the bf16 copy goes with an fp16 output");
    CLDRD_CHECK(x_f32 || out32 == nullptr, "layernorm_fwd: an fp32 output copy goes with an fp32 input (the fp32 residual stream)");
    ln_dispatch(d, x_f32 != 0, [&](auto dc, auto x32) {
        hipLaunchKernelGGL((ln_fwd_kernel<decltype(dc)::value, decltype(x32)::value>), dim3((T + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, beta,
                           (bf16_t*)out, out32, mean, rstd, T, d, eps, cls_out, cls_stride > 0 ? cls_stride : 1, out_f16, (bf16_t*)out_bf16_copy);
    });
    CLDRD_LAUNCH_CHECK();
    return 0;
}

extern "C" int cldrd_embed_ln_fwd(const long long* ids, const float* word, const float* pos, const float* type0,
                                  const float* gamma, const float* beta, void* out, float* mean, float* rstd, int T, int L,
                                  int d, int vocab, float eps, float dropout_p, unsigned long long seed, float* out32, int out_f16,
                                  const int* pos_idx, void* out_bf16_copy, void* stream) {
    CLDRD_CHECK(out_bf16_copy == nullptr || out_f16, "embed_ln_fwd: the bf16 copy goes with an fp16 output");
    CLDRD_CHECK(T > 0 && d > 0 && d <= 1024 && d % 4 == 0 && L > 0, "embed_ln_fwd: bad shape");
    const uint32_t th = dropout_p > 0.f ? dropout_thresh16(dropout_p) : 0u;
    ln_dispatch(d, th != 0, [&](auto dc, auto dr) {
        hipLaunchKernelGGL((embed_ln_fwd_kernel<decltype(dc)::value, decltype(dr)::value>), dim3((T + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                           (const int64_t*)ids, word, pos, type0, gamma, beta, (bf16_t*)out, out32, mean, rstd, T, L, d, vocab, eps, th,
                           1.0f / (1.0f - dropout_p), seed_arg(seed), out_f16, pos_idx, (bf16_t*)out_bf16_copy);
    });
    CLDRD_LAUNCH_CHECK();
    return 0;
}

// The same reduction for up to LN_GROUP_MAX LayerNorms in one launch (blockIdx.y = job): the parameter gradients of a LayerNorm are
// not on the backward's critical path, so the trainer parks the per-block partials of every cldrd_layernorm_bwd call of a tower
// and reduces them together (one launch instead of one 7-us launch per LayerNorm behind every ln_bwd_kernel).
constexpr int LN_GROUP_MAX = 32;
struct LnReduceGroup {
    const float* partial[LN_GROUP_MAX];
    float* out[LN_GROUP_MAX][3];
    int nblk[LN_GROUP_MAX];
};
__global__ __launch_bounds__(1024) void reduce_partials_group_kernel(LnReduceGroup g, int n, int seg, int accumulate, float* __restrict__ sq_out) {
    constexpr int RG = 16;
    __shared__ float red[RG][64];
    const float* __restrict__ partial = g.partial[blockIdx.y];
    const int nblk = g.nblk[blockIdx.y];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f;
    if (c < n) {                                 // same summation order as reduce_partials_kernel: identical bits
        int b = rg;
        for (; b + RG < nblk; b += 2 * RG) { s0 += partial[(size_t)b * n + c]; s1 += partial[(size_t)(b + RG) * n + c]; }
        if (b < nblk) s0 += partial[(size_t)b * n + c];
    }
    red[rg][threadIdx.x & 63] = s0 + s1;
    __syncthreads();
    if (rg != 0) return;           // wave 0 from here on (whole: lanes past n carry zeros)
    float v = 0.f;
    if (c < n) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < RG; i += 4) s += (red[i][threadIdx.x] + red[i + 1][threadIdx.x]) + (red[i + 2][threadIdx.x] + red[i + 3][threadIdx.x]);
        float* out = g.out[blockIdx.y][c < seg ? 0 : (c < 2 * seg ? 1 : 2)];
        if (out) {
            const int i = c % seg;
            v = accumulate ? out[i] + s : s;
            out[i] = v;
        }
    }
    if (sq_out) {                  // the clip norm's share of these 64 columns (capi.hip: cldrd_set_norm_sink)
        const float q = wave_sum(v * v);
        if (threadIdx.x == 0) sq_out[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = q;
    }
}

static int launch_reduce(const float* partial, int nblk, int d, float* o0, float* o1, float* o2, int accumulate, hipStream_t st) {
    if (!o0 && !o1 && !o2) return 0;             // deferred: the caller reduces `partial` later (cldrd_ln_reduce_group)
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((3 * d + 63) / 64), dim3(1024), 0, st, partial, nblk, 3 * d, o0, o1, o2, d, accumulate);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

// partial must hold cldrd_ln_partial_blocks(T) * 3 * d floats.  dgamma/dbeta/dbias are accumulated (+=) when accumulate != 0.
// All three null: the reduction is deferred - `partial` keeps the per-block sums for a later cldrd_ln_reduce_group call.
extern "C" int cldrd_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                   void* dx, void* dx_dropped, float* dgamma, float* dbeta, float* dbias, float* partial, int T,
                                   int d, float dropout_p, unsigned long long seed, int accumulate, int x_f32, const void* dy_branch, void* stream) {
    CLDRD_CHECK(T > 0 && d > 0 && d <= 1024 && d % 4 == 0, "layernorm_bwd: need 0 < d <= 1024, d % 4 == 0");
    CLDRD_CHECK(dy_branch == nullptr || (x_f32 & (2 | 8)), "layernorm_bwd: dy_branch goes with the fp32 / fp16 gradient stream (x_f32 bit 1 / 3)");
    // x_f32 bit 2 (round 4): dx_dropped and dy_branch are fp16, not bf16 (the all-fp16 training mode)
    // x_f32 bit 3 (round 5): dy and dx are FP16 rows (the fp16 gradient stream; implies bit 2); dx_dropped may be null when no dropout
    //   separates the stream from the MFMA operand - dx then serves as both
    const bool g_f16 = (x_f32 & 8) != 0;
    const int h16 = ((x_f32 & 4) || g_f16) ? 1 : 0;
    const float* inv_scale = g_cldrd_loss_scale ? g_cldrd_loss_scale + 1 : nullptr;
    // x_f32: bit 0 = x holds fp32 pre-LN sums; bit 1 = dy and dx are fp32 rows (fp32 gradient stream; dx_dropped stays 16-bit and is required:
    // it is the MFMA operand of the next data-gradient GEMM)
    const bool g_f32 = (x_f32 & 2) != 0;
    CLDRD_CHECK(!(g_f32 && g_f16), "layernorm_bwd: the gradient stream is fp32 or fp16, not both");
    CLDRD_CHECK(!g_f32 || ((x_f32 & 1) && dx_dropped != nullptr), "layernorm_bwd: the fp32 gradient stream needs fp32 x and the 16-bit operand copy (dx_dropped)");
    CLDRD_CHECK(!g_f16 || (x_f32 & 1), "layernorm_bwd: the fp16 gradient stream needs fp32 x (the pre-LN sums of the fp32 residual stream)");
    const int nb = ln_bwd_blocks(T);
    const uint32_t th = dropout_p > 0.f ? dropout_thresh16(dropout_p) : 0u;
    ln_dispatch(d, th != 0, [&](auto dc, auto dr) {
        constexpr int DCV = decltype(dc)::value;
        constexpr bool DRV = decltype(dr)::value;
        const size_t lds = (512 / 128) * 3 * MAX_IT * 64 * sizeof(float4);
        const float sc = 1.0f / (1.0f - dropout_p);
        hipStream_t st = (hipStream_t)stream;
        bf16_t* d2 = (bf16_t*)dx_dropped;
        if (g_f16)
            hipLaunchKernelGGL((ln_bwd_kernel<DCV, DRV, true, 2>), dim3(nb), dim3(512), lds, st, dy, x, mean, rstd, gamma, dx, d2, partial, T, d, th, sc, seed_arg(seed), (const bf16_t*)dy_branch, 1, inv_scale);
        else if (g_f32)     // fp32 gradient stream: only with the fp32 pre-LN sums of the fp32 residual stream
            hipLaunchKernelGGL((ln_bwd_kernel<DCV, DRV, true, 1>), dim3(nb), dim3(512), lds, st, dy, x, mean, rstd, gamma, dx, d2, partial, T, d, th, sc, seed_arg(seed), (const bf16_t*)dy_branch, h16, inv_scale);
        else if (x_f32 & 1)
            hipLaunchKernelGGL((ln_bwd_kernel<DCV, DRV, true>), dim3(nb), dim3(512), lds, st, dy, x, mean, rstd, gamma, dx, d2, partial, T, d, th, sc, seed_arg(seed), (const bf16_t*)nullptr, 0, inv_scale);
        else
            hipLaunchKernelGGL((ln_bwd_kernel<DCV, DRV, false>), dim3(nb), dim3(512), lds, st, dy, x, mean, rstd, gamma, dx, d2, partial, T, d, th, sc, seed_arg(seed), (const bf16_t*)nullptr, 0, inv_scale);
    });
    CLDRD_LAUNCH_CHECK();
    return launch_reduce(partial, nb, d, dgamma, dbeta, dbias, accumulate, (hipStream_t)stream);
}

// dgamma[i] / dbeta[i] / dbias[i] (each may be null) (+)= column sums of partial[i] = the scratch a cldrd_layernorm_bwd call with all
// three outputs null left behind (T[i] rows went into it).  Bit-identical to the reduction that call would have run itself.
extern "C" int cldrd_ln_reduce_group(const float* const* partial, const int* T, float* const* dgamma, float* const* dbeta,
                                     float* const* dbias, int n, int d, int accumulate, void* stream) {
    CLDRD_CHECK(n > 0 && d > 0 && d <= 1024 && d % 4 == 0, "ln_reduce_group: bad arguments");
    for (int lo = 0; lo < n; lo += LN_GROUP_MAX) {
        const int m = n - lo < LN_GROUP_MAX ? n - lo : LN_GROUP_MAX;
        LnReduceGroup g;
        for (int i = 0; i < m; ++i) {
            CLDRD_CHECK(partial[lo + i] != nullptr && T[lo + i] > 0, "ln_reduce_group: empty job");
            g.partial[i] = partial[lo + i];
            g.nblk[i] = ln_bwd_blocks(T[lo + i]);
            g.out[i][0] = dgamma[lo + i]; g.out[i][1] = dbeta[lo + i]; g.out[i][2] = dbias[lo + i];
        }
        float* sq = cldrd_norm_sink_take(((3 * d + 63) / 64) * m);      // clip-norm partial sums of what this launch writes, when a sink is set
        hipLaunchKernelGGL(reduce_partials_group_kernel, dim3((3 * d + 63) / 64, m), dim3(1024), 0, (hipStream_t)stream, g, 3 * d, d, accumulate, sq);
        CLDRD_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int cldrd_embed_ln_bwd(const void* dy, const long long* ids, const float* word, const float* pos, const float* type0,
                                  const float* gamma, const float* mean, const float* rstd, float* dword, float* dpos,
                                  float* dtype0, float* dgamma, float* dbeta, float* partial, int T, int L, int d, int vocab,
                                  float dropout_p, unsigned long long seed, int accumulate, const int* pos_idx, int dy_f32, const void* dy_branch,
                                  void* stream) {
    CLDRD_CHECK(T > 0 && d > 0 && d <= 1024 && d % 4 == 0 && L > 0, "embed_ln_bwd: bad shape");
    // dy_f32: bit 0 = dy is fp32; bit 2 = dy_branch is fp16; bit 3 (round 5) = dy is FP16 (the fp16 gradient stream; dy_branch fp16 then)
    CLDRD_CHECK(dy_branch == nullptr || (dy_f32 & (1 | 8)), "embed_ln_bwd: dy_branch goes with an fp32 / fp16 dy");
    CLDRD_CHECK((dy_f32 & 9) != 9, "embed_ln_bwd: dy is fp32 or fp16, not both");
    int nb = ln_bwd_blocks(T);                      // the caller sized `partial` for this many blocks; fewer is fine
    int g4 = 4, r = L;                              // gcd(4, L)
    while (r) { const int t = g4 % r; g4 = r; r = t; }
    const int step = L / g4;                        // grids that are multiples of this make the wave row stride 4*nb a multiple of L
    const int pos_uniform = nb >= step && pos_idx == nullptr;       // packed rows: a wave's rows sit at different positions
    if (pos_uniform) nb = (nb / step) * step;
    const uint32_t th = dropout_p > 0.f ? dropout_thresh16(dropout_p) : 0u;
    ln_dispatch(d, th != 0, [&](auto dc, auto dr) {
        const size_t lds = (256 / 128) * 3 * MAX_IT * 64 * sizeof(float4);
        if (dy_f32 & 8)
            hipLaunchKernelGGL((embed_ln_bwd_kernel<decltype(dc)::value, decltype(dr)::value, 2>), dim3(nb), dim3(256), lds,
                               (hipStream_t)stream, dy, (const int64_t*)ids, word, pos, type0, gamma, mean, rstd, dword, dpos, partial,
                               T, L, d, vocab, th, 1.0f / (1.0f - dropout_p), seed_arg(seed), pos_uniform, pos_idx, (const bf16_t*)dy_branch,
                               1, g_cldrd_loss_scale ? g_cldrd_loss_scale + 1 : nullptr);
        else if (dy_f32 & 1)
            hipLaunchKernelGGL((embed_ln_bwd_kernel<decltype(dc)::value, decltype(dr)::value, 1>), dim3(nb), dim3(256), lds,
                               (hipStream_t)stream, dy, (const int64_t*)ids, word, pos, type0, gamma, mean, rstd, dword, dpos, partial,
                               T, L, d, vocab, th, 1.0f / (1.0f - dropout_p), seed_arg(seed), pos_uniform, pos_idx, (const bf16_t*)dy_branch,
                               (dy_f32 & 4) ? 1 : 0, g_cldrd_loss_scale ? g_cldrd_loss_scale + 1 : nullptr);
        else
            hipLaunchKernelGGL((embed_ln_bwd_kernel<decltype(dc)::value, decltype(dr)::value>), dim3(nb), dim3(256), lds,
                               (hipStream_t)stream, dy, (const int64_t*)ids, word, pos, type0, gamma, mean, rstd, dword, dpos, partial,
                               T, L, d, vocab, th, 1.0f / (1.0f - dropout_p), seed_arg(seed), pos_uniform, pos_idx, (const bf16_t*)nullptr, 0,
                               g_cldrd_loss_scale ? g_cldrd_loss_scale + 1 : nullptr);
    });
    CLDRD_LAUNCH_CHECK();
    return launch_reduce(partial, nb, d, dgamma, dbeta, dtype0, accumulate, (hipStream_t)stream);
}

// bias gradient: out[N] (+)= column sums of x[T, N].  partial must hold ceil(T/128) * N floats.
extern "C" int cldrd_colsum_bf16(const void* x, float* out, float* partial, int T, int N, int ld, int accumulate, void* stream) {
    CLDRD_CHECK(T > 0 && N > 0 && N % 4 == 0 && ld % 4 == 0, "colsum: N and ld must be multiples of 4");
    const int rows = 128;
    const int ny = (T + rows - 1) / rows;
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 1023) / 1024, ny), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, partial, T, N, ld, rows);
    CLDRD_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((N + 63) / 64), dim3(1024), 0, (hipStream_t)stream, (const float*)partial, ny, N,
                       out, (float*)nullptr, (float*)nullptr, N, accumulate);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

extern "C" int cldrd_scatter_cls_grad(const float* dcls, void* g, int R, int d, int stride, int T, int g_f32, void* stream) {
    CLDRD_CHECK(R > 0 && d > 0 && stride > 0 && (long long)R * stride <= (long long)T + stride - 1, "scatter_cls_grad: bad shape");
    // g_f32: 0 = bf16 rows, 1 = fp32, 2 = fp16
    if (hipMemsetAsync(g, 0, (size_t)T * d * (g_f32 == 1 ? sizeof(float) : sizeof(bf16_t)), (hipStream_t)stream) != hipSuccess) return cldrd_set_error("scatter_cls_grad: memset failed");
    if (g_f32 == 1) hipLaunchKernelGGL(scatter_cls_kernel<1>, dim3(R), dim3(256), 0, (hipStream_t)stream, dcls, g, R, d, stride);
    else if (g_f32 == 2) hipLaunchKernelGGL(scatter_cls_kernel<2>, dim3(R), dim3(256), 0, (hipStream_t)stream, dcls, g, R, d, stride);
    else hipLaunchKernelGGL(scatter_cls_kernel<0>, dim3(R), dim3(256), 0, (hipStream_t)stream, dcls, g, R, d, stride);
    CLDRD_LAUNCH_CHECK();
    return 0;
}
