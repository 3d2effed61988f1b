// N-way scoring and the list-wise distillation losses, forward + analytic backward, fp32.
//
// Reference: models/nway_dual_encoder.py:30-47 (dot-product scoring, optional in-batch negatives),
// losses/kl_div.py:11-22, losses/margin_mse.py:8-19, losses/ranknet.py:3-44, losses/lambda_rank.py:3-96
// (SURVEY.md K6-K10).  The reference materialises three [B,N,N] tensors, sorts, and synchronises with the
// host twice per call (ranknet.py:16,40); here one workgroup owns one row: ranks come from an in-LDS count
// (stable: ties by index), the pair loop runs over LDS, loss and pair count are wave-reduced, and the
// gradient is written directly (no autograd graph, no host sync).
#include "common.h"

namespace {

enum { LOSS_KL = 0, LOSS_MSE = 1, LOSS_RANKNET = 2, LOSS_LAMBDA = 3, LOSS_WPOINT = 4 };

__device__ __forceinline__ float NEG_INF_F() { return -__builtin_inff(); }

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) s = fmaxf(s, red[w]);
    return s;
}

// idx of the passage behind logit column j of row b.  mode 0: own N; 1: own N then every other sample's passages in
// global order; 2: own N then the next sample's N (cyclic)   (reference models/nway_dual_encoder.py:30-44)
__device__ __forceinline__ int col_to_passage(int mode, int b, int j, int B, int N) {
    if (j < N) return b * N + j;
    const int jj = j - N;
    if (mode == 1) return jj < b * N ? jj : jj + N;
    return ((b + 1) % B) * N + jj;
}

__global__ __launch_bounds__(256) void score_fwd_kernel(const float* __restrict__ q, const float* __restrict__ p,
                                                         float* __restrict__ logits, int B, int N, int Np, int d, int mode) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= B * Np) return;
    const int lane = threadIdx.x & 63;
    const int b = o / Np, j = o % Np;
    const float* qr = q + (size_t)b * d;
    const float* pr = p + (size_t)col_to_passage(mode, b, j, B, N) * d;
    float s = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        const float4 a = *(const float4*)(qr + c), w = *(const float4*)(pr + c);
        s += a.x * w.x + a.y * w.y + a.z * w.z + a.w * w.w;
    }
    s = wave_sum(s);
    if (lane == 0) logits[o] = s;
}

// dq[b] = sum_j dl[b][j] * p[passage(b, j)]
__global__ __launch_bounds__(256) void score_bwd_q_kernel(const float* __restrict__ dl, const float* __restrict__ p,
                                                           float* __restrict__ dq, int B, int N, int Np, int d, int mode) {
    // grid (B, column chunks of 256): with one block per query a thread walked its Np passages one dependent load after the other
    // (26 us at B = 8, N = 32 on the step's critical path between the loss and the first backward GEMM); the loads of eight passages are
    // now in flight together.  Same summation order (j ascending): same bits.
    const int b = blockIdx.x;
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= d) return;
    float s = 0.f;
    int j = 0;
    for (; j + 8 <= Np; j += 8) {
        float v[8], w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { w[u] = dl[(size_t)b * Np + j + u]; v[u] = p[(size_t)col_to_passage(mode, b, j + u, B, N) * d + c]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += w[u] * v[u];
    }
    for (; j < Np; ++j) s += dl[(size_t)b * Np + j] * p[(size_t)col_to_passage(mode, b, j, B, N) * d + c];
    dq[(size_t)b * d + c] = s;
}
// dp[m] = sum over (b, j) with passage(b, j) == m of dl[b][j] * q[b]
__global__ __launch_bounds__(256) void score_bwd_p_kernel(const float* __restrict__ dl, const float* __restrict__ q,
                                                           float* __restrict__ dp, int B, int N, int Np, int d, int mode) {
    const int m = blockIdx.x;
    const int bo = m / N, o = m % N;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        float s = dl[(size_t)bo * Np + o] * q[(size_t)bo * d + c];
        if (mode == 1) {
            for (int b = 0; b < B; ++b)
                if (b != bo) s += dl[(size_t)b * Np + N + (m < b * N ? m : m - N)] * q[(size_t)b * d + c];
        } else if (mode == 2 && B > 0) {
            const int b = (bo - 1 + B) % B;
            s += dl[(size_t)b * Np + N + o] * q[(size_t)b * d + c];
        }
        dp[(size_t)m * d + c] = s;
    }
}

// One block per row.  Writes the un-normalised gradient and the row's {loss sum, pair count}.
__global__ __launch_bounds__(256) void loss_row_kernel(int kind, const float* __restrict__ y_pred, const float* __restrict__ y_true,
                                                        const float* __restrict__ bweight, float* __restrict__ grad,
                                                        float* __restrict__ row_out, int B, int N, float T, float pad) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s = sm;            // [N]
    float* t = s + N;         // [N]
    float* rk = t + N;        // [N] 1/rank
    __shared__ float red[4];
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < N; i += blockDim.x) { s[i] = y_pred[(size_t)b * N + i]; t[i] = y_true[(size_t)b * N + i]; }
    __syncthreads();
    float lsum = 0.f, cnt = 0.f;
    if (kind == LOSS_KL) {
        const float iT = 1.0f / T;
        float ms = NEG_INF_F(), mt = ms;
        for (int i = threadIdx.x; i < N; i += blockDim.x) { ms = fmaxf(ms, s[i] * iT); mt = fmaxf(mt, t[i] * iT); }
        ms = block_max(ms, red); mt = block_max(mt, red);
        float es = 0.f, et = 0.f;
        for (int i = threadIdx.x; i < N; i += blockDim.x) { es += __expf(s[i] * iT - ms); et += __expf(t[i] * iT - mt); }
        es = block_sum(es, red); et = block_sum(et, red);
        const float lzs = ms + __logf(es), lzt = mt + __logf(et);
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            const float ls = s[i] * iT - lzs, lt = t[i] * iT - lzt;
            const float pt = __expf(lt);
            lsum += pt * (lt - ls);
            grad[(size_t)b * N + i] = (__expf(ls) - pt) * iT / (float)B;
        }
        cnt = 0.f;
    } else if (kind == LOSS_MSE) {
        float sd = 0.f;
        for (int i = threadIdx.x; i < N; i += blockDim.x) sd += s[i] - t[i];
        sd = block_sum(sd, red);
        const float mean = sd / (float)N;
        const float k = 4.0f / ((float)B * (float)N);     // 4/(B N^2) * N
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            const float dc = (s[i] - t[i]) - mean;
            lsum += dc * dc;
            grad[(size_t)b * N + i] = k * dc;
        }
        lsum *= 2.0f / ((float)B * (float)N);              // 2/(B N^2) * N * sum (d - mean)^2
    } else if (kind == LOSS_WPOINT) {
        // weighted_pointwise_loss (reference losses/weighted_pointwise.py:3-14): mean over B*N of softplus(-y/T) * weight
        const float iT = 1.0f / T, k = 1.0f / ((float)B * (float)N);
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            const float z = -s[i] * iT, w = t[i];
            const float e = __expf(-fabsf(z));
            lsum += (log1pf(e) + fmaxf(z, 0.f)) * w;
            grad[(size_t)b * N + i] = -(z > 0.f ? 1.f / (1.f + e) : e / (1.f + e)) * iT * w * k;      // -sigmoid(z)/T * w / (B N)
        }
        lsum *= k;
    } else {
        const bool by_rank = (kind == LOSS_LAMBDA);
        if (by_rank) {
            for (int i = threadIdx.x; i < N; i += blockDim.x) {
                const bool pi = t[i] == pad;
                const float ki = pi ? NEG_INF_F() : s[i];
                int r = 1;
                for (int j = 0; j < N; ++j) {
                    const float kj = (t[j] == pad) ? NEG_INF_F() : s[j];
                    r += (kj > ki) || (kj == ki && j < i);
                }
                rk[i] = 1.0f / (float)r;
            }
            __syncthreads();
        }
        const float bw = bweight ? bweight[b] : 1.0f;
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            const float si = s[i], ti = t[i];
            float gi = 0.f;
            if (ti != pad) {
                for (int j = 0; j < N; ++j) {
                    const float tj = t[j];
                    if (tj == pad || tj == ti) continue;
                    const float w = (by_rank ? fabsf(rk[i] - rk[j]) : 1.0f) * bw;
                    if (ti > tj) {            // pair (i, j)
                        const float df = fminf(fmaxf(si - s[j], -1e8f), 1e8f);
                        const float e = __expf(-fabsf(df));
                        lsum += w * (log1pf(e) + fmaxf(-df, 0.f));
                        cnt += 1.f;
                        gi -= w * (df > 0.f ? e / (1.f + e) : 1.f / (1.f + e));      // sigma(-df)
                    } else {                  // pair (j, i)
                        const float df = fminf(fmaxf(s[j] - si, -1e8f), 1e8f);
                        const float e = __expf(-fabsf(df));
                        gi += w * (df > 0.f ? e / (1.f + e) : 1.f / (1.f + e));
                    }
                }
            }
            grad[(size_t)b * N + i] = gi;
        }
    }
    lsum = block_sum(lsum, red);
    cnt = block_sum(cnt, red);
    if (threadIdx.x == 0) { row_out[2 * b] = lsum; row_out[2 * b + 1] = cnt; }
}

// loss_out[0] = loss, loss_out[1] = pair count; scales grad for the 'mean' reduction of the pairwise losses.
__global__ __launch_bounds__(256) void loss_finalize_kernel(int kind, const float* __restrict__ row_out, float* __restrict__ grad,
                                                             float* __restrict__ loss_out, int B, int N, int mean_reduction) {
    __shared__ float tot[2];
    if (threadIdx.x == 0) {
        float l = 0.f, c = 0.f;
        for (int b = 0; b < B; ++b) { l += row_out[2 * b]; c += row_out[2 * b + 1]; }
        tot[0] = l; tot[1] = c;
    }
    __syncthreads();
    const float l = tot[0], c = tot[1];
    if (kind == LOSS_KL) {
        if (threadIdx.x == 0) { loss_out[0] = l / (float)B; loss_out[1] = 0.f; }
        return;
    }
    if (kind == LOSS_MSE || kind == LOSS_WPOINT) {
        if (threadIdx.x == 0) { loss_out[0] = l; loss_out[1] = 0.f; }
        return;
    }
    if (mean_reduction) {
        const float inv = 1.0f / c;           // c == 0 -> inf/nan, as torch.mean of an empty selection
        if (threadIdx.x == 0) { loss_out[0] = c > 0.f ? l * inv : __int_as_float(0x7fc00000); loss_out[1] = c; }
        for (int i = threadIdx.x; i < B * N; i += blockDim.x) grad[i] = c > 0.f ? grad[i] * inv : __int_as_float(0x7fc00000);
    } else if (threadIdx.x == 0) {
        loss_out[0] = l; loss_out[1] = c;
    }
}

// LambdaLoss framework (reference losses/standard_lambda_rank.py:3-117, allRank's lambda_loss): one block per slate.
// Everything lives in "sorted by prediction" positions p = 0..N-1 (padded items sort last: their keys are -inf):
//   pair (p, q) counts iff both are real items, p < k and q < k, and (except ndcgLoss1) true[p] > true[q];
//   term = -log( clamp( clamp(sigmoid(sigma (s_p - s_q)), eps) ^ W_pq, eps) ), natural or base-2 log;
//   W per weighing scheme from gains G = (2^true - 1 | true - 1) / maxDCG@k and discounts D_p = log2(2 + p).
// The gradient flows through the score difference only (sort indices and weights are constants for autograd); clamp passes
// the gradient where its input is >= the bound, as torch.clamp does.
enum { LL_NONE = 0, LL_NDCG1 = 1, LL_NDCG2 = 2, LL_LAMBDARANK = 3, LL_NDCG2PP = 4, LL_RANKNET = 5, LL_GTDIFF = 6, LL_GTDIFF_POW = 7 };

__global__ __launch_bounds__(256) void lambda_loss_row_kernel(const float* __restrict__ y_pred, const float* __restrict__ y_true,
                                                               float* __restrict__ grad, float* __restrict__ row_out, int N, int scheme,
                                                               int k, float eps, float sigma, float mu, float pad, int log2_red,
                                                               int gain_linear) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* ps = sm;                  // [N] prediction at sorted position (padded: -inf)
    float* ts = ps + N;              // [N] label at sorted position (padded: -inf)
    float* G = ts + N;               // [N] gain at sorted position
    float* gp = G + N;               // [N] gradient at sorted position
    int* item = (int*)(gp + N);      // [N] original column at sorted position
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float* yp = y_pred + (size_t)b * N;
    const float* yt = y_true + (size_t)b * N;
    const float NINF = NEG_INF_F();
    // ---- positions: stable descending order of the (masked) predictions; maxDCG from the descending labels
    float dcg = 0.f;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const bool padded = yt[i] == pad;
        const float si = padded ? NINF : yp[i], ti = padded ? NINF : yt[i];
        int pos_s = 0, pos_t = 0;
        for (int j = 0; j < N; ++j) {
            const bool pj = yt[j] == pad;
            const float sj = pj ? NINF : yp[j], tj = pj ? NINF : yt[j];
            pos_s += (sj > si) || (sj == si && j < i);
            pos_t += (tj > ti) || (tj == ti && j < i);
        }
        ps[pos_s] = si; ts[pos_s] = ti; item[pos_s] = i;
        if (pos_t < k) {
            const float tc = fmaxf(ti, 0.f);
            dcg += (gain_linear ? tc - 1.f : exp2f(tc) - 1.f) / log2f(2.f + (float)pos_t);
        }
    }
    const float maxdcg = fmaxf(block_sum(dcg, red), eps);      // block_sum synchronises: ps / ts / item are complete after it
    for (int p = threadIdx.x; p < N; p += blockDim.x) {
        const float tc = fmaxf(ts[p], 0.f);
        G[p] = (gain_linear ? tc - 1.f : exp2f(tc) - 1.f) / maxdcg;
    }
    __syncthreads();
    const float logk = log2_red ? 1.4426950408889634f : 1.0f;
    auto weight = [&](int p, int q) -> float {
        const float Dp = log2f(2.f + (float)p), Dq = log2f(2.f + (float)q);
        float lr = 0.f, n2 = 0.f;
        if (scheme == LL_LAMBDARANK || scheme == LL_NDCG2PP) lr = fabsf(1.f / Dp - 1.f / Dq) * fabsf(G[p] - G[q]);
        if (scheme == LL_NDCG2 || scheme == LL_NDCG2PP) {
            const int dl = p > q ? p - q : q - p;
            n2 = dl == 0 ? 0.f : fabsf(1.f / log2f(1.f + (float)dl) - 1.f / log2f(2.f + (float)dl)) * fabsf(G[p] - G[q]);
        }
        switch (scheme) {
            case LL_NDCG1: return G[p] / Dp;
            case LL_NDCG2: return n2;
            case LL_LAMBDARANK: return lr;
            case LL_NDCG2PP: return mu * n2 + lr;
            case LL_GTDIFF: return fabsf(fmaxf(ts[p], 0.f) - fmaxf(ts[q], 0.f));
            case LL_GTDIFF_POW: { const float a = fmaxf(ts[p], 0.f), c = fmaxf(ts[q], 0.f); return fabsf(a * a - c * c); }
            default: return 1.f;
        }
    };
    // term(p, q) and d term / d (s_p - s_q)
    auto term = [&](int p, int q, float& dterm) -> float {
        const float W = weight(p, q);
        const float d = fminf(fmaxf(ps[p] - ps[q], -1e8f), 1e8f);
        const float u = 1.f / (1.f + __expf(-sigma * d));
        const float a = fmaxf(u, eps);
        const float bw = powf(a, W);
        const float c = fmaxf(bw, eps);
        // d(-log c)/dd = -(1/c) [bw >= eps] W a^(W-1) [u >= eps] sigma u (1-u), and the clamp of d passes inside (-1e8, 1e8)
        const bool live = bw >= eps && u >= eps && fabsf(ps[p] - ps[q]) <= 1e8f;
        dterm = live ? -logk * (W * bw / a) / c * sigma * u * (1.f - u) : 0.f;
        return -logk * __logf(c);
    };
    const int kk = k < N ? k : N;
    float lsum = 0.f, cnt = 0.f;
    for (int p = threadIdx.x; p < N; p += blockDim.x) {
        float g = 0.f;
        if (p < kk && ts[p] != NINF) {
            for (int q = 0; q < kk; ++q) {
                if (q == p && scheme != LL_NDCG1) continue;
                if (ts[q] == NINF) continue;
                const float td = ts[p] - ts[q];
                float dt;
                if (scheme == LL_NDCG1 || td > 0.f) {          // pair (p, q): p is the first index
                    lsum += term(p, q, dt);
                    cnt += 1.f;
                    if (q != p) g += dt;                       // (p, p) only exists for ndcgLoss1: s_p - s_p has no gradient
                }
                if (q != p && (scheme == LL_NDCG1 || td < 0.f)) {          // pair (q, p): p is the second index
                    (void)term(q, p, dt);
                    g -= dt;
                }
            }
        }
        gp[p] = g;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < N; p += blockDim.x) grad[(size_t)b * N + item[p]] = gp[p];
    lsum = block_sum(lsum, red);
    cnt = block_sum(cnt, red);
    if (threadIdx.x == 0) { row_out[2 * b] = lsum; row_out[2 * b + 1] = cnt; }
}

// Logit-norm regulariser of the multistep-curriculum trainers (reference nway_listwise_1.py:348-350):
//   reg = lambda * ||logits||_2 over the whole [B, N] matrix;  loss_out[0] += reg;  grad += lambda * logits / ||logits||_2;
//   reg_out = reg (the trainer logs it and its ratio to the loss).  One workgroup: B * N is a few hundred numbers.
__global__ __launch_bounds__(256) void logit_reg_kernel(const float* __restrict__ logits, int n, float lambda, float* __restrict__ loss_out,
                                                         float* __restrict__ grad, float* __restrict__ reg_out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += logits[i] * logits[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    const float inv = norm > 0.f ? lambda / norm : 0.f;      // d||x||/dx at 0: torch gives 0 (subgradient)
    for (int i = threadIdx.x; i < n; i += blockDim.x) grad[i] += inv * logits[i];
    if (threadIdx.x == 0) {
        loss_out[0] += lambda * norm;
        if (reg_out) *reg_out = lambda * norm;
    }
}

}  // namespace

extern "C" int cldrd_score_fwd(const float* q, const float* p, float* logits, int B, int N, int d, int mode, void* stream) {
    CLDRD_CHECK(B > 0 && N > 0 && d > 0 && d % 4 == 0 && mode >= 0 && mode <= 2, "score_fwd: bad arguments");
    const int Np = mode == 0 ? N : (mode == 1 ? B * N : 2 * N);
    hipLaunchKernelGGL(score_fwd_kernel, dim3((B * Np + 3) / 4), dim3(256), 0, (hipStream_t)stream, q, p, logits, B, N, Np, d, mode);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

extern "C" int cldrd_score_bwd(const float* dlogits, const float* q, const float* p, float* dq, float* dp, int B, int N, int d,
                               int mode, void* stream) {
    CLDRD_CHECK(B > 0 && N > 0 && d > 0 && mode >= 0 && mode <= 2, "score_bwd: bad arguments");
    const int Np = mode == 0 ? N : (mode == 1 ? B * N : 2 * N);
    hipLaunchKernelGGL(score_bwd_q_kernel, dim3(B, (d + 255) / 256), dim3(256), 0, (hipStream_t)stream, dlogits, p, dq, B, N, Np, d, mode);
    CLDRD_LAUNCH_CHECK();
    hipLaunchKernelGGL(score_bwd_p_kernel, dim3(B * N), dim3(256), 0, (hipStream_t)stream, dlogits, q, dp, B, N, Np, d, mode);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

// kind: 0 KLDiv(T), 1 MarginMSE, 2 ranknet, 3 lambda_mrr (batch_weight != null -> bweight_lambda_mrr).
// loss_out: device float[2] = {loss, number of valid pairs}; grad: device [B, N] = d loss / d y_pred;
// workspace: device float[2*B].
extern "C" int cldrd_loss_fwd_bwd(int kind, const float* y_pred, const float* y_true, const float* batch_weight, float* loss_out,
                                  float* grad, float* workspace, int B, int N, float T, float pad_indicator, int mean_reduction,
                                  void* stream) {
    CLDRD_CHECK(kind >= 0 && kind <= 4, "loss: unknown kind");
    CLDRD_CHECK(B > 0 && N > 0 && N <= 8192, "loss: need 0 < N <= 8192");
    CLDRD_CHECK(T > 0.f, "loss: T must be positive");
    hipLaunchKernelGGL(loss_row_kernel, dim3(B), dim3(256), 3 * N * sizeof(float), (hipStream_t)stream, kind, y_pred, y_true,
                       batch_weight, grad, workspace, B, N, T, pad_indicator);
    CLDRD_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, kind, (const float*)workspace, grad, loss_out, B, N,
                       mean_reduction);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

// loss_out[0] += lambda * ||logits||_2, grad += its gradient, *reg_out = the added term (reference nway_listwise_1.py:348-350).
// Call after cldrd_loss_fwd_bwd on the same stream.
extern "C" int cldrd_logit_norm_reg(const float* logits, int n, float reg_lambda, float* loss_out, float* grad, float* reg_out,
                                    void* stream) {
    CLDRD_CHECK(n > 0 && reg_lambda >= 0.f, "logit_norm_reg: bad arguments");
    hipLaunchKernelGGL(logit_reg_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, n, reg_lambda, loss_out, grad, reg_out);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

// lambda_loss of reference losses/standard_lambda_rank.py:3-95.  scheme: 0 None, 1 ndcgLoss1, 2 ndcgLoss2, 3 lambdaRank,
// 4 ndcgLoss2PP, 5 rankNet, 6 rankNetWeightedByGTDiff, 7 rankNetWeightedByGTDiffPowed; k <= 0 = no truncation.
// loss_out[2] = {loss, number of pairs}; grad [B, N]; workspace 2*B floats.
extern "C" int cldrd_lambda_loss_fwd_bwd(const float* y_pred, const float* y_true, float* loss_out, float* grad, float* workspace,
                                         int B, int N, int scheme, int k, float eps, float sigma, float mu, float pad_indicator,
                                         int mean_reduction, int log2_reduction, int gain_linear, void* stream) {
    CLDRD_CHECK(B > 0 && N > 0 && N <= 4096, "lambda_loss: need 0 < N <= 4096");
    CLDRD_CHECK(scheme >= 0 && scheme <= 7, "lambda_loss: unknown weighing scheme");
    hipLaunchKernelGGL(lambda_loss_row_kernel, dim3(B), dim3(256), 5 * N * sizeof(float), (hipStream_t)stream, y_pred, y_true, grad,
                       workspace, N, scheme, k > 0 ? k : N, eps, sigma, mu, pad_indicator, log2_reduction, gain_linear);
    CLDRD_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (int)LOSS_RANKNET, (const float*)workspace, grad,
                       loss_out, B, N, mean_reduction);
    CLDRD_LAUNCH_CHECK();
    return 0;
}
