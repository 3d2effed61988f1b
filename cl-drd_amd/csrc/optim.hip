// Optimizer step of the trainer over ONE flat fp32 parameter buffer (both towers), fused:
//   pass 1: global gradient square-norm (deterministic two-stage reduction) -> clip coefficient on device
//   pass 2: clip + legacy-HF AdamW + linear-schedule lr + bf16 shadow weights, one read/write of (p, g, m, v)
//
// Reference: trainer/multistep-curriculum/nway_listwise_1.py:353-367 (unscale_, clip_grad_norm_(1.0), AdamW step,
// scheduler.step) and :258-266 (two param groups: no weight decay for names containing "bias"/"LayerNorm.weight").
// the reference spends ~4 extra passes over 0.5 GB x (p, g, m, v); this is HBM-bound at
// 28 B/param (+2 B/param for the bf16 shadow the MFMA GEMMs consume).  No host synchronisation: the clip
// coefficient stays in device memory.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, size_t n4, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = ((const float4*)g)[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// out[0] = total L2 norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6)), out[2] = 1 if the norm is not finite
// scale_state (optional, device float[8] = {S, 1 / S, good steps since the headroom last changed, skipped steps, headroom exponent h <= 0, ...}):
// the loss scale of the all-fp16 training mode (cldrd_loss_scale_adapt below sets S every step from the gradient that enters the towers).
// HERE - after the backward, the last reader of S in a step, and before AdamW - only the safety net runs, the analogue of
// torch.cuda.amp.GradScaler's skip rule (the reference's scaler, nway_listwise_1.py:355-359): a non-finite gradient norm skips the step
// (AdamW reads out[2]) and gives the next steps 4x more headroom (h -= 2); `interval` finite steps in a row give one factor 2 back.
__global__ void clip_coef_kernel(const float* __restrict__ partial, int nblk, float max_norm, float* __restrict__ out, float* __restrict__ scale_state,
                                 int interval) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(red[0]);
        out[0] = norm;
        const float c = max_norm / (norm + 1e-6f);
        out[1] = (max_norm > 0.f) ? (c < 1.f ? c : 1.f) : 1.f;
        const bool bad = !(norm == norm && norm < 3.0e38f);
        out[2] = bad ? 1.f : 0.f;
        if (scale_state) {
            float good = scale_state[2], h = scale_state[4];
            if (bad) { h = fmaxf(h - 2.f, -24.f); good = 0.f; scale_state[3] += 1.f; }
            else if (++good >= (float)interval) { h = fminf(h + 1.f, 0.f); good = 0.f; }
            scale_state[2] = good; scale_state[4] = h;
        }
    }
}

// ---- loss scale of the all-fp16 training mode ------------------------------------------------------------------------------------
// The activation gradients of that mode are fp16 (csrc/*: every 16-bit tensor of the backward), so they must sit inside fp16's range:
// S = 2^(12 + h - ceil(log2 max|dCLS|)) puts the largest element of the gradient that ENTERS the towers (dL/dCLS of both, fp32, from
// cldrd_score_bwd) at 2^11..2^12 - a factor 16+ below fp16's maximum for what the layers add, 2^26 above its smallest normal number.
// A power of two: scaling and unscaling are exact.  Where the reference's GradScaler finds its scale by overflowing and backing off over
// many steps, this one is recomputed from the data every step; h <= 0 is the safety net's extra headroom (clip_coef_kernel).
constexpr int SCALE_BLOCKS = 64;
__global__ __launch_bounds__(256) void scale_amax_kernel(const float* __restrict__ a, size_t na, const float* __restrict__ b, size_t nb,
                                                        float* __restrict__ scratch) {
    __shared__ float red[4];
    float m = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = i0; i < na; i += stride) m = fmaxf(m, fabsf(a[i]));          // fmaxf drops NaNs: a NaN gradient is caught by the norm later
    for (size_t i = i0; i < nb; i += stride) m = fmaxf(m, fabsf(b[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) scratch[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__global__ __launch_bounds__(256) void scale_apply_kernel(float* __restrict__ a, size_t na, float* __restrict__ b, size_t nb,
                                                         float* __restrict__ state, const float* __restrict__ scratch) {
    float m = 0.f;
    for (int i = threadIdx.x & 63; i < SCALE_BLOCKS; i += 64) m = fmaxf(m, scratch[i]);      // every wave: the same maximum
    m = wave_max(m);
    float S = 1.0f;
    if (m > 0.f && m < 3.0e38f) {
        int e;
        (void)frexpf(m, &e);                               // m = f 2^e, 0.5 <= f < 1: ceil(log2 m) <= e
        int k = 12 - e + (int)state[4];
        k = k < -24 ? -24 : (k > 24 ? 24 : k);
        S = ldexpf(1.0f, k);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { state[0] = S; state[1] = 1.0f / S; }
    const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = i0; i < na; i += stride) a[i] *= S;
    for (size_t i = i0; i < nb; i += stride) b[i] *= S;
}

struct AdamArgs {
    float* p; const float* g; float* m; float* v; const uint8_t* decay; bf16_t* shadow;
    size_t n4; float lr, beta1, beta2, eps, wd, step_size; const float* coef;
    int reverse;
    uint16_t* shadow16; size_t h_lo4, h_hi4;      // fp16 copy of parameters [4 h_lo4, 4 h_hi4) (shadow16[0] = parameter 4 h_lo4), or null
    const float* hyper;                           // device {lr, step_size} replacing the by-value pair (a captured step reads them at replay), or null
};

typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ntload4(const float4* p) {
    const f32x4_t t = __builtin_nontemporal_load((const f32x4_t*)p);
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void ntstore4(float4* p, const float4& v) {
    const f32x4_t t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, (f32x4_t*)p);
}

__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
    const float coef = a.coef ? a.coef[1] : 1.0f;
    const bool skip = a.coef && a.coef[2] != 0.f;        // non-finite gradient norm: leave the weights untouched
    const float lr = a.hyper ? a.hyper[0] : a.lr, step_size = a.hyper ? a.hyper[1] : a.step_size;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < a.n4; i0 += stride) {
        // the sweep runs from the END of the buffer: the norm pass in front of this kernel has just read g front to back, so its
        // last ~256 MB are what the Infinity Cache still holds (elementwise: the order changes no result)
        const size_t i = a.reverse ? a.n4 - 1 - i0 : i0;
        // p, g, m, v are streamed once per step (4 GB): non-temporal, so they do not push the bf16 shadows (read by the next
        // step's GEMMs) and the activations out of the Infinity Cache
        float4 p = ntload4((const float4*)a.p + i);
        if (!skip) {
            const float4 g = ntload4((const float4*)a.g + i);
            float4 m = ntload4((const float4*)a.m + i), v = ntload4((const float4*)a.v + i);
            const bool dec = (a.decay[i >> 4] & 1) != 0;   // flag byte per 64 elements: bit 0 weight decay, bit 1 no 16-bit shadows
            const float gg[4] = {g.x * coef, g.y * coef, g.z * coef, g.w * coef};
            float pp[4] = {p.x, p.y, p.z, p.w}, mm[4] = {m.x, m.y, m.z, m.w}, vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mm[j] = a.beta1 * mm[j] + (1.f - a.beta1) * gg[j];
                vv[j] = a.beta2 * vv[j] + (1.f - a.beta2) * gg[j] * gg[j];
                pp[j] -= step_size * mm[j] / (sqrtf(vv[j]) + a.eps);
                if (dec) pp[j] -= lr * a.wd * pp[j];
            }
            p = make_float4(pp[0], pp[1], pp[2], pp[3]);
            ntstore4((float4*)a.p + i, p);
            ntstore4((float4*)a.m + i, make_float4(mm[0], mm[1], mm[2], mm[3]));
            ntstore4((float4*)a.v + i, make_float4(vv[0], vv[1], vv[2], vv[3]));
        }
        // embedding tables (35 % of the parameters) are read by the embedding kernel in fp32: nobody reads their 16-bit shadows
        const bool no_shadow = (a.decay[i >> 4] & 2) != 0;
        if (a.shadow && !no_shadow) {
            uint2 o; o.x = pack2bf(p.x, p.y); o.y = pack2bf(p.z, p.w);
            ((uint2*)a.shadow)[i] = o;
        }
        if (a.shadow16 && !no_shadow && i >= a.h_lo4 && i < a.h_hi4) {
            const _Float16 h0 = (_Float16)p.x, h1 = (_Float16)p.y, h2 = (_Float16)p.z, h3 = (_Float16)p.w;
            uint2 o;
            o.x = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
            o.y = (uint32_t)__builtin_bit_cast(uint16_t, h2) | ((uint32_t)__builtin_bit_cast(uint16_t, h3) << 16);
            ((uint2*)a.shadow16)[i - a.h_lo4] = o;
        }
    }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 p = ((const float4*)src)[i];
        uint2 o; o.x = pack2bf(p.x, p.y); o.y = pack2bf(p.z, p.w);
        ((uint2*)dst)[i] = o;
    }
}

// Batched fp32 [rows, cols] -> bf16 [cols, rows] transposes (the K-contiguous weight shadows of the data-gradient
// GEMMs).  desc[i] = {src offset, dst offset, rows, cols} in elements; tile_prefix[i] = first 32x32 tile of matrix i.
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                              const long long* __restrict__ desc, const int* __restrict__ tile_prefix,
                                                              int ndesc) {
    __shared__ float tile[32][33];
    int i = 0;
    while (i + 1 < ndesc && (int)blockIdx.x >= tile_prefix[i + 1]) ++i;
    const long long so = desc[4 * i], dof = desc[4 * i + 1];
    const int rows = (int)desc[4 * i + 2], cols = (int)desc[4 * i + 3];
    const int t = blockIdx.x - tile_prefix[i];
    const int tcols = (cols + 31) / 32;
    const int r0 = (t / tcols) * 32, c0 = (t % tcols) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < rows && c < cols) ? src[so + (long long)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (c < cols && r < rows) dst[dof + (long long)c * rows + r] = f2bf(tile[tx][k]);
    }
}

// Same batch of transposes from the bf16 shadow (already written by the optimizer step) instead of the fp32 master: half the
// bytes read, 64 x 64 tiles, 16-byte global accesses on both sides.  Needs rows % 64 == 0 and cols % 64 == 0; tile_prefix
// counts 64 x 64 tiles here.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                              const long long* __restrict__ desc, const int* __restrict__ tile_prefix,
                                                              int ndesc) {
    // 128-byte rows, the 16-byte chunk index XOR-swizzled with (row / 8): a column gather below reads rows 8 part .. 8 part + 7 of one column, so
    // the eight `part`s of a wave land in eight different chunks = disjoint bank quads (the padded [64][72] layout of rounds 2-5 put all of
    // them on ONE bank - 8 part rows x 36 dwords = 288 = 9 x 32 - and ran with SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.82)
    __shared__ __attribute__((aligned(16))) bf16_t tile[64][64];
    int i = 0;
    while (i + 1 < ndesc && (int)blockIdx.x >= tile_prefix[i + 1]) ++i;
    const long long so = desc[4 * i], dof = desc[4 * i + 1];
    const int rows = (int)desc[4 * i + 2], cols = (int)desc[4 * i + 3];
    const int t = blockIdx.x - tile_prefix[i];
    const int tcols = cols / 64;
    const int r0 = (t / tcols) * 64, c0 = (t % tcols) * 64;
    const int part = threadIdx.x & 7, line = threadIdx.x >> 3;            // 8 threads x 16 B per 64-element line, 32 lines per pass
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int r = pass * 32 + line;
        *(uint4*)&tile[r][(part ^ ((r >> 3) & 7)) * 8] = *(const uint4*)(src + so + (long long)(r0 + r) * cols + c0 + part * 8);
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int c = pass * 32 + line;                                    // output line c holds source rows r0 .. r0 + 63
        uint32_t w[4];
        const int cs = (((c >> 3) ^ part) << 3) | (c & 7);               // column c of rows 8 part .. 8 part + 7 (their swizzle key is `part`)
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = (uint32_t)tile[part * 8 + 2 * j][cs] | ((uint32_t)tile[part * 8 + 2 * j + 1][cs] << 16);
        *(uint4*)(dst + dof + (long long)(c0 + c) * rows + r0 + part * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

}  // namespace

static inline int stream_blocks(size_t n4) {
    size_t b = (n4 + 255) / 256;
    return (int)(b < 2048 ? (b ? b : 1) : 2048);
}

extern "C" int cldrd_sqnorm_blocks(void) { return 2048; }

// state: device float[8 + 64] = {S, 1 / S, good steps, skipped steps, headroom exponent h, -, -, -, scratch[64]} (the block
// cldrd_set_loss_scale points at).  Sets S from max(|a|, |b|) (both fp32, either may be null / empty) and multiplies a and b by it in place.
extern "C" int cldrd_loss_scale_adapt(float* a, size_t na, float* b, size_t nb, float* state, void* stream) {
    CLDRD_CHECK(state != nullptr && (na == 0 || a != nullptr) && (nb == 0 || b != nullptr), "loss_scale_adapt: bad arguments");
    hipLaunchKernelGGL(scale_amax_kernel, dim3(SCALE_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const float*)a, na, (const float*)b, nb, state + 8);
    CLDRD_LAUNCH_CHECK();
    hipLaunchKernelGGL(scale_apply_kernel, dim3(SCALE_BLOCKS), dim3(256), 0, (hipStream_t)stream, a, na, b, nb, state, (const float*)(state + 8));
    CLDRD_LAUNCH_CHECK();
    return 0;
}

// out: device float[3] = {norm, clip coefficient, non-finite flag}; partial: device float[cldrd_sqnorm_blocks()].
extern "C" int cldrd_grad_clip_coef(const float* g, size_t n, float max_norm, float* partial, float* out, void* stream) {
    CLDRD_CHECK(n > 0 && n % 4 == 0 && ((uintptr_t)g % 16 == 0), "grad_clip_coef: n must be a multiple of 4, g 16-byte aligned");
    const int nb = stream_blocks(n / 4);
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, g, n / 4, partial);
    CLDRD_LAUNCH_CHECK();
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partial, nb, max_norm, out, (float*)g_cldrd_loss_scale,
                       g_cldrd_loss_scale_interval);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

// The two halves of cldrd_grad_clip_coef on their own, for a norm taken in pieces: cldrd_sqnorm_partial writes the sums of squares of
// g[0, n) into partial[0, nblk) (exactly nblk blocks: every slot is written), cldrd_clip_coef turns nblk_total slots into {norm, clip
// coefficient, non-finite flag}.  The trainer takes the norm of the gradients that are complete early (the query tower, the passage
// tower's embedding block) on its second stream UNDER the passage tower's weight-gradient launch and only the rest after it.
extern "C" int cldrd_sqnorm_partial(const float* g, size_t n, float* partial, int nblk, void* stream) {
    CLDRD_CHECK(n > 0 && n % 4 == 0 && ((uintptr_t)g % 16 == 0), "sqnorm_partial: n must be a multiple of 4, g 16-byte aligned");
    CLDRD_CHECK(nblk >= 1 && nblk <= 65535 && partial != nullptr, "sqnorm_partial: 1 <= nblk <= 65535 partial sums");
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, g, n / 4, partial);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

extern "C" int cldrd_clip_coef(const float* partial, int nblk_total, float max_norm, float* out, void* stream) {
    CLDRD_CHECK(nblk_total >= 1 && partial != nullptr && out != nullptr, "clip_coef: no partial sums");
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nblk_total, max_norm, out, (float*)g_cldrd_loss_scale,
                       g_cldrd_loss_scale_interval);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

// Legacy transformers.AdamW (correct_bias=True), step is 1-based.  decay_flags: one byte per 64 parameters.
// clip: device float[3] from cldrd_grad_clip_coef or null.  shadow: bf16 copy of the updated parameters or null.
// shadow16 (optional): fp16 copy (RNE, as cldrd_cast_f16) of the updated parameters [h16_begin, h16_end) - the high-precision forward
// of the query tower reads it; both bounds multiples of 4, shadow16[0] = parameter h16_begin.
extern "C" int cldrd_adamw_step_h16(float* p, const float* g, float* m, float* v, const unsigned char* decay_flags, void* shadow,
                                    size_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                    const float* clip, void* shadow16, size_t h16_begin, size_t h16_end, void* stream);
extern "C" int cldrd_adamw_step(float* p, const float* g, float* m, float* v, const unsigned char* decay_flags, void* shadow,
                                size_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                const float* clip, void* stream) {
    return cldrd_adamw_step_h16(p, g, m, v, decay_flags, shadow, n, lr, beta1, beta2, eps, weight_decay, step, clip, nullptr, 0, 0, stream);
}
extern "C" int cldrd_adamw_step_h16(float* p, const float* g, float* m, float* v, const unsigned char* decay_flags, void* shadow,
                                    size_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                    const float* clip, void* shadow16, size_t h16_begin, size_t h16_end, void* stream) {
    CLDRD_CHECK(n > 0 && n % 64 == 0, "adamw_step: n must be a multiple of 64");
    CLDRD_CHECK(step >= 1, "adamw_step: step is 1-based");
    CLDRD_CHECK(shadow16 == nullptr || (h16_begin % 4 == 0 && h16_end % 4 == 0 && h16_begin <= h16_end && h16_end <= n && (uintptr_t)shadow16 % 8 == 0),
                "adamw_step: fp16 shadow range must be 4-aligned and inside the buffer");
    AdamArgs a;
    a.shadow16 = (uint16_t*)shadow16; a.h_lo4 = h16_begin / 4; a.h_hi4 = h16_end / 4;
    a.reverse = true;         // back to front: the norm pass has just read g front to back (+0.3 %, profiles/r02_microbench.txt)
    a.p = p; a.g = g; a.m = m; a.v = v; a.decay = decay_flags; a.shadow = (bf16_t*)shadow; a.n4 = n / 4;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay; a.coef = clip;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    a.step_size = (float)((double)lr * sqrt(bc2) / bc1);
    a.hyper = g_cldrd_optim_hyper;
    hipLaunchKernelGGL(adamw_kernel, dim3(stream_blocks(a.n4)), dim3(256), 0, (hipStream_t)stream, a);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

extern "C" int cldrd_cast_bf16(const float* src, void* dst, size_t n, void* stream) {
    CLDRD_CHECK(n > 0 && n % 4 == 0, "cast_bf16: n must be a multiple of 4");
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(stream_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n / 4);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

extern "C" int cldrd_transpose_cast_batched(const float* src, void* dst, const long long* desc, const int* tile_prefix, int ndesc,
                                            int total_tiles, void* stream) {
    CLDRD_CHECK(ndesc > 0 && total_tiles > 0, "transpose_cast_batched: empty");
    hipLaunchKernelGGL(transpose_cast_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, desc, tile_prefix, ndesc);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

// dst[cols, rows] (bf16) = src[rows, cols]^T (bf16) for a batch of matrices; desc / tile_prefix as above with 64 x 64 tiles.
extern "C" int cldrd_transpose_bf16_batched(const void* src, void* dst, const long long* desc, const int* tile_prefix, int ndesc,
                                            int total_tiles, void* stream) {
    CLDRD_CHECK(ndesc > 0 && total_tiles > 0, "transpose_bf16_batched: empty");
    CLDRD_CHECK(((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0), "transpose_bf16_batched: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, desc,
                       tile_prefix, ndesc);
    CLDRD_LAUNCH_CHECK();
    return 0;
}
