// Error reporting + version for the C-ABI library (include/cldrd_hip.h).
#include "common.h"
#include <math.h>
#include <string.h>

static thread_local char g_err[512] = "";

int cldrd_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "unknown error", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
    return 1;
}

extern "C" const char* cldrd_last_error(void) { return g_err; }

#ifdef CLDRD_DEV_BUILD
#include <stdlib.h>
int cldrd_dev_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
extern "C" unsigned long long g_dev_stamps_host[1024 * 8];
unsigned long long g_dev_stamps_host[1024 * 8];
extern "C" const unsigned long long* cldrd_dev_stamps() { return g_dev_stamps_host; }      // after a stream synchronize (tools/epi_stamps.py)
#endif
int g_cldrd_tune_splitk = 0, g_cldrd_tune_attn_fwd2 = 1, g_cldrd_tune_attn_bwd2 = 1, g_cldrd_tune_nt64 = 1;
// key: "gemm_splitk" | "attn_fwd2" | "attn_bwd2" (meanings: common.h).  Every choice computes the same function; tests use it to
// reach the alternative kernels.  Process-wide, not thread-safe against concurrent launches.
extern "C" int cldrd_set_tuning(const char* key, int value) {
    CLDRD_CHECK(key != nullptr, "set_tuning: null key");
    if (!strcmp(key, "gemm_splitk")) { CLDRD_CHECK(value >= 0, "set_tuning: gemm_splitk >= 0"); g_cldrd_tune_splitk = value; return 0; }
    if (!strcmp(key, "attn_fwd2")) { g_cldrd_tune_attn_fwd2 = value != 0; return 0; }
    if (!strcmp(key, "attn_bwd2")) { g_cldrd_tune_attn_bwd2 = value != 0; return 0; }
    if (!strcmp(key, "gemm_nt64")) { g_cldrd_tune_nt64 = value != 0; return 0; }
    return cldrd_set_error("set_tuning: unknown key");
}
extern "C" int cldrd_version(void) { return 100; }
extern "C" int cldrd_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}


// ---- per-step state in device memory (HIP-graph replay of the training step) --------------------------------------------------
thread_local const unsigned long long* g_cldrd_seed_base = nullptr;
thread_local const float* g_cldrd_optim_hyper = nullptr;

// Every launch made by this thread from now on adds *base (read on the device, at run time) to its dropout seed; null: off.
extern "C" void cldrd_set_seed_base(const unsigned long long* base) { g_cldrd_seed_base = base; }
// cldrd_adamw_step* launched by this thread from now on take {lr, step size = lr sqrt(1 - beta2^t) / (1 - beta1^t)} from this device
// float[2] instead of their by-value arguments; null: off.
extern "C" void cldrd_set_optim_hyper(const float* hyper) { g_cldrd_optim_hyper = hyper; }
// Loss scaling of the all-fp16 training mode (reference: torch.cuda.amp.GradScaler around nway_listwise_1.py:334-359).  scale = device
// float[72] (ALL 72 are written: a shorter buffer is overrun) = {[0] S, [1] 1 / S, [2] finite steps since the headroom last changed,
// [3] skipped steps, [4] headroom exponent h <= 0, [5..7] unused, [8..71] scratch of cldrd_loss_scale_adapt}.  cldrd_loss_scale_adapt
// sets S every step from dL/dCLS and multiplies dL/dCLS by it (so every 16-bit gradient tensor of the backward carries S);
// cldrd_clip_coef / cldrd_grad_clip_coef run the safety net ([2..4]).  Launches made by this thread from now on that produce PARAMETER
// gradients multiply by 1 / S where they write them: cldrd_wgrad_group (epilogue / slab reduction), cldrd_layernorm_bwd (inside its
// per-block partial sums, so cldrd_ln_reduce_group adds unscaled numbers), cldrd_embed_ln_bwd (atomics) - flat_g never carries the scale.  Read on the device at run time (a replayed graph sees the current value).  null: off.
thread_local const float* g_cldrd_loss_scale = nullptr;
thread_local int g_cldrd_loss_scale_interval = 2000;
extern "C" void cldrd_set_loss_scale(const float* scale, int growth_interval) {
    g_cldrd_loss_scale = scale;
    g_cldrd_loss_scale_interval = growth_interval > 0 ? growth_interval : 2000;
}

// Clip-norm partial sums from the kernels that WRITE the gradients (round 5).  While a sink is set, launches of this thread that produce final
// parameter gradients - the slab reduction of cldrd_wgrad_group, cldrd_ln_reduce_group - also write one sum of squares per workgroup of what
// they wrote to slots[cursor ...] and advance the (host-side, thread-local) cursor; cldrd_clip_coef then reduces those slots next to the ones of
// cldrd_sqnorm_partial.  The trainer uses it for the passage tower's layer gradients, the only ones that are complete AFTER the last
// weight-gradient group: the separate norm pass over them (265 MB re-read, ~90 us on the critical path of every step) is gone.  A launch that
// cannot contribute (a weight-gradient group that writes its tiles directly, without slabs) marks the sink incomplete: cldrd_norm_sink_used()
// then returns -1 and the caller takes the norm of that range the old way.
thread_local float* g_cldrd_norm_sink = nullptr;
thread_local int g_cldrd_norm_sink_cap = 0, g_cldrd_norm_sink_used = 0, g_cldrd_norm_sink_bad = 0;
extern "C" void cldrd_set_norm_sink(float* slots, int capacity) {
    g_cldrd_norm_sink = slots;
    g_cldrd_norm_sink_cap = slots ? capacity : 0;
    g_cldrd_norm_sink_used = 0;
    g_cldrd_norm_sink_bad = 0;
}
// slots written since cldrd_set_norm_sink; -1: a launch could not contribute (cldrd_norm_sink_miss); -2: the sink was too small for a launch
extern "C" int cldrd_norm_sink_used(void) { return g_cldrd_norm_sink_bad ? -g_cldrd_norm_sink_bad : g_cldrd_norm_sink_used; }
// reserve n slots for a launch; null when no sink is set or it is full (the sink is then marked incomplete)
float* cldrd_norm_sink_take(int n) {
    if (!g_cldrd_norm_sink) return nullptr;
    if (n <= 0 || g_cldrd_norm_sink_used + n > g_cldrd_norm_sink_cap) { g_cldrd_norm_sink_bad = 2; return nullptr; }
    float* p = g_cldrd_norm_sink + g_cldrd_norm_sink_used;
    g_cldrd_norm_sink_used += n;
    return p;
}
void cldrd_norm_sink_miss(void) { if (g_cldrd_norm_sink && !g_cldrd_norm_sink_bad) g_cldrd_norm_sink_bad = 1; }

namespace {
__global__ void step_state_kernel(unsigned long long* seeds, unsigned long long s0, unsigned long long s1, float* hyper, float lr, float step_size,
                                  float beta1, float beta2, int adam_step, const float* scale_state) {
    if (threadIdx.x == 0) {
        if (seeds) { seeds[0] = s0; seeds[1] = s1; }
        if (hyper) {
            if (scale_state) {
                // GradScaler semantics (reference nway_listwise_1.py:357: scaler.step() does not call optimizer.step() on a non-finite
                // gradient, so Adam's per-parameter `step` - the bias-correction exponent - only counts the steps that were applied):
                // the skipped count lives on the device (clip_coef_kernel), so the exponent is formed here, not on the host
                int t = adam_step - (int)scale_state[3];
                if (t < 1) t = 1;
                const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
                step_size = (float)((double)lr * sqrt(bc2) / bc1);
            }
            hyper[0] = lr; hyper[1] = step_size;
        }
    }
}
}  // namespace

namespace {
struct CopySegs { const void* src[8]; void* dst[8]; unsigned long long bytes[8]; int n; };
__global__ __launch_bounds__(256) void copy_segments_kernel(CopySegs c) {
    const int seg = blockIdx.y;
    if (seg >= c.n) return;
    const unsigned long long nb = c.bytes[seg];
    const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((((uintptr_t)c.src[seg] | (uintptr_t)c.dst[seg] | nb) & 15) == 0) {
        const uint4* s = (const uint4*)c.src[seg];
        uint4* d = (uint4*)c.dst[seg];
        for (size_t i = i0; i < nb / 16; i += stride) d[i] = s[i];
    } else {
        const unsigned char* s = (const unsigned char*)c.src[seg];
        unsigned char* d = (unsigned char*)c.dst[seg];
        for (size_t i = i0; i < nb; i += stride) d[i] = s[i];
    }
}
}  // namespace

// Up to 8 small device-to-device copies in ONE launch (the inputs of a captured training step go to the graph's static buffers: five
// 5-us copy launches in front of every replay otherwise).  Host arrays of n pointers / byte counts; segments must not overlap.
extern "C" int cldrd_copy_segments(const void* const* src, void* const* dst, const size_t* bytes, int n, void* stream) {
    CLDRD_CHECK(n >= 1 && n <= 8, "copy_segments: 1..8 segments");
    CopySegs c;
    size_t biggest = 0;
    for (int i = 0; i < 8; ++i) {
        c.src[i] = i < n ? src[i] : nullptr; c.dst[i] = i < n ? dst[i] : nullptr; c.bytes[i] = i < n ? bytes[i] : 0;
        if (i < n) { CLDRD_CHECK(src[i] != nullptr && dst[i] != nullptr, "copy_segments: null segment"); biggest = biggest > bytes[i] ? biggest : bytes[i]; }
    }
    c.n = n;
    size_t blocks = (biggest / 16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(copy_segments_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, (hipStream_t)stream, c);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

namespace {
struct ZeroSegs { void* dst[8]; unsigned long long bytes[8]; int n; };
__global__ __launch_bounds__(256) void zero_segments_kernel(ZeroSegs z) {
    const int seg = blockIdx.y;
    if (seg >= z.n) return;
    uint4* d = (uint4*)z.dst[seg];
    const size_t n16 = z.bytes[seg] / 16, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) d[i] = make_uint4(0, 0, 0, 0);
}
}  // namespace

// n <= 8 device ranges set to zero in ONE launch (16-byte aligned, sizes multiples of 16): the embedding-table gradients of both towers at
// the start of a training step (everything else of the gradient buffer is written, not accumulated; two FillFunctor launches until round 5).
extern "C" int cldrd_zero_segments(void* const* dst, const size_t* bytes, int n, void* stream) {
    CLDRD_CHECK(n >= 1 && n <= 8, "zero_segments: 1..8 segments");
    ZeroSegs z;
    size_t biggest = 0;
    for (int i = 0; i < 8; ++i) {
        z.dst[i] = i < n ? dst[i] : nullptr; z.bytes[i] = i < n ? bytes[i] : 0;
        if (i < n) {
            CLDRD_CHECK(dst[i] != nullptr && ((uintptr_t)dst[i] % 16 == 0) && bytes[i] % 16 == 0, "zero_segments: 16-byte aligned ranges");
            biggest = biggest > bytes[i] ? biggest : bytes[i];
        }
    }
    z.n = n;
    size_t blocks = (biggest / 16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(zero_segments_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, (hipStream_t)stream, z);
    CLDRD_LAUNCH_CHECK();
    return 0;
}

// One tiny launch that writes this step's values (two seed words, lr, Adam step size for bias-correction step `adam_step`) to device
// memory, in stream order in front of the replay that reads them.
// scale_state (optional): the float[72] loss-scale block; the bias-correction exponent becomes adam_step - skipped steps (state[3]), read on the device.
extern "C" int cldrd_write_step_state(unsigned long long* seeds, unsigned long long seed0, unsigned long long seed1, float* hyper, float lr,
                                      float beta1, float beta2, int adam_step, const float* scale_state, void* stream) {
    CLDRD_CHECK(adam_step >= 1, "write_step_state: the Adam step is 1-based");
    const double bc1 = 1.0 - pow((double)beta1, (double)adam_step), bc2 = 1.0 - pow((double)beta2, (double)adam_step);
    hipLaunchKernelGGL(step_state_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, seeds, seed0, seed1, hyper, lr, (float)((double)lr * sqrt(bc2) / bc1),
                       beta1, beta2, adam_step, scale_state);
    CLDRD_LAUNCH_CHECK();
    return 0;
}
