// How fast can ONE workgroup per CU pull L2- / Infinity-Cache-resident data: LDS-DMA (global_load_lds_dwordx4) against plain
// global_load_dwordx4 into registers?  (The GEMM K loops stop at ~41 GB/s per CU of operand traffic: is that the DMA path?)
// build: hipcc --offload-arch=gfx950 -O3 -o dma_rate tools/micro/dma_rate.hip ; run: ./dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// every wave issues PIECES 1-KiB pieces per round into its own LDS region and keeps at most `DEPTH` rounds in flight
template <int PIECES, int DEPTH>
__global__ __launch_bounds__(512) void dma_kernel(const char* __restrict__ src, size_t wg_bytes, int rounds, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (size_t)blockIdx.x * wg_bytes;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wid * (DEPTH * PIECES * 1024);
    const size_t round_bytes = (size_t)8 * PIECES * 1024;                 // all 8 waves
    size_t off = 0;
    for (int r = 0; r < rounds; ++r) {
        const int slot = r % DEPTH;
        const char* p = base + off + (size_t)wid * PIECES * 1024;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const uint32_t voff = (uint32_t)(i * 1024 + lane * 16);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds0 + (slot * PIECES + i) * 1024), "v"(voff), "s"(p) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * PIECES) : "memory");
        off += round_bytes;
        if (off + round_bytes > wg_bytes) off = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = ((unsigned*)smem)[5];
}

template <int PIECES, int DEPTH>
__global__ __launch_bounds__(512) void reg_kernel(const char* __restrict__ src, size_t wg_bytes, int rounds, unsigned* sink) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const char* base = src + (size_t)blockIdx.x * wg_bytes;
    const size_t round_bytes = (size_t)8 * PIECES * 1024;
    size_t off = 0;
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint4 buf[DEPTH][PIECES];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < PIECES; ++i) buf[d][i] = make_uint4(0, 0, 0, 0);
    for (int r = 0; r < rounds; r += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            // consume what this slot held, then refill it: DEPTH - 1 rounds stay in flight
#pragma unroll
            for (int i = 0; i < PIECES; ++i) { acc.x ^= buf[d][i].x; acc.y ^= buf[d][i].y; acc.z ^= buf[d][i].z; acc.w ^= buf[d][i].w; }
            const char* p = base + off + (size_t)wid * PIECES * 1024 + lane * 16;
#pragma unroll
            for (int i = 0; i < PIECES; ++i) buf[d][i] = *(const uint4*)(p + i * 1024);
            off += round_bytes;
            if (off + round_bytes > wg_bytes) off = 0;
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < PIECES; ++i) { acc.x ^= buf[d][i].x; acc.y ^= buf[d][i].y; }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[blockIdx.x] = acc.x;
}

template <typename F>
static float time_ms(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main() {
    const int ncu = 256;
    char* src; unsigned* sink;
    const size_t total = (size_t)1 << 30;
    CK(hipMalloc(&src, total)); CK(hipMalloc(&sink, 4096)); CK(hipMemset(src, 1, total));
    const int rounds = 4096;
    for (size_t wg_bytes : {(size_t)64 << 10, (size_t)512 << 10, (size_t)4 << 20}) {
        printf("footprint per workgroup %zu KiB (total %zu MiB)\n", wg_bytes >> 10, (wg_bytes * ncu) >> 20);
#define RUN(KERN, P, D, LDS) { \
        (void)hipFuncSetAttribute((const void*)KERN<P, D>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); \
        float ms = time_ms([&] { hipLaunchKernelGGL((KERN<P, D>), dim3(ncu), dim3(512), LDS, 0, src, wg_bytes, rounds, sink); }, 5); \
        double bytes = (double)ncu * rounds * 8 * P * 1024; \
        printf("  %-10s pieces/wave/round %d, rounds in flight %d (%3d KiB per CU): %7.1f GB/s per CU, %6.2f TB/s\n", #KERN, P, D - 1, (D - 1) * P * 8, bytes / ms / 1e6 / ncu, bytes / ms / 1e9); }
        RUN(dma_kernel, 4, 2, 2 * 4 * 8 * 1024)
        RUN(dma_kernel, 4, 3, 3 * 4 * 8 * 1024)
        RUN(dma_kernel, 4, 4, 4 * 4 * 8 * 1024)
        RUN(dma_kernel, 8, 2, 2 * 8 * 8 * 1024)
        RUN(dma_kernel, 2, 8, 8 * 2 * 8 * 1024)
        RUN(reg_kernel, 4, 2, 0)
        RUN(reg_kernel, 4, 3, 0)
        RUN(reg_kernel, 4, 4, 0)
        RUN(reg_kernel, 8, 2, 0)
        RUN(reg_kernel, 2, 8, 0)
    }
    CK(hipDeviceSynchronize());
    return 0;
}
