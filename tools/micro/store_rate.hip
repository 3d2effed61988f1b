// What does one global_store_dwordx4 (1 KiB per wave-instruction) cost a CU?  One 8-wave workgroup per CU stores `PER` instructions per
// round into a private window (wg_bytes per workgroup: small = L2-resident, large = streams to HBM), keeping at most DEPTH*PER stores in
// flight per wave.  Patterns: 0 = 1 KiB contiguous per instruction; 1 = 8 rows x 128 B (row pitch `pitch` bytes: the fp16 epilogue of a
// 256 x 256 tile); 2 = 10 rows x 96 B halves of 192-B segments (the fp32 epilogue of a 256 x 192 tile: two instructions per 32-B lane slot).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/store_rate tools/micro/store_rate.hip ; run: /tmp/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int PAT, int NT>
__global__ __launch_bounds__(512) void store_kernel(char* __restrict__ dst, size_t wg_bytes, int pitch, int rounds, int per, int nwaves) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (wid >= nwaves) return;
    char* base = dst + (size_t)blockIdx.x * wg_bytes;
    u32x4 v = {(unsigned)lane, (unsigned)wid, blockIdx.x, 7u};
    size_t lane_off;
    size_t instr_step;       // bytes between consecutive instructions of a wave
    if (PAT == 0) { lane_off = (size_t)lane * 16; instr_step = 1024; }
    else if (PAT == 1) { lane_off = (size_t)(lane >> 3) * pitch + (lane & 7) * 16; instr_step = (size_t)8 * pitch; }
    else { lane_off = (size_t)(lane / 6) * pitch + (lane % 6) * 32; instr_step = (size_t)10 * pitch; }
    const size_t wave_window = wg_bytes / nwaves;
    size_t off = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < per; ++i) {
            char* p = base + (size_t)wid * wave_window + off + lane_off;
            if (PAT == 2 && lane >= 60) p = base;      // 60 active lanes in that pattern (the others rewrite one place)
            if (NT) __builtin_nontemporal_store(v, (u32x4*)p); else *(u32x4*)p = v;
            if (PAT == 2) { if (NT) __builtin_nontemporal_store(v, (u32x4*)(p + 16)); else *(u32x4*)(p + 16) = v; }
            off += instr_step;
            if (off + instr_step + 64 * 16 > wave_window) off = 0;
        }
        v.x += 1;
    }
}

int main() {
    const int nwg = 256;
    char* buf;
    const size_t big = (size_t)nwg * (64u << 20);
    CK(hipMalloc(&buf, big));
    CK(hipMemset(buf, 0, big));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { const char* name; int pat, nt; size_t wg_bytes; int pitch; int nwaves; };
    const Case cases[] = {
        {"contiguous 1 KiB, L2-resident window (64 KiB / CU)", 0, 0, 64u << 10, 0, 8},
        {"contiguous 1 KiB, streaming (64 MiB / CU)", 0, 0, 64u << 20, 0, 8},
        {"contiguous 1 KiB, streaming, nt", 0, 1, 64u << 20, 0, 8},
        {"8 rows x 128 B, pitch 6144 (fp16 N=3072), resident 1 MiB / CU", 1, 0, 1u << 20, 6144, 8},
        {"8 rows x 128 B, pitch 6144, streaming", 1, 0, 64u << 20, 6144, 8},
        {"8 rows x 128 B, pitch 6144, streaming, nt", 1, 1, 64u << 20, 6144, 8},
        {"10 rows x 2 x 16 B of 32-B slots, pitch 3072 (fp32 N=768), resident", 2, 0, 1u << 20, 3072, 8},
        {"10 rows x 2 x 16 B of 32-B slots, pitch 3072, streaming", 2, 0, 64u << 20, 3072, 8},
        {"contiguous 1 KiB, streaming, 4 waves", 0, 0, 64u << 20, 0, 4},
        {"contiguous 1 KiB, streaming, 2 waves", 0, 0, 64u << 20, 0, 2},
        {"contiguous 1 KiB, streaming, 1 wave", 0, 0, 64u << 20, 0, 1},
    };
    for (const Case& c : cases) {
        const int rounds = 200, per = 16;
        auto launch = [&]() {
            if (c.pat == 0 && !c.nt) hipLaunchKernelGGL((store_kernel<0, 0>), dim3(nwg), dim3(512), 0, 0, buf, c.wg_bytes, c.pitch, rounds, per, c.nwaves);
            else if (c.pat == 0) hipLaunchKernelGGL((store_kernel<0, 1>), dim3(nwg), dim3(512), 0, 0, buf, c.wg_bytes, c.pitch, rounds, per, c.nwaves);
            else if (c.pat == 1 && !c.nt) hipLaunchKernelGGL((store_kernel<1, 0>), dim3(nwg), dim3(512), 0, 0, buf, c.wg_bytes, c.pitch, rounds, per, c.nwaves);
            else if (c.pat == 1) hipLaunchKernelGGL((store_kernel<1, 1>), dim3(nwg), dim3(512), 0, 0, buf, c.wg_bytes, c.pitch, rounds, per, c.nwaves);
            else hipLaunchKernelGGL((store_kernel<2, 0>), dim3(nwg), dim3(512), 0, 0, buf, c.wg_bytes, c.pitch, rounds, per, c.nwaves);
        };
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_cu = (double)rounds * per * c.nwaves * (c.pat == 2 ? 2 : 1);
        const double bytes = instr_per_cu * nwg * (c.pat == 2 ? 960.0 : 1024.0);
        printf("%-72s %8.1f us  %6.1f ns per store instruction per CU (%5.0f cycles at 2.4 GHz)  %6.2f TB/s\n", c.name, ms * 1e3, ms * 1e6 / instr_per_cu,
               ms * 1e6 / instr_per_cu * 2.4, bytes / (ms * 1e-3) / 1e12);
    }
    return 0;
}
