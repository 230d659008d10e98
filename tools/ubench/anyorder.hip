// Does hipExtAnyOrderLaunch (AQL packet without the barrier bit) let a kernel start inside the TAIL of the kernel in front of it on the
// same stream on gfx950?  (hip_ext.h says the flag is "not supported on GFX9xx" for the module-launch form.)
//   A: 256 workgroups x 512 threads, 160 KiB of LDS each (one per CU, like gemm256_kernel); workgroup i spins 100 us, the last 32 spin 160 us.
//   B: 64 small workgroups, 10 us each.
// Stamps (wall_clock64, 100 MHz): A's per-workgroup end, B's per-workgroup start.  Build: hipcc --offload-arch=gfx950 -O2 anyorder.hip -o anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(512) void kern_a(unsigned long long* end, int long_from, int t_short, int t_long) {
    extern __shared__ char smem[];
    const unsigned long long t0 = wall_clock64();
    const unsigned long long dl = (unsigned long long)((int)blockIdx.x >= long_from ? t_long : t_short);
    if (threadIdx.x == 0) smem[0] = 1;
    while (wall_clock64() - t0 < dl) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) end[blockIdx.x] = wall_clock64();
}
__global__ void kern_b(unsigned long long* start, int t) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) start[blockIdx.x] = t0;
    while (wall_clock64() - t0 < (unsigned long long)t) __builtin_amdgcn_s_sleep(8);
}

int main() {
    unsigned long long *end, *start;
    hipMalloc(&end, 256 * 8);
    hipMalloc(&start, 64 * 8);
    hipFuncSetAttribute((const void*)kern_a, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipStream_t s;
    hipStreamCreate(&s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, s);
            for (int it = 0; it < 20; ++it) {
                hipLaunchKernelGGL(kern_a, dim3(256), dim3(512), 160 * 1024, s, end, 224, 10000, 16000);
                if (mode) hipExtLaunchKernelGGL(kern_b, dim3(64), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, start, 1000);
                else hipLaunchKernelGGL(kern_b, dim3(64), dim3(256), 0, s, start, 1000);
            }
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> he(256), hs(64);
            hipMemcpy(he.data(), end, 256 * 8, hipMemcpyDeviceToHost);
            hipMemcpy(hs.data(), start, 64 * 8, hipMemcpyDeviceToHost);
            const unsigned long long a_last = *std::max_element(he.begin(), he.end()), a_first = *std::min_element(he.begin(), he.end());
            const unsigned long long b_first = *std::min_element(hs.begin(), hs.end()), b_last = *std::max_element(hs.begin(), hs.end());
            printf("%s: 20 x (A + B) = %.1f us per pair; last pair: A ends %.1f..%.1f us, B starts %.1f..%.1f us (relative to A's first end)\n",
                   mode ? "any-order B" : "ordinary  B", ms * 1000 / 20, 0.0, (a_last - a_first) / 100.0, ((double)b_first - (double)a_first) / 100.0,
                   ((double)b_last - (double)a_first) / 100.0);
        }
    }
    return 0;
}
