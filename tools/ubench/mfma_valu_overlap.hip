// Micro-benchmark: can one SIMD run a wave's MFMAs and another wave's VALU work at the same time?
// 8 waves per workgroup, one workgroup per CU: waves 0-3 (one per SIMD) run role A, waves 4-7 run role B.
// roles: 0 idle, 1 = 32x32x16 bf16 MFMA stream (4 independent accumulators), 2 = v_fma_f32 stream, 3 = v_exp_f32 stream,
//        4 = 16x16x32 MFMA stream, 5 = mixed (2 MFMA + 10 VALU interleaved in ONE wave)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int r>
__device__ __forceinline__ void role(float* out, long long* cyc, int slot) {
    long long t0 = 0, t1 = 0;
    float res = 0.f;
    if (r == 1 || r == 5) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (threadIdx.x + e)); b[e] = (__bf16)(0.02f * e); }
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = 0.5f + i;
        t0 = clock64();
        for (int it = 0; it < 256; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                if (r == 5) {
#pragma unroll
                    for (int j = 0; j < 5; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(i * 5 + j) & 7]) : "v"(0.999f));
                }
            }
        }
        t1 = clock64();
        for (int i = 0; i < 4; ++i) res += acc[i][0];
        for (int i = 0; i < 8; ++i) res += v[i];
    } else if (r == 4) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (threadIdx.x + e)); b[e] = (__bf16)(0.02f * e); }
        t0 = clock64();
        for (int it = 0; it < 256; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
        t1 = clock64();
        for (int i = 0; i < 8; ++i) res += acc[i][0];
    } else if (r == 2 || r == 3) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = 0.5f + i * 0.01f + threadIdx.x * 1e-6f;
        t0 = clock64();
        for (int it = 0; it < 1024; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (r == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(0.999f));
                else asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            }
        }
        t1 = clock64();
        for (int i = 0; i < 8; ++i) res += v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + slot] = t1 - t0;
}

template <int RA, int RB>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc) {
    const int w = threadIdx.x >> 6;
    if (w < 4) role<RA>(out, cyc, w); else role<RB>(out, cyc, w);
}
template <int RA, int RB>
void run(float* out, long long* cyc, const char** names) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, out, cyc);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(256 * 8);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < 256; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? a : b) += h[i * 8 + w];
    printf("A = %-26s B = %-26s : A %9.0f cycles   B %9.0f cycles\n", names[RA], names[RB], a / 1024, b / 1024);
}

int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    const char* names[] = {"idle", "mfma 32x32x16 x1024", "v_fma x8192", "v_exp x8192", "mfma 16x16x32 x2048", "mfma x1024 + 5 fma each"};
    run<1, 0>(out, cyc, names); run<4, 0>(out, cyc, names); run<2, 0>(out, cyc, names); run<3, 0>(out, cyc, names); run<5, 0>(out, cyc, names);
    run<1, 2>(out, cyc, names); run<1, 3>(out, cyc, names); run<4, 2>(out, cyc, names); run<4, 3>(out, cyc, names); run<1, 1>(out, cyc, names);
    run<2, 2>(out, cyc, names); run<3, 3>(out, cyc, names); run<5, 5>(out, cyc, names); run<5, 2>(out, cyc, names); run<5, 3>(out, cyc, names);
    return 0;
}
