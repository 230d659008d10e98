// Micro-benchmark: issue cost (shader cycles per wave-instruction) of the VALU / transcendental ops the attention softmax uses.
// One workgroup of NW waves per CU-quarter...: launched as 1 block of 64*NW threads per CU (grid 256) so NW waves share a SIMD when NW > 4.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
template <int OP>
__global__ void k(float* out, long long* cyc, float a, float b) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i * 0.001f + threadIdx.x * 1e-6f;
    unsigned u[8];
    for (int i = 0; i < 8; ++i) u[i] = __float_as_uint(v[i]);
    long long t0 = clock64();
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            if (OP == 2) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
            if (OP == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
            if (OP == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            if (OP == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
            if (OP == 6) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(v[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]));
            if (OP == 7) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(v[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]));
            if (OP == 8) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v[i]) : "v"(u[0] & 1));
            if (OP == 9) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
            if (OP == 10) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
            if (OP == 11) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
            if (OP == 12) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&v[i & 6]) : "v"(*(double*)&v[(i + 2) & 6]));
            if (OP == 13) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
            if (OP == 14) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(b));
            if (OP == 15) asm volatile("v_pk_mov_b32 %0, %1, %1" : "+v"(*(double*)&v[i & 6]) : "v"(*(double*)&v[(i + 2) & 6]));
            if (OP == 16) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&v[i & 6]) : "v"(*(double*)&v[(i + 2) & 6]));
            if (OP == 17) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
            if (OP == 18) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&v[i & 6]) : "v"(*(double*)&v[(i + 2) & 6]));
            if (OP == 19) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
            if (OP == 20) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(const char* name, int nw) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 64 * nw * 4); hipMalloc(&cyc, 256 * nw * 8);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * nw), 0, 0, out, cyc, 0.5f, 0.25f);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * nw), 0, 0, out, cyc, 0.5f, 0.25f);
    hipDeviceSynchronize();
    std::vector<long long> h(256 * nw);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto c : h) m += c; m /= h.size();
    // per-SIMD throughput: waves per SIMD = nw / 4 (>= 1); cycles per wave-instr seen by the SIMD = m / (2048 * waves_per_simd)
    double wps = nw >= 4 ? nw / 4.0 : 1.0;
    printf("%-22s waves/CU %2d: %7.2f cycles per instr per wave, %6.2f SIMD-cycles per wave-instr\n", name, nw, m / 2048.0, m / 2048.0 / wps);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int nw : {4, 16}) {
        run<0>("v_fma_f32", nw); run<3>("v_add_f32", nw); run<10>("v_mul_f32", nw); run<13>("v_sub_f32", nw); run<1>("v_exp_f32", nw);
        run<2>("v_exp_f16", nw); run<9>("v_rcp_f32", nw); run<4>("v_max3_f32", nw); run<5>("v_cvt_pk_bf16_f32", nw); run<11>("v_cvt_pkrtz_f16_f32", nw);
        run<6>("v_dot2_f32_bf16", nw); run<7>("v_dot2_f32_f16", nw); run<8>("v_ldexp_f32", nw); run<12>("v_pk_mul_f32", nw);
        run<14>("v_mov_b32", nw); run<15>("v_pk_mov_b32", nw); run<16>("v_pk_add_f32", nw); run<17>("v_cvt_pk_f16_f32", nw); run<18>("v_pk_fma_f32", nw);
        run<19>("v_max_f32", nw); run<20>("v_fmac_f32", nw);
    }
    return 0;
}
