// Minimal reproducer ATTEMPT for the round-3 "SLP" wrong-result finding (csrc/mhmr_common.h).
//
// What failed in the product (vit_cls.hip's folded-LayerNorm epilogue, SLP-vectorised build): the LOW halves of
//     v_pk_fma_f32 vD[0:1], vA[0:1], vB[0:1], vC[0:1] op_sel:[0,1,0]        (multiplier = HIGH dword of vB for both halves)
// in lanes 48..63, in 1 of 40..600 forwards, and only while a second stream kept MFMA-heavy foreign waves on the same SIMDs.
//
// This file isolates exactly that: kernel P executes the instruction (inline asm, the product's operand pattern: the multiplier pair and
// the addend pair arrive by global loads right before their first use) next to the two scalar v_fma_f32 it stands for and counts lanes
// whose packed result differs bitwise from the scalar one; kernel M is a persistent v_mfma_f32_16x16x32_f16 loop on every SIMD, launched
// on a second stream so that P's waves share SIMDs with it.  P runs alone (control) and under M, N launches each.
//
//   hipcc --offload-arch=gfx950 -O3 -o slp_repro tools/ubench/slp_repro.hip && ./slp_repro [launches] > profiles/r05_slp_repro.txt
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                                   \
    do {                                                                        \
        hipError_t e__ = (x);                                                   \
        if (e__ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e__, __FILE__, __LINE__); return 1; } \
    } while (0)

// persistent MFMA load: `iters` x 64 dependent-free MFMAs per wave, operands fixed in registers (random f16 bit patterns: power matters)
__global__ __launch_bounds__(256) void mfma_kernel(const f16x8* __restrict__ src, float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a = src[lane], b = src[64 + lane];
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;          // never true: keeps the loop alive
}

// the packed instruction against its scalar meaning.  Per iteration and lane: x (pair), m (pair; only its HIGH dword multiplies), c (pair)
// come from global memory (fresh addresses every iteration, so the loads and their waits sit right in front of the use, as in the product's
// epilogue); bad[0] counts lane-iterations whose LOW half differs, bad[1] HIGH half, bad[2 + lane] per-lane low-half failures.
template <int NOPS>
__global__ __launch_bounds__(256) void pk_kernel(const f32x2* __restrict__ xs, const f32x2* __restrict__ ms, const f32x2* __restrict__ cs,
                                                 int n, int iters, unsigned* __restrict__ bad, float* __restrict__ ex) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    unsigned lo_bad = 0, hi_bad = 0;
    for (int it = 0; it < iters; ++it) {
        const int i = (tid + it * 977) % n;
        const f32x2 x = xs[i], m = ms[(i * 7 + 3) % n], c = cs[(i * 13 + 5) % n];
        // (the result leaves the asm as ONE 64-bit integer and is split by shifts: taken as a float2, element 1 of the asm output was
        // read from the register of element 0 by this compiler -- v_cmp_ne_u32 v10, v22 for d[1] != r1 with d in v[10:11] -- which made
        // every high half "differ" in the first version of this file)
        uint64_t dq;
        if (NOPS == 0)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(dq) : "v"(x), "v"(m), "v"(c));
        else
            asm volatile("s_nop 4\n\tv_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(dq) : "v"(x), "v"(m), "v"(c));
        const float d[2] = {__builtin_bit_cast(float, (uint32_t)dq), __builtin_bit_cast(float, (uint32_t)(dq >> 32))};
        float r0, r1;
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(x[0]), "v"(m[1]), "v"(c[0]));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r1) : "v"(x[1]), "v"(m[1]), "v"(c[1]));
        const bool lb = __builtin_bit_cast(uint32_t, d[0]) != __builtin_bit_cast(uint32_t, r0);
        const bool hb = __builtin_bit_cast(uint32_t, d[1]) != __builtin_bit_cast(uint32_t, r1);
        lo_bad += lb;
        hi_bad += hb;
        if ((lb || hb) && bad[66 + (lb ? 0 : 1)] < 8) {       // the first few offending tuples of each kind (plain read first: no atomic storm): [kind, lane, x0, x1, m0, m1, c0, c1, d0, d1, r0, r1]
            const unsigned slot = atomicAdd(bad + 66 + (lb ? 0 : 1), 1u);
            if (slot < 8) {
                float* e = ex + ((lb ? 0 : 8) + slot) * 12;
                e[0] = lb ? 0.f : 1.f; e[1] = (float)lane; e[2] = x[0]; e[3] = x[1]; e[4] = m[0]; e[5] = m[1]; e[6] = c[0]; e[7] = c[1];
                e[8] = d[0]; e[9] = d[1]; e[10] = r0; e[11] = r1;
            }
        }
    }
    if (lo_bad) { atomicAdd(bad, lo_bad); atomicAdd(bad + 2 + lane, lo_bad); }
    if (hi_bad) atomicAdd(bad + 1, hi_bad);
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 400;
    const int n = 1 << 20, iters = 64, grid = 4096;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    f32x2 *xs, *ms, *cs;
    f16x8* src;
    float* sink;
    unsigned* bad;
    CK(hipMalloc(&xs, n * sizeof(f32x2))); CK(hipMalloc(&ms, n * sizeof(f32x2))); CK(hipMalloc(&cs, n * sizeof(f32x2)));
    CK(hipMalloc(&src, 128 * sizeof(f16x8))); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&bad, 68 * 4));
    float* ex;
    CK(hipMalloc(&ex, 16 * 12 * 4));
    {
        float* h = (float*)malloc(3 * n * 2 * sizeof(float));
        srand(7);
        for (int i = 0; i < 3 * n * 2; ++i) h[i] = (float)rand() / RAND_MAX * 4.f - 2.f;
        CK(hipMemcpy(xs, h, n * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(ms, h + 2 * n, n * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(cs, h + 4 * n, n * 8, hipMemcpyHostToDevice));
        uint16_t hs[128 * 8];
        for (int i = 0; i < 128 * 8; ++i) hs[i] = (uint16_t)(0x3000 + (rand() & 0x0fff));      // f16 values in [0.125, 0.5)
        CK(hipMemcpy(src, hs, sizeof(hs), hipMemcpyHostToDevice));
        free(h);
    }
    hipStream_t sp, sm;
    CK(hipStreamCreate(&sp)); CK(hipStreamCreate(&sm));
    unsigned hb[68];
    float hex_[16 * 12];
    for (int nops = 0; nops < 2; ++nops) {
        for (int mode = 0; mode < 2; ++mode) {       // 0: pk kernel alone (control), 1: under the MFMA kernel on a second stream
            CK(hipMemset(bad, 0, 68 * 4));
            CK(hipMemset(ex, 0, sizeof(hex_)));
            CK(hipDeviceSynchronize());
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, sp));
            for (int l = 0; l < launches; ++l) {
                // one MFMA workgroup per CU (4 waves: one per SIMD), ~2 ms each, re-launched so that it is ALWAYS there while P runs
                if (mode == 1 && (l % 4) == 0) hipLaunchKernelGGL(mfma_kernel, dim3(prop.multiProcessorCount), dim3(256), 0, sm, src, sink, 6000);
                if (nops == 0) hipLaunchKernelGGL((pk_kernel<0>), dim3(grid), dim3(256), 0, sp, xs, ms, cs, n, iters, bad, ex);
                else hipLaunchKernelGGL((pk_kernel<1>), dim3(grid), dim3(256), 0, sp, xs, ms, cs, n, iters, bad, ex);
            }
            CK(hipEventRecord(e1, sp));
            CK(hipDeviceSynchronize());
            float msec = 0.f;
            CK(hipEventElapsedTime(&msec, e0, e1));
            CK(hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost));
            const double total = (double)launches * grid * 256 * iters;
            printf("%s, %s: %d launches, %.3g packed FMAs, %.1f ms: low-half mismatches %u, high-half mismatches %u\n",
                   nops ? "s_nop 4 in front" : "bare instruction", mode ? "UNDER the MFMA kernel (second stream)" : "alone (control)", launches, total,
                   msec, hb[0], hb[1]);
            CK(hipMemcpy(hex_, ex, sizeof(hex_), hipMemcpyDeviceToHost));
            for (int k = 0; k < 16; ++k) {
                const float* e = hex_ + k * 12;
                if (e[2] == 0.f && e[3] == 0.f) continue;
                if (k >= 8 && k >= 10) continue;         // two high-half examples are enough
                printf("  %s-half example, lane %2d: x = (%.9g, %.9g) m = (%.9g, %.9g) c = (%.9g, %.9g) -> packed (%.9g, %.9g), scalar fma(x0, m1, c0) = %.9g, fma(x1, m1, c1) = %.9g;"
                       " fma(x0, m0, c0) = %.9g, fma(x1, m0, c1) = %.9g\n", e[0] == 0.f ? "LOW" : "HIGH", (int)e[1], e[2], e[3], e[4], e[5], e[6], e[7], e[8], e[9], e[10], e[11],
                       fmaf(e[2], e[4], e[6]), fmaf(e[3], e[4], e[7]));
            }
            if (hb[0]) {
                printf("  per-lane low-half failures:");
                for (int i = 0; i < 64; ++i) if (hb[2 + i]) printf(" %d:%u", i, hb[2 + i]);
                printf("\n");
            }
        }
    }
    printf("(the product is built without the SLP vectoriser: libmhmr.so contains no op_sel-swizzled packed fp32 instruction)\n");
    return 0;
}
