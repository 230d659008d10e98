// Common device helpers for the Multi-HMR gfx950 (MI355X / CDNA4) kernels.
// wave = 64 lanes; MFMA 32x32x16 (bf16|f16 operands, fp32 accumulate); LDS tiles are [rows][64] 16-bit
// (128-byte rows) filled by global_load_lds_dwordx4 with an XOR swizzle applied on the SOURCE address.
#pragma once
// Every translation unit of this library must be compiled with -fno-slp-vectorize (multi_hmr_amd/_lib.py passes -DMHMR_NO_SLP beside it).
// hipcc's SLP vectoriser turns groups of scalar fp32 operations into v_pk_{fma,mul,add}_f32 with op_sel operand swizzles; two kernels built
// that way returned wrong values on gfx950 although the packed and the scalar instruction streams are arithmetically the same when read
// line by line:
//   lbs.hip      (round 2)  a projection computed with a zero focal length for about one (person, vertex tile) pair in 10^4, the same
//                           pairs on every run of one build (tests/test_gpu_kernels.py::test_lbs_max_abs_gate_160_persons_x_20_seeds);
//   vit_cls.hip  (round 3)  the class-row linear with the folded-LayerNorm epilogue: elements 0 / 2 of lanes 48 ... 63 -- the LOW halves of
//                           v_pk_fma_f32 ... op_sel:[0,1,0] (multiplier = the high dword of a register pair, for both halves) -- came
//                           out wrong in 1 of 40 ... 600 forwards, and ONLY while a second stream kept foreign waves (MFMA GEMMs,
//                           attention) on the same SIMDs: serial runs are bit-reproducible.  Re-assembling the same device code with
//                           s_nop in front of every such instruction, or between the operand's load wait and its first use, lowers the
//                           rate but does not remove it (profiles/r03_slp_waitstate_variants.txt), so it is not a missing wait state
//                           of the dependent-instruction kind; without the SLP vectoriser: 0 of 720 (profiles/
//                           r03_two_stream_slp_vs_noslp.txt).  Gate: tests/test_gpu_model.py::
//                           test_two_host_threads_two_streams_are_independent; tool: tools/two_stream_check.py.
// Round 5 reproduced it in isolation (tools/ubench/slp_repro.hip, profiles/r05_slp_repro_with_tuples.txt): v_pk_fma_f32 ... op_sel:[0,1,0] by
// inline assembly next to the two v_fma_f32 it stands for, 6.7e9 executions: alone 0 wrong results; while a persistent MFMA kernel on a second
// stream shares the SIMDs, 0.7 % of all executions are wrong -- ALL in lanes 48 ... 63, always the LOW half, and the wrong value is exactly
// the addend (the product term is lost); s_nop in front changes nothing.  An erratum of this gfx950 stack, avoided by construction:
// The no-SLP build has no op_sel-swizzled packed fp32 instruction left in any kernel (they were in all of them: 1472 in gemm256.hip,
// 484 in attention.hip, 360 in hph.hip) and is also 2.1 % FASTER on the whole forward (138.9 vs 141.9 ms, same box, profiles/
// r03_slp_ab.txt): scalar fp32 beside MFMAs is what MI355X_MICROARCH.md recommends.  Any other build recipe fails here, not at run time.
#ifndef MHMR_NO_SLP
#error "libmhmr: build every .hip file with -fno-slp-vectorize -DMHMR_NO_SLP (see the comment above)"
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mhmr.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// Operand-type traits: the two 16-bit MFMA operand formats run at the same rate on gfx950; bf16 is the
// north-star dtype, f16 is what meets 1e-3 parity against the fp32 CPU reference.
template <int DT> struct Op;
template <> struct Op<MHMR_DT_BF16> {
    typedef __bf16 T;
    typedef bf16x8 V8;
    typedef bf16x4 V4;
    typedef bf16x2 V2;
    static __device__ __forceinline__ f32x16 mfma32(V8 a, V8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(V8 a, V8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float pair_sum(uint32_t packed, float acc) {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, packed), __builtin_bit_cast(bf16x2, 0x3F803F80u), acc, false);
    }
};
template <> struct Op<MHMR_DT_F16> {
    typedef _Float16 T;
    typedef f16x8 V8;
    typedef f16x4 V4;
    typedef f16x2 V2;
    static __device__ __forceinline__ f32x16 mfma32(V8 a, V8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(V8 a, V8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    // acc + lo + hi of one packed pair (v_dot2_f32_f16 against (1, 1)): two exact products, fp32 accumulation.
    // The BUILTIN, not inline assembly (rounds 3-6 had `asm("v_dot2_f32_f16 ...")` here): on gfx90a+ a VALU instruction that is not the
    // same dot opcode needs THREE wait states before it reads a dot product's result, and the compiler's hazard recogniser does not look
    // inside an asm statement.  The shipped attention loop happened to have exactly three instructions in between (the next query block's
    // dot product, a wait, an MFMA: checked in the ISA); the class-query role of round 6 (attention.hip cls_dot2) did not and read stale
    // sums.  With the builtin the compiler places the s_nop itself.
    static __device__ __forceinline__ float pair_sum(uint32_t packed, float acc) {
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, packed), __builtin_bit_cast(f16x2, 0x3C003C00u), acc, false);
    }
};

// fp32 x4 -> four bf8 (e5m2) bytes, round to nearest even, clamped to the format's finite range first (+-57344)
__device__ __forceinline__ uint32_t pack_bf8x4(float a, float b, float c, float d) {
    const float lim = 57344.f;
    a = __builtin_amdgcn_fmed3f(a, -lim, lim); b = __builtin_amdgcn_fmed3f(b, -lim, lim);
    c = __builtin_amdgcn_fmed3f(c, -lim, lim); d = __builtin_amdgcn_fmed3f(d, -lim, lim);
    int r = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false);
    r = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, r, true);
    return (uint32_t)r;
}

// C/D fragment of a 32x32 MFMA: lane l holds column (l & 31) and rows crow(r, l >> 5), r = 0..15.
__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// 16-byte-chunk XOR swizzle of a [rows][64 x 16-bit] LDS tile (8 chunks per 128-byte row).  A lane group of
// ds_read_b128 reading one logical chunk over rows (lane & 31) then touches 16 distinct 16-byte slots.
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// Asynchronous 16 B/lane global -> LDS copy.  LDS destination = wave-uniform base + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// GELU for 16-bit outputs.  x Phi(x) = max(x, 0) - |x| Q(|x|), Q(a) = erfc(a / sqrt 2) / 2 the upper tail of the normal law (erf is
// odd: no sign select).  Round 6: Q(a) = exp2(P(a)) with P a degree-5 polynomial fitted to log2 Q -- a smooth, concave function: -1 at
// 0, ~ -a^2 / (2 ln 2) far out -- by weighted minimax on the ABSOLUTE error of a exp2(P(a)) over [0, 12] (tools/gelu_fit.py makes the
// constants; tests/test_host.py restates them).  |abs err on gelu| < 4.4e-7 as a formula, < 7e-7 evaluated in fp32 -- 40x closer than
// the form of rounds 4-5 (Abramowitz-Stegun 7.1.25, three terms: 2.6e-5) -- for 8 VALU instructions with ONE transcendental instead of
// 11 with two (rcp + exp2): the epilogue's arithmetic was 0.14 ms of an fc1 launch at the headline.  The leading coefficient is
// negative, so P -> -inf and the tail term -> 0 for any large |x| (massive activations); exp2 of < -126 flushes to zero.
__device__ __forceinline__ float gelu_fast(float x) {
    const float ax = fabsf(x);
    float p = __builtin_fmaf(ax, -0.0004733092791866511f, 0.007084557320922613f);
    p = __builtin_fmaf(p, ax, -0.051827382296323776f);
    p = __builtin_fmaf(p, ax, -0.4599924385547638f);
    p = __builtin_fmaf(p, ax, -1.1507878303527832f);
    p = __builtin_fmaf(p, ax, -1.000037670135498f);
    return __builtin_fmaf(-ax, __builtin_amdgcn_exp2f(p), fmaxf(x, 0.f));
}

// XCD-aware remap of a linear workgroup id: the dispatcher places block b on XCD b % 8 (speed only, never
// correctness); give each XCD a contiguous chunk of the logical grid so neighbouring tiles share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

#define MHMR_CHECK_LAUNCH()                          \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)
