// Internal (non-ABI) declarations shared by the kernel translation units of libmhmr.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mhmr.h"

enum GemmEpilogue {
    EPI_OP16 = 0,       // out16[m][n] = acc + bias
    EPI_OP16_GELU = 1,  // out16[m][n] = gelu(acc + bias)
    EPI_OP16_RELU = 2,  // out16[m][n] = relu(acc + bias)
    EPI_RESID = 3,      // out32[m][n] += gamma[n] * (acc + bias[n])        (LayerScale + residual, in place)
    EPI_PATCH = 4,      // out32[row(m)][n] = acc + bias + pos[1 + m % Np][n],  row(m) = (m / Np) * Tp + 1 + m % Np
    EPI_F32 = 5,        // out32[m][n] = acc (+ bias)
    EPI_VT = 6,         // vt[b][h][d][swap23(t)] = acc + bias               (V^T for the attention kernel)
    EPI_OP16_QK = 7,    // out16[m][n] = (acc + bias) * (n < N/2 ? MHMR_ATTN_QSCALE : 1)   (Q | K projection, softmax scale folded into Q)
};

struct GemmArgs {
    const void* A; int lda;
    const void* W; int ldw;
    int M, N, K;
    const float* bias;
    const float* gamma;
    void* out; int ldo;
    const float* pos;
    int Np, Tp, H, Mvalid;
    int epi;
    int stagger_ticks = 0;   // gemm256: CU quarter q starts q * stagger_ticks (100 MHz wall clock) late
};

int mhmr_launch_gemm(const GemmArgs& g, int dtype, hipStream_t s);

// ---- per-kernel-family hipEvent profiling (bench.py roofline leg) ----
enum ProfKind { PROF_GEMM = 0, PROF_ATTN = 1, PROF_LBS = 2, PROF_KINDS = 3 };
void prof_begin(int kind, hipStream_t s);
void prof_end(int kind, hipStream_t s, double work);
