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
    // Token-row map of the activation / output rows (ViT blocks): logical row m = b * img_rows + n lives at physical row
    // b * img_stride + n.  The big linears then run over the B * N patch rows only (exactly 8 / 16 / 32 rounds of 256 tiles at
    // 896^2 x 32) and skip the class + padding rows of every image (vit_cls.hip computes the class rows).  0 = rows are physical.
    int img_rows = 0, img_stride = 0;
    unsigned img_magic = 0;  // floor(2^32 / (img_rows / 256)) + 1 (0 when img_rows == 256), set by mhmr_launch_gemm: image of row tile tm = umulhi(tm, img_magic)
    // Low-half weight pass: W = [W_hi | W_lo] along k (ldw >= K, K = 2 * a_k): k tiles >= a_k / 64 re-read the activation's k tiles
    // from the start, so acc = A . W_hi^T + A . W_lo^T in one accumulator chain.  0 = off.
    // K = 3 * a_k (the f16x3 precision mode): W = [W_hi | W_lo | W_hi], A = [A_hi | A_lo] (lda >= 2 a_k): the k tiles of the third range
    // read the activation's SECOND a_k columns, so acc = A_hi W_hi^T + A_hi W_lo^T + A_lo W_hi^T -- three 16-bit products per term.
    int a_k = 0;
    // fp8 low-half range (gemm256 only; the weight's LOW half needs three significant bits, not eleven).  lo8 != 0: BOTH operand rows are
    // [a_k op16 values | a_k bytes of fp8] (lda, ldw >= K = a_k + a_k / 2 in 16-bit units), and the k tiles behind a_k are 128-deep fp8 tiles
    // on v_mfma_scale_f32_16x16x128_f8f6f4 (twice the 16-bit rate) -- same 16 KiB half-tile slots, same copies, same fragment reads, no
    // activation wrap-around: only the MFMA differs.  Weight bytes: e4m3 of W_lo * 2^-e with w8_scale = 127 + e (E8M0); activation bytes:
    // bf8 (e5m2) of the SAME values as the 16-bit part, unscaled, written by the activation's producer (x8_off below; attention's out8).
    int lo8 = 0;
    int w8_scale = 127;
    // producer side (EPI_RESID with x16): row pitch of x16 in elements (0 = ldo) and, if x8_off > 0, the byte offset inside an x16 row where
    // the bf8 copy of the row's N values goes (the next linear's fp8 low-half range)
    int ldx16 = 0, x8_off = 0;
    int colgroup = 0;        // gemm256: column-group tile order for wide outputs: log2 of the weight column panels an XCD keeps (2 = four panels, 3 = eight; 0 = off); set by mhmr_launch_gemm, MHMR_COLGROUP overrides
    // LayerNorm folded into the neighbouring GEMMs (gemm256 only): the LayerNorm pass of its own disappears.
    //   producer (EPI_RESID): besides the fp32 residual rows it writes x16 = their 16-bit copy (RAW, un-normalised: the next linear's A
    //   operand) and pstats[row][N / 64][2] = (sum, sum of squares) of every 64-column block of the row (summed over the row by
    //   ln_stats_kernel -> rowstats[row] = (mean, rstd));
    //   consumer (EPI_OP16_QK / EPI_VT / EPI_OP16_GELU with rowstats != null): W carries the LayerNorm weight (W' = W diag(w_ln)),
    //   out = rstd_m * (acc - mean_m * colsum_n) + fbias_n,  colsum_n = sum_k W'[n][k],  fbias = b + W . b_ln;  `bias` must be null.
    void* x16 = nullptr;
    float* pstats = nullptr;
    const float* rowstats = nullptr;
    const float* colsum = nullptr;
    const float* fbias = nullptr;
    // Split-k (gemm256 only, EPI_F32, no bias): ksplit > 0 = k tiles (of 64) per slice, nslices slices; slice s writes its fp32 partial
    // product into out + s * M * ldo (ldo == N).  For launches of fewer tiles than half the CUs (mhmr_splitk_plan, capi.hip).
    int ksplit = 0, nslices = 0;
    // Masked output width (gemm256 only, EPI_RESID / EPI_VT): n_valid > 0 and != N: the real width, N - 128 (N = roundup(n_valid, 256));
    // W and every per-column array (bias, gamma, colsum, fbias) must be readable -- zero-padded -- up to N; `out`, x16 and pstats have the
    // real width (ldo = n_valid or wider; pstats[row][n_valid / 64][2]); the V^T output has n_valid / 64 heads.
    int n_valid = 0;
    // Merged qkv linear (gemm256 only, EPI_OP16_QK): out2 != null: output columns >= split_col go row-major to out2 [M, ldo2] (column
    // n - split_col), the others to out; the softmax scale applies to columns < qcols.
    void* out2 = nullptr;
    int ldo2 = 0, split_col = 0, qcols = 0;
};

// physical row of logical activation row m (see GemmArgs::img_rows)
__host__ __device__ inline long long mhmr_phys_row(int m, int img_rows, int img_stride) {
    return img_rows > 0 ? (long long)(m / img_rows) * img_stride + (m % img_rows) : (long long)m;
}

int mhmr_launch_gemm(const GemmArgs& g, int dtype, hipStream_t s);

// Per-DEVICE one-time kernel attributes (hipFuncAttributeMaxDynamicSharedMemorySize is a property of the function ON A DEVICE: a process
// that drives a second GPU must set it there too).  One bit per device id in a launcher-local mask; setting an attribute twice from two
// host threads is harmless, so a relaxed fetch_or is enough.  Device ids >= 64 set the attribute on every call.
#include <atomic>
struct DeviceOnce {
    std::atomic<uint64_t> done{0};
    // -> the current device id if the caller must (re)apply its attribute there, -1 if already applied, -2 on a HIP error
    int need(int* dev_out) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return -2;
        *dev_out = dev;
        if (dev >= 0 && dev < 64 && ((done.load(std::memory_order_relaxed) >> dev) & 1ull)) return -1;
        return dev;
    }
    void mark(int dev) { if (dev >= 0 && dev < 64) done.fetch_or(1ull << dev, std::memory_order_relaxed); }
};
// multiProcessorCount of the CURRENT device (cached per device id)
int mhmr_cu_count();

// "Any-order" launches (hipExtAnyOrderLaunch: the AQL packet goes out WITHOUT the barrier bit, so the command processor does not wait for
// the previous kernel of the stream to finish before dispatching this one).  capi.hip sets the thread-local flag around launches whose
// inputs were complete before the PREVIOUS launch started and whose outputs nothing touches until the next ordinary launch (which waits
// for everything before it): the V projection behind the Q | K projection, the class-row linears behind the big GEMM of the same
// linear.  MEASURED (round 6, tools/ubench/anyorder.hip + a kernel trace of the forward, profiles/r06_session_a.txt / _b.txt): on gfx950 /
// ROCm 7.2 such a kernel starts about when the kernel in front of it ends -- a persistent GEMM holds every CU until then, so nothing
// overlaps for long (the trace shows ~8 boundaries per forward where the next kernel starts 0.1-0.2 us BEFORE its predecessor's end;
// hip_ext.h calls the flag "not supported on GFX9xx") -- but the ordering IS relaxed: a class-row launch that also read the block sums
// of the GEMM in front of it, launched this way by mistake for one session, returned wrong, run-to-run different statistics.  The
// headline step is 0.1-0.8 % faster in six of six interleaved A/B pairs.  On by default, ONLY for launches independent of their predecessor.  MHMR_ANYORDER=0 switches it off (A/B measurements); never set while a stream is being captured.
extern thread_local int g_mhmr_anyorder;
#include <hip/hip_ext.h>
template <typename F, typename... Args>
inline void mhmr_launch_kernel(F kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, Args... args) {
    if (g_mhmr_anyorder) hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, s, nullptr, nullptr, hipExtAnyOrderLaunch, args...);
    else hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
}

// Row statistics carried by the class-row kernel's launches (vit_cls.hip, round 6).  cls_pstats [rows][cls_nblk][2]: block sums of the class
// rows (written by epi = 1, read by epi = 0 / 2 in place of `rowstats`); st_*: the patch rows' statistics as extra workgroups of an epi = 1
// launch (st_pstats = the big GEMM's block sums, st_rowstats = its consumers' (mean, rstd); st_B images x st_N patch rows of st_Tp-row images).
struct ClsStats {
    float* cls_pstats = nullptr;
    int cls_nblk = 0, cls_C = 0;
    float eps = 1e-6f;
    const float* st_pstats = nullptr;
    float* st_rowstats = nullptr;
    int st_B = 0, st_N = 0, st_Tp = 0, st_C = 0;
};

// ---- per-kernel-family hipEvent profiling (bench.py roofline leg) ----
enum ProfKind { PROF_GEMM = 0, PROF_ATTN = 1, PROF_LBS = 2, PROF_KINDS = 3 };
void prof_begin(int kind, hipStream_t s);
void prof_end(int kind, hipStream_t s, double work);
