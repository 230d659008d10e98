// Flash-style multi-head self-attention forward for the DINOv2 blocks, head_dim = 64, non-causal,
// T = N + 1 tokens padded to Tp (multiple of 128); keys >= T are masked.
//
// gfx950 structure: one workgroup = 4 waves = 128 query rows of one (image, head); each wave owns 32 query
// rows.  Per 64-key tile:  S^T = K . Q^T  (8 x v_mfma_f32_32x32x16, "swapped" so a lane owns ONE query column
// and 32 of the 64 keys -> row max / row sum are lane-local plus one lane^32 exchange), online softmax in
// fp32 registers, P packed to 16-bit in place as the B operand of  O^T += V^T . P^T  (8 MFMAs).  V arrives
// already transposed and key-permuted (bits 2<->3 of the key index swapped inside every 16-key group, written
// that way by the V GEMM epilogue), so the lane that holds P for keys {16s+4hi+0..3, 16s+8+4hi+0..3} reads the
// matching V^T operand as ONE ds_read_b128 -- no cross-lane shuffle and no LDS transpose.
// K and V^T tiles ([64][64] 16-bit = 128-byte rows) are DMA'd by global_load_lds_dwordx4 into a 2-slot LDS ring with
// the chunk XOR swizzle on the source address, one tile ahead (the whole consume phase of tile j covers tile j+1's
// flight); one barrier per KV tile; 32 KiB LDS and 114 VGPRs put 4 workgroups = 16 waves on a CU (measured +2.8 % over a
// 3-slot ring at 3 workgroups/CU).  The landing wait is an EXPLICIT `s_waitcnt vmcnt(0)`: inside a loop hipcc
// (ROCm 7.2) does NOT emit the vmcnt wait for LDS-DMA in front of __syncthreads() -- it hoisted it out of the loop -- and
// the kernel then read tiles that had not landed (run-to-run different results; tools/determinism.py).
#include "mhmr_common.h"
#include "mhmr_internal.h"

namespace {

// max over the lane pair {l, l ^ 32} on the VALU (v_permlane32_swap): a ds_bpermute would queue behind the fragment reads in the LDS
__device__ __forceinline__ float max_lane32(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

constexpr int KB = 64;
constexpr int KV_TILE_BYTES = KB * 64 * 2;  // 8 KiB

// NW waves x 32 query rows per workgroup; RING K/V tile slots in LDS (tile j + RING - 1 is in flight while tile j is consumed)
template <int DT, int NW, int RING>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 4 : 2) void attn_kernel(const void* __restrict__ qk_, const void* __restrict__ vt_,
                                                      void* __restrict__ out_, int T, int Tp, int C, int H,
                                                      int nqt, float scale_log2e) {
    typedef typename Op<DT>::T Tt;
    typedef typename Op<DT>::V8 V8;
    typedef typename Op<DT>::V4 V4;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][K tile | Vt tile]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;

    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int qt = bid % nqt, bh = bid / nqt;
    const int b = bh / H, h = bh - b * H;

    const Tt* qk = (const Tt*)qk_;
    const Tt* vt = (const Tt*)vt_;
    const int ldq = 2 * C;
    const size_t row0 = (size_t)b * Tp;

    // ---- Q fragments (B operand: lane (q = l31, hi) holds Q[q][16 ks + 8 hi + 0..7]) ----
    constexpr int QB = 32 * NW;
    const int q_row = qt * QB + 32 * w + l31;
    const bool active = qt * QB + 32 * w < T;  // a wave whose 32 query rows are all padding (T = 64 n + 1: three of the four waves
                                               // of every image's last workgroup) skips the arithmetic and stores zeros
    const bool in_buf = qt * QB + 32 * w < Tp;   // QB = 256: the last workgroup's upper waves lie past the padded rows
    V8 qf[4] = {};
    if (in_buf) {
        const Tt* qp = qk + (row0 + q_row) * ldq + h * 64 + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const V8*)(qp + 16 * ks);
    }

    // ---- staging addresses: one glds16 per thread moves 64 * NW chunks of 16 B = 8 * NW tile rows ----
    constexpr int PASSES = 8 / NW, ROWS_PER_PASS = 8 * NW;        // NW = 4: 2 passes of 32 rows; NW = 8: 1 pass of 64 rows
    constexpr int OPS = 2 * PASSES;                                // DMA instructions per tile per thread
    const int srow = tid >> 3;
    const int schunk = (tid & 7) ^ ((tid >> 4) & 7);
    const Tt* k_src = qk + (row0 + srow) * ldq + C + h * 64 + schunk * 8;                 // + key0 * ldq
    const Tt* v_src = vt + ((size_t)(b * H + h) * 64 + srow) * Tp + schunk * 8;           // + key0
    auto stage = [&](int j, int buf) {
        char* sk = smem + buf * (2 * KV_TILE_BYTES) + w * 1024;
        char* sv = sk + KV_TILE_BYTES;
        const Tt* kp = k_src + (size_t)j * KB * ldq;
        const Tt* vp = v_src + j * KB;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) glds16(kp + (size_t)(ROWS_PER_PASS * ps) * ldq, sk + ps * (ROWS_PER_PASS * 128));
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) glds16(vp + (size_t)(ROWS_PER_PASS * ps) * Tp, sv + ps * (ROWS_PER_PASS * 128));
    };

    const int fsw = (lane >> 1) & 7;

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;  // running max (scaled, log2 domain) and this lane's partial row sum

    const int ntile = (T + KB - 1) / KB;
#pragma unroll
    for (int t = 0; t < RING - 1; ++t) stage(t < ntile ? t : ntile - 1, t);
    int buf = 0, nbuf = RING - 1;                // ring slot of tile j, and of tile j + RING - 1
    for (int j = 0; j < ntile; ++j) {
        // tile j has landed (this wave's part: the counted vmcnt -- only the RING - 2 newer tiles may still be in flight; every
        // wave's: the barrier), and every wave is done reading tile j-1, whose slot takes tile j + RING - 1
        if constexpr (RING == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if constexpr (OPS * (RING - 2) == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        else if constexpr (OPS * (RING - 2) == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else if constexpr (OPS * (RING - 2) == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        static_assert(OPS * (RING - 2) <= 8, "extend the counted waits");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const char* sk = smem + buf * (2 * KV_TILE_BYTES);
        const char* sv = sk + KV_TILE_BYTES;
        const int nbuf_now = nbuf;
        buf = buf == RING - 1 ? 0 : buf + 1;
        nbuf = nbuf == RING - 1 ? 0 : nbuf + 1;
        if (!active) {          // wave-uniform: the wave still stages its share of every tile and meets every barrier
            stage(j + RING - 1 < ntile ? j + RING - 1 : ntile - 1, nbuf_now);
            continue;
        }
        // ---- S^T = K . Q^T ----
        f32x16 s[2];   // the two key halves alternate in issue order: back-to-back MFMAs never share an accumulator
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[sub][r] = 0.f;
        __builtin_amdgcn_s_setprio(1);   // matrix sections outrank the other waves' VALU work at the issue arbiter (+1-2 %)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            V8 kf[2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) kf[sub] = *(const V8*)(sk + (32 * sub + l31) * 128 + (((2 * ks + hi) ^ fsw) * 16));
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) s[sub] = Op<DT>::mfma32(kf[sub], qf[ks], s[sub]);
        }
        __builtin_amdgcn_s_setprio(0);
        // the next tile's DMA is issued behind the score MFMAs (its address arithmetic no longer delays the first fragment reads);
        // tail: harmless re-load, keeps the wait count uniform
        stage(j + RING - 1 < ntile ? j + RING - 1 : ntile - 1, nbuf_now);
        __builtin_amdgcn_sched_barrier(0);
        // ---- mask keys >= T (only the last tile can contain them) ----
        if (j * KB + KB > T) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (j * KB + 32 * sub + crow(r, hi) >= T) s[sub][r] = -INFINITY;
        }
        // ---- online softmax (one query row per lane pair {l, l^32}) ----
        float mt = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[1][r]);
        mt = max_lane32(mt);
        const float m_new = fmaxf(m_run, mt * scale_log2e);
        if (!__all(m_new == m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
            m_run = m_new;
        }
        float psum = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[sub][r] * scale_log2e - m_run);
                s[sub][r] = p;
                psum += p;
            }
        l_run += psum;

        // ---- O^T += V^T . P^T ----
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            V8 pf;
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[e] = (Tt)s[st >> 1][8 * (st & 1) + e];
#pragma unroll
            for (int ds = 0; ds < 2; ++ds) {
                const V8 vf = *(const V8*)(sv + (32 * ds + l31) * 128 + (((2 * st + hi) ^ fsw) * 16));
                o[ds] = Op<DT>::mfma32(vf, pf, o[ds]);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- normalise and store: lane (q, hi) holds O[q][32 ds + 8 rg + 4 hi + 0..3] ----
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = active ? 1.0f / l_tot : 0.f;
    if (!in_buf) return;
    Tt* op = (Tt*)out_ + (row0 + q_row) * C + h * 64 + 4 * hi;
#pragma unroll
    for (int ds = 0; ds < 2; ++ds)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            V4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (Tt)(o[ds][4 * rg + e] * inv);
            *(V4*)(op + 32 * ds + 8 * rg) = v;
        }
}


}  // namespace

template <int NW, int RING>
static int launch_attn(const void* qk, const void* vt, void* out, int B, int T, int Tp, int C, int H, int dtype, hipStream_t s) {
    constexpr int QB = 32 * NW;
    const int nqt = (Tp + QB - 1) / QB;
    const int grid = nqt * H * B;
    const size_t lds = RING * 2 * KV_TILE_BYTES;
    const float scale_log2e = 0.125f * 1.44269504088896340736f;
    if (dtype == MHMR_DT_F16)
        hipLaunchKernelGGL((attn_kernel<MHMR_DT_F16, NW, RING>), dim3(grid), dim3(64 * NW), lds, s, qk, vt, out, T, Tp, C, H, nqt, scale_log2e);
    else
        hipLaunchKernelGGL((attn_kernel<MHMR_DT_BF16, NW, RING>), dim3(grid), dim3(64 * NW), lds, s, qk, vt, out, T, Tp, C, H, nqt, scale_log2e);
    return 0;
}

// Measured at ViT-L 896 b32 (tools/kbench.py): <4,2> 852, <4,3> 832, <8,2> 855, <8,3> 855-860, <8,4> 854 TFLOP/s -- halving the
// L2->LDS traffic (NW = 8) or deepening the ring changes nothing: the kernel is bound by the per-wave issue mix (16 MFMA, ~150 VALU
// incl. 32 v_exp, 16 ds_read_b128 per tile), and only waves/SIMD moved it (3 -> 4: +2.8 %).  <4,2> is the shipped configuration.
// Role-structured variants were built, passed the parity and determinism tests, and lost: next tile's scores issued in the same
// basic block as this tile's exps (two score tiles in registers, 3 waves/SIMD) 784; 8-wave two-group ping-pong (VALU phase /
// 16-MFMA phase, fragments pre-read into registers) 640-740; 12-wave three-group rotation (one matrix wave + two VALU waves per
// SIMD at any time, 4-slot ring) 795-822 TFLOP/s.  A lone wave issues VALU at ~5.5 cycles per instruction (tools/ubench), so the
// softmax of one tile is 1100+ cycles beside 512 matrix-pipe cycles; four unsynchronised waves per SIMD hide that best.
int mhmr_launch_attention(const void* qk, const void* vt, void* out, int B, int T, int Tp, int C, int H, int dtype,
                          hipStream_t s) {
    if (C != H * 64 || Tp % 128 || T > Tp || T <= 0) return MHMR_ERR_BAD_SHAPE;
    prof_begin(PROF_ATTN, s);
    launch_attn<4, 2>(qk, vt, out, B, T, Tp, C, H, dtype, s);
    prof_end(PROF_ATTN, s, 4.0 * B * H * (double)T * T * 64);
    MHMR_CHECK_LAUNCH();
    return 0;
}
