// Flash-style multi-head self-attention forward for the DINOv2 blocks, head_dim = 64, non-causal,
// T = N + 1 tokens padded to Tp (multiple of 64 = the key tile; query rows past Tp in the last workgroup are skipped); keys >= T are masked.
//
// Input contract: the Q half of `qk` arrives PRE-SCALED by MHMR_ATTN_QSCALE = 0.125 * log2(e) (the QK projection's epilogue
// multiplies in fp32 before the one rounding to 16 bits, MHMR_EPI_OP16_QK), so the scores leave the matrix pipe already in the
// exp2 domain and the softmax needs no per-element multiply.
//
// gfx950 structure: one workgroup = 4 waves = 128 query rows of one (image, head); each wave owns 32 query
// rows.  Per 64-key tile:  S^T = K . Q^T  (8 x v_mfma_f32_32x32x16, "swapped" so a lane owns ONE query column
// and 32 of the 64 keys -> row max / row sum are lane-local plus one lane^32 exchange), online softmax in
// fp32 registers, P packed to 16-bit in place as the B operand of  O^T += V^T . P^T  (8 MFMAs).  V arrives
// already transposed and key-permuted (bits 2<->3 of the key index swapped inside every 16-key group, written
// that way by the V GEMM epilogue), so the lane that holds P for keys {16s+4hi+0..3, 16s+8+4hi+0..3} reads the
// matching V^T operand as ONE ds_read_b128 -- no cross-lane shuffle and no LDS transpose.
//
// Softmax.  The VALU, not the matrix pipe, bounds this kernel at d = 64: 16 MFMAs = 512 pipe cycles per tile beside ~400 VALU
// cycles in the textbook online-softmax form (tools/ubench: exp 5.3, max / max3 / cvt_pk 2.85, add / sub / mov 1.8 SIMD cycles per
// wave instruction at 4 waves per SIMD; one wave alone issues at most one VALU instruction per ~5.3 cycles).  The per-tile VALU
// work is cut by keeping, per query, a reference level m_ref that is SUBTRACTED INSIDE THE MATRIX PIPE: the score accumulators
// start at -m_ref instead of 0 (16 v_mov: both key halves take the same tuple as their first C operand), so p = exp2(s) needs
// no per-element subtract or multiply, and m_ref only has to be NEAR the row maximum, not equal to it:
//   MODE 2  m_ref moves when the running row maximum leaves [m_ref - 8, m_ref + 8] (wave-uniform branch; exact: O, l and the
//           tile's scores are shifted by the same power of two).  Self-contained.
//   MODE 3  m_ref = the exact row maximum of key tile 0, then fixed: no row maximum at all on later tiles.  Every p is positive,
//           so "this lane's tile sum <= 2^15" proves every p of the lane finite in f16 / bf16; a wave that ever sees a larger
//           sum (a later key beats tile 0's maximum by more than 2^15: rare) raises its workgroup's flag, and the textbook
//           kernel (MODE 1), launched right behind on the same stream, recomputes exactly the flagged workgroups and returns at
//           once everywhere else.
//   MODE 1  textbook: exact running maximum and a per-element subtract every tile.
// MODE 3 at T = 64 n + 1 (every DINOv2 grid + class token): the lone key of the last tile is a rank-1 update after the loop (fp32 p,
// V row through a wave-private LDS strip) instead of a 65th tile with 63 masked keys: 2.309 -> 2.275 ms at B = 32, T = 4097.
// Nothing is approximated in any mode: p keeps its full significand at any magnitude, l and O accumulate in fp32, and O / l is
// invariant to the reference level; the modes differ in rounding order only (tests: all three against fp64 on inputs that force
// every branch).
//
// K and V^T tiles ([64][64] 16-bit = 128-byte rows) are DMA'd by global_load_lds_dwordx4 into a 2-slot LDS ring with
// the chunk XOR swizzle on the source address, one tile ahead (the whole consume phase of tile j covers tile j+1's
// flight); one barrier per KV tile.  The landing wait is an EXPLICIT `s_waitcnt vmcnt(0)`: inside a loop hipcc
// (ROCm 7.2) does NOT emit the vmcnt wait for LDS-DMA in front of __syncthreads() -- it hoisted it out of the loop -- and
// the kernel then read tiles that had not landed (run-to-run different results; tools/determinism.py).
//
// Round 4: mhmr_vit_forward runs attn16_kernel (variant 6, further down): the MODE 3 arithmetic of this file on v_mfma_f32_16x16x32 -- the
// 32x32x16 shape used by the kernels up here costs 5-8 % more power per flop on this power-clocked chip.  1116-1128 against 973-1048 TF/s
// alone, 1052-1059 against 1015 inside the forward (f16, one box, interleaved: profiles/r04_session_i_attention_16x16x32.txt).  The forms
// above stay as selectable variants (0-5) with their tests; MODE 1 remains the fallback of every gated form.
//
// Measured and rejected on the MODE 3 form (B = 32, H = 16, T = 4097, same process, interleaved): K fragments double-buffered one
// k-step ahead with pinned issue order (120 VGPRs) 933-936 vs 900-953 TFLOP/s f16 for this form, 990-997 vs 945-1016 bf16; 8 waves
// with a 3-slot ring 862-908; both together 954 / 1006: all inside the run-to-run spread of the plain 4-wave form, which stays.
#include <stdlib.h>
#include <type_traits>
#include "mhmr_common.h"
#include "mhmr_internal.h"

namespace {

// max over the lane pair {l, l ^ 32} on the VALU (v_permlane32_swap): a ds_bpermute would queue behind the fragment reads in the LDS
__device__ __forceinline__ float max_lane32(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

constexpr int KB = 64;
#ifndef MHMR_ATTN_DEFAULT_VARIANT
#define MHMR_ATTN_DEFAULT_VARIANT 6          // what mhmr_vit_forward runs (MHMR_ATTN_VARIANT overrides at run time: A/B measurements)
#endif
constexpr int KV_TILE_BYTES = KB * 64 * 2;  // 8 KiB
constexpr float BAND = 8.f;                 // MODE 2: half-width of the band around the reference level (exp2 domain)

// NW waves x 32 query rows per workgroup; RING K/V tile slots in LDS (tile j + RING - 1 is in flight while tile j is consumed).
// flags: four ints per workgroup (logical id), one per wave.  MODE 3 writes every one of them (1 = this workgroup's result must be
// recomputed, else 0: no memset in front of the launch); MODE 1 with a
// non-null flags pointer returns immediately unless flags[wg] != 0.
// (attn_body: the kernel for logical workgroup `bid`; attn_kernel = one workgroup per block; attn_fallback_kernel = the gated MODE 1 pass of
// the MODE 3 forms as a SMALL grid that scans the flags -- further down)
template <int DT, int NW, int RING, int MODE>
__device__ __forceinline__ void attn_body(
    const void* __restrict__ qk_, const void* __restrict__ vt_, void* __restrict__ out_, int T, int Tp, int C, int H, int nqt,
    float limit, int* __restrict__ flags, int ldo, int o8, const int bid) {
    // ldo: row pitch of `out` in elements (C, or wider when a row also carries its bf8 copy); o8 > 0: byte offset of that copy inside a row
    typedef typename Op<DT>::T Tt;
    typedef typename Op<DT>::V8 V8;
    typedef typename Op<DT>::V4 V4;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [RING][K tile | Vt tile]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;

    const int qt = bid % nqt, bh = bid / nqt;
    const int b = bh / H, h = bh - b * H;
    constexpr int QB = 32 * NW;
    // a workgroup whose query rows are ALL padding (rows per image padded to a multiple of 256: 1288^2 has 239 of them) has nothing to
    // compute or store -- its output rows are never written by anyone and stay as allocated (zero); MODE 3 still owes its flags
    if (qt * QB >= T) {
        if constexpr (MODE == 3) {
            if (lane == 0) flags[4 * bid + w] = 0;
        }
        return;
    }

    const Tt* qk = (const Tt*)qk_;
    const Tt* vt = (const Tt*)vt_;
    const int ldq = 2 * C;
    const size_t row0 = (size_t)b * Tp;

    // ---- Q fragments (B operand: lane (q = l31, hi) holds Q[q][16 ks + 8 hi + 0..7]) ----
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
    // m_ref: this query's reference level (exp2 domain; MODE 1: the exact running maximum); r_run (MODE 2): running row maximum
    // relative to m_ref; l_run: this lane's partial row sum of exp2(s - m_ref)
    float m_ref = MODE == 1 ? -1e30f : 0.f, r_run = -INFINITY, l_run = 0.f;
    bool bad = false;                                         // MODE 3: a p may have left the 16-bit range

    // MODE 3, T = 64 n + 1 (every DINOv2 grid: n patches + the class token): the lone key of the last tile is folded in as a
    // rank-1 update after the loop instead of a 65th tile of which 63 keys are masked (1.5 % of the kernel at T = 4097)
    const bool tail1 = MODE == 3 && (T & (KB - 1)) == 1 && T > KB;
    const int ntile = tail1 ? T / KB : (T + KB - 1) / KB;
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
        // ---- S^T = K . Q^T - m_ref ----
        f32x16 s[2];   // the two key halves alternate in issue order: back-to-back MFMAs never share an accumulator
        {
            const float init = MODE == 1 ? 0.f : -m_ref;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[sub][r] = init;
        }
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
        // ---- reference level ----
        if (MODE != 3 || j == 0) {
            // row maximum of the tile (one query row per lane pair {l, l^32}); every tile holds at least one unmasked key
            float mt = s[0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[1][r]);
            mt = max_lane32(mt);
            if constexpr (MODE == 1) {
                const float m_new = fmaxf(m_ref, mt);
                if (!__all(m_new == m_ref)) {
                    const float alpha = __builtin_amdgcn_exp2f(m_ref - m_new);
                    l_run *= alpha;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
                    m_ref = m_new;
                }
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[sub][r] -= m_ref;
            } else if constexpr (MODE == 2) {
                // exact: everything exponentiated against the old level (O, l) and the scores of THIS tile (not yet exponentiated)
                // are shifted by the same delta
                const float r_new = fmaxf(r_run, mt);
                if (!__all(r_new <= BAND && r_new >= -BAND)) {
                    const float delta = r_new;
                    const float alpha = r_run == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(-delta);   // first tile: O = l = 0
                    l_run *= alpha;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[sub][r] -= delta;
                    m_ref += delta;
                    r_run = 0.f;
                } else {
                    r_run = r_new;
                }
            } else {        // MODE 3, tile 0 (m_ref = 0, O = l = 0): the level becomes the exact row maximum
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[sub][r] -= mt;
                m_ref = mt;
            }
        }
        // ---- p = exp2(s - m_ref) ----
        // The row sum is taken from the ROUNDED values the PV product multiplies (numerator and denominator of the softmax then see the
        // same p), two per instruction: v_dot2 of each packed pair against (1, 1), in two independent chains -- 16 VALU instructions per
        // tile instead of 34 adds; the kernel is VALU-bound.  Measured against fp32 adds on one box: 1036-1039 vs
        // 1034-1035 TF/s alone, 1007-1013 vs 1001 TF/s inside the forward (f16), 1105 vs 1091 alone (bf16); profiles/r03_attention_rowsum_dot2.txt.
        float psum0 = 0.f, psum1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[0][r] = __builtin_amdgcn_exp2f(s[0][r]);
            s[1][r] = __builtin_amdgcn_exp2f(s[1][r]);
        }

        // ---- O^T += V^T . P^T ----
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            V8 pf;
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[e] = (Tt)s[st >> 1][8 * (st & 1) + e];
            {
                const u32x4 pw = __builtin_bit_cast(u32x4, pf);
                psum0 = Op<DT>::pair_sum(pw[0], psum0);
                psum1 = Op<DT>::pair_sum(pw[1], psum1);
                psum0 = Op<DT>::pair_sum(pw[2], psum0);
                psum1 = Op<DT>::pair_sum(pw[3], psum1);
            }
#pragma unroll
            for (int ds = 0; ds < 2; ++ds) {
                const V8 vf = *(const V8*)(sv + (32 * ds + l31) * 128 + (((2 * st + hi) ^ fsw) * 16));
                o[ds] = Op<DT>::mfma32(vf, pf, o[ds]);
            }
        }
        {
            const float psum = psum0 + psum1;
            l_run += psum;
            if constexpr (MODE == 3) bad |= !(psum <= limit);     // all p > 0: a lane sum <= limit proves every p finite in 16 bits
        }
        __builtin_amdgcn_s_setprio(0);
    }

    if constexpr (MODE == 3) {
        if (tail1 && active) {
            // key T - 1 against this wave's 32 queries: its K row is read like a Q fragment (lane (q, hi) holds the same 32 of the 64
            // dimensions of both), the dot product closes over the lane pair; p in fp32 (not rounded to 16 bits: this key is not an
            // MFMA operand); its V row goes through a wave-private 256-byte LDS strip so that a lane can pick the 32 output
            // dimensions its accumulators hold
            // (every address below is derived HERE from values made opaque to the optimiser: hoisted above the loop, the
            // loop-invariant parts of these addresses were five more live registers -> 24 B of scratch per lane, 100 MB per call)
            int kl = T - 1, lane2 = lane;
            asm volatile("" : "+s"(kl), "+v"(lane2));
            const int hi2 = lane2 >> 5;
            const Tt* kp = qk + (row0 + kl) * ldq + C + h * 64 + 8 * hi2;
            float dot = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const V8 kf = *(const V8*)(kp + 16 * ks);
#pragma unroll
                for (int e = 0; e < 8; ++e) dot = __builtin_fmaf((float)qf[ks][e], (float)kf[e], dot);
            }
            dot += __shfl_xor(dot, 32);
            const float p = __builtin_amdgcn_exp2f(dot - m_ref);
            bad |= !(p <= limit);
            if (hi == 0) l_run += p;                            // (the row sum is the sum over the lane pair)
            const int klp = (kl & ~12) | ((kl & 4) << 1) | ((kl & 8) >> 1);          // V^T columns are key-permuted (bits 2 <-> 3)
            float* vl = (float*)(smem + RING * 2 * KV_TILE_BYTES) + w * 64;
            vl[lane2] = (float)vt[((size_t)(b * H + h) * 64 + lane2) * Tp + klp];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the strip is written and read by this wave only
#pragma unroll
            for (int ds = 0; ds < 2; ++ds)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const f32x4 v4 = *(const f32x4*)(vl + 32 * ds + 8 * rg + 4 * hi2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[ds][4 * rg + e] = __builtin_fmaf(v4[e], p, o[ds][4 * rg + e]);
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (MODE == 3) {
        static_assert(MODE != 3 || NW == 4, "four flags per workgroup");
        const bool anybad = __any(bad);
        if (lane == 0) flags[4 * bid + w] = anybad ? 1 : 0;   // every wave writes its flag (no memset in front of the launch); the
                                                              // MODE 1 launch behind this one recomputes a workgroup with any flag set
    }
    // ---- normalise and store: lane (q, hi) holds O[q][32 ds + 8 rg + 4 hi + 0..3] ----
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = active ? 1.0f / l_tot : 0.f;
    if (!in_buf) return;
    int lane3 = lane;
    asm volatile("" : "+v"(lane3));                               // (the store address is formed here, not carried through the loop)
    Tt* op = (Tt*)out_ + (row0 + (qt * QB + 32 * w + (lane3 & 31))) * (size_t)ldo + h * 64 + 4 * (lane3 >> 5);
    char* op8 = (char*)((Tt*)out_ + (row0 + (qt * QB + 32 * w + (lane3 & 31))) * (size_t)ldo) + o8 + h * 64 + 4 * (lane3 >> 5);
#pragma unroll
    for (int ds = 0; ds < 2; ++ds)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            V4 v;
            float f[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { f[e] = o[ds][4 * rg + e] * inv; v[e] = (Tt)f[e]; }
            *(V4*)(op + 32 * ds + 8 * rg) = v;
            if (o8 > 0) *(uint32_t*)(op8 + 32 * ds + 8 * rg) = pack_bf8x4(f[0], f[1], f[2], f[3]);      // the output projection's fp8 low-half range
        }
}

template <int DT, int NW, int RING, int MODE>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 4 : 2) void attn_kernel(
    const void* __restrict__ qk_, const void* __restrict__ vt_, void* __restrict__ out_, int T, int Tp, int C, int H, int nqt,
    float limit, int* __restrict__ flags, int ldo, int o8) {
    attn_body<DT, NW, RING, MODE>(qk_, vt_, out_, T, Tp, C, H, nqt, limit, flags, ldo, o8, xcd_remap(blockIdx.x, gridDim.x));
}

// The gated textbook pass behind a MODE 3 form (round 6).  Rounds 2-5 launched the textbook kernel over the WHOLE grid and let every
// workgroup read its four flags and return: 16 896 workgroups at the headline, 9-11 us per launch for nothing (nothing is ever flagged on
// real or random data), 24 times per forward.  Here a workgroup looks at the flags of FALLBACK_SLICE = 16 workgroups (one 16-byte load in each
// of 16 lanes) -- the grid is ceil(nwg / 16) workgroups: 1 056 at the headline = ONE round of the chip's 1 024 workgroup slots instead of
// sixteen and a half, 34 for a batch of one -- and recomputes the flagged ones of its slice one after the other.  Worst case (steep weights
// with the precision forced to plain f16: every workgroup flagged): 16 recomputations in a row on every slot = what the full grid took.
// (First form of the round: slices of 256, 66 workgroups -- 1 us less in the common case, 16 x the time when everything is flagged:
// 170 ms against 19 for the forced-f16 hostile-weights forward, profiles/r06_final_validation_summary_lib_909c1f76.txt.)
constexpr int FALLBACK_SLICE = 16;
template <int DT>
__global__ __launch_bounds__(256, 4) void attn_fallback_kernel(const void* __restrict__ qk_, const void* __restrict__ vt_, void* __restrict__ out_,
                                                               int T, int Tp, int C, int H, int nqt, float limit, int* __restrict__ flags,
                                                               int ldo, int o8, int nwg) {
    __shared__ unsigned int smask;
    const int tid = threadIdx.x, wg = blockIdx.x * FALLBACK_SLICE + tid;
    if (tid < 64) {
        int any = 0;
        if (tid < FALLBACK_SLICE && wg < nwg) {
            const int4 f = *(const int4*)(flags + 4 * wg);    // (one flag per wave of the MODE 3 workgroup, written unconditionally)
            any = (f.x | f.y | f.z | f.w) != 0;
        }
        const unsigned long long m = __ballot(any);
        if (tid == 0) smask = (unsigned int)m;
    }
    __syncthreads();
    unsigned int mm = smask;
    while (mm) {
        const int bit = __builtin_ctz(mm);
        mm &= mm - 1;
        attn_body<DT, 4, 2, 1>(qk_, vt_, out_, T, Tp, C, H, nqt, limit, nullptr, ldo, o8, blockIdx.x * FALLBACK_SLICE + bit);
        __syncthreads();                                      // the K / V ring of the next flagged workgroup re-uses this one's LDS
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// 64 queries per wave (round 3).  The 32-query form above runs four waves per SIMD, each with ONE score block: a wave's softmax can
// only start when its own score MFMAs have finished, so inside a wave the matrix pipe and the VALU strictly alternate, and the pipe is
// busy only when ANOTHER wave of the SIMD happens to be in a matrix section (measured 53 % busy with 51 % of the wave cycles "ready but
// stalled").  Here a wave owns TWO 32-query blocks A and B (256 registers, two waves per SIMD, 256 queries per workgroup) and its own
// instruction stream overlaps the two pipes:
//     QK_A | QK_B + softmax_A | PV_A + softmax_B | PV_B            (matrix | matrix + VALU | matrix + VALU | matrix)
// Per key tile a workgroup still stages 16 KiB once -- for twice the queries, so the LDS-DMA / L2 traffic per flop halves.
// Same arithmetic as MODE 3 above (reference level = exact row maximum of key tile 0, then fixed; flagged workgroups are recomputed by
// the textbook MODE 1 kernel; the lone last key of T = 64 n + 1 as a rank-1 update).  flags are written in the 128-query MODE 1
// workgroup numbering (four per workgroup, one per 32-query block).
template <int DT>
struct QBlock {
    typename Op<DT>::V8 qf[4];
    f32x16 o[2];
    float mneg;         // -m_ref; splatted into the first C operand of the score MFMAs at the top of every QK section (16 v_mov per block
                        // and tile; a resident 16-register tuple per block does not fit beside two score blocks in 256 registers)
    float l_run;
    bool bad;
};

template <int DT, int RING>
__global__ __launch_bounds__(256, 2) void attn64_kernel(const void* __restrict__ qk_, const void* __restrict__ vt_, void* __restrict__ out_,
                                                        int T, int Tp, int C, int H, int nqt, float limit, int* __restrict__ flags) {
    typedef typename Op<DT>::T Tt;
    typedef typename Op<DT>::V8 V8;
    typedef typename Op<DT>::V4 V4;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [RING][K tile | Vt tile] + 4 x 256 B strips

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
    const int qr0 = qt * 256 + 64 * w;          // first query row of the wave
    const bool active = qr0 < T;                // (wave-uniform) a wave without a real query row only stages and meets the barriers
    QBlock<DT> A, Bq;
    auto init_block = [&](QBlock<DT>& x, int first_row) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) x.qf[ks] = V8{};
        if (first_row < Tp) {                   // rows past the padded rows of the image are not read (and not stored)
            const Tt* qp = qk + (row0 + first_row + l31) * ldq + h * 64 + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) x.qf[ks] = *(const V8*)(qp + 16 * ks);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) x.o[i][r] = 0.f;
        x.mneg = 0.f;
        x.l_run = 0.f;
        x.bad = false;
    };
    init_block(A, qr0);
    init_block(Bq, qr0 + 32);

    const int srow = tid >> 3;
    const int schunk = (tid & 7) ^ ((tid >> 4) & 7);
    const Tt* k_src = qk + (row0 + srow) * ldq + C + h * 64 + schunk * 8;
    const Tt* v_src = vt + ((size_t)(b * H + h) * 64 + srow) * Tp + schunk * 8;
    auto stage = [&](int j, int buf) {
        char* sk = smem + buf * (2 * KV_TILE_BYTES) + w * 1024;
        char* sv = sk + KV_TILE_BYTES;
        const Tt* kp = k_src + (size_t)j * KB * ldq;
        const Tt* vp = v_src + j * KB;
        glds16(kp, sk);
        glds16(kp + (size_t)32 * ldq, sk + 4096);
        glds16(vp, sv);
        glds16(vp + (size_t)32 * Tp, sv + 4096);
    };
    const int fsw = (lane >> 1) & 7;
    const bool tail1 = (T & (KB - 1)) == 1 && T > KB;
    const int ntile = tail1 ? T / KB : (T + KB - 1) / KB;
    const int nfull = tail1 ? ntile : T / KB;                      // tiles without masked keys

    // ---- building blocks of one key tile ----
    auto splat = [&](float m) {                 // (opaque per tile: hoisted out of the tile loop it would be a resident tuple again)
        asm volatile("" : "+v"(m));
        f32x16 t;
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = m;
        return t;
    };
    auto mask_block = [&](f32x16 (&s)[2], int j) {                 // keys >= T of the last, partial tile
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (j * KB + 32 * sub + crow(r, hi) >= T) s[sub][r] = -INFINITY;
    };
    auto level_block = [&](QBlock<DT>& x, f32x16 (&s)[2]) {        // tile 0: the level becomes the exact row maximum
        float mt = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[1][r]);
        mt = max_lane32(mt);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[sub][r] -= mt;
        x.mneg = -mt;
    };
#pragma unroll
    for (int t = 0; t < RING - 1; ++t) stage(t < ntile ? t : ntile - 1, t);
    int buf = 0, nbuf = RING - 1;
    // ring bookkeeping of one key tile: landing wait + barrier, slot pointers, and the next DMA (issued by the caller where it suits)
    auto tile_begin = [&](const char*& sk, const char*& sv, int& nbuf_now) {
        if constexpr (RING == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if constexpr (RING == 3) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        static_assert(RING >= 2 && RING <= 4, "counted waits");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        sk = smem + buf * (2 * KV_TILE_BYTES);
        sv = sk + KV_TILE_BYTES;
        nbuf_now = nbuf;
        buf = buf == RING - 1 ? 0 : buf + 1;
        nbuf = nbuf == RING - 1 ? 0 : nbuf + 1;
    };
    // one key tile, hand-pipelined (FIRST: tile 0 sets the reference level; MASKED: a last tile with keys >= T): four sections of four steps; a step = the NEXT step's two operand fragments requested from
    // LDS (double-buffered: 16 registers), two MFMAs on the fragments requested one step earlier, and -- in the two middle sections --
    // a quarter of the other block's softmax (8 exp, 8 adds, 4 converts) in the shadow of those MFMAs.  A scheduling barrier closes
    // every step: the register footprint is what is written here (hipcc's own schedule of the same work requested all eight K
    // fragments up front and spilled the Q fragments: 16 registers of scratch and a vmcnt(0) in front of every reload, which also
    // drained the LDS-DMA queue once per tile).
    auto tile = [&](int j, auto first_c, auto masked_c) {
        constexpr bool FIRST = decltype(first_c)::value, MASKED = decltype(masked_c)::value;
        const char *sk, *sv;
        int nbuf_now;
        tile_begin(sk, sv, nbuf_now);
        const int jn = j + RING - 1 < ntile ? j + RING - 1 : ntile - 1;
        if (!active) {
            stage(jn, nbuf_now);
            return;
        }
        V8 fr[2][2];                                  // [buffer][key half | dim half]
        auto rdK = [&](int bi, int ks) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) fr[bi][sub] = *(const V8*)(sk + (32 * sub + l31) * 128 + (((2 * ks + hi) ^ fsw) * 16));
        };
        auto rdV = [&](int bi, int st) {
#pragma unroll
            for (int ds = 0; ds < 2; ++ds) fr[bi][ds] = *(const V8*)(sv + (32 * ds + l31) * 128 + (((2 * st + hi) ^ fsw) * 16));
        };
        f32x16 sA[2], sB[2];
        V8 pA[4], pB[4];
        float ps0 = 0.f, ps1 = 0.f;
        // a quarter of a block's softmax: scores s[c >> 1][8 (c & 1) .. + 7] -> p -> the c-th packed operand of the PV MFMAs
        auto sm_quarter = [&](f32x16 (&s)[2], V8 (&pf)[4], int c) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const float p0 = __builtin_amdgcn_exp2f(s[c >> 1][8 * (c & 1) + e]);
                const float p1 = __builtin_amdgcn_exp2f(s[c >> 1][8 * (c & 1) + e + 1]);
                ps0 += p0;
                ps1 += p1;
                pf[c][e] = (Tt)p0;
                pf[c][e + 1] = (Tt)p1;
            }
        };
        auto sm_close = [&](QBlock<DT>& x) {
            const float psum = ps0 + ps1;
            x.l_run += psum;
            x.bad |= !(psum <= limit);
            ps0 = 0.f;
            ps1 = 0.f;
        };
        rdK(0, 0);
        const f32x16 initA = splat(A.mneg);
        // (inside a step the two MFMAs come FIRST -- their fragments were requested a whole step ago -- then the next step's requests and
        // the softmax quarter, which issue in the shadow of those MFMAs)
#define MHMR_STEP_SPLIT() __builtin_amdgcn_sched_barrier(0)
        // ---- section 1: QK_A (no VALU work to hide behind: the next step's fragments are requested BEFORE this step's MFMAs) ----
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            rdK((ks + 1) & 1, ks < 3 ? ks + 1 : 0);                   // (the last step requests QK_B's first fragments)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) sA[sub] = Op<DT>::mfma32(fr[ks & 1][sub], A.qf[ks], ks == 0 ? initA : sA[sub]);
            if (ks == 0) stage(jn, nbuf_now);                        // the next tile's DMA behind the first MFMAs
            MHMR_STEP_SPLIT();
        }
        if constexpr (MASKED) mask_block(sA, j);
        if constexpr (FIRST) level_block(A, sA);
        if constexpr (MASKED || FIRST) __builtin_amdgcn_sched_barrier(0);
        const f32x16 initB = splat(Bq.mneg);
        // ---- section 2: QK_B beside softmax_A ----
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) sB[sub] = Op<DT>::mfma32(fr[ks & 1][sub], Bq.qf[ks], ks == 0 ? initB : sB[sub]);
            MHMR_STEP_SPLIT();
            if (ks < 3) rdK((ks + 1) & 1, ks + 1);
            else rdV(0, 0);
            sm_quarter(sA, pA, ks);
            MHMR_STEP_SPLIT();
        }
        sm_close(A);
        if constexpr (MASKED) mask_block(sB, j);
        if constexpr (FIRST) level_block(Bq, sB);
        if constexpr (MASKED || FIRST) __builtin_amdgcn_sched_barrier(0);
        // ---- section 3: PV_A beside softmax_B ----
#pragma unroll
        for (int st = 0; st < 4; ++st) {
#pragma unroll
            for (int ds = 0; ds < 2; ++ds) A.o[ds] = Op<DT>::mfma32(fr[st & 1][ds], pA[st], A.o[ds]);
            MHMR_STEP_SPLIT();
            rdV((st + 1) & 1, st < 3 ? st + 1 : 0);
            sm_quarter(sB, pB, st);
            MHMR_STEP_SPLIT();
        }
        sm_close(Bq);
        // ---- section 4: PV_B (requests first, as in section 1) ----
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            if (st < 3) rdV((st + 1) & 1, st + 1);
#pragma unroll
            for (int ds = 0; ds < 2; ++ds) Bq.o[ds] = Op<DT>::mfma32(fr[st & 1][ds], pB[st], Bq.o[ds]);
            MHMR_STEP_SPLIT();
        }
#undef MHMR_STEP_SPLIT
    };
    {
        using TT = std::true_type;
        using FF = std::false_type;
        if (nfull > 0) tile(0, TT{}, FF{});
        else tile(0, TT{}, TT{});
        for (int j = 1; j < nfull; ++j) tile(j, FF{}, FF{});
        if (ntile > nfull && nfull > 0) tile(nfull, FF{}, TT{});
    }

    // ---- the lone last key of T = 64 n + 1 (the class token, stored last): rank-1 update, fp32 p ----
    float* vl = (float*)(smem + RING * 2 * KV_TILE_BYTES) + w * 64;
    auto tail_block = [&](QBlock<DT>& x, int kl) {
        int lane2 = lane;
        asm volatile("" : "+s"(kl), "+v"(lane2));
        const int hi2 = lane2 >> 5;
        const Tt* kp = qk + (row0 + kl) * ldq + C + h * 64 + 8 * hi2;
        float dot = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const V8 kf = *(const V8*)(kp + 16 * ks);
#pragma unroll
            for (int e = 0; e < 8; ++e) dot = __builtin_fmaf((float)x.qf[ks][e], (float)kf[e], dot);
        }
        dot += __shfl_xor(dot, 32);
        const float p = __builtin_amdgcn_exp2f(dot + x.mneg);
        x.bad |= !(p <= limit);
        if (hi == 0) x.l_run += p;
#pragma unroll
        for (int ds = 0; ds < 2; ++ds)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x4 v4 = *(const f32x4*)(vl + 32 * ds + 8 * rg + 4 * hi2);
#pragma unroll
                for (int e = 0; e < 4; ++e) x.o[ds][4 * rg + e] = __builtin_fmaf(v4[e], p, x.o[ds][4 * rg + e]);
            }
    };
    if (tail1 && active) {
        const int kl = T - 1;
        const int klp = (kl & ~12) | ((kl & 4) << 1) | ((kl & 8) >> 1);          // V^T columns are key-permuted (bits 2 <-> 3)
        vl[lane] = (float)vt[((size_t)(b * H + h) * 64 + lane) * Tp + klp];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // the strip is written and read by this wave only
        tail_block(A, kl);
        tail_block(Bq, kl);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- flags (MODE 1 numbering: 128-query workgroups, one flag per 32-query block), normalise, store ----
    const int nqt1 = (Tp + 127) / 128, qt1 = 2 * qt + (w >> 1);
    auto finish_block = [&](QBlock<DT>& x, int blk) {
        const int first_row = qr0 + 32 * blk;
        const bool anybad = __any(x.bad);
        if (lane == 0 && qt1 < nqt1) flags[4 * (bh * nqt1 + qt1) + 2 * (w & 1) + blk] = (anybad && first_row < T) ? 1 : 0;
        const float l_tot = x.l_run + __shfl_xor(x.l_run, 32);
        const float inv = first_row < T ? 1.0f / l_tot : 0.f;                    // blocks of padding rows store zeros
        if (first_row >= Tp) return;
        int lane3 = lane;
        asm volatile("" : "+v"(lane3));
        Tt* op = (Tt*)out_ + (row0 + (first_row + (lane3 & 31))) * C + h * 64 + 4 * (lane3 >> 5);
#pragma unroll
        for (int ds = 0; ds < 2; ++ds)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                V4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (Tt)(x.o[ds][4 * rg + e] * inv);
                *(V4*)(op + 32 * ds + 8 * rg) = v;
            }
    };
    finish_block(A, 0);
    finish_block(Bq, 1);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The lone query of T = 128 n + 1 (every DINOv2 grid whose side is a multiple of 16 patches, + the class token, which is the LAST token
// row) as a role of its own (round 6, variant 10).  As a 128-query workgroup it was a "ghost": one real query in 128 rows, the whole key
// loop, a workgroup slot for about as long as a full workgroup -- 512 of 16 896 workgroups at the headline, measured 2.2 % of the launch
// (tools/kbench.py --tokens 4096 against 4097, profiles/r06_session_n_lone_key_prefetch.txt).  Here a workgroup of four waves takes the
// class query of one (image, head) on the vector ALU -- exact online softmax in fp32, no reference-level flags -- 32 keys per step, wave w
// the steps w, w + 4, ...:
//   scores  lane (key = lane & 31, half = lane >> 5) holds 32 of the key's 64 d (64 B of its K row, four 16-byte loads), 16 v_dot2 against
//           the query's packed pairs (registers), the halves meet by one lane exchange;
//   output  lane = d: 64 B of V^T row d (the step's 32 key columns; columns are key-permuted, bits 2 <-> 3), p of column c read from lane
//           perm(c) by v_readlane -- one FMA per (d, key), the running o[d] is ONE register;
//   merge   the four waves' (level, sum, o) meet in LDS, wave 0 writes the row.
// Every global load is an asm statement with counted waits: two register sets per wave, the next step's 8 loads in flight under this
// step's arithmetic (the first form left the loads to the compiler, which waited for ALL of them before every step -- a full memory round
// trip per step, 160 us per query: slower than the ghost, profiles/r06_session_o_class_query_role_v1_slower.txt).
// B H such workgroups in FRONT of the grid (raw block ids: spread over the XCDs), the 128-query workgroups behind.
// Cost: K and V^T of every head are read once more (0.54 GB per launch at the headline, in the shadow of a compute-bound kernel).
// (the dot-product BUILTINS, not inline assembly: on gfx90a+ a VALU instruction that reads a dot product's result needs three wait states
// behind it, and the hazard recogniser cannot see inside an asm statement -- the very first form of this role, with asm, read stale sums)
template <int DT>
__device__ __forceinline__ float cls_dot2(uint32_t a, uint32_t b, float acc) {
    typedef typename Op<DT>::V2 V2;
    if constexpr (DT == MHMR_DT_F16) return __builtin_amdgcn_fdot2(__builtin_bit_cast(V2, a), __builtin_bit_cast(V2, b), acc, false);
    else return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(V2, a), __builtin_bit_cast(V2, b), acc, false);
}
template <int DT>
__device__ __forceinline__ float cls_half(uint32_t w, int hi) {      // the low / high 16-bit value of a packed pair as fp32
    typedef typename Op<DT>::T Tt;
    return (float)__builtin_bit_cast(Tt, (uint16_t)(hi ? w >> 16 : w));
}
// 64 contiguous bytes per lane, requested and NOT waited for (the compiler does not know these registers are in flight: every use must sit
// behind a cls_wait on the same registers)
__device__ __forceinline__ void cls_gload64(u32x4& d0, u32x4& d1, u32x4& d2, u32x4& d3, const void* p) {
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48"
                 : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void cls_wait(u32x4* a, u32x4* b) {       // at most N of this wave's loads still in flight; a[0..3], b[0..3] are ready
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N) : "memory");
}
template <int DT>
__device__ __forceinline__ void attn_cls_role(const void* __restrict__ qk_, const void* __restrict__ vt_, void* __restrict__ out_, int T, int Tp,
                                              int C, int H, int nqt, int* __restrict__ flags, int ldo, int bh, int w, int lane, char* smem) {
    typedef typename Op<DT>::T Tt;
    const int b = bh / H, h = bh - b * H;
    const Tt* qk = (const Tt*)qk_;
    const Tt* vt = (const Tt*)vt_;
    const int ldq = 2 * C;
    const size_t row0 = (size_t)b * Tp;
    const int key = lane & 31, half = lane >> 5;
    const Tt* kbase = qk + (row0 + key) * ldq + C + h * 64 + 32 * half;
    const Tt* vbase = vt + ((size_t)(b * H + h) * 64 + lane) * Tp;
    const int nst = (T + 31) / 32, last = nst - 1;
    const int nmine = (nst - w + 3) / 4;                    // this wave's steps: w, w + 4, ... (T >= 129: at least one each)
    u32x4 q4[4], ka[4], va[4], kb[4], vb[4];
    auto issue = [&](int st, u32x4* k4, u32x4* v4) {
        if (st > last) st = last;                           // (past the end: a harmless re-load keeps the load count per step fixed)
        cls_gload64(k4[0], k4[1], k4[2], k4[3], kbase + (size_t)st * 32 * ldq);
        cls_gload64(v4[0], v4[1], v4[2], v4[3], vbase + st * 32);
    };
    float m_ref = 0.f, l_lane = 0.f, o = 0.f;
    bool first = true;
    auto step = [&](int st, const u32x4* k4, const u32x4* v4) {
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sa = cls_dot2<DT>(k4[i][0], q4[i][0], sa);
            sb = cls_dot2<DT>(k4[i][1], q4[i][1], sb);
            sa = cls_dot2<DT>(k4[i][2], q4[i][2], sa);
            sb = cls_dot2<DT>(k4[i][3], q4[i][3], sb);
        }
        float sc = sa + sb;
        sc += __shfl_xor(sc, 32);
        if (st * 32 + key >= T) sc = -INFINITY;
        if (first || __any(sc > m_ref + 64.f)) {            // (wave-uniform; after a wave's first step taken only when a key beats the level by 2^64)
            float mx = sc;
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
            const float f = first ? 0.f : __builtin_amdgcn_exp2f(m_ref - mx);
            o *= f;
            l_lane *= f;
            m_ref = mx;
            first = false;
        }
        const float pv = __builtin_amdgcn_exp2f(sc - m_ref);
        l_lane += pv;
        const int pbits = __builtin_bit_cast(int, pv);
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const int src = (c & ~12) | ((c & 4) << 1) | ((c & 8) >> 1);           // column c of V^T holds key src of the step
            const float pc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(pbits, src));
            o = __builtin_fmaf(pc, cls_half<DT>(v4[c >> 3][(c >> 1) & 3], c & 1), o);
        }
    };
    cls_gload64(q4[0], q4[1], q4[2], q4[3], qk + (row0 + T - 1) * ldq + h * 64 + 32 * half);
    issue(w, ka, va);
    for (int i = 0; i < nmine; i += 2) {
        const int st = w + 4 * i;
        issue(st + 4, kb, vb);
        cls_wait<8>(ka, va);                                // (the query's four loads are older still)
        asm volatile("" : "+v"(q4[0]), "+v"(q4[1]), "+v"(q4[2]), "+v"(q4[3]));
        step(st, ka, va);
        if (i + 1 < nmine) {
            issue(st + 8, ka, va);
            cls_wait<8>(kb, vb);
            step(st + 4, kb, vb);
        }
    }
    cls_wait<0>(ka, va);                                    // nothing may stay in flight into registers the compiler believes free
    cls_wait<0>(kb, vb);
    // ---- the four waves' partial results meet in LDS (the K / V^T ring of the other role: unused here) ----
    float l = half == 0 ? l_lane : 0.f;                     // (the two halves of a key hold the same p)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) l += __shfl_xor(l, d);
    float* part = (float*)smem;                             // [4 waves][64 o | level | sum]
    part[w * 66 + lane] = o;
    if (lane == 0) { part[w * 66 + 64] = m_ref; part[w * 66 + 65] = l; }
    __syncthreads();
    if (w != 0) return;
    float M = part[64];
#pragma unroll
    for (int i = 1; i < 4; ++i) M = fmaxf(M, part[i * 66 + 64]);
    float O = 0.f, L = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float f = __builtin_amdgcn_exp2f(part[i * 66 + 64] - M);
        O = __builtin_fmaf(part[i * 66 + lane], f, O);
        L = __builtin_fmaf(part[i * 66 + 65], f, L);
    }
    ((Tt*)out_)[(row0 + T - 1) * (size_t)ldo + h * 64 + lane] = (Tt)(O / L);
    // the 128-query numbering keeps a workgroup for this query (and for tiles of padding rows behind it); the fallback pass scans their
    // flags: never flagged
    if (flags != nullptr)
        for (int i = 4 * (T / 128) + lane; i < 4 * nqt; i += 64) flags[4 * bh * nqt + i] = 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The MODE 3 arithmetic on v_mfma_f32_16x16x32 (round 4, variant 6).  tools/ubench/gemm4w.hip -DMFMA32 showed that the 32x32x16 shape
// costs 5-8 % more power per flop than 16x16x32 (profiles/r04_mfma_shape_prototype.txt: slower on random data, equal on zeros), and this
// kernel runs at the lowest clock of the forward (1.6 GHz).  Same workgroup (4 waves x 32 queries, 64-key tiles, 2-slot K / V^T ring,
// MODE 3 level + flags, the same V^T layout), other fragment geometry:
//   a wave's 32 queries are two 16-query blocks qb; lane (j = lane & 15, g = lane >> 4) holds, per block, query 16 qb + j and FOUR
//   lane-groups' worth of keys: of every 32-key half m of the tile the score blocks X (keys 0-3 | 4-7 | 16-19 | 20-23 of the half, one
//   quad per g) and Y (the same + 8), so that [X quad | Y quad] of lane (j, g) are the eight keys whose V^T values sit in ONE 16-byte
//   chunk of the key-permuted V^T row: P feeds the PV product in place, as in the 32x32 form.  S^T = K . Q^T: 16 MFMAs (2 halves x X / Y
//   x 2 query blocks x 2 k steps), O^T += V^T . P^T: 16 MFMAs (2 halves x 4 d blocks x 2 query blocks); 8 + 8 ds_read_b128 per tile.
//   The K tile's rows are read in the order 0-7, 16-23 (X) / 8-15, 24-31 (Y): its LDS swizzle is ((row >> 1) & 3) | ((row >> 4) & 1) << 2
//   (applied on the copy's source address), which keeps those sixteen rows on sixteen different 16-byte slots; V^T keeps (row >> 1) & 7.
//   A query's keys are spread over the four lanes j, j + 16, j + 32, j + 48: the row maximum of tile 0, the row sums at the end and the
//   lone last key's dot product close over them with two lane exchanges each.
// Round-6 experiment forms (variants 7 / 8 / 9 of mhmr_attention16_ex; what mhmr_vit_forward runs is <DT, false, 2>):
//   EARLY     the next tile's copies are issued right behind the barrier, IN FRONT of the score MFMAs (the slot they fill was last read in
//             the previous tile, which every wave has left): ~300 cycles more lead for the landing wait at the top of the next tile;
//   RING = 3  a third K / V^T slot (48 KiB per workgroup: three workgroups per CU instead of four), copies two tiles ahead behind a COUNTED
//             wait (the newest tile's four copies per thread may still be in flight).
// CLSQ (variant 10): launched with ncls > 0 leading workgroups in the class-query role (attn_cls_role above), the 128-query workgroups of the
//   nfull = T / 128 full query tiles behind them; flags keep the numbering bid = (image, head) * nqt + tile.
// The body of attn16_kernel behind its early exits, for NQB = 2 (a wave's two 16-query blocks) or NQB = 1 (round 6, session R: the LAST workgroup
// of an image whose real queries all sit in wave 0's first block -- T = 128 n + 1 ... 128 n + 16; at T = 128 n + 1, every DINOv2 grid with a side
// of a multiple of 16 patches, that workgroup holds ONE real query and was measured at 2.2 % of the launch: half the per-tile work of its only
// active wave was rows of padding).  Rows 16 ... 31 of that workgroup's first wave are then never stored (they keep what was allocated: zero).
// The split became possible (no spill beside the 128-register key loop) with the copies' addresses as a scalar base + a 32-bit lane offset.
template <int DT, bool EARLY, int RING, int NQB>
__device__ __forceinline__ void attn16_tail(const typename Op<DT>::T* __restrict__ qk, const typename Op<DT>::T* __restrict__ vt, void* __restrict__ out_,
                                            int T, int Tp, int C, int H, float limit, int* __restrict__ flags, int ldo, int o8, int bid, int qt, int b, int h,
                                            int w, int lane, int j15, int g, int tid, char* smem, size_t row0, int ldq, bool active, bool in_buf) {
    typedef typename Op<DT>::T Tt;
    typedef typename Op<DT>::V8 V8;
    typedef typename Op<DT>::V4 V4;
    constexpr int QB = 128;
    // Q fragments (second operand): lane (j, g) holds Q[16 qb + j][32 ks + 8 g + 0..7]
    V8 qf[2][2] = {};
    if (in_buf) {
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            const Tt* qp = qk + (row0 + qt * QB + 32 * w + 16 * qb + j15) * ldq + h * 64 + 8 * g;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) qf[qb][ks] = *(const V8*)(qp + 32 * ks);
        }
    }
    // staging: one copy per thread moves 32 tile rows; K with this kernel's swizzle, V^T with the common one
    const int srow = tid >> 3;
    const int kchunk = (tid & 7) ^ (((srow >> 1) & 3) | (((srow >> 4) & 1) << 2));
    const int vchunk = (tid & 7) ^ ((srow >> 1) & 7);
    // (addresses as a wave-uniform 64-bit base, advanced on the scalar unit, + a 32-bit lane offset that never changes: the copies take the
    // SGPR-base form of global_load_lds -- no 64-bit vector add per copy, two registers less than two 64-bit lane pointers)
    const uint32_t k_lane = (uint32_t)(srow * ldq + kchunk * 8) * 2u;
    const uint32_t v_lane = (uint32_t)(srow * Tp + vchunk * 8) * 2u;
    const char* k_base = (const char*)(qk + row0 * ldq + C + h * 64);
    const char* v_base = (const char*)(vt + (size_t)(b * H + h) * 64 * Tp);
    auto stage = [&](int jt, int buf) {
        char* sk = smem + buf * (2 * KV_TILE_BYTES) + w * 1024;
        char* sv = sk + KV_TILE_BYTES;
        const char* kp = k_base + (size_t)jt * KB * ldq * 2;
        const char* vp = v_base + (size_t)jt * KB * 2;
        // (the 32-bit lane offsets pass through an empty asm HERE: instruction selection works per basic block, and a zero-extension
        // hoisted out of the key loop would leave it a 64-bit vector add per copy again)
        uint32_t kl = k_lane, vl_ = v_lane;
        asm volatile("" : "+v"(kl), "+v"(vl_));
        const char* kp1 = kp + (size_t)32 * ldq * 2;
        const char* vp1 = vp + (size_t)32 * Tp * 2;
        asm volatile("" : "+s"(kp1), "+s"(vp1));          // (scalar registers: base + row offset is added on the scalar unit, not per lane)
        glds16(kp + kl, sk);
        glds16(kp1 + kl, sk + 4096);
        glds16(vp + vl_, sv);
        glds16(vp1 + vl_, sv + 4096);
    };
    // fragment addresses inside a tile.  K: row = 32 m + 8 xy + i + (i & 8), chunk 4 ks + g; V^T: row = 16 db + i, chunk 4 m + g.  Both
    // swizzle terms depend on the lane only ((32 m + 8 xy) and 16 db leave the bits they use alone), and chunk 4 + c is chunk c with bit 2
    // flipped: two lane offsets per tile, everything else is an immediate offset of the ds_read
    const int krow = j15 + (j15 & 8);
    const int kb_g = (g & 1) * 4 + (g >> 1) * 16;                  // first key (inside a half, X block) of this lane's quad
    const int kf_sw = g ^ (((krow >> 1) & 3) | (((krow >> 4) & 1) << 2));
    const int vf_sw = g ^ ((j15 >> 1) & 7);
    int kofs[2] = {krow * 128 + kf_sw * 16, krow * 128 + (kf_sw ^ 4) * 16};           // [ks]
    int vofs[2] = {j15 * 128 + vf_sw * 16, j15 * 128 + (vf_sw ^ 4) * 16};             // [m]

    f32x4 o[4][2];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) o[db][qb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_ref[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
    bool bad = false;
    const bool tail1 = (T & (KB - 1)) == 1 && T > KB;
    const int ntile = tail1 ? T / KB : (T + KB - 1) / KB;
    // round 6: the lone last key's K row (128 B) and V^T column (64 values, Tp apart: the aligned dword around each) travel to LDS by DMA
    // NOW, in front of tile 0's copies (wave 0 for the workgroup; landed and visible behind the first tile's wait + barrier), instead of
    // being fetched from global memory behind the key loop, where every wave then sat out a full memory round trip with nothing else to do
    char* lone = smem + RING * 2 * KV_TILE_BYTES + 4 * 256;              // [K row: 32 dwords | V^T column: 64 dwords]
    if (tail1 && w == 0) {
        const int kl = T - 1;
        const int klp = (kl & ~12) | ((kl & 4) << 1) | ((kl & 8) >> 1);          // V^T columns are key-permuted (bits 2 <-> 3)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vt + ((size_t)(b * H + h) * 64 + lane) * Tp + (klp & ~1)),
                                         (__attribute__((address_space(3))) void*)(lone + 128), 4, 0, 0);
        if (lane < 32)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qk + (row0 + kl) * ldq + C + h * 64 + 2 * lane),
                                             (__attribute__((address_space(3))) void*)lone, 4, 0, 0);
    }
    stage(0, 0);
    if constexpr (RING == 3) stage(ntile > 1 ? 1 : 0, 1);
    int buf = 0, nbuf = RING - 1;
    for (int jt = 0; jt < ntile; ++jt) {
        // tile jt has landed: RING == 2: everything this thread issued; RING == 3: all but the newest tile's four copies
        if constexpr (RING == 3) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const char* sk = smem + buf * (2 * KV_TILE_BYTES);
        const char* sv = sk + KV_TILE_BYTES;
        const int nbuf_now = nbuf;
        if constexpr (RING == 3) { buf = buf == 2 ? 0 : buf + 1; nbuf = nbuf == 2 ? 0 : nbuf + 1; }
        else { buf ^= 1; nbuf ^= 1; }
        const int jnext = jt + RING - 1 < ntile ? jt + RING - 1 : ntile - 1;      // (past the end: a harmless re-copy keeps the copy count per tile fixed)
        if (!active) {
            stage(jnext, nbuf_now);
            continue;
        }
        if constexpr (EARLY) {
            stage(jnext, nbuf_now);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- S^T = K . Q^T - m_ref : s[m][xy][qb], lane (j, g) <- keys 32 m + 8 xy + kb_g + 0..3 of query 16 qb + j ----
        f32x4 sc[2][2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int xy = 0; xy < 2; ++xy)
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) sc[m][xy][qb] = (f32x4){-m_ref[qb], -m_ref[qb], -m_ref[qb], -m_ref[qb]};
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                V8 kf[2];
#pragma unroll
                for (int xy = 0; xy < 2; ++xy) kf[xy] = *(const V8*)(sk + kofs[ks] + (32 * m + 8 * xy) * 128);
#pragma unroll
                for (int xy = 0; xy < 2; ++xy)
#pragma unroll
                    for (int qb = 0; qb < NQB; ++qb) sc[m][xy][qb] = Op<DT>::mfma16(kf[xy], qf[qb][ks], sc[m][xy][qb]);
            }
        __builtin_amdgcn_s_setprio(0);
        if constexpr (!EARLY) stage(jnext, nbuf_now);
        __builtin_amdgcn_sched_barrier(0);
        // ---- mask keys >= T (only the last tile can contain them) ----
        if (jt * KB + KB > T) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int xy = 0; xy < 2; ++xy)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (jt * KB + 32 * m + 8 * xy + kb_g + r >= T) { sc[m][xy][0][r] = -INFINITY; sc[m][xy][1][r] = -INFINITY; }
        }
        // ---- reference level: the exact row maximum of tile 0 (over the four lanes of a query), then fixed ----
        if (jt == 0) {
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
                float mt = sc[0][0][qb][0];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int xy = 0; xy < 2; ++xy)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mt = fmaxf(mt, sc[m][xy][qb][r]);
                mt = fmaxf(mt, __shfl_xor(mt, 16));
                mt = fmaxf(mt, __shfl_xor(mt, 32));
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int xy = 0; xy < 2; ++xy)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sc[m][xy][qb][r] -= mt;
                m_ref[qb] = mt;
            }
        }
        // ---- p = exp2(s - m_ref) ----
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int xy = 0; xy < 2; ++xy)
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[m][xy][qb][r] = __builtin_amdgcn_exp2f(sc[m][xy][qb][r]);
        // ---- O^T += V^T . P^T ; row sums from the rounded values (v_dot2 of each packed pair against (1, 1)) ----
        float psum[2] = {0.f, 0.f};
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            V8 pf[2];
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { pf[qb][e] = (Tt)sc[m][0][qb][e]; pf[qb][4 + e] = (Tt)sc[m][1][qb][e]; }
                const u32x4 pw = __builtin_bit_cast(u32x4, pf[qb]);
                psum[qb] = Op<DT>::pair_sum(pw[0], psum[qb]);
                psum[qb] = Op<DT>::pair_sum(pw[1], psum[qb]);
                psum[qb] = Op<DT>::pair_sum(pw[2], psum[qb]);
                psum[qb] = Op<DT>::pair_sum(pw[3], psum[qb]);
            }
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const V8 vf = *(const V8*)(sv + vofs[m] + db * 2048);
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) o[db][qb] = Op<DT>::mfma16(vf, pf[qb], o[db][qb]);
            }
        }
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            l_run[qb] += psum[qb];
            bad |= !(psum[qb] <= limit);          // all p > 0: a lane sum <= limit proves every p of the lane finite in 16 bits
        }
        __builtin_amdgcn_s_setprio(0);
    }

    if (tail1 && active) {
        // the lone key T - 1 as a rank-1 update (fp32 p; its V row through a wave-private LDS strip), closed over a query's four lanes
        int kl = T - 1, lane2 = lane;
        asm volatile("" : "+s"(kl), "+v"(lane2));
        const int g2 = lane2 >> 4;
        const Tt* kp = (const Tt*)lone + 8 * g2;                                  // (the row is in LDS since tile 0)
        float dot[2] = {0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const V8 kf = *(const V8*)(kp + 32 * ks);
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                for (int e = 0; e < 8; ++e) dot[qb] = __builtin_fmaf((float)qf[qb][ks][e], (float)kf[e], dot[qb]);
        }
        const int klp = (kl & ~12) | ((kl & 4) << 1) | ((kl & 8) >> 1);          // V^T columns are key-permuted (bits 2 <-> 3)
        float* vl = (float*)(smem + RING * 2 * KV_TILE_BYTES) + w * 64;
        {
            const uint32_t raw = ((const uint32_t*)(lone + 128))[lane2];          // the aligned dword around V^T[d = lane][klp]
            const uint16_t bits = (uint16_t)((klp & 1) ? raw >> 16 : raw);
            vl[lane2] = (float)__builtin_bit_cast(Tt, bits);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                        // the strip is written and read by this wave only
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            float d = dot[qb];
            d += __shfl_xor(d, 16);
            d += __shfl_xor(d, 32);
            const float p = __builtin_amdgcn_exp2f(d - m_ref[qb]);
            bad |= !(p <= limit);
            if (g2 == 0) l_run[qb] += p;                                          // (the row sum is the sum over the query's four lanes)
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const f32x4 v4 = *(const f32x4*)(vl + 16 * db + 4 * g2);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[db][qb][e] = __builtin_fmaf(v4[e], p, o[db][qb][e]);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        const bool anybad = __any(bad);
        if (lane == 0) flags[4 * bid + w] = anybad ? 1 : 0;
    }
    // ---- normalise and store: lane (j, g) holds O[16 qb + j][16 db + 4 g + 0..3] ----
    if (!in_buf) return;
    int lane3 = lane;
    asm volatile("" : "+v"(lane3));
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        float l_tot = l_run[qb];
        l_tot += __shfl_xor(l_tot, 16);
        l_tot += __shfl_xor(l_tot, 32);
        const float inv = active ? 1.0f / l_tot : 0.f;
        Tt* op = (Tt*)out_ + (row0 + (qt * QB + 32 * w + 16 * qb + (lane3 & 15))) * (size_t)ldo + h * 64 + 4 * (lane3 >> 4);
        char* op8 = (char*)((Tt*)out_ + (row0 + (qt * QB + 32 * w + 16 * qb + (lane3 & 15))) * (size_t)ldo) + o8 + h * 64 + 4 * (lane3 >> 4);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            V4 v;
            float f[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { f[e] = o[db][qb][e] * inv; v[e] = (Tt)f[e]; }
            *(V4*)(op + 16 * db) = v;
            if (o8 > 0) *(uint32_t*)(op8 + 16 * db) = pack_bf8x4(f[0], f[1], f[2], f[3]);      // the output projection's fp8 low-half range
        }
    }
}

template <int DT, bool EARLY = false, int RING = 2, bool CLSQ = false>
__global__ __launch_bounds__(256, RING == 2 ? 4 : 3) void attn16_kernel(const void* __restrict__ qk_, const void* __restrict__ vt_, void* __restrict__ out_,
                                                        int T, int Tp, int C, int H, int nqt, float limit, int* __restrict__ flags, int ldo,
                                                        int o8, int ncls = 0, int nbh = 0) {
    typedef typename Op<DT>::T Tt;
    typedef typename Op<DT>::V8 V8;
    typedef typename Op<DT>::V4 V4;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][K tile | Vt tile] + one 64-float strip per wave
    constexpr int QB = 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j15 = lane & 15, g = lane >> 4;
    int bid;
    if constexpr (CLSQ) {
        if ((int)blockIdx.x < ncls) {
            if ((int)blockIdx.x >= nbh) return;             // (ncls = the (image, head) count rounded up to a multiple of 8)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            attn_cls_role<DT>(qk_, vt_, out_, T, Tp, C, H, nqt, flags, ldo, (int)blockIdx.x, w, ln, smem);
            return;
        }
        const int nfull = T / QB;
        const int r = xcd_remap((int)blockIdx.x - ncls, (int)gridDim.x - ncls);
        bid = (r / nfull) * nqt + r % nfull;
    } else {
        bid = xcd_remap(blockIdx.x, gridDim.x);
    }
    const int qt = bid % nqt, bh = bid / nqt;
    const int b = bh / H, h = bh - b * H;
    if (qt * QB >= T) {
        if (lane == 0) flags[4 * bid + w] = 0;
        return;
    }
    const Tt* qk = (const Tt*)qk_;
    const Tt* vt = (const Tt*)vt_;
    const int ldq = 2 * C;
    const size_t row0 = (size_t)b * Tp;
    const bool active = qt * QB + 32 * w < T;
    const bool in_buf = qt * QB + 32 * w < Tp;
    if (T - qt * QB <= 16) attn16_tail<DT, EARLY, RING, 1>(qk, vt, out_, T, Tp, C, H, limit, flags, ldo, o8, bid, qt, b, h, w, lane, j15, g, tid, smem, row0, ldq, active, in_buf);
    else attn16_tail<DT, EARLY, RING, 2>(qk, vt, out_, T, Tp, C, H, limit, flags, ldo, o8, bid, qt, b, h, w, lane, j15, g, tid, smem, row0, ldq, active, in_buf);
}

template <bool EARLY = false, int RING = 2>
int launch_attn16(const void* qk, const void* vt, void* out, int B, int T, int Tp, int C, int H, int dtype, float limit, int* flags,
                  hipStream_t s, int ldo, int o8) {
    const int nqt = (Tp + 127) / 128;
    const int grid = nqt * H * B;
    const size_t lds = RING * 2 * KV_TILE_BYTES + 4 * 256 + 384;         // + the lone last key's K row and V^T column
    if (dtype == MHMR_DT_F16)
        hipLaunchKernelGGL((attn16_kernel<MHMR_DT_F16, EARLY, RING>), dim3(grid), dim3(256), lds, s, qk, vt, out, T, Tp, C, H, nqt, limit, flags, ldo, o8);
    else
        hipLaunchKernelGGL((attn16_kernel<MHMR_DT_BF16, EARLY, RING>), dim3(grid), dim3(256), lds, s, qk, vt, out, T, Tp, C, H, nqt, limit, flags, ldo, o8);
    MHMR_CHECK_LAUNCH();
    return 0;
}

// variant 10: the class query (T = 128 n + 1) on workgroups of its own in front of the grid; any other shape runs variant 6's launch
int launch_attn16_clsq(const void* qk, const void* vt, void* out, int B, int T, int Tp, int C, int H, int dtype, float limit, int* flags,
                       hipStream_t s, int ldo, int o8) {
    if (T % 128 != 1 || T < 129 || o8 != 0) return launch_attn16<false, 2>(qk, vt, out, B, T, Tp, C, H, dtype, limit, flags, s, ldo, o8);
    const int nqt = (Tp + 127) / 128, nbh = B * H;
    const int ncls = (nbh + 7) / 8 * 8;                                  // one workgroup per (image, head); a multiple of 8: the tile workgroups keep their XCD chunks
    const int grid = ncls + (T / 128) * nbh;
    const size_t lds = 2 * 2 * KV_TILE_BYTES + 4 * 256 + 384;
    if (dtype == MHMR_DT_F16)
        hipLaunchKernelGGL((attn16_kernel<MHMR_DT_F16, false, 2, true>), dim3(grid), dim3(256), lds, s, qk, vt, out, T, Tp, C, H, nqt, limit, flags, ldo, o8, ncls, nbh);
    else
        hipLaunchKernelGGL((attn16_kernel<MHMR_DT_BF16, false, 2, true>), dim3(grid), dim3(256), lds, s, qk, vt, out, T, Tp, C, H, nqt, limit, flags, ldo, o8, ncls, nbh);
    MHMR_CHECK_LAUNCH();
    return 0;
}

template <int RING>
int launch_attn64(const void* qk, const void* vt, void* out, int B, int T, int Tp, int C, int H, int dtype, float limit, int* flags,
                  hipStream_t s) {
    const int nqt = (Tp + 255) / 256;
    const int grid = nqt * H * B;
    const size_t lds = RING * 2 * KV_TILE_BYTES + 4 * 256;
    if (dtype == MHMR_DT_F16)
        hipLaunchKernelGGL((attn64_kernel<MHMR_DT_F16, RING>), dim3(grid), dim3(256), lds, s, qk, vt, out, T, Tp, C, H, nqt, limit, flags);
    else
        hipLaunchKernelGGL((attn64_kernel<MHMR_DT_BF16, RING>), dim3(grid), dim3(256), lds, s, qk, vt, out, T, Tp, C, H, nqt, limit, flags);
    MHMR_CHECK_LAUNCH();
    return 0;
}

// the gated textbook pass over the flags a MODE 3 form (128-query workgroup numbering) has left
int launch_attn_fallback(const void* qk, const void* vt, void* out, int B, int T, int Tp, int C, int H, int dtype, float limit, int* flags,
                         hipStream_t s, int ldo = 0, int o8 = 0) {
    if (ldo <= 0) ldo = C;
    const int nqt = (Tp + 127) / 128, nwg = nqt * H * B;
    const int grid = (nwg + FALLBACK_SLICE - 1) / FALLBACK_SLICE;
    const size_t lds = 2 * 2 * KV_TILE_BYTES + 4 * 256;
    if (dtype == MHMR_DT_F16)
        hipLaunchKernelGGL((attn_fallback_kernel<MHMR_DT_F16>), dim3(grid), dim3(256), lds, s, qk, vt, out, T, Tp, C, H, nqt, limit, flags, ldo, o8, nwg);
    else
        hipLaunchKernelGGL((attn_fallback_kernel<MHMR_DT_BF16>), dim3(grid), dim3(256), lds, s, qk, vt, out, T, Tp, C, H, nqt, limit, flags, ldo, o8, nwg);
    MHMR_CHECK_LAUNCH();
    return 0;
}

template <int NW, int RING, int MODE>
int launch_attn(const void* qk, const void* vt, void* out, int B, int T, int Tp, int C, int H, int dtype, float limit, int* flags,
                hipStream_t s, int ldo = 0, int o8 = 0) {
    if (ldo <= 0) ldo = C;
    constexpr int QB = 32 * NW;
    const int nqt = (Tp + QB - 1) / QB;
    const int grid = nqt * H * B;
    const size_t lds = RING * 2 * KV_TILE_BYTES + NW * 256;        // K / V^T ring + one 64-float strip per wave (MODE 3's last key)
    if (dtype == MHMR_DT_F16)
        hipLaunchKernelGGL((attn_kernel<MHMR_DT_F16, NW, RING, MODE>), dim3(grid), dim3(64 * NW), lds, s, qk, vt, out, T, Tp, C, H, nqt, limit, flags, ldo, o8);
    else
        hipLaunchKernelGGL((attn_kernel<MHMR_DT_BF16, NW, RING, MODE>), dim3(grid), dim3(64 * NW), lds, s, qk, vt, out, T, Tp, C, H, nqt, limit, flags, ldo, o8);
    MHMR_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// Number of ints of `flags` workspace the default attention form needs for a problem (one per wave of its 128-query workgroups).
int mhmr_attention_flag_count_impl(int B, int Tp, int H) { return 4 * ((Tp + 127) / 128) * H * B; }

// limit_log2 (variant 0): a workgroup is recomputed by the textbook kernel when a lane's tile sum of exp2(s - level) exceeded
// 2^limit_log2 (0 <= limit_log2 <= 15; 15 = the shipped value "would leave the 16-bit range", 0 = nearly every workgroup).
// variant: 0 = MODE 3 + gated MODE 1 fallback (needs `flags`), 1 = textbook (MODE 1), 2 = banded running maximum (MODE 2),
// 3 = MODE 2 with 8-wave workgroups, 4 / 5 = 64 queries per wave (attn64_kernel, MODE 3 arithmetic + gated fallback; needs `flags`),
// 6 = MODE 3 arithmetic on 16x16x32 MFMAs (attn16_kernel + gated fallback; needs `flags`): what mhmr_vit_forward runs.
// Measured at ViT-L 896 b32, f16 / bf16 TFLOP/s (tools/kbench.py, interleaved rounds): textbook 855-875 / 895-929; banded
// maximum 930-940 / 973-1003; the same with the level as a 16-register C tuple (3 waves per SIMD) 918-920 / 975; with all 8 K
// fragments and the V^T fragments requested ahead of their MFMAs (sched_barrier-pinned; 3 waves per SIMD, or 4 with spills)
// 840-918: LDS latency is not what bounds the kernel, the VALU work per tile and the 128-register budget are (a form that kept
// the exact rescale inside the loop next to the sum test needed all 128 registers and fell to one LDS read per MFMA: 850).
int mhmr_launch_attention_pitch(const void* qk, const void* vt, void* out, int B, int T, int Tp, int C, int H, int dtype, float limit_log2,
                                int variant, int* flags, hipStream_t s, int ldo, int o8);
int mhmr_launch_attention_ex(const void* qk, const void* vt, void* out, int B, int T, int Tp, int C, int H, int dtype, float limit_log2,
                             int variant, int* flags, hipStream_t s) {
    return mhmr_launch_attention_pitch(qk, vt, out, B, T, Tp, C, H, dtype, limit_log2, variant, flags, s, C, 0);
}
// ldo: row pitch of `out` in elements (>= C); o8 > 0: byte offset inside an out row where the bf8 (e5m2) copy of the row's C values goes
// (GemmArgs::lo8 of the output projection).  Only the default form (variant 6 + its fallback) takes a pitch other than C.
int mhmr_launch_attention_pitch(const void* qk, const void* vt, void* out, int B, int T, int Tp, int C, int H, int dtype, float limit_log2,
                                int variant, int* flags, hipStream_t s, int ldo, int o8) {
    if ((ldo != C || o8 != 0) && (variant < 6 || variant > 10 || ldo < C || o8 < 0 || (o8 > 0 && (o8 < 2 * C || o8 + C > 2 * ldo)))) return MHMR_ERR_BAD_ARG;
    if (C != H * 64 || Tp % 64 || T > Tp || T <= 0 || limit_log2 < 0.f || limit_log2 > 15.f) return MHMR_ERR_BAD_SHAPE;
    if ((variant == 0 || (variant >= 6 && variant <= 10)) && flags == nullptr) return MHMR_ERR_BAD_ARG;
    const float limit = exp2f(limit_log2);
    prof_begin(PROF_ATTN, s);
    int rc = 0;
    switch (variant) {
        case 0: {
            rc = launch_attn<4, 2, 3>(qk, vt, out, B, T, Tp, C, H, dtype, limit, flags, s);
            if (!rc) rc = launch_attn_fallback(qk, vt, out, B, T, Tp, C, H, dtype, limit, flags, s);      // recomputes the flagged workgroups only
            break;
        }
        case 1: rc = launch_attn<4, 2, 1>(qk, vt, out, B, T, Tp, C, H, dtype, limit, nullptr, s); break;
        case 2: rc = launch_attn<4, 2, 2>(qk, vt, out, B, T, Tp, C, H, dtype, limit, nullptr, s); break;
        case 3: rc = launch_attn<8, 3, 2>(qk, vt, out, B, T, Tp, C, H, dtype, limit, nullptr, s); break;
        case 4:
        case 5: {     // 64 queries per wave (attn64_kernel) + the gated textbook fallback; 4: 3-slot K/V ring, 5: 2-slot ring
            if (flags == nullptr) return MHMR_ERR_BAD_ARG;
            rc = variant == 4 ? launch_attn64<3>(qk, vt, out, B, T, Tp, C, H, dtype, limit, flags, s)
                              : launch_attn64<2>(qk, vt, out, B, T, Tp, C, H, dtype, limit, flags, s);
            if (!rc) rc = launch_attn_fallback(qk, vt, out, B, T, Tp, C, H, dtype, limit, flags, s);
            break;
        }
        case 6:       // MODE 3 arithmetic on v_mfma_f32_16x16x32 (attn16_kernel) + the gated textbook fallback
        case 7:       // ... with the next tile's copies in front of the score MFMAs
        case 8:       // ... with a three-slot K / V^T ring (three workgroups per CU)
        case 9: {     // ... both
            if (flags == nullptr) return MHMR_ERR_BAD_ARG;
            rc = variant == 6 ? launch_attn16<false, 2>(qk, vt, out, B, T, Tp, C, H, dtype, limit, flags, s, ldo, o8)
               : variant == 7 ? launch_attn16<true, 2>(qk, vt, out, B, T, Tp, C, H, dtype, limit, flags, s, ldo, o8)
               : variant == 8 ? launch_attn16<false, 3>(qk, vt, out, B, T, Tp, C, H, dtype, limit, flags, s, ldo, o8)
                              : launch_attn16<true, 3>(qk, vt, out, B, T, Tp, C, H, dtype, limit, flags, s, ldo, o8);
            if (!rc) rc = launch_attn_fallback(qk, vt, out, B, T, Tp, C, H, dtype, limit, flags, s, ldo, o8);
            break;
        }
        case 10: {    // variant 6 with the class query of T = 128 n + 1 on workgroups of its own (attn_cls_role) instead of a 128-query workgroup
            rc = launch_attn16_clsq(qk, vt, out, B, T, Tp, C, H, dtype, limit, flags, s, ldo, o8);
            if (!rc) rc = launch_attn_fallback(qk, vt, out, B, T, Tp, C, H, dtype, limit, flags, s, ldo, o8);
            break;
        }
        default: return MHMR_ERR_BAD_ARG;
    }
    prof_end(PROF_ATTN, s, 4.0 * B * H * (double)T * T * 64);
    return rc;
}

int mhmr_launch_attention(const void* qk, const void* vt, void* out, int B, int T, int Tp, int C, int H, int dtype, int* flags,
                          hipStream_t s, int ldo = 0, int o8 = 0) {
    static const char* v = getenv("MHMR_ATTN_VARIANT");      // A/B measurements only
    const int variant = v ? atoi(v) : (flags ? MHMR_ATTN_DEFAULT_VARIANT : 2);
    return mhmr_launch_attention_pitch(qk, vt, out, B, T, Tp, C, H, dtype, 15.f, variant, flags, s, ldo > 0 ? ldo : C, o8);
}
