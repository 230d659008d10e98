// The CLASS-token rows of the DINOv2 block linears (reference blocks/dinov2.py:16-26 -> DinoVisionTransformer blocks: the class
// token runs through the same qkv / proj / fc1 / fc2 as the patch tokens).
//
// With the class token stored LAST in every image (row b*Tp + N, vit_misc.hip) the big 256x256-tile GEMMs run over the B*N patch
// rows only -- a whole number of tile rounds (896^2 x 32: 2048 / 4096 / 8192 tiles on 256 CUs) instead of 8.1 / 16.25 / 32.5 rounds
// paid as 9 / 17 / 33 -- and the B class rows (one per image, Tp rows apart) are computed here: a "skinny" GEMM, M = B <= 32 rows per
// block pass, which is a weight-streaming problem (2 ... 8 MB of 16-bit weights per linear against 67 ... 268 MFLOP).
//
//   grid = (N / 16, ceil(B / 32)); one workgroup = NW = 8 (4 when K % 256 != 0) waves = 16 output columns x 32 rows; the k range is split
//   over the waves (each streams its share of the 16 weight rows straight into MFMA operand registers: lane (n = lane & 15, k group = lane >> 4)
//   holds W[n][32 s + 8 g .. + 7] as one 16-byte load; the activations' fragments come the same way, L2-resident), partial
//   accumulators meet in LDS and are summed in wave order (bit-reproducible).  v_mfma_f32_16x16x32 with the WEIGHT as the first
//   operand: a lane ends up with 4 consecutive output columns of one row -- the same orientation as gemm256.hip, so the epilogues
//   (bias, Q scale, V^T scatter, LayerScale + residual, GELU) are the same arithmetic in the same order.
//
// Round 6: the row statistics of the LayerNorm fold ride in THIS kernel's residual launches instead of a launch of their own (47 per
// forward, 11.8 us each + a boundary):
//   * the statistics of the PATCH rows (ln_stats.h, from the big GEMM's block sums) are extra workgroups behind the N / 16 class-row
//     workgroups of a CLS_RESID launch (ClsArgs::st_*): other CUs, same launch;
//   * the CLASS rows get block sums of their own: a CLS_RESID workgroup leaves (sum, sum of squares) of its 16 columns of every row in
//     cls_pstats[row][N / 16][2], and the consumers (CLS_QKV / CLS_GELU with ClsArgs::cls_pstats) add the N / 16 pairs of a row in a
//     fixed order -> (mean, rstd).  (Before: one wave per class row re-read the fp32 row in ln_stats_kernel.)
#include "mhmr_common.h"
#include "mhmr_internal.h"
#include "ln_stats.h"

namespace {

struct ClsArgs {
    const void* A; long long a_stride;      // activation row b at A + b * a_stride (elements); k contiguous
    const void* W; int ldw;                 // [N][ldw] weight rows (k contiguous)
    int B, N, K, a_k;                       // K = total k (2 * a_k with a low-half weight pass: A's k index wraps at a_k), K % 128 == 0
    const float* bias; const float* gamma;
    void* out; long long o_stride;          // output row b at out + b * o_stride (elements of the output type)
    int n_base, C;                          // CLS_QKV: global column of local column 0; embed dim (column regions Q | K | V)
    void* vt; int H, Tp, vcol;              // CLS_QKV: V^T [B][H][64][Tp], the (already key-permuted) column of the class token
    // LayerNorm fold (GemmArgs in mhmr_internal.h): consumer side -- rowstats (mean, rstd) of row b at rowstats + b * rs_stride floats,
    // out = rstd * (acc - mean * colsum_n) + fbias_n (bias null); producer side (CLS_RESID) -- x16: 16-bit copy of the updated rows
    const float* rowstats; long long rs_stride; const float* colsum; const float* fbias;
    void* x16; long long x_stride;
    // class-row block sums: CLS_RESID writes cls_pstats[row][N / 16][2]; a consumer with cls_pstats != null takes (mean, rstd) of its rows
    // from them (cls_nblk = the producing linear's N / 16 = embed_dim / 16, cls_C = embed_dim) instead of from `rowstats`
    float* cls_pstats; int cls_nblk, cls_C; float cls_eps;
    // statistics role (CLS_RESID only): st_blocks > 0 extra workgroups at blockIdx.x >= N / 16 run ln_stats_patch_rows over st_B x st_N rows
    const float* st_pstats; float* st_rowstats; int st_blocks, st_B, st_N, st_Tp, st_C;
};

enum { CLS_QKV = 0, CLS_RESID = 1, CLS_GELU = 2 };

// U = k steps (of 32) per batch of loads: a wave requests 3 U fragments (weight + two row blocks) before the first MFMA of the batch, so
// a batch costs ONE L2 round trip (hipcc does not unroll the run-time k loop by itself: one round trip per k step, 32 in a row for fc2)
template <int DT, int EPI, int U, int NW>
__global__ __launch_bounds__(64 * NW) void cls_linear_kernel(const ClsArgs a) {
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V8 V8;
    typedef typename Op<DT>::V4 V4;
    __shared__ f32x4 red[NW - 1][2][64];                  // partial accumulators of waves 1..NW-1
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g4 = lane >> 4, l15 = lane & 15;
    if constexpr (EPI == CLS_RESID) {
        if ((int)blockIdx.x >= a.N / 16) {               // statistics role: the patch rows' (mean, rstd) from the big GEMM's block sums
            if (blockIdx.y == 0)
                ln_stats_patch_rows(((int)blockIdx.x - a.N / 16) * (64 * NW) + (int)threadIdx.x, a.st_pstats, a.st_rowstats, a.st_B, a.st_N, a.st_Tp,
                                    a.st_C, a.st_C / 64, a.cls_eps);
            return;
        }
    }
    const int n0 = blockIdx.x * 16, rb = blockIdx.y * 32;
    // rows past B read row B - 1 (their results are dropped)
    const int r0 = min(rb + l15, a.B - 1), r1 = min(rb + 16 + l15, a.B - 1);
    const T* wp = (const T*)a.W + (size_t)(n0 + l15) * a.ldw + 8 * g4;
    const T* a0p = (const T*)a.A + (size_t)r0 * a.a_stride + 8 * g4;
    const T* a1p = (const T*)a.A + (size_t)r1 * a.a_stride + 8 * g4;
    const int kq = a.K / NW, k0 = w * kq;
    const int awrap = a.a_k > 0 ? a.a_k : a.K;
    // consumer of class-row block sums: wave 0 (the one that runs the epilogue) requests its rows' pairs now; they land under the k loop.
    // Lane (l15, g4) takes pairs g4 * nq .. + nq - 1 of rows rb + l15 and rb + 16 + l15 (nq = cls_nblk / 4 <= 16: C <= 1024)
    [[maybe_unused]] f32x4 cp0[8], cp1[8];
    [[maybe_unused]] const int nq2 = EPI != CLS_RESID && a.cls_pstats ? a.cls_nblk / 8 : 0;       // f32x4 loads (two pairs each) per lane and row
    if constexpr (EPI != CLS_RESID) {
        if (w == 0 && a.cls_pstats) {
            const float* p0 = a.cls_pstats + ((size_t)r0 * a.cls_nblk + g4 * (a.cls_nblk / 4)) * 2;
            const float* p1 = a.cls_pstats + ((size_t)r1 * a.cls_nblk + g4 * (a.cls_nblk / 4)) * 2;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                cp0[i] = i < nq2 ? *(const f32x4*)(p0 + 4 * i) : (f32x4){0.f, 0.f, 0.f, 0.f};
                cp1[i] = i < nq2 ? *(const f32x4*)(p1 + 4 * i) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int kk = k0; kk < k0 + kq; kk += 32 * U) {          // (a batch never straddles the wrap point: a_k % (32 U) == 0, launcher)
        const int ka = kk >= awrap ? kk - awrap : kk;
        V8 wf[U], x0[U], x1[U];
#pragma unroll
        for (int i = 0; i < U; ++i) {
            wf[i] = *(const V8*)(wp + kk + 32 * i);
            x0[i] = *(const V8*)(a0p + ka + 32 * i);
            x1[i] = *(const V8*)(a1p + ka + 32 * i);
        }
#pragma unroll
        for (int i = 0; i < U; ++i) {
            acc0 = Op<DT>::mfma16(wf[i], x0[i], acc0);
            acc1 = Op<DT>::mfma16(wf[i], x1[i], acc1);
        }
    }
    if (w > 0) { red[w - 1][0][lane] = acc0; red[w - 1][1][lane] = acc1; }
    __syncthreads();
    if (w > 0) return;
#pragma unroll
    for (int i = 0; i < NW - 1; ++i) { acc0 += red[i][0][lane]; acc1 += red[i][1][lane]; }      // wave order: bit-reproducible
    // lane holds output columns n0 + 4 g4 + 0..3 of rows rb + l15 (acc0) and rb + 16 + l15 (acc1)
    const int nl = n0 + 4 * g4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f}, cs = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bv = *(const f32x4*)(a.bias + nl);
    const bool folded = a.rowstats != nullptr || (EPI != CLS_RESID && a.cls_pstats != nullptr);
    if (folded) { bv = *(const f32x4*)(a.fbias + nl); cs = *(const f32x4*)(a.colsum + nl); }
    // (mean, rstd) of this lane's two rows from the class rows' block sums: the lane's share in index order, then the four shares of a row
    // (lanes l15, l15 + 16, + 32, + 48) in lane order -- fixed, bit-reproducible
    [[maybe_unused]] f32x2 mrc[2] = {{0.f, 1.f}, {0.f, 1.f}};
    if constexpr (EPI != CLS_RESID) {
        if (a.cls_pstats) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x4 c4 = half ? cp1[i] : cp0[i];
                    s1 += c4[0] + c4[2];
                    s2 += c4[1] + c4[3];
                }
                const float a1 = __shfl(s1, l15), b1 = __shfl(s1, l15 + 16), c1 = __shfl(s1, l15 + 32), d1 = __shfl(s1, l15 + 48);
                const float a2 = __shfl(s2, l15), b2 = __shfl(s2, l15 + 16), c2 = __shfl(s2, l15 + 32), d2 = __shfl(s2, l15 + 48);
                const float t1 = ((a1 + b1) + c1) + d1, t2 = ((a2 + b2) + c2) + d2;
                const float mean = t1 * (1.0f / a.cls_C);
                const float var = fmaxf(t2 * (1.0f / a.cls_C) - mean * mean, 0.f);
                mrc[half] = (f32x2){mean, rsqrtf(var + a.cls_eps)};
            }
        }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int m = rb + 16 * half + l15;
        if (m >= a.B) continue;
        f32x4 v = half ? acc1 : acc0;
        if (folded) {
            f32x2 mr;
            if (EPI != CLS_RESID && a.cls_pstats) mr = mrc[half];
            else mr = *(const f32x2*)(a.rowstats + (size_t)m * a.rs_stride);
            const float t = -mr[0] * mr[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(v[e], mr[1], __builtin_fmaf(t, cs[e], bv[e]));
        } else {
            v += bv;
        }
        if constexpr (EPI == CLS_RESID) {
            float* op = (float*)a.out + (size_t)m * a.o_stride + nl;
            const f32x4 gm = *(const f32x4*)(a.gamma + nl);
            const f32x4 nv = *(const f32x4*)op + gm * v;
            *(f32x4*)op = nv;
            if (a.x16) {
                V4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (T)nv[e];
                *(V4*)((T*)a.x16 + (size_t)m * a.x_stride + nl) = o;
            }
            if (a.cls_pstats) {
                // this workgroup's 16 columns of row m: four lanes hold four columns each
                float s1 = (nv[0] + nv[1]) + (nv[2] + nv[3]);
                float s2 = (nv[0] * nv[0] + nv[1] * nv[1]) + (nv[2] * nv[2] + nv[3] * nv[3]);
                s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                if (g4 == 0) *(f32x2*)(a.cls_pstats + ((size_t)m * a.cls_nblk + blockIdx.x) * 2) = (f32x2){s1, s2};
            }
        } else if constexpr (EPI == CLS_GELU) {
            V4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (T)gelu_fast(v[e]);
            *(V4*)((T*)a.out + (size_t)m * a.o_stride + nl) = o;
        } else {
            const int ng = a.n_base + nl;                 // the 16 columns of a workgroup lie in one of the regions Q | K | V
            if (ng < 2 * a.C) {
                const float sc = ng < a.C ? MHMR_ATTN_QSCALE : 1.f;
                V4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (T)(v[e] * sc);
                *(V4*)((T*)a.out + (size_t)m * a.o_stride + ng) = o;
            } else {
                const int c = ng - 2 * a.C, h = c >> 6, d = c & 63;
                T* vp = (T*)a.vt + ((size_t)(m * a.H + h) * 64 + d) * a.Tp + a.vcol;
#pragma unroll
                for (int e = 0; e < 4; ++e) vp[(size_t)e * a.Tp] = (T)v[e];
            }
        }
    }
}

template <int DT, int U, int NW>
int launch_cls_u(const ClsArgs& a, int epi, hipStream_t s) {
    const dim3 grid(a.N / 16 + (epi == CLS_RESID ? a.st_blocks : 0), (a.B + 31) / 32);
    switch (epi) {
        case CLS_QKV: mhmr_launch_kernel(cls_linear_kernel<DT, CLS_QKV, U, NW>, grid, dim3(64 * NW), 0, s, a); break;
        case CLS_RESID: mhmr_launch_kernel(cls_linear_kernel<DT, CLS_RESID, U, NW>, grid, dim3(64 * NW), 0, s, a); break;
        case CLS_GELU: mhmr_launch_kernel(cls_linear_kernel<DT, CLS_GELU, U, NW>, grid, dim3(64 * NW), 0, s, a); break;
        default: return MHMR_ERR_BAD_ARG;
    }
    MHMR_CHECK_LAUNCH();
    return 0;
}

template <int DT, int NW>
int launch_cls_nw(const ClsArgs& a, int epi, hipStream_t s) {
    // the largest batch of k steps that divides a wave's k share and the wrap point
    const int kq = a.K / NW, wrap = a.a_k > 0 ? a.a_k : kq;
    auto ok = [&](int u) { return kq % (32 * u) == 0 && wrap % (32 * u) == 0; };
    if (ok(8)) return launch_cls_u<DT, 8, NW>(a, epi, s);
    if (ok(6)) return launch_cls_u<DT, 6, NW>(a, epi, s);
    if (ok(4)) return launch_cls_u<DT, 4, NW>(a, epi, s);
    if (ok(3)) return launch_cls_u<DT, 3, NW>(a, epi, s);
    return launch_cls_u<DT, 1, NW>(a, epi, s);
}

template <int DT>
int launch_cls(const ClsArgs& a, int epi, hipStream_t s) {
    return a.K % 256 == 0 ? launch_cls_nw<DT, 8>(a, epi, s) : launch_cls_nw<DT, 4>(a, epi, s);
}

}  // namespace

// epi: 0 = Q | K | V projection of the class rows (Q pre-scaled, V scattered into column `vcol` of V^T), 1 = out32 += gamma * (acc + bias),
// 2 = out16 = gelu(acc + bias)
int mhmr_launch_cls_linear_fold(const void* A, long long a_stride, const void* W, int ldw, int B, int N, int K, int a_k, const float* bias,
                                const float* gamma, void* out, long long o_stride, int n_base, int C, void* vt, int H, int Tp, int vcol, int epi,
                                int dtype, const float* rowstats, long long rs_stride, const float* colsum, const float* fbias, void* x16,
                                long long x_stride, hipStream_t s, const ClsStats* st) {
    if (B <= 0 || N <= 0 || N % 16 || K <= 0 || K % 128 || ldw < K || (a_k > 0 && K != 2 * a_k) || a_stride % 8 || ldw % 8) return MHMR_ERR_BAD_SHAPE;
    if (epi == CLS_RESID && !gamma) return MHMR_ERR_BAD_ARG;
    if (epi == CLS_QKV && (C % 64 || n_base % 16 || !vt || Tp <= vcol)) return MHMR_ERR_BAD_SHAPE;
    if (rowstats && (!colsum || !fbias || bias || epi == CLS_RESID)) return MHMR_ERR_BAD_ARG;
    ClsArgs a{A, a_stride, W, ldw, B, N, K, a_k, bias, gamma, out, o_stride, n_base, C, vt, H, Tp, vcol, rowstats, rs_stride, colsum, fbias,
              x16, x_stride, nullptr, 0, 0, 1e-6f, nullptr, nullptr, 0, 0, 0, 0, 0};
    if (st) {
        if (st->cls_pstats) {
            // producer: one pair per 16-column workgroup; consumers: N / 16 pairs of a row, four lanes x at most eight 16-byte loads
            if (epi == CLS_RESID ? st->cls_nblk != N / 16 : (st->cls_nblk % 8 || st->cls_nblk > 64 || st->cls_C != 16 * st->cls_nblk || !colsum || !fbias || bias))
                return MHMR_ERR_BAD_ARG;
            a.cls_pstats = st->cls_pstats; a.cls_nblk = st->cls_nblk; a.cls_C = st->cls_C; a.cls_eps = st->eps;
        }
        if (st->st_pstats && epi == CLS_RESID) {
            if (!st->st_rowstats || st->st_C % 128 || st->st_C > 1024 || st->st_B <= 0 || st->st_N <= 0) return MHMR_ERR_BAD_ARG;
            const int threads = 64 * (K % 256 == 0 ? 8 : 4);
            a.st_pstats = st->st_pstats; a.st_rowstats = st->st_rowstats; a.st_B = st->st_B; a.st_N = st->st_N; a.st_Tp = st->st_Tp; a.st_C = st->st_C;
            a.cls_eps = st->eps;
            a.st_blocks = (int)(((long long)st->st_B * st->st_N * 8 + threads - 1) / threads);
        }
    }
    return dtype == MHMR_DT_F16 ? launch_cls<MHMR_DT_F16>(a, epi, s) : launch_cls<MHMR_DT_BF16>(a, epi, s);
}

int mhmr_launch_cls_linear(const void* A, long long a_stride, const void* W, int ldw, int B, int N, int K, int a_k, const float* bias,
                           const float* gamma, void* out, long long o_stride, int n_base, int C, void* vt, int H, int Tp, int vcol, int epi,
                           int dtype, hipStream_t s) {
    return mhmr_launch_cls_linear_fold(A, a_stride, W, ldw, B, N, K, a_k, bias, gamma, out, o_stride, n_base, C, vt, H, Tp, vcol, epi, dtype,
                                       nullptr, 0, nullptr, nullptr, nullptr, 0, s, nullptr);
}
