// 256x256x64 "8-phase ping-pong" GEMM for the large ViT linears (same math and epilogues as gemm.hip).
//
//   workgroup = 8 waves (2 x 4) on one CU (1 block/CU, 2 waves per SIMD), 128 KiB LDS = 2 K-tile buffers (even/odd)
//   x 4 half-tile slots (P0, P1, Q0, Q1; 128 rows x 64 k, 16 KiB each).  A wave owns 64 P-rows in EACH P half and
//   32 Q-rows in EACH Q half, so its 128x64 output splits into 4 quadrants and every half-tile slot is read in
//   exactly one phase by all waves:
//
//     phase   ds_read (slot -> regs)        MFMA (8 x v_mfma_f32_32x32x16)        LDS-DMA issued (2 x glds16 / thread)
//     1 / 5   P0 -> PR (8), Q0 -> QA (4)    acc[0][0] += PR x QA                  Q1(odd,  t1) / Q1(even, t2)
//     2 / 6   Q1 -> QB (4)                  acc[0][1] += PR x QB                  P1(odd,  t1) / P1(even, t2)
//     3 / 7   P1 -> PR (8)                  acc[1][1] += PR x QB                  Q0(even, t2) / Q0(odd,  t3)
//     4 / 8   --                            acc[1][0] += PR x QA                  P0(even, t2) / P0(odd,  t3)
//
//   Every phase is  [ds_reads ; DMA issue ; s_waitcnt vmcnt(8)] s_barrier [MFMAs] s_barrier.  The two wave groups
//   (wp = 0 / 1, one wave of each per SIMD) run ONE barrier apart, so while one group's 8 MFMAs occupy the SIMD's
//   matrix pipe the other group issues its LDS reads and DMA -- the pipe never waits for a load phase.
//   vmcnt(8) after each issue = "the half-tile issued 4 phases ago has landed": every slot is waited for one phase
//   before it is first read (RAW needs the wait + a barrier) and is re-filled >= 2 phases after its last read
//   (WAR), never draining the DMA queue inside the loop.  Tail iterations re-issue the last K tile into slots that
//   are no longer read, which keeps the counted waits uniform.
#include "mhmr_common.h"
#include "mhmr_internal.h"

namespace {

constexpr int HT = 16384;           // bytes per half-tile slot (128 rows x 128 B)
constexpr int BUF = 4 * HT;         // one K-tile buffer: P0 | P1 | Q0 | Q1
constexpr int SLOT_P0 = 0, SLOT_P1 = HT, SLOT_Q0 = 2 * HT, SLOT_Q1 = 3 * HT;

template <int DT, int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GemmArgs g) {
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V8 V8;
    typedef typename Op<DT>::V4 V4;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = w >> 2, wq = w & 3;
    const int hi = lane >> 5, l31 = lane & 31;

    constexpr bool ROWMAJOR = (EPI != EPI_VT);
    const int nbn = g.N / 256;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % nbn, tm = bid / nbn;
    const int m0 = tm * 256, n0 = tn * 256;
    // P = first MFMA operand (D rows, 4 consecutive per accumulator quad), Q = second (D columns, one per lane)
    const T* Pm = ROWMAJOR ? (const T*)g.W : (const T*)g.A;
    const T* Qm = ROWMAJOR ? (const T*)g.A : (const T*)g.W;
    const int ldp = ROWMAJOR ? g.ldw : g.lda, ldq = ROWMAJOR ? g.lda : g.ldw;
    const int p0 = ROWMAJOR ? n0 : m0, q0 = ROWMAJOR ? m0 : n0;

    // ---- DMA source addressing: one half-tile = 2 passes of 64 rows; lane-linear LDS image, swizzle on the source ----
    const int srow = tid >> 3;                                  // 0..63
    const int schunk = (tid & 7) ^ ((tid >> 4) & 7);
    const T* p_src = Pm + (size_t)(p0 + srow) * ldp + schunk * 8;
    const T* q_src = Qm + (size_t)(q0 + srow) * ldq + schunk * 8;
    const int nt = g.K / 64;

    auto dma = [&](const T* src, int ld, int half, int kt, int lds_off) {
        kt = kt < nt ? kt : nt - 1;                             // tail: harmless re-load of the last tile
        const T* s = src + (size_t)(128 * half) * ld + kt * 64;
        char* d = smem + lds_off + w * 1024;
        glds16(s, d);
        glds16(s + (size_t)64 * ld, d + 8192);
    };

    // ---- fragment read addressing ----
    const int fsw = (lane >> 1) & 7;
    const int pr_off = (64 * wp + l31) * 128, q_off = (32 * wq + l31) * 128;
    int co[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) co[ks] = ((2 * ks + hi) ^ fsw) * 16;

    f32x16 acc[2][2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][s][r] = 0.f;
    V8 PR[2][4], QA[4], QB[4];

    auto rdP = [&](int buf, int slot) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) PR[s][ks] = *(const V8*)(smem + buf * BUF + slot + pr_off + s * 4096 + co[ks]);
    };
    auto rdQ = [&](V8 (&Q)[4], int buf, int slot) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) Q[ks] = *(const V8*)(smem + buf * BUF + slot + q_off + co[ks]);
    };
    auto mma = [&](f32x16 (&c)[2], const V8 (&Q)[4]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int s = 0; s < 2; ++s) c[s] = Op<DT>::mfma32(PR[s][ks], Q[ks], c[s]);
        __builtin_amdgcn_s_setprio(0);
    };
#define MHMR_SYNC()                          \
    __builtin_amdgcn_s_barrier();            \
    __builtin_amdgcn_sched_barrier(0)
#define MHMR_WAIT_DMA() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")

    // ---- prologue: tile 0 -> even buffer (all four halves), tile 1 -> odd buffer (Q0, P0) ----
    dma(q_src, ldq, 0, 0, SLOT_Q0);
    dma(p_src, ldp, 0, 0, SLOT_P0);
    dma(q_src, ldq, 1, 0, SLOT_Q1);
    dma(p_src, ldp, 1, 0, SLOT_P1);
    dma(q_src, ldq, 0, 1, BUF + SLOT_Q0);
    dma(p_src, ldp, 0, 1, BUF + SLOT_P0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MHMR_SYNC();
    if (wp == 1) { MHMR_SYNC(); }           // stagger: group 1 runs one barrier behind group 0

    for (int t = 0; t < nt; t += 2) {
        const int t1 = t + 1, t2 = t + 2, t3 = t + 3;
        // phase 1
        rdP(0, SLOT_P0); rdQ(QA, 0, SLOT_Q0);
        dma(q_src, ldq, 1, t1, BUF + SLOT_Q1); MHMR_WAIT_DMA();
        MHMR_SYNC(); mma(acc[0][0], QA); MHMR_SYNC();
        // phase 2
        rdQ(QB, 0, SLOT_Q1);
        dma(p_src, ldp, 1, t1, BUF + SLOT_P1); MHMR_WAIT_DMA();
        MHMR_SYNC(); mma(acc[0][1], QB); MHMR_SYNC();
        // phase 3
        rdP(0, SLOT_P1);
        dma(q_src, ldq, 0, t2, SLOT_Q0); MHMR_WAIT_DMA();
        MHMR_SYNC(); mma(acc[1][1], QB); MHMR_SYNC();
        // phase 4
        dma(p_src, ldp, 0, t2, SLOT_P0); MHMR_WAIT_DMA();
        MHMR_SYNC(); mma(acc[1][0], QA); MHMR_SYNC();
        // phase 5
        rdP(1, SLOT_P0); rdQ(QA, 1, SLOT_Q0);
        dma(q_src, ldq, 1, t2, SLOT_Q1); MHMR_WAIT_DMA();
        MHMR_SYNC(); mma(acc[0][0], QA); MHMR_SYNC();
        // phase 6
        rdQ(QB, 1, SLOT_Q1);
        dma(p_src, ldp, 1, t2, SLOT_P1); MHMR_WAIT_DMA();
        MHMR_SYNC(); mma(acc[0][1], QB); MHMR_SYNC();
        // phase 7
        rdP(1, SLOT_P1);
        dma(q_src, ldq, 0, t3, BUF + SLOT_Q0); MHMR_WAIT_DMA();
        MHMR_SYNC(); mma(acc[1][1], QB); MHMR_SYNC();
        // phase 8
        dma(p_src, ldp, 0, t3, BUF + SLOT_P0); MHMR_WAIT_DMA();
        MHMR_SYNC(); mma(acc[1][0], QA); MHMR_SYNC();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wp == 0) { MHMR_SYNC(); }           // re-align the two groups
    __syncthreads();                        // all DMA landed, all fragment reads done: LDS is free for the epilogue
#undef MHMR_SYNC
#undef MHMR_WAIT_DMA

    // ---- epilogue: per quadrant, transpose the wave's [32 Q-rows][64 P-cols] block through a private 8 KiB LDS
    //      region so that global accesses are whole contiguous row segments ----
    char* wl = smem + w * 8192;
    constexpr bool OUT16 = (EPI == EPI_OP16 || EPI == EPI_OP16_GELU || EPI == EPI_OP16_RELU || EPI == EPI_VT);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pb = p0 + 128 * h + 64 * wp;      // first P index of the block (n for row-major, token for V^T)
            const int qb = q0 + 128 * j + 32 * wq;      // first Q index (m for row-major, channel for V^T)
            if constexpr (OUT16) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int pc = 32 * s + 8 * rg + 4 * hi;
                        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                        if (g.bias) {
                            if constexpr (ROWMAJOR) bv = *(const f32x4*)(g.bias + pb + pc);
                            else { const float b1 = g.bias[qb + l31]; bv = (f32x4){b1, b1, b1, b1}; }
                        }
                        V4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[h][j][s][4 * rg + e] + bv[e];
                            if constexpr (EPI == EPI_OP16_GELU) v = gelu_fast(v);
                            if constexpr (EPI == EPI_OP16_RELU) v = fmaxf(v, 0.f);
                            o[e] = (T)v;
                        }
                        const int pos = ROWMAJOR ? pc : ((pc & ~12) | ((pc & 4) << 1) | ((pc & 8) >> 1));  // V^T: swap bits 2,3
                        *(V4*)(wl + l31 * 128 + (((pos >> 2) ^ ((l31 & 7) << 1)) * 8)) = o;
                    }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = 8 * it + (lane >> 3), c16 = lane & 7;
                    const u32x4 v = *(const u32x4*)(wl + row * 128 + ((c16 ^ (row & 7)) * 16));
                    if constexpr (ROWMAJOR) {
                        *(u32x4*)((T*)g.out + (size_t)(qb + row) * g.ldo + pb + c16 * 8) = v;
                    } else {
                        const int n = qb + row, b = pb / g.Tp, tl = pb - b * g.Tp;
                        *(u32x4*)((T*)g.out + ((size_t)(b * g.H + (n >> 6)) * 64 + (n & 63)) * g.Tp + tl + c16 * 8) = v;
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int ch = 8 * s + 2 * rg + hi;
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[h][j][s][4 * rg + e];
                        *(f32x4*)(wl + l31 * 256 + ((ch ^ (l31 & 15)) * 16)) = v;
                    }
                const int c = lane & 15;
                const int n = pb + 4 * c;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f}, gm = {1.f, 1.f, 1.f, 1.f};
                if (g.bias) bv = *(const f32x4*)(g.bias + n);
                if constexpr (EPI == EPI_RESID) gm = *(const f32x4*)(g.gamma + n);
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int row = 4 * it + (lane >> 4);
                    f32x4 v = *(const f32x4*)(wl + row * 256 + ((c ^ (row & 15)) * 16)) + bv;
                    const int m = qb + row;
                    if constexpr (EPI == EPI_RESID) {
                        float* op = (float*)g.out + (size_t)m * g.ldo + n;
                        *(f32x4*)op = *(const f32x4*)op + gm * v;
                    } else if constexpr (EPI == EPI_PATCH) {
                        if (m < g.Mvalid) {
                            const int b = m / g.Np, n_in = m - b * g.Np;
                            v += *(const f32x4*)(g.pos + (size_t)(1 + n_in) * g.N + n);
                            *(f32x4*)((float*)g.out + ((size_t)b * g.Tp + 1 + n_in) * g.ldo + n) = v;
                        }
                    } else {
                        *(f32x4*)((float*)g.out + (size_t)m * g.ldo + n) = v;
                    }
                }
            }
        }
    }
}

template <int DT>
int launch256_dt(const GemmArgs& g, hipStream_t s) {
    const int grid = (g.M / 256) * (g.N / 256);
    const size_t lds = 2 * BUF;
#define MHMR_GEMM_CASE(E)                                                                                      \
    case E: {                                                                                                  \
        static bool attr_set = false;                                                                          \
        if (!attr_set) {                                                                                       \
            (void)hipFuncSetAttribute((const void*)gemm256_kernel<DT, E>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds);                                                               \
            attr_set = true;                                                                                   \
        }                                                                                                      \
        hipLaunchKernelGGL((gemm256_kernel<DT, E>), dim3(grid), dim3(512), lds, s, g);                         \
        break;                                                                                                 \
    }
    switch (g.epi) {
        MHMR_GEMM_CASE(EPI_OP16)
        MHMR_GEMM_CASE(EPI_OP16_GELU)
        MHMR_GEMM_CASE(EPI_OP16_RELU)
        MHMR_GEMM_CASE(EPI_RESID)
        MHMR_GEMM_CASE(EPI_PATCH)
        MHMR_GEMM_CASE(EPI_F32)
        MHMR_GEMM_CASE(EPI_VT)
        default:
            return MHMR_ERR_BAD_ARG;
    }
#undef MHMR_GEMM_CASE
    MHMR_CHECK_LAUNCH();
    return 0;
}

}  // namespace

bool mhmr_gemm256_eligible(const GemmArgs& g) {
    return g.M % 256 == 0 && g.N % 256 == 0 && g.K % 128 == 0 && (g.epi != EPI_VT || g.Tp % 64 == 0);
}

int mhmr_launch_gemm256(const GemmArgs& g, int dtype, hipStream_t s) {
    return dtype == MHMR_DT_F16 ? launch256_dt<MHMR_DT_F16>(g, s) : launch256_dt<MHMR_DT_BF16>(g, s);
}
