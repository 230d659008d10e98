// 16-bit-operand GEMM with fused epilogues for the DINOv2 ViT blocks and the heads.
//
//   C[m][n] = sum_k A[m][k] * W[n][k]          A: [M,K] activations, W: [N,K] nn.Linear weight (K contiguous)
//
// gfx950 structure: 128x128 block tile, BK = 64, 4 waves (2x2), each wave a 64x64 sub-tile as 2x2
// v_mfma_f32_32x32x16 fragments (fp32 accumulate).  Both operand tiles are staged by
// global_load_lds_dwordx4 (16 B/lane, no VGPR round trip) into double-buffered LDS with the XOR swizzle
// applied on the per-lane SOURCE address (the DMA destination is lane-linear), read back conflict-free with
// ds_read_b128; one barrier per K tile, next tile's DMA in flight under the current tile's 16 MFMAs.
//
// Orientation: the MFMA's first operand indexes D's rows (4 consecutive rows per accumulator quad), the second
// D's columns (one per lane).  For row-major [m][n] outputs the WEIGHT tile is the first operand so every lane
// owns 4 consecutive n of one m (8/16-byte stores); for the transposed V^T output the ACTIVATION tile is first
// so every lane owns 4 consecutive tokens of one channel.
#include <stdlib.h>
#include "mhmr_common.h"
#include <string.h>
#include <stdlib.h>
#include "mhmr_internal.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;  // 16 KiB per operand tile

template <int DT, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs g) {
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V8 V8;
    typedef typename Op<DT>::V4 V4;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A tile | W tile]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;

    const int nbn = g.N / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % nbn, tm = bid / nbn;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging addresses (per thread constant; advance by BK per K tile) ----
    const int srow = tid >> 3;                                   // 0..31 (+32 per pass)
    const int schunk = (tid & 7) ^ ((tid >> 4) & 7);             // logical chunk landing at phys chunk tid&7
    const int m0p = (int)mhmr_phys_row(m0, g.img_rows, g.img_stride);      // physical first activation / output row (mhmr_internal.h)
    // (round 6: a wave-uniform 64-bit tile base, advanced on the scalar unit, + a 32-bit BYTE offset per lane that never changes: the copies take
    // the SGPR-base form of global_load_lds -- rounds 1-5 kept two 64-bit lane pointers and paid a 64-bit vector add per copy, eight per k tile)
    const char* a_base = (const char*)((const T*)g.A + (size_t)m0p * g.lda);
    const char* w_base = (const char*)((const T*)g.W + (size_t)n0 * g.ldw);
    const uint32_t a_lane = ((uint32_t)srow * (uint32_t)g.lda + (uint32_t)(schunk * 8)) * (uint32_t)sizeof(T);
    const uint32_t w_lane = ((uint32_t)srow * (uint32_t)g.ldw + (uint32_t)(schunk * 8)) * (uint32_t)sizeof(T);
    const size_t a_pass = (size_t)32 * g.lda * sizeof(T), w_pass = (size_t)32 * g.ldw * sizeof(T);

    const int nta = g.a_k > 0 ? g.a_k / BK : g.K / BK;
    auto stage = [&](int kt, int buf) {
        char* sa = smem + buf * (2 * TILE_BYTES) + w * 1024;
        char* sw = sa + TILE_BYTES;
        const char* ap = a_base + (size_t)((kt >= nta ? kt - nta : kt) * BK) * sizeof(T);      // low-half weight pass: the activation's k tiles wrap around
        const char* wp = w_base + (size_t)(kt * BK) * sizeof(T);
        // (the lane offsets pass through an empty asm HERE: instruction selection works per basic block, a zero-extension hoisted out of the
        // k loop would leave a 64-bit vector add per copy; the row-pass bases are pinned to scalar registers for the same reason)
        uint32_t al = a_lane, wl = w_lane;
        asm volatile("" : "+v"(al), "+v"(wl));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char* api = ap + i * a_pass;
            const char* wpi = wp + i * w_pass;
            asm volatile("" : "+s"(api), "+s"(wpi));
            glds16(api + al, sa + i * 4096);
            glds16(wpi + wl, sw + i * 4096);
        }
    };

    // ---- fragment read offsets ----
    constexpr bool ROWMAJOR = (EPI != EPI_VT);
    // first operand (D rows): W tile for row-major outputs, A tile for V^T; second operand: the other one
    const int wp_ = w >> 1, wq_ = w & 1;
    const int p_tile_off = ROWMAJOR ? TILE_BYTES : 0;
    const int q_tile_off = ROWMAJOR ? 0 : TILE_BYTES;
    const int fsw = (lane >> 1) & 7;  // ((row >> 1) & 7) for row = 32*s + l31
    const int p_row = 64 * wp_ + l31, q_row = 64 * wq_ + l31;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = g.K / BK;
    stage(0, 0);
    for (int t = 0; t < nt; ++t) {
        // tile t landed and everyone is done reading the other buffer.  The vmcnt wait is EXPLICIT: inside a loop hipcc
        // (ROCm 7.2) does not emit it for LDS-DMA in front of __syncthreads() (it hoists it out of the loop).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) stage(t + 1, (t + 1) & 1);
        const char* sb = smem + (t & 1) * (2 * TILE_BYTES);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ph = ((2 * ks + hi) ^ fsw) * 16;
            V8 pf[2], qf[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                pf[s] = *(const V8*)(sb + p_tile_off + (p_row + 32 * s) * 128 + ph);
                qf[s] = *(const V8*)(sb + q_tile_off + (q_row + 32 * s) * 128 + ph);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = Op<DT>::mfma32(pf[i], qf[j], acc[i][j]);
        }
    }

    // ---- epilogue ----
    // Row-major outputs go through LDS so that global accesses are whole contiguous row segments (a direct store
    // from the MFMA layout touches 32 rows x 8..16 B per instruction at a power-of-two row stride, which camps on
    // one memory channel).  Each wave transposes its own 64(m) x 64(n) sub-tile in a private 16 KiB LDS region:
    // write in fragment order with an XOR swizzle, read back row-contiguous (16 B per lane).
    if constexpr (ROWMAJOR) {
        __syncthreads();  // every wave is done reading the last K tile
        char* wl = smem + w * 16384;
        const int mb = m0p + 64 * wq_, nb = n0 + 64 * wp_;
        constexpr bool OUT16 = (EPI == EPI_OP16 || EPI == EPI_OP16_GELU || EPI == EPI_OP16_RELU || EPI == EPI_OP16_QK);
        if constexpr (OUT16) {
            // bias + activation + convert here (lane owns 4 consecutive n), LDS rows = 64 n x 2 B = 128 B
            const float qscale = (EPI == EPI_OP16_QK && nb < (g.N >> 1)) ? MHMR_ATTN_QSCALE : 1.f;   // the wave's 64 columns are all Q or all K
#pragma unroll
            for (int pi = 0; pi < 2; ++pi)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int nl = 32 * pi + 8 * rg + 4 * hi;
                    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                    if (g.bias) bv = *(const f32x4*)(g.bias + nb + nl);
#pragma unroll
                    for (int qj = 0; qj < 2; ++qj) {
                        const int ml = 32 * qj + l31;
                        V4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[pi][qj][4 * rg + e] + bv[e];
                            if constexpr (EPI == EPI_OP16_GELU) v = gelu_fast(v);
                            if constexpr (EPI == EPI_OP16_RELU) v = fmaxf(v, 0.f);
                            if constexpr (EPI == EPI_OP16_QK) v *= qscale;
                            o[e] = (T)v;
                        }
                        *(V4*)(wl + ml * 128 + (((nl >> 2) ^ ((ml & 7) << 1)) * 8)) = o;
                    }
                }
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): wave-private region, no barrier needed
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = 8 * it + (lane >> 3), c16 = lane & 7;
                const u32x4 v = *(const u32x4*)(wl + row * 128 + ((c16 ^ (row & 7)) * 16));
                *(u32x4*)((T*)g.out + (size_t)(mb + row) * g.ldo + nb + c16 * 8) = v;
            }
        } else {
            // fp32 tile: LDS rows = 64 n x 4 B = 256 B
#pragma unroll
            for (int pi = 0; pi < 2; ++pi)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int ch = 8 * pi + 2 * rg + hi;  // 16-byte chunk = (n_local / 4)
#pragma unroll
                    for (int qj = 0; qj < 2; ++qj) {
                        const int ml = 32 * qj + l31;
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[pi][qj][4 * rg + e];
                        *(f32x4*)(wl + ml * 256 + ((ch ^ (ml & 15)) * 16)) = v;
                    }
                }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            const int c = lane & 15;
            const int n = nb + 4 * c;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f}, gm = {1.f, 1.f, 1.f, 1.f};
            if (g.bias) bv = *(const f32x4*)(g.bias + n);
            if constexpr (EPI == EPI_RESID) gm = *(const f32x4*)(g.gamma + n);
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = 4 * it + (lane >> 4);
                f32x4 v = *(const f32x4*)(wl + row * 256 + ((c ^ (row & 15)) * 16)) + bv;
                const int m = mb + row;
                if constexpr (EPI == EPI_RESID) {
                    float* op = (float*)g.out + (size_t)m * g.ldo + n;
                    *(f32x4*)op = *(const f32x4*)op + gm * v;
                } else if constexpr (EPI == EPI_PATCH) {
                    if (m < g.Mvalid) {
                        const int b = m / g.Np, n_in = m - b * g.Np;
                        v += *(const f32x4*)(g.pos + (size_t)(1 + n_in) * g.N + n);
                        *(f32x4*)((float*)g.out + ((size_t)b * g.Tp + n_in) * g.ldo + n) = v;      // class token LAST: patch n at row n
                    }
                } else {  // EPI_F32
                    *(f32x4*)((float*)g.out + (size_t)m * g.ldo + n) = v;
                }
            }
        }
    } else {
        // V^T: vt[((b*H + h)*64 + d) * Tp + swap23(t)], 4 consecutive tokens per lane (8-byte store)
        const int rpi = g.img_rows > 0 ? g.img_rows : g.Tp;
        const int b = m0 / rpi, t0 = m0 - b * rpi;
#pragma unroll
        for (int qj = 0; qj < 2; ++qj) {
            const int n = n0 + 64 * wq_ + 32 * qj + l31;
            const float bv = g.bias ? g.bias[n] : 0.f;
            const int h = n >> 6, d = n & 63;
            T* base = (T*)g.out + ((size_t)(b * g.H + h) * 64 + d) * g.Tp;
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) {
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int t = t0 + 64 * wp_ + 32 * pi + 8 * rg + 4 * hi;
                    const int ts = (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1);  // swap key bits 2 and 3
                    V4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (T)(acc[pi][qj][4 * rg + e] + bv);
                    *(V4*)(base + ts) = o;
                }
            }
        }
    }
}

template <int DT>
int launch_dt(const GemmArgs& g, hipStream_t s) {
    const int grid = (g.M / BM) * (g.N / BN);
    const size_t lds = 4 * TILE_BYTES;
#define MHMR_GEMM_CASE(E)                                                                                   \
    case E: {                                                                                               \
        static DeviceOnce once;                                                                             \
        int dev = 0;                                                                                        \
        const int need = once.need(&dev);                                                                   \
        if (need == -2) return MHMR_ERR_BAD_ARG;                                                            \
        if (need >= 0) {                                                                                    \
            hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<DT, E>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               (int)lds);                                                   \
            if (e != hipSuccess) return (int)e;                                                             \
            once.mark(dev);                                                                                 \
        }                                                                                                   \
        hipLaunchKernelGGL((gemm_kernel<DT, E>), dim3(grid), dim3(256), lds, s, g);                         \
        break;                                                                                              \
    }
    switch (g.epi) {
        MHMR_GEMM_CASE(EPI_OP16)
        MHMR_GEMM_CASE(EPI_OP16_GELU)
        MHMR_GEMM_CASE(EPI_OP16_RELU)
        MHMR_GEMM_CASE(EPI_RESID)
        MHMR_GEMM_CASE(EPI_PATCH)
        MHMR_GEMM_CASE(EPI_F32)
        MHMR_GEMM_CASE(EPI_VT)
        MHMR_GEMM_CASE(EPI_OP16_QK)
        default:
            return MHMR_ERR_BAD_ARG;
    }
#undef MHMR_GEMM_CASE
    MHMR_CHECK_LAUNCH();
    return 0;
}

}  // namespace

bool mhmr_gemm256_eligible(const GemmArgs& g);
int mhmr_launch_gemm256(const GemmArgs& g, int dtype, hipStream_t s);

int mhmr_cu_count() {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (dev >= 0 && dev < 64) {
        const int c = cus[dev].load(std::memory_order_relaxed);
        if (c > 0) return c;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    if (dev >= 0 && dev < 64) cus[dev].store(prop.multiProcessorCount, std::memory_order_relaxed);
    return prop.multiProcessorCount;
}

namespace {
// log2(panels per column group) for the two wide output widths; default four panels (2) for both
struct ColGroupEnv {
    int narrow = 2, wide = 2;
    static int lg(int v) { return v <= 0 ? 0 : v == 1 ? 2 : v == 2 ? 1 : v == 4 ? 2 : v == 8 ? 3 : v == 16 ? 4 : v == 32 ? 5 : 0; }
    ColGroupEnv() {
        const char* e = getenv("MHMR_COLGROUP");
        if (!e) return;
        narrow = wide = lg(atoi(e));
        if (const char* c = strchr(e, ',')) wide = lg(atoi(c + 1));
    }
};
}  // namespace
// MHMR_GEMM128=1 forces the 128x128 kernel everywhere (A/B measurements, bisecting)
static const bool g_force_gemm128 = getenv("MHMR_GEMM128") != nullptr;

// Split-k for a SHORT launch (gemm256.hip SPLITK): when the tiles of an [M, N] output fill at most half of the chip, cut K into nslices
// slices of ksplit k tiles (whole pairs; the last slice may be shorter) so that tiles x slices fill one round of CUs.  false = do not split.
// MHMR_SPLITK=0 switches it off (A/B measurements).
bool mhmr_splitk_plan(int M, int N, int K, int* ksplit, int* nslices) {
    static const bool on = !(getenv("MHMR_SPLITK") && atoi(getenv("MHMR_SPLITK")) == 0);
    if (!on || g_force_gemm128 || M <= 0 || M % 256 || N % 256 || K % 128) return false;
    const int ncu = mhmr_cu_count();
    const int tiles = (M / 256) * (N / 256), nt = K / 64;
    if (ncu <= 0 || tiles <= 0) return false;
    int S = ncu / tiles;
    if (S > 8) S = 8;
    if (S > nt / 4) S = nt / 4;                 // at least four k tiles per slice
    if (S < 2) return false;
    int ks = (nt + S - 1) / S;
    ks += ks & 1;                               // whole pairs of k tiles
    S = (nt + ks - 1) / ks;
    if (S < 2) return false;
    *ksplit = ks;
    *nslices = S;
    return true;
}

int mhmr_launch_gemm(const GemmArgs& g, int dtype, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.M % BM || g.N % BN || g.K % BK) return MHMR_ERR_BAD_SHAPE;
    if (g.lda % 8 || g.ldw % 8) return MHMR_ERR_BAD_SHAPE;
    if (g.epi == EPI_VT && (g.Tp % 64 || g.N % 64)) return MHMR_ERR_BAD_SHAPE;
    if (g.epi == EPI_VT && (g.img_rows > 0 ? g.img_rows : g.Tp) % BM && (g_force_gemm128 || !mhmr_gemm256_eligible(g))) return MHMR_ERR_BAD_SHAPE;   // (this kernel's row tile)
    if (g.img_rows > 0 && (g.img_rows % BM || g.M % g.img_rows || g.img_stride < g.img_rows || g.epi == EPI_PATCH)) return MHMR_ERR_BAD_SHAPE;
    if (g.a_k > 0 && !g.lo8 && (g.a_k % BK || (g.K != 2 * g.a_k && g.K != 3 * g.a_k) || g.ldw < g.K || (g.K == 3 * g.a_k && g.lda < 2 * g.a_k))) return MHMR_ERR_BAD_SHAPE;
    if (g.epi == EPI_OP16_QK && g.N % 128) return MHMR_ERR_BAD_SHAPE;      // Q | K halves are whole 64-column blocks
    if (g.lo8 && (g_force_gemm128 || !mhmr_gemm256_eligible(g))) return MHMR_ERR_BAD_SHAPE;      // fp8 low-half range: 256x256 kernel only
    if ((g.x8_off > 0 || g.ldx16 > 0) && !g.x16) return MHMR_ERR_BAD_ARG;
    if (g.ksplit > 0 && (g_force_gemm128 || !mhmr_gemm256_eligible(g))) return MHMR_ERR_BAD_SHAPE;   // split-k: 256x256 kernel only
    if (g.out2 && (g_force_gemm128 || !mhmr_gemm256_eligible(g))) return MHMR_ERR_BAD_SHAPE;         // merged qkv linear: 256x256 kernel only
    if (g.n_valid > 0 && g.n_valid != g.N && (g_force_gemm128 || !mhmr_gemm256_eligible(g))) return MHMR_ERR_BAD_SHAPE;   // masked output halves likewise
    if (g.x16 || g.pstats || g.rowstats) {       // LayerNorm fold: 256x256 kernel only
        if (g_force_gemm128 || !mhmr_gemm256_eligible(g)) return MHMR_ERR_BAD_SHAPE;
        if (g.rowstats && g.K < 256) return MHMR_ERR_BAD_SHAPE;      // (the strip DMA of a tile needs a barrier-separated k pair in front of it)
        if (g.rowstats && (g.bias || !g.colsum || !g.fbias || !(g.epi == EPI_OP16_QK || g.epi == EPI_VT || g.epi == EPI_OP16_GELU))) return MHMR_ERR_BAD_ARG;
        if ((g.x16 != nullptr) != (g.pstats != nullptr) || (g.x16 && g.epi != EPI_RESID)) return MHMR_ERR_BAD_ARG;
    }
    prof_begin(PROF_GEMM, s);
    int rc;
    if (!g_force_gemm128 && mhmr_gemm256_eligible(g)) {
        GemmArgs g2 = g;
        // The residual epilogue moves 512 KiB per tile and all CUs reach it together: the round's 134 MB burst runs at the HBM
        // floor while the MFMAs idle.  Starting the CU quarters 0/1/2/3 quarter-periods apart interleaves the bursts with the
        // other quarters' K loops (proj 0.375 -> 0.350 ms, fc2 0.967 -> 0.953 ms; bias/GELU/V^T epilogues measured no gain).
        // Period estimate: 1.4 us per K tile + 18 us epilogue.  MHMR_STAGGER_PCT scales it (0 = off).  With the token-row map (exact
        // tile rounds, 8 tiles per CU) the late quarters' tail costs what the interleaving saves: off there (round 3, one box:
        // 136.7 ms per step without, 137.3 with; profiles/r03_gemm_negative_results.txt).
        static const char* st = getenv("MHMR_STAGGER_PCT");
        const int pct = st ? atoi(st) : (g.img_rows > 0 ? 0 : 100);
        if (g.epi == EPI_RESID && (g.M / 256) * (g.N / 256) >= 1024 && pct > 0)      // only when every CU walks several tiles
            g2.stagger_ticks = (int)((g.K / 64 * 1.4 + 18.0) * 25.0 * pct / 100.0);
        // column-group order (gemm256.hip): MHMR_COLGROUP = "a[,b]" panels per group for N = 2048 (QK) [, N >= 4096 (fc1)]; 0 = off, 1 = 4
        static const ColGroupEnv cge;
        const int nbn = g.N / 256;
        g2.colgroup = g.ksplit > 0 ? 0 : nbn >= 16 ? cge.wide : cge.narrow;
        // image of a row tile = umulhi(tm, magic), exact while tm * tiles_per_image < 2^32; one tile per image: magic 0 = identity
        if (g.img_rows > 256) g2.img_magic = (unsigned)((1ull << 32) / (unsigned)(g.img_rows >> 8)) + 1u;
        rc = mhmr_launch_gemm256(g2, dtype, s);
    }
    else rc = dtype == MHMR_DT_F16 ? launch_dt<MHMR_DT_F16>(g, s) : launch_dt<MHMR_DT_BF16>(g, s);
    prof_end(PROF_GEMM, s, 2.0 * g.M * g.N * (g.lo8 ? 2.0 * g.a_k : (double)g.K));      // (an fp8 range covers a_k more k in a_k / 2 units)
    return rc;
}
