// 256x256x64 persistent "8-phase ping-pong" GEMM for the large ViT linears (same math and epilogues as gemm.hip).
//
//   workgroup = 8 waves (2 x 4) on one CU (1 block/CU, 2 waves per SIMD); LDS = 128 KiB DMA ring (2 K-tile buffers
//   even/odd x 4 half-tile slots P0, P1, Q0, Q1; 128 rows x 64 k, 16 KiB each) + 32 KiB epilogue staging.  A wave owns
//   64 P-rows in EACH P half and 32 Q-rows in EACH Q half, so its 128x64 output splits into 4 quadrants and every
//   half-tile slot is read in exactly one phase by all waves.  One pair of K tiles (t in the even buffer e, t+1 in the odd
//   buffer o) = 8 phases; the quadrant order is chosen so that the LDS reads alternate 8 / 4 per phase (a 12 / 4 / 8 / 0
//   pattern made the reading group outlast the other group's 16 MFMAs) without any extra fragment registers:
//
//     phase   ds_read (slot -> regs)   MFMA (16 x v_mfma_f32_16x16x32)   LDS-DMA issued (2 x glds16 / thread)
//     1       P0e -> PR (8)            acc[0][0] += PR x QA              P1o <- tile t+1
//     2       Q1e -> QB (4)            acc[0][1] += PR x QB              Q0e <- tile t+2
//     3       P1e -> PR (8)            acc[1][1] += PR x QB              P0e <- tile t+2
//     4       Q1o -> QB (4)            acc[1][0] += PR x QA              Q1e <- tile t+2
//     5       P0o -> PR (8)            acc[0][1] += PR x QB              P1e <- tile t+2
//     6       Q0o -> QA (4)            acc[0][0] += PR x QA              Q1o <- tile t+3
//     7       P1o -> PR (8)            acc[1][0] += PR x QA              P0o <- tile t+3
//     8       Q0e(t+2) -> QA (4)       acc[1][1] += PR x QB              Q0o <- tile t+3
//
//   Every phase is  [ds_reads ; DMA issue ; s_waitcnt vmcnt(10)] s_barrier [MFMAs] s_barrier.  The two wave groups
//   (wp = 0 / 1, one wave of each per SIMD) run ONE barrier apart, so while one group's 16 MFMAs (8 independent
//   accumulators: no dependent-issue stalls with only one computing wave per SIMD) occupy the SIMD's matrix pipe the
//   other group issues its LDS reads and DMA.  Each phase re-fills the slot that was read TWO phases earlier (WAR: two
//   barriers and the readers' lgkmcnt wait lie between) with the data that slot serves six phases later; vmcnt(10) after
//   each issue = "everything but the five newest half-tiles has landed" = the slot the NEXT phase reads (RAW = wait +
//   barrier).  The DMA queue is never drained inside the K loop.
//
//   Persistent: a block walks its output tiles; the K-tile indices t2, t3 that run past the end of one tile are the
//   first K tiles of the NEXT tile, so the ring stays full across tile boundaries and the next tile's operands stream
//   in underneath the epilogue (which transposes through its own LDS region and finishes with one vmcnt(0)).
#include <type_traits>
#include "mhmr_common.h"
#include "mhmr_internal.h"

namespace {

constexpr int HT = 16384;           // bytes per half-tile slot (128 rows x 128 B)
constexpr int BUF = 4 * HT;         // one K-tile buffer: P0 | P1 | Q0 | Q1
constexpr int SLOT_P0 = 0, SLOT_P1 = HT, SLOT_Q0 = 2 * HT, SLOT_Q1 = 3 * HT;
constexpr int STAGE_OFF = 2 * BUF;  // 8 waves x 4 KiB epilogue staging (the LayerNorm-foldable epilogues: 8 x 2 KiB + the 4 KiB strip)
constexpr int STRIP_OFF = STAGE_OFF + 8 * 2048;
constexpr int LDS_BYTES = 2 * BUF + 8 * 4096;

// An operand fragment of one 16-row sub-tile and one 64-k tile: two 16-byte halves (k steps 0 / 1 of the 16-bit MFMA).  The LO8 form keeps
// them as ONE 256-bit value -- the eight consecutive registers v_mfma_scale_f32_16x16x128_f8f6f4 takes as an operand.
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef int v4i_t __attribute__((ext_vector_type(4)));
template <typename V8, bool L8> struct Frag;
template <typename V8> struct Frag<V8, false> {
    V8 k[2];
    __device__ __forceinline__ void set(int ks, V8 x) { k[ks] = x; }
    __device__ __forceinline__ V8 get(int ks) const { return k[ks]; }
    // (never executed: the fp8 form exists in LO8 kernels only; keeps the generic k-pair lambda well-formed for both fragment types)
    __device__ __forceinline__ v8i_t raw() const {
        return __builtin_shufflevector(__builtin_bit_cast(v4i_t, k[0]), __builtin_bit_cast(v4i_t, k[1]), 0, 1, 2, 3, 4, 5, 6, 7);
    }
};
template <typename V8> struct Frag<V8, true> {
    v8i_t v;
    __device__ __forceinline__ void set(int ks, V8 x) {
        const v4i_t t = __builtin_bit_cast(v4i_t, x);
        if (ks == 0) v = __builtin_shufflevector(t, __builtin_shufflevector(v, v, 4, 5, 6, 7), 0, 1, 2, 3, 4, 5, 6, 7);
        else v = __builtin_shufflevector(__builtin_shufflevector(v, v, 0, 1, 2, 3), t, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    __device__ __forceinline__ V8 get(int ks) const {
        return __builtin_bit_cast(V8, ks == 0 ? __builtin_shufflevector(v, v, 0, 1, 2, 3) : __builtin_shufflevector(v, v, 4, 5, 6, 7));
    }
    __device__ __forceinline__ v8i_t raw() const { return v; }
};

// acc += A(16 x 128 fp8) . B(128 x 16 fp8), IN PLACE.  Inline assembly on purpose: this LLVM has no tied-accumulator ("mac") form of the
// builtin __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4 -- every call writes a FRESH destination quad, which doubles the live
// accumulators of the k loop (500-670 B of scratch per lane, reloaded between the MFMAs); "+v" ties destination and addend.  AFMT / BFMT: 0 =
// e4m3, 1 = e5m2 (cbsz / blgp); sa / sb: E8M0 scale bytes of the lane's 32-k block (byte 0 of the register).  Hazards: the eight
// accumulators of a phase are distinct and next touched a barrier later; the operands come from LDS reads (the compiler's waitcnt pass
// covers inline-asm uses) and the scale registers are loop invariants.  (Also tried: ONE k loop with a per-phase branch between this and the
// 16-bit builtin -- the accumulators became phi values and neither arm updated them in place; both arms as inline assembly -- 130-220 B of
// scratch; hence two k loops, the 16-bit one on the builtin.)
template <int AFMT, int BFMT>
__device__ __forceinline__ void mfma_fp8_inplace(f32x4& c, v8i_t a, v8i_t b, int sa, int sb) {
    static_assert((AFMT == 0 && BFMT == 1) || (AFMT == 1 && BFMT == 0), "weight e4m3 x activation e5m2");
    if constexpr (AFMT == 0)
        asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] blgp:1" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
    else
        asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:1" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
}

// sum over the 16 lanes of a DPP row, on the VALU (quad xor 1, quad xor 2, half-row mirror, row mirror); every lane ends with the sum
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);     // row_half_mirror
    v += dpp_f32<0x140>(v);     // row_mirror
    return v;
}

#ifdef MHMR_GEMM_STAMPS     // tools/gemm_timeline.py: per-workgroup, per-tile wall-clock stamps of wave 0 (debug build only, never in libmhmr.so)
__device__ unsigned long long* g_gemm_stamps;
__device__ int g_gemm_sametile;      // 1: every tile reads the operands of tile 0 (all L2 hits: the k loop without memory stalls); 2: ... of its first tile
#define GEMM_STAMP(r, i)                                                                                                    \
    do {                                                                                                                    \
        if (g_gemm_stamps && threadIdx.x == 0 && (r) < 64) g_gemm_stamps[((size_t)blockIdx.x * 64 + (r)) * 8 + (i)] = wall_clock64(); \
    } while (0)
#else
#define GEMM_STAMP(r, i)
#endif

// FOLD: a consumer of a folded LayerNorm (GemmArgs::rowstats; EPI_OP16_QK / EPI_VT / EPI_OP16_GELU only) -- a kernel of its own, so that
// the plain epilogues keep their 32-row staging passes and carry no run-time switches
// LO8 (GemmArgs::lo8): the k tiles behind a_k are fp8 tiles of 128 k.  A lane's two 16-byte fragments of a row (k steps 0 and 1 of a 16-bit
// tile) are exactly the 32 bytes v_mfma_scale_f32_16x16x128_f8f6f4 wants from it (which 32 of the row's 128 k a lane holds is a permutation
// of k applied to both operands alike), so the ring, the copies and the fragment reads are those of the 16-bit tiles: only the MFMA differs.
// SPLITK (GemmArgs::ksplit; EPI_F32 only): a SHORT launch -- fewer tiles than half the CUs: the residual linears of a batch of one -- walks
// VIRTUAL tiles (k slice, tile): slice s covers the k tiles [s * ksplit, min((s + 1) * ksplit, K / 64)) and writes its fp32 partial tile into
// slab s of `out` ([nslices][M][ldo]); the slices of one tile are summed in slice order by the kernel that follows (vit_misc.hip:
// splitk_resid_kernel, which also runs the residual epilogue and leaves the row statistics) -- deterministic, no inter-workgroup hand-off.
// NMASK (GemmArgs::n_valid; EPI_RESID and EPI_VT only): an output width that is a multiple of 128 but not of 256 (ViT-S: N = 384) runs as
// N = 512 with the weight rows (and every per-column array: bias, gamma, colsum, fbias) zero-padded by the caller; the 128-column half
// tiles at or behind n_valid are computed (zeros) and not stored -- 25 % of such a launch's matrix work is padding, at three times the
// rate the 128x128 kernel reaches on the same shape.
// QKV (GemmArgs::out2; EPI_OP16_QK only): ONE launch for the whole qkv linear of a short batch (N = 3C: Q | K | V).  Column tiles in front of
// split_col (= 2C) go to `out` (the Q | K rows the attention kernel reads, Q scaled: columns < qcols = C), the others -- V -- row-major to
// out2 [M, ldo2]; vit_misc.hip's vt_transpose_kernel makes the key-permuted V^T of them.  For a batch of one the Q | K projection (136 of 256
// CUs at 896^2) and the V projection (68) were two launches of half a round each.
template <int DT, int EPI, bool FOLD = false, bool LO8 = false, bool SPLITK = false, bool NMASK = false, bool QKV = false>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GemmArgs g) {
    static_assert(!QKV || (EPI == EPI_OP16_QK && !LO8 && !SPLITK && !NMASK), "merged qkv linear: the Q | K epilogue");
    static_assert(!SPLITK || (EPI == EPI_F32 && !FOLD && !LO8), "split-k: fp32 partial tiles only");
    static_assert(!NMASK || ((EPI == EPI_RESID || EPI == EPI_VT) && !LO8 && !SPLITK), "masked output halves: residual and V^T epilogues");
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V8 V8;
    typedef typename Op<DT>::V4 V4;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = w >> 2, wq = w & 3;
    const int g4 = lane >> 4, l15 = lane & 15;   // 16x16x32 fragments: row/col = lane & 15, k-group / row-quad = lane >> 4

    constexpr bool ROWMAJOR = (EPI != EPI_VT);
    constexpr bool FOLDABLE = FOLD;
    static_assert(!FOLD || EPI == EPI_OP16_QK || EPI == EPI_VT || EPI == EPI_OP16_GELU, "consumers of a folded LayerNorm");
    // P = first MFMA operand (D rows, 4 consecutive per accumulator quad), Q = second (D columns, one per lane)
    const T* Pm = ROWMAJOR ? (const T*)g.W : (const T*)g.A;
    const T* Qm = ROWMAJOR ? (const T*)g.A : (const T*)g.W;
    const int ldp = ROWMAJOR ? g.ldw : g.lda, ldq = ROWMAJOR ? g.lda : g.ldw;

    // ---- persistent tile walk: XCD x owns a contiguous run of G/8 tiles in every round (neighbouring tiles share an L2) ----
    const int nbn = g.N / 256, ntiles_real = (g.M / 256) * nbn, ntiles = SPLITK ? ntiles_real * g.nslices : ntiles_real;
    const int G = gridDim.x, b = blockIdx.x;
    const int first = (G & 7) == 0 ? (b & 7) * (G >> 3) + (b >> 3) : b;
    // Column-group order (wide outputs: QK N = 2048, fc1 N = 4096): an XCD keeps the SAME four weight column panels in every round
    // (2 MB of W stay in its L2 for the whole launch) and walks 8 row tiles of them per round, instead of 2 row tiles x all 16
    // column panels (every XCD re-fetched the whole 8 MB of fc1's W each round: 2.5 GB per launch for 0.28 GB of operands).
    // g.colgroup = log2 of the panels per group (2: four panels x 8 row tiles per XCD and round, 3: eight panels x 4 row tiles)
    const int cgl = g.colgroup, ncg = cgl > 0 ? nbn >> cgl : 0;      // column groups
    const bool colgroup = cgl >= 1 && cgl <= 5 && G == 256 && (nbn & ((1 << cgl) - 1)) == 0 && ncg > 1 && ncg <= 8 && (8 % ncg) == 0;

    // ---- DMA source addressing: one half-tile = 2 passes of 64 rows; lane-linear LDS image, swizzle on the source ----
    const int srow = tid >> 3;                                  // 0..63
    const int schunk = (tid & 7) ^ ((tid >> 4) & 7);
    const int nt = g.K / 64;
    const int nta = g.a_k > 0 ? g.a_k / 64 : nt;                     // k tiles of the activation operand (low-half weight pass: nt = 2 nta)
    auto ka = [&](int t) { return (!LO8 && t >= nta) ? t - nta : t; };       // (LO8: the activation row carries its own fp8 range, no wrap-around)
    // an operand position = a wave-uniform 64-bit TILE base (scalar registers; any M x K) + a 32-bit element offset inside the tile
    // (< 256 rows: fits for every ld below 2^22): half the vector registers of per-lane 64-bit pointers
    const uint32_t p_lane = (uint32_t)srow * (uint32_t)ldp + (uint32_t)(schunk * 8);
    const uint32_t q_lane = (uint32_t)srow * (uint32_t)ldq + (uint32_t)(schunk * 8);
    // p0 / q0: LOGICAL first index of the tile on the P / Q side; ar: PHYSICAL first row of its activation rows (GemmArgs::img_rows)
    // (the image of a row tile by a host-made reciprocal, GemmArgs::img_magic: a run-time integer division here costs a dozen live
    // vector registers in a kernel that has none to spare)
    auto tile_src = [&](int tix, const T*& pb, const T*& qb, int& p0, int& q0, int& ar, int& bimg, [[maybe_unused]] int& slc) {
        if constexpr (SPLITK) {          // virtual tile = slice * ntiles_real + tile (slice-major: neighbouring CUs share a slice's operand panels)
            slc = 0;
            while (tix >= ntiles_real) { tix -= ntiles_real; ++slc; }
        }
#ifdef MHMR_GEMM_STAMPS
        if (g_gemm_sametile == 1) tix = 0;
        if (g_gemm_sametile == 2) tix = (int)blockIdx.x;
#endif
        const int tn = tix % nbn, tm = tix / nbn;
        p0 = ROWMAJOR ? tn * 256 : tm * 256;
        q0 = ROWMAJOR ? tm * 256 : tn * 256;
        bimg = 0;
        ar = tm * 256;
        if (g.img_rows > 0) {
            bimg = g.img_magic ? (int)__umulhi((unsigned)tm, g.img_magic) : tm;        // (magic 0: one tile per image)
            ar = bimg * g.img_stride + (tm - bimg * (g.img_rows >> 8)) * 256;
        }
        pb = Pm + (size_t)(ROWMAJOR ? p0 : ar) * (size_t)ldp;
        qb = Qm + (size_t)(ROWMAJOR ? ar : q0) * (size_t)ldq;
    };
    // LO8 kernels: the lane-derived address terms (DMA lane offset, fragment offsets) are RE-MADE where they are used, from an mbcnt the
    // optimiser cannot hoist (a dozen VALU operations per phase, nothing beside 16 MFMAs): kept live across the k loops they were eleven
    // registers too many for the residual-epilogue variant -- spilled, and every reload's vmcnt(0) drained the DMA ring.
    auto fresh_lane = [&]() {
        int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(ln));
        return ln;
    };
    auto dma = [&](const T* base, uint32_t lane_off, int ld, int half, int kt, int lds_off) {
        if constexpr (LO8) {
            const int t2 = w * 64 + fresh_lane();
            lane_off = (uint32_t)(t2 >> 3) * (uint32_t)ld + (uint32_t)((((t2 & 7) ^ ((t2 >> 4) & 7))) * 8);
        }
        // (BYTE offsets in 32 bits: the copies then take the SGPR-base form of global_load_lds -- scalar base + zero-extended lane offset --
        // instead of a 64-bit vector shift-and-add per copy: element offsets are shifted AFTER the extension, which that form cannot express)
        const uint32_t o = (lane_off + (uint32_t)(128 * half) * (uint32_t)ld + (uint32_t)(kt * 64)) * (uint32_t)sizeof(T);
        char* d = smem + lds_off + w * 1024;
        glds16((const char*)base + o, d);
        glds16((const char*)base + (o + 64u * (uint32_t)sizeof(T) * (uint32_t)ld), d + 8192);
    };

    // ---- fragment read addressing (16-row sub-tiles; chunk = 4*ks + g4 within the 128-byte row) ----
    const int fsw = l15 >> 1;
    const int pr_off = (64 * wp + l15) * 128, q_off = (32 * wq + l15) * 128;
    int co[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) co[ks] = ((4 * ks + g4) ^ fsw) * 16;

    f32x4 acc[2][2][4][2];   // [P half][Q half][16-row P sub-tile][16-row Q sub-tile]
    typedef Frag<V8, LO8> Fr;
    Fr PR[4], QA[2], QB[2];

    auto rdP = [&](int buf, int slot) {
        if constexpr (LO8) {
            const int ln = fresh_lane(), a15 = ln & 15, sw = a15 >> 1, gg = ln >> 4;
            const int po = (64 * wp + a15) * 128, c0 = (gg ^ sw) * 16, c1 = ((4 + gg) ^ sw) * 16;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) PR[ps].set(ks, *(const V8*)(smem + buf * BUF + slot + po + ps * 2048 + (ks ? c1 : c0)));
        } else {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) PR[ps].set(ks, *(const V8*)(smem + buf * BUF + slot + pr_off + ps * 2048 + co[ks]));
        }
    };
    auto rdQ = [&](Fr (&Q)[2], int buf, int slot) {
        if constexpr (LO8) {
            const int ln = fresh_lane(), a15 = ln & 15, sw = a15 >> 1, gg = ln >> 4;
            const int qo = (32 * wq + a15) * 128, c0 = (gg ^ sw) * 16, c1 = ((4 + gg) ^ sw) * 16;
#pragma unroll
            for (int qs = 0; qs < 2; ++qs)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) Q[qs].set(ks, *(const V8*)(smem + buf * BUF + slot + qo + qs * 2048 + (ks ? c1 : c0)));
        } else {
#pragma unroll
            for (int qs = 0; qs < 2; ++qs)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) Q[qs].set(ks, *(const V8*)(smem + buf * BUF + slot + q_off + qs * 2048 + co[ks]));
        }
    };
    // E8M0 scales of the fp8 tiles, loop invariants pinned in registers (not re-made by a v_mov in front of an MFMA)
    [[maybe_unused]] int w8s = g.w8_scale * 0x01010101, a8s = 0x7F7F7F7F;
    if constexpr (LO8) asm volatile("" : "+v"(w8s), "+v"(a8s));
    // LOWC: std::true_type for a pair of fp8 k tiles -- a compile-time choice: the k loop exists twice (16-bit tiles, then fp8 tiles), so
    // that no run-time branch makes the accumulators phi values
    auto mma = [&](f32x4 (&c)[4][2], const Fr (&Q)[2], auto lowc) {
        constexpr bool LOW = decltype(lowc)::value;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (LOW) {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps)
#pragma unroll
                for (int qs = 0; qs < 2; ++qs) {
                    if constexpr (ROWMAJOR) mfma_fp8_inplace<0, 1>(c[ps][qs], PR[ps].raw(), Q[qs].raw(), w8s, a8s);
                    else mfma_fp8_inplace<1, 0>(c[ps][qs], PR[ps].raw(), Q[qs].raw(), a8s, w8s);
                }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int ps = 0; ps < 4; ++ps)
#pragma unroll
                    for (int qs = 0; qs < 2; ++qs) c[ps][qs] = Op<DT>::mfma16(PR[ps].get(ks), Q[qs].get(ks), c[ps][qs]);
        }
        __builtin_amdgcn_s_setprio(0);
    };
#define MHMR_SYNC()                          \
    __builtin_amdgcn_s_barrier();            \
    __builtin_amdgcn_sched_barrier(0)
#define MHMR_WAIT_DMA() asm volatile("s_waitcnt vmcnt(10)" ::: "memory")

    // The bias is the accumulators' initial value: the epilogue then has no bias loads at all (a global load there is followed by
    // its s_waitcnt vmcnt(0), which also waits for every store already issued -- 16 serialized HBM round trips per tile).  The
    // next tile's values are requested in the epilogue as each quadrant's accumulators are staged out, and land by its drain.
    auto acc_init = [&](int h, int j, int qs, int pp0, int qq0) {
        if constexpr (ROWMAJOR) {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps)
                acc[h][j][ps][qs] = g.bias ? *(const f32x4*)(g.bias + pp0 + 128 * h + 64 * wp + 16 * ps + 4 * g4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        } else {
            const float b1 = g.bias ? g.bias[qq0 + 128 * j + 32 * wq + 16 * qs + l15] : 0.f;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) acc[h][j][ps][qs] = (f32x4){b1, b1, b1, b1};
        }
    };

    const T *p_src, *q_src, *p_nxt, *q_nxt;
    int p0, q0, p0n, q0n, ar, arn, bimg, bimgn;
    [[maybe_unused]] int sl = 0, sln = 0;          // SPLITK: the k slice of this / the next virtual tile
    if (first >= ntiles) return;             // (never with the launcher's grid: G <= ntiles; keeps barrier counts trivially equal)
    if (g.stagger_ticks > 0) {       // CU quarters start 0/1/2/3 x stagger_ticks late so their epilogue bursts interleave (gemm.hip)
        const uint64_t t0 = wall_clock64(), dl = (uint64_t)g.stagger_ticks * (uint64_t)(first * 4 / G);
        while (wall_clock64() - t0 < dl) __builtin_amdgcn_s_sleep(32);
    }
    // this workgroup's tiles: one per full round, and the tiles of the last, partial round (none under the token-row map) go to the
    // LOWEST block indices (b < ntiles % G): the workgroups dispatched first are then the ones with the extra tile
    const int full = ntiles / G, nmine = full + (b < ntiles - full * G ? 1 : 0);
    // (walking the rounds of qk / v / fc1 downwards, so that they start with the rows the residual GEMM before them wrote last -- still in
    // the Infinity Cache -- measured +0.1 %: not kept)
    auto tile_of = [&](int r) {
        if (r >= full) return full * G + b;
        if (colgroup) {
            const int x = b & 7, slot = b >> 3;                       // XCD (speed only), slot 0..31 inside it
            const int row = r * (G / nbn) + (x / ncg) * (32 >> cgl) + (slot >> cgl), col = ((x % ncg) << cgl) + (slot & ((1 << cgl) - 1));
            return row * nbn + col;
        }
        return first + r * G;
    };
    tile_src(tile_of(0), p_src, q_src, p0, q0, ar, bimg, sl);
    // SPLITK: first k tile of this tile's slice (kb; kbn = the next tile's) and its number of k tiles (ntc); otherwise 0 / 0 / nt
    int kb = 0, kbn = 0, ntc = nt;
    if constexpr (SPLITK) { kb = sl * g.ksplit; ntc = nt - kb < g.ksplit ? nt - kb : g.ksplit; }

    // ---- prologue (first tile only): K tile 0 -> even buffer (all four halves), K tile 1 -> odd buffer (Q1, P0, Q0; P1 follows
    //      in phase 1 like in every later pair) ----
    {
        // (the activation side -- Q for row-major outputs -- wraps around at nta k tiles; a slice may start behind the wrap)
        const int k0p = !SPLITK ? 0 : ROWMAJOR ? kb : ka(kb), k1p = !SPLITK ? 1 : ROWMAJOR ? kb + 1 : ka(kb + 1);
        const int k0q = !SPLITK ? 0 : ROWMAJOR ? ka(kb) : kb, k1q = !SPLITK ? 1 : ROWMAJOR ? ka(kb + 1) : kb + 1;
        dma(q_src, q_lane, ldq, 0, k0q, SLOT_Q0);
        dma(p_src, p_lane, ldp, 0, k0p, SLOT_P0);
        dma(q_src, q_lane, ldq, 1, k0q, SLOT_Q1);
        dma(p_src, p_lane, ldp, 1, k0p, SLOT_P1);
        dma(q_src, q_lane, ldq, 1, k1q, BUF + SLOT_Q1);
        dma(p_src, p_lane, ldp, 0, k1p, BUF + SLOT_P0);
        dma(q_src, q_lane, ldq, 0, k1q, BUF + SLOT_Q0);
    }
#pragma unroll
    for (int a = 0; a < 8; ++a) acc_init(a >> 2, (a >> 1) & 1, a & 1, p0, q0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MHMR_SYNC();

    for (int r = 0; r < nmine; ++r) {
        GEMM_STAMP(r, 0);
        // Q0 of this tile's first K tile (landed and published by the previous pair's phase-8 wait + barrier, or by the prologue)
        rdQ(QA, 0, SLOT_Q0);
        if (wp == 1) { MHMR_SYNC(); }       // stagger: during the K loop group 1 runs one barrier behind group 0
        const bool has_next = r + 1 < nmine;
        if (has_next) tile_src(tile_of(r + 1), p_nxt, q_nxt, p0n, q0n, arn, bimgn, sln);
        else { p_nxt = p_src; q_nxt = q_src; p0n = p0; q0n = q0; arn = ar; bimgn = bimg; sln = sl; }     // last tile: harmless re-load into slots nobody reads
        if constexpr (SPLITK) kbn = sln * g.ksplit;
        auto kpair = [&](int t, auto lowc) {
            // K-tile indices past the end of this tile are the first K tiles of the next one
            // (t counts this tile's k tiles, 0 .. ntc - 1; the copies take ABSOLUTE k-tile indices: + kb / kbn, both 0 unless SPLITK)
            const bool wrap = t + 2 >= ntc;
            const T* p2 = wrap ? p_nxt : p_src;
            const T* q2 = wrap ? q_nxt : q_src;
            const int t1 = kb + t + 1, t2 = wrap ? (has_next ? kbn : kb + ntc - 1) : kb + t + 2, t3 = wrap ? (has_next ? kbn + 1 : kb + ntc - 1) : kb + t + 3;
            // the activation side (Q for row-major outputs, P for V^T) wraps around at nta k tiles (low-half weight pass)
            const int t1p = ROWMAJOR ? t1 : ka(t1), t2p = ROWMAJOR ? t2 : ka(t2), t3p = ROWMAJOR ? t3 : ka(t3);
            const int t2q = ROWMAJOR ? ka(t2) : t2, t3q = ROWMAJOR ? ka(t3) : t3;
            if constexpr (FOLDABLE) {
                // LayerNorm fold: this tile's column sums / folded bias (256 output columns) and row statistics (256 rows) travel by
                // LDS-DMA into the strip while the last pair of k tiles is computed: 1 KiB per wave 0..3, sixteen ring copies younger
                // than it by the end of the pair, so the phases' counted waits cover its landing (and the barriers of the k loop
                // separate it from the previous tile's epilogue, which read the strip: K >= 256, mhmr_launch_gemm)
                if (wrap && w < 4) {
                    const int nbase = ROWMAJOR ? p0 : q0;
                    const float* sp = w == 0 ? g.colsum + nbase : w == 1 ? g.fbias + nbase : g.rowstats + 2 * (size_t)ar + (w - 2) * 256;
                    // (the lane index is recomputed here: a register kept live across the k loop for it was spilled, and the reload's
                    // vmcnt(0) drained the DMA ring once per tile)
                    const int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                    glds16(sp + ln * 4, smem + STRIP_OFF + w * 1024);
                }
            }
            // phase 1
            rdP(0, SLOT_P0);
            dma(p_src, p_lane, ldp, 1, t1p, BUF + SLOT_P1); MHMR_WAIT_DMA();
            MHMR_SYNC(); mma(acc[0][0], QA, lowc); MHMR_SYNC();
            // phase 2
            rdQ(QB, 0, SLOT_Q1);
            dma(q2, q_lane, ldq, 0, t2q, SLOT_Q0); MHMR_WAIT_DMA();
            MHMR_SYNC(); mma(acc[0][1], QB, lowc); MHMR_SYNC();
            // phase 3
            rdP(0, SLOT_P1);
            dma(p2, p_lane, ldp, 0, t2p, SLOT_P0); MHMR_WAIT_DMA();
            MHMR_SYNC(); mma(acc[1][1], QB, lowc); MHMR_SYNC();
            // phase 4
            rdQ(QB, 1, SLOT_Q1);
            dma(q2, q_lane, ldq, 1, t2q, SLOT_Q1); MHMR_WAIT_DMA();
            MHMR_SYNC(); mma(acc[1][0], QA, lowc); MHMR_SYNC();
            // phase 5
            rdP(1, SLOT_P0);
            dma(p2, p_lane, ldp, 1, t2p, SLOT_P1); MHMR_WAIT_DMA();
            MHMR_SYNC(); mma(acc[0][1], QB, lowc); MHMR_SYNC();
            // phase 6
            rdQ(QA, 1, SLOT_Q0);
            dma(q2, q_lane, ldq, 1, t3q, BUF + SLOT_Q1); MHMR_WAIT_DMA();
            MHMR_SYNC(); mma(acc[0][0], QA, lowc); MHMR_SYNC();
            // phase 7
            rdP(1, SLOT_P1);
            dma(p2, p_lane, ldp, 0, t3p, BUF + SLOT_P0); MHMR_WAIT_DMA();
            MHMR_SYNC(); mma(acc[1][0], QA, lowc); MHMR_SYNC();
            // phase 8 (the next pair's Q0; at the end of a tile it is read after the epilogue instead, so that QA's registers are
            // free for the epilogue)
            if (!wrap) rdQ(QA, 0, SLOT_Q0);
            dma(q2, q_lane, ldq, 0, t3q, BUF + SLOT_Q0); MHMR_WAIT_DMA();
            MHMR_SYNC(); mma(acc[1][1], QB, lowc); MHMR_SYNC();
        };
        // the 16-bit k tiles, then (LO8) the fp8 k tiles behind a_k: whole pairs either way (a_k % 256 == 0)
        for (int t = 0; t < (LO8 ? nta : ntc); t += 2) kpair(t, std::false_type{});
        if constexpr (LO8)
            for (int t = nta; t < nt; t += 2) kpair(t, std::true_type{});

        GEMM_STAMP(r, 1);
        if (wp == 0) { MHMR_SYNC(); }       // re-align: both groups run the memory-bound epilogue together
        GEMM_STAMP(r, 2);
        // ---- epilogue (wave-private staging; the next tile's DMA is in flight underneath) ----
        // per quadrant, the wave's [32 Q-rows][64 P-cols] block is transposed through LDS so that global accesses are
        // whole contiguous row segments (a direct store from the MFMA layout camps on one memory channel).
        char* wl = smem + STAGE_OFF + w * 4096;
        // (the folded-LayerNorm epilogues and the residual epilogue with its extra outputs need more address arithmetic than the others: with
        // the lane index recomputed HERE, none of it is hoisted in front of the tile loop and kept alive -- i.e. spilled -- across the k loop)
        constexpr bool RELANE = FOLD || EPI == EPI_RESID;
        int lane_e = 0;
        if constexpr (RELANE) {
            lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            asm volatile("" : "+v"(lane_e));
        }
        const int lane_outer = lane, l15_outer = l15, g4_outer = g4;
        {
        const int lane = RELANE ? lane_e : lane_outer;
        const int l15 = RELANE ? (lane_e & 15) : l15_outer, g4 = RELANE ? (lane_e >> 4) : g4_outer;
        constexpr bool OUT16 = (EPI == EPI_OP16 || EPI == EPI_OP16_GELU || EPI == EPI_OP16_RELU || EPI == EPI_VT || EPI == EPI_OP16_QK);
        if constexpr (EPI == EPI_RESID) {
            // out32 += gamma * (acc + bias), in place.  8 passes (h, j, qs) of [16 rows][64 cols] fp32 per wave; a pass that
            // loads its residual lines only when it needs them pays one HBM round trip per pass (8 in series: 22-25 us per
            // tile).  The residual of pass s + 1 is requested before pass s is staged (registers: the operand fragments, dead
            // between K loops): proj 0.414 -> 0.365 ms, fc2 1.01 -> 0.975 ms.  Deeper prefetch (2 passes, or growing as staged
            // passes free their accumulators) measured no better: with every CU in its epilogue at once the 134 MB round is at
            // the HBM floor.  Also measured and rejected (round 2): the residual tile as the accumulators' INITIAL value (LayerScale
            // folded into W, epilogue write-only) -- the 32 fragment-shaped loads per lane (16 rows x 64 B each) cost more than the
            // read-modify-write they replace: proj 0.351 -> 0.372 ms, fc2 0.977 -> 1.006 ms (f16).
            constexpr int RD = 1;          // (RD = 2, 247 VGPRs, re-measured in round 3 with the staggered quarters in place: no change; RD = 3 spills)
            const int c = lane & 15;
            // 32-bit byte offsets from the uniform base (eligibility guarantees M * ldo * 4 < 2^32): one VGPR per address
            const uint32_t off0 = ((uint32_t)(ar + 32 * wq + (lane >> 4)) * (uint32_t)g.ldo + (uint32_t)(p0 + 64 * wp + 4 * c)) * 4u;
            const uint32_t rstep = (uint32_t)g.ldo * 16u;          // 4 rows
            // NMASK: the 128-column half h = 1 of the LAST column tile lies behind n_valid (n_valid = N - 128).  Its passes run like any other
            // -- their residual loads are pointed at half 0's columns (valid memory: no conditional loads, which cost this kernel 392 B of
            // scratch) -- and only their stores are skipped (wave-uniform branch).
            [[maybe_unused]] const int nvalid = NMASK ? g.n_valid : g.N;
            [[maybe_unused]] const bool dead1 = NMASK && p0 + 128 >= nvalid;
            auto live = [&](int sidx) { return !NMASK || !(dead1 && (sidx >> 2) == 1); };
            auto rptr = [&](int sidx, int it) {
                const int h = sidx >> 2, j = (sidx >> 1) & 1, qs = sidx & 1;
                const uint32_t hoff = NMASK ? (live(sidx) ? (uint32_t)(512 * h) : 0u) : (uint32_t)(512 * h);
                const uint32_t off = off0 + (uint32_t)(32 * j + 4 * qs + it) * rstep + hoff;
                return (f32x4*)((char*)g.out + off);
            };
            f32x4 r[8][4];
#pragma unroll
            for (int sidx = 0; sidx < RD; ++sidx)
#pragma unroll
                for (int it = 0; it < 4; ++it) r[sidx][it] = *rptr(sidx, it);
            f32x4 gm[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) gm[h] = *(const f32x4*)(g.gamma + p0 + 128 * h + 64 * wp + 4 * c);
#pragma unroll
            for (int sidx = 0; sidx < 8; ++sidx) {
                const int h = sidx >> 2, j = (sidx >> 1) & 1, qs = sidx & 1;
                if (sidx + RD < 8) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) r[sidx + RD][it] = *rptr(sidx + RD, it);
                }
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) *(f32x4*)(wl + l15 * 256 + (((4 * ps + g4) ^ l15) * 16)) = acc[h][j][ps][qs];
                acc_init(h, j, qs, p0n, q0n);
                if (!live(sidx)) continue;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = 4 * it + (lane >> 4);
                    const f32x4 v = *(const f32x4*)(wl + row * 256 + ((c ^ row) * 16));
                    const f32x4 nv = r[sidx][it] + gm[h] * v;
                    f32x4* rp = rptr(sidx, it);
                    *rp = nv;
                    if (g.x16 != nullptr) {
                        // LayerNorm fold, producer side: the 16-bit copy of the new residual values (the next linear's A operand) and this
                        // 64-column block's (sum, sum of squares) per row.  16 lanes (c = 0..15) hold one row's 64 columns.
                        const uint32_t off = (uint32_t)((char*)rp - (char*)g.out);
                        V4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (T)nv[e];
                        // x16 rows may be wider than the output's (GemmArgs::ldx16: a row also carries its bf8 copy at byte x8_off): the
                        // same walk as rptr's with the x16 pitch -- no division (with ldx16 = ldo this is off >> 1)
                        if (g.ldx16 > 0) {
                            const uint32_t prow = (uint32_t)(ar + 32 * wq + (lane >> 4)) + (uint32_t)(4 * (32 * j + 4 * qs + it));
                            const uint32_t pcol = (uint32_t)(p0 + 64 * wp + 4 * c + 128 * h);
                            char* xrow = (char*)g.x16 + (size_t)prow * ((uint32_t)g.ldx16 * 2u);
                            *(V4*)(xrow + pcol * 2u) = o;
                            if (g.x8_off > 0) *(uint32_t*)(xrow + (uint32_t)g.x8_off + pcol) = pack_bf8x4(nv[0], nv[1], nv[2], nv[3]);
                        } else {
                            *(V4*)((char*)g.x16 + (off >> 1)) = o;
                        }
                        float s1 = (nv[0] + nv[1]) + (nv[2] + nv[3]);
                        float s2 = (nv[0] * nv[0] + nv[1] * nv[1]) + (nv[2] * nv[2] + nv[3] * nv[3]);
                        s1 = row16_sum(s1);
                        s2 = row16_sum(s2);
                        if (c == 0) {
                            const int prow = ar + 128 * j + 32 * wq + 16 * qs + row;
                            const int slot = (p0 + 128 * h + 64 * wp) >> 6;
                            *(f32x2*)(g.pstats + ((size_t)prow * (nvalid >> 6) + slot) * 2) = (f32x2){s1, s2};
                        }
                    }
                }
            }
        } else {
        // LayerNorm fold, consumer side.  Row side = the activation rows (m), column side = the output columns (n).  The tile's column
        // values (colsum_n, fbias_n) and row values (mean_m, rstd_m) were DMA'd into the strip (smem + STRIP_OFF) during the last k pair:
        // [colsum 256 | fbias 256 | (mean, rstd) x 256 rows] floats;  out = acc * rstd_m + t_m * colsum_n + fbias_n,  t_m = -mean_m rstd_m.
        // The foldable epilogues stage 16-row passes (2 KiB per wave instead of 4), which leaves the strip its own 4 KiB of LDS: it is
        // read where it is used (a register-resident strip handed around by ds_bpermute cost 144 LDS-crossbar operations per tile and
        // wave, and the residual epilogue's 16-lane sums as ds_bpermute chains another 256: together they ate the whole LayerNorm pass).
        constexpr bool fold = FOLD;
        [[maybe_unused]] const float* strip = (const float*)(smem + STRIP_OFF);
        constexpr int NPASS = FOLDABLE ? 2 : 1;              // staging passes per (h, j): 16 or 32 Q rows
        if constexpr (FOLDABLE) wl = smem + STAGE_OFF + w * 2048;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int pb = p0 + 128 * h + 64 * wp;      // first P index of the block (n for row-major, LOGICAL token row for V^T)
                const int qb = q0 + 128 * j + 32 * wq;      // first Q index (logical m for row-major, channel for V^T)
                const int qphys = ar + 128 * j + 32 * wq;   // row-major outputs: physical row of qb
                if constexpr (NMASK && EPI == EPI_VT) {
                    // V^T: the Q side is the weight (output channel): half j of this tile lies behind n_valid -> nothing to store
                    if (q0 + 128 * j >= g.n_valid) {
                        acc_init(h, j, 0, p0n, q0n);
                        acc_init(h, j, 1, p0n, q0n);
                        continue;
                    }
                }
                if constexpr (OUT16) {
                    const float qscale = (EPI == EPI_OP16_QK && pb < (QKV ? g.qcols : (g.N >> 1))) ? MHMR_ATTN_QSCALE : 1.f;   // 64 columns: all Q or all K
#pragma unroll
                    for (int qh = 0; qh < NPASS; ++qh) {
                        // Q-side values of this pass's rows (fold): one position per lane and qs
                        [[maybe_unused]] float qa2[2] = {0.f, 0.f}, qb2[2] = {0.f, 0.f};
                        if constexpr (FOLDABLE) {
                            if (fold) {
                                const int qpos = 128 * j + 32 * wq + 16 * qh + l15;
                                if constexpr (ROWMAJOR) { const f32x2 mr = *(const f32x2*)(strip + 512 + 2 * qpos); qa2[qh] = mr[1]; qb2[qh] = -mr[0] * mr[1]; }
                                else { qa2[qh] = strip[qpos]; qb2[qh] = strip[256 + qpos]; }
                            }
                        }
#pragma unroll
                        for (int ps = 0; ps < 4; ++ps) {
                            const int pc = 16 * ps + 4 * g4;    // lane owns P columns pc..pc+3 of Q rows 16*qs + l15
                            [[maybe_unused]] f32x4 pa4 = {0.f, 0.f, 0.f, 0.f}, pb4 = {0.f, 0.f, 0.f, 0.f};      // P-side values of those four positions
                            if constexpr (FOLDABLE) {
                                if (fold) {
                                    const int ppos = 128 * h + 64 * wp + pc;
                                    if constexpr (ROWMAJOR) { pa4 = *(const f32x4*)(strip + ppos); pb4 = *(const f32x4*)(strip + 256 + ppos); }     // colsum, fbias
                                    else {
                                        const f32x4 m0 = *(const f32x4*)(strip + 512 + 2 * ppos), m1 = *(const f32x4*)(strip + 512 + 2 * ppos + 4);     // (mean, rstd) x 4 rows
                                        pa4 = (f32x4){m0[1], m0[3], m1[1], m1[3]};
                                        pb4 = (f32x4){-m0[0] * m0[1], -m0[2] * m0[3], -m1[0] * m1[1], -m1[2] * m1[3]};
                                    }
                                }
                            }
                            const int gs = ((g4 & 1) << 1) | (g4 >> 1);
                            const int pos = ROWMAJOR ? pc : 16 * ps + 4 * gs;          // V^T: swap key bits 2 and 3
#pragma unroll
                            for (int qs = FOLDABLE ? qh : 0; qs < (FOLDABLE ? qh + 1 : 2); ++qs) {
                                const int qr = FOLDABLE ? l15 : 16 * qs + l15;          // row of the staging image
                                V4 o;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    float v = acc[h][j][ps][qs][e];
                                    if constexpr (FOLDABLE) {
                                        if (fold) {
                                            if constexpr (ROWMAJOR) v = __builtin_fmaf(v, qa2[qs], __builtin_fmaf(qb2[qs], pa4[e], pb4[e]));   // rows = Q side
                                            else v = __builtin_fmaf(v, pa4[e], __builtin_fmaf(pb4[e], qa2[qs], qb2[qs]));                      // rows = P side
                                        }
                                    }
                                    if constexpr (EPI == EPI_OP16_GELU) v = gelu_fast(v);
                                    if constexpr (EPI == EPI_OP16_RELU) v = fmaxf(v, 0.f);
                                    if constexpr (EPI == EPI_OP16_QK) v *= qscale;
                                    o[e] = (T)v;
                                }
                                *(V4*)(wl + qr * 128 + (((pos >> 2) ^ ((qr & 7) << 1)) * 8)) = o;
                            }
                        }
                        if constexpr (FOLDABLE) acc_init(h, j, qh, p0n, q0n);
                        else { acc_init(h, j, 0, p0n, q0n); acc_init(h, j, 1, p0n, q0n); }
#pragma unroll
                        for (int it = 0; it < (FOLDABLE ? 2 : 4); ++it) {
                            const int srow = 8 * it + (lane >> 3), c16 = lane & 7;      // row of the staging image
                            const int row = (FOLDABLE ? 16 * qh : 0) + srow;            // Q row of the (h, j) block
                            const u32x4 v = *(const u32x4*)(wl + srow * 128 + ((c16 ^ (srow & 7)) * 16));
                            if constexpr (ROWMAJOR && QKV) {
                                const bool second = pb >= g.split_col;          // (wave-uniform: a 64-column block is all Q | K or all V)
                                T* ob = second ? (T*)g.out2 : (T*)g.out;
                                const int ld = second ? g.ldo2 : g.ldo, col = second ? pb - g.split_col : pb;
                                *(u32x4*)(ob + (size_t)(qphys + row) * ld + col + c16 * 8) = v;
                            } else if constexpr (ROWMAJOR) {
                                *(u32x4*)((T*)g.out + (size_t)(qphys + row) * g.ldo + pb + c16 * 8) = v;
                            } else {
                                // image and token-in-image of the block's first (logical) token row
                                const int n = qb + row, bi = g.img_rows > 0 ? bimg : pb / g.Tp, tl = pb - bi * (g.img_rows > 0 ? g.img_rows : g.Tp);
                                *(u32x4*)((T*)g.out + ((size_t)(bi * g.H + (n >> 6)) * 64 + (n & 63)) * g.Tp + tl + c16 * 8) = v;
                            }
                        }
                    }
                } else {
                    const int c = lane & 15;
                    const int n = pb + 4 * c;
#pragma unroll
                    for (int qs = 0; qs < 2; ++qs) {      // 16 Q-rows x 64 P-cols x 4 B = 4 KiB per pass
#pragma unroll
                        for (int ps = 0; ps < 4; ++ps) {
                            const int ch = 4 * ps + g4;
                            *(f32x4*)(wl + l15 * 256 + ((ch ^ l15) * 16)) = acc[h][j][ps][qs];
                        }
                        acc_init(h, j, qs, p0n, q0n);
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int row = 4 * it + (lane >> 4);
                            f32x4 v = *(const f32x4*)(wl + row * 256 + ((c ^ row) * 16));
                            const int m = qb + 16 * qs + row;
                            if constexpr (EPI == EPI_PATCH) {
                                if (m < g.Mvalid) {
                                    const int bi = m / g.Np, n_in = m - bi * g.Np;
                                    v += *(const f32x4*)(g.pos + (size_t)(1 + n_in) * g.N + n);
                                    *(f32x4*)((float*)g.out + ((size_t)bi * g.Tp + n_in) * g.ldo + n) = v;      // class token LAST: patch n at row n
                                }
                            } else {
                                if constexpr (SPLITK) *(f32x4*)((float*)g.out + ((size_t)sl * g.M + (size_t)(qphys + 16 * qs + row)) * g.ldo + n) = v;
                                else *(f32x4*)((float*)g.out + (size_t)(qphys + 16 * qs + row) * g.ldo + n) = v;
                            }
                        }
                    }
                }
            }
        }
        }
        }
        // drain: the epilogue's stores and the (long landed) next-tile DMA; re-establishes exact vmcnt accounting
        GEMM_STAMP(r, 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GEMM_STAMP(r, 4);
        p_src = p_nxt; q_src = q_nxt; p0 = p0n; q0 = q0n; ar = arn; bimg = bimgn;
        if constexpr (SPLITK) { sl = sln; kb = kbn; ntc = nt - kb < g.ksplit ? nt - kb : g.ksplit; }
    }
#undef MHMR_SYNC
#undef MHMR_WAIT_DMA
}

#ifdef MHMR_GEMM_STAMPS
}  // namespace
extern "C" int mhmr_debug_gemm_stamps(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamps), &p, sizeof(p)); }
extern "C" int mhmr_debug_gemm_sametile(int v) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_sametile), &v, sizeof(v)); }
namespace {
#endif

template <int DT>
int launch256_dt(const GemmArgs& g, hipStream_t s) {
    const int ntiles = (g.M / 256) * (g.N / 256) * (g.ksplit > 0 ? g.nslices : 1);      // (split-k: virtual tiles)
    const int ncu = mhmr_cu_count();                   // of the CURRENT device
    if (ncu <= 0) return MHMR_ERR_BAD_ARG;
    const int grid = ntiles < ncu ? ntiles : ncu;      // one persistent block per CU
    // 160 KiB of dynamic LDS: the attribute is per device (DeviceOnce, mhmr_internal.h)
#define MHMR_GEMM_LAUNCH11(E, F, L8, SK, NM, QV)                                                               \
    {                                                                                                          \
        static DeviceOnce once;                                                                                \
        int dev = 0;                                                                                           \
        const int need = once.need(&dev);                                                                      \
        if (need == -2) return MHMR_ERR_BAD_ARG;                                                               \
        if (need >= 0) {                                                                                       \
            hipError_t e = hipFuncSetAttribute((const void*)gemm256_kernel<DT, E, F, L8, SK, NM, QV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               LDS_BYTES);                                                     \
            if (e != hipSuccess) return (int)e;                                                                \
            once.mark(dev);                                                                                    \
        }                                                                                                      \
        mhmr_launch_kernel(gemm256_kernel<DT, E, F, L8, SK, NM, QV>, dim3(grid), dim3(512), LDS_BYTES, s, g);  \
    }
#define MHMR_GEMM_LAUNCH10(E, F, L8, SK, NM) MHMR_GEMM_LAUNCH11(E, F, L8, SK, NM, false)
#define MHMR_GEMM_LAUNCH9(E, F, L8, SK) MHMR_GEMM_LAUNCH10(E, F, L8, SK, false)
#define MHMR_GEMM_LAUNCH8(E, F, L8) MHMR_GEMM_LAUNCH9(E, F, L8, false)
#define MHMR_GEMM_LAUNCH(E, F) MHMR_GEMM_LAUNCH8(E, F, false)
#define MHMR_GEMM_CASE(E) \
    case E: MHMR_GEMM_LAUNCH(E, false) break;
    if (g.ksplit > 0) {                     // split-k: fp32 partial tiles of a short launch (eligibility: mhmr_gemm256_eligible)
        MHMR_GEMM_LAUNCH9(EPI_F32, false, false, true)
        MHMR_CHECK_LAUNCH();
        return 0;
    }
    if (g.out2) {                                   // merged qkv linear (plain or as the consumer of a folded LayerNorm)
        if (g.epi != EPI_OP16_QK) return MHMR_ERR_BAD_ARG;
        if (g.rowstats) MHMR_GEMM_LAUNCH11(EPI_OP16_QK, true, false, false, false, true)
        else MHMR_GEMM_LAUNCH11(EPI_OP16_QK, false, false, false, false, true)
        MHMR_CHECK_LAUNCH();
        return 0;
    }
    if (g.n_valid > 0 && g.n_valid != g.N) {        // masked output halves (N = 384 as 512): residual and V^T epilogues
        if (g.epi == EPI_RESID && !g.rowstats) MHMR_GEMM_LAUNCH10(EPI_RESID, false, false, false, true)
        else if (g.epi == EPI_VT && g.rowstats) MHMR_GEMM_LAUNCH10(EPI_VT, true, false, false, true)
        else if (g.epi == EPI_VT) MHMR_GEMM_LAUNCH10(EPI_VT, false, false, false, true)
        else return MHMR_ERR_BAD_ARG;
        MHMR_CHECK_LAUNCH();
        return 0;
    }
    if (g.lo8) {                            // fp8 low-half range: the V^T projection (plain or folded) and the residual projection
        if (g.epi == EPI_VT && g.rowstats != nullptr) MHMR_GEMM_LAUNCH8(EPI_VT, true, true)
        else if (g.epi == EPI_VT) MHMR_GEMM_LAUNCH8(EPI_VT, false, true)
        else if (g.epi == EPI_RESID) MHMR_GEMM_LAUNCH8(EPI_RESID, false, true)
        else return MHMR_ERR_BAD_ARG;
        MHMR_CHECK_LAUNCH();
        return 0;
    }
    if (g.rowstats != nullptr) {            // consumers of a folded LayerNorm
        switch (g.epi) {
            case EPI_OP16_GELU: MHMR_GEMM_LAUNCH(EPI_OP16_GELU, true) break;
            case EPI_VT: MHMR_GEMM_LAUNCH(EPI_VT, true) break;
            case EPI_OP16_QK: MHMR_GEMM_LAUNCH(EPI_OP16_QK, true) break;
            default: return MHMR_ERR_BAD_ARG;
        }
        MHMR_CHECK_LAUNCH();
        return 0;
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
#undef MHMR_GEMM_LAUNCH
#undef MHMR_GEMM_LAUNCH8
#undef MHMR_GEMM_LAUNCH9
#undef MHMR_GEMM_LAUNCH10
#undef MHMR_GEMM_LAUNCH11
    MHMR_CHECK_LAUNCH();
    return 0;
}

}  // namespace

bool mhmr_gemm256_eligible(const GemmArgs& g) {
    const uint64_t mphys = g.img_rows > 0 ? (uint64_t)(g.M / g.img_rows) * (uint64_t)g.img_stride : (uint64_t)g.M;
    if (g.epi == EPI_RESID && mphys * (uint64_t)g.ldo * 4u >= (1ull << 32)) return false;   // 32-bit residual offsets
    if (g.img_rows > 0 && (g.img_rows % 256 || g.M % g.img_rows || g.epi == EPI_PATCH)) return false;   // a tile never straddles two images
    if (g.lo8) {       // fp8 low-half range: K = a_k + a_k / 2 (the low range as bytes), whole PAIRS of 128-deep fp8 tiles
        if (g.a_k <= 0 || g.a_k % 256 || g.K != g.a_k + g.a_k / 2 || g.lda < g.K || g.ldw < g.K || !(g.epi == EPI_VT || g.epi == EPI_RESID)) return false;
    } else if (g.a_k > 0 && (g.a_k % 128 || (g.K != 2 * g.a_k && g.K != 3 * g.a_k))) return false;
    if (g.lda >= (1 << 22) || g.ldw >= (1 << 22)) return false;      // 32-bit operand offsets INSIDE a 256-row tile (tile bases are 64-bit)
    if (g.n_valid > 0 && g.n_valid != g.N) {   // masked output halves: the last 128 columns of the padded width are dead
        if (g.n_valid % 128 || g.n_valid > g.N || g.N - g.n_valid != 128 || g.lo8 || g.ksplit > 0 || !(g.epi == EPI_RESID || g.epi == EPI_VT)) return false;
    }
    if (g.out2) {              // merged qkv linear: whole 64-column blocks on either side of split_col, Q columns in front of it
        if (g.epi != EPI_OP16_QK || g.split_col <= 0 || g.split_col % 64 || g.split_col >= g.N || g.qcols < 0 || g.qcols % 64 || g.qcols > g.split_col ||
            g.ldo2 < g.N - g.split_col || g.lo8 || g.ksplit > 0 || g.n_valid > 0)
            return false;
    }
    if (g.ksplit > 0) {        // split-k: whole PAIRS of k tiles per slice, the last slice included; fp32 partials, no bias, no row map
        const int nt = g.K / 64;
        if (g.epi != EPI_F32 || g.bias || g.lo8 || g.img_rows > 0 || g.ksplit % 2 || g.nslices < 2 || g.nslices > 16 ||
            (g.nslices - 1) * g.ksplit >= nt || g.nslices * g.ksplit < nt || g.ldo != g.N)
            return false;
    }
    return g.M % 256 == 0 && g.N % 256 == 0 && g.K % 128 == 0 && (g.epi != EPI_VT || g.Tp % 64 == 0);
}

int mhmr_launch_gemm256(const GemmArgs& g, int dtype, hipStream_t s) {
    return dtype == MHMR_DT_F16 ? launch256_dt<MHMR_DT_F16>(g, s) : launch256_dt<MHMR_DT_BF16>(g, s);
}
