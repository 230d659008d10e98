// fp32 flash attention of the "f16x3" precision mode: the attention of a checkpoint whose softmax logits are
// too steep for 16-bit Q / K / P (pack-time statistic vit.logit_gain).  Every product runs on v_mfma_f32_16x16x4_f32 -- exact fp32
// multiply-adds, the f32 VECTOR rate (157 TF/s dense): ~1/16 of the 16-bit kernel's rate, which is what the mode costs.
//
//   qkv   [B*Tp, 3C] fp32 = (Q | K | V) rows as the QKV linear leaves them (bias included, NOT scaled); head h = columns h*64 .. +63
//   out   op16 PAIR [B*Tp, 2C]: column h*64 + d holds hi = op16(o), column C + h*64 + d holds lo = op16(o - hi): the A operand of the
//         three-product output projection (GemmArgs::a_k with K = 3 a_k)
//
// One wave = 16 queries, one workgroup = 4 waves = 64 consecutive query rows of one (image, head); keys in tiles of 16, K / V rows read
// straight from global memory into MFMA operand registers (the four waves of a workgroup and the workgroups of a head share them
// through L1 / L2; no LDS).  Fragment geometry (c = lane & 15, g = lane >> 4):
//   S^T = K Q^T   A = K  [key c ][d = 16 g + s]   B = Q^T [d = 16 g + s][query c]   s = 0..15      (any d order is a dot product)
//                 D: lane holds S^T[key 4 g + i][query c], i = 0..3            -> a lane owns ONE query, four of the tile's keys
//   O^T = V^T P^T A = V^T[d = 4 c + db][key 4 g + i]   B = P^T[key 4 g + i][query c] = the lane's own p_i     (i = 0..3, db = 0..3)
//                 D: lane holds O^T[d = 4 (4 g + i') + db][query c]            -> 16 consecutive d of one query per lane
// Online softmax in the exp2 domain; the row maximum / row sum of a query close over its four lanes with two lane exchanges.
#include "mhmr_common.h"
#include "mhmr_internal.h"

namespace {

template <int DT>
__global__ __launch_bounds__(256) void attn_f32_kernel(const float* __restrict__ qkv, void* __restrict__ out_, int T, int Tp, int C) {
    typedef typename Op<DT>::T Tt;
    typedef typename Op<DT>::V8 V8;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * 64 + 16 * w;                 // the wave's first query row inside the image
    const size_t ld = 3 * (size_t)C;
    const float* img = qkv + (size_t)b * Tp * ld + (size_t)h * 64;
    Tt* orow = (Tt*)out_ + ((size_t)b * Tp + q0 + c) * (size_t)(2 * C) + h * 64 + 16 * g;

    if (q0 >= T) {            // sixteen padding rows: zeros (no dependence on how the caller allocated `out`)
        V8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (Tt)0.f;
        *(V8*)orow = z; *(V8*)(orow + 8) = z; *(V8*)(orow + C) = z; *(V8*)(orow + C + 8) = z;
        return;
    }

    // Q fragment, pre-scaled by head_dim^-0.5 * log2(e) in fp32
    float qf[16];
    {
        const float* qp = img + (size_t)(q0 + c) * ld + 16 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 v = *(const f32x4*)(qp + 4 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) qf[4 * j + e] = v[e] * MHMR_ATTN_QSCALE;
        }
    }
    const float* kbase = img + C + (size_t)c * ld + 16 * g;             // K[key0 + c][16 g ..]
    const float* vbase = img + 2 * C + (size_t)(4 * g) * ld + 4 * c;    // V[key0 + 4 g + i][4 c ..]

    f32x4 o[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) o[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    const int ntile = (T + 15) >> 4;
    f32x4 kn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kn[j] = *(const f32x4*)(kbase + 4 * j);
    for (int t = 0; t < ntile; ++t) {
        const int key0 = 16 * t;
        f32x4 kf[4], vf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) kf[j] = kn[j];
#pragma unroll
        for (int i = 0; i < 4; ++i) vf[i] = *(const f32x4*)(vbase + (size_t)(key0 + i) * ld);
        {   // next tile's K rows (the last iteration re-reads its own: rows < Tp either way)
            const int kn0 = t + 1 < ntile ? key0 + 16 : key0;
#pragma unroll
            for (int j = 0; j < 4; ++j) kn[j] = *(const f32x4*)(kbase + (size_t)kn0 * ld + 4 * j);
        }
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j][e], qf[4 * j + e], s, 0, 0, 0);
        // keys >= T (the tail of the last tile) are masked
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (key0 + 4 * g + i >= T) s[i] = -INFINITY;
            mx = fmaxf(mx, s[i]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);                  // finite from tile 0 on (key 0 is real)
        const float scale = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float p[4], ps = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            p[i] = __builtin_amdgcn_exp2f(s[i] - m_new);
            ps += p[i];
        }
        l_run = l_run * scale + ps;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[db][e] *= scale;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int db = 0; db < 4; ++db) o[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[i][db], p[i], o[db], 0, 0, 0);
    }
    l_run += __shfl_xor(l_run, 16);
    l_run += __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_run;
    // lane (c, g): query q0 + c, d = 16 g + 4 i' + db  (o[db][i'])
    V8 hi[2], lo[2];
#pragma unroll
    for (int ip = 0; ip < 4; ++ip)
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const float v = o[db][ip] * inv;
            const int j = 4 * ip + db;
            const Tt hv = (Tt)v;
            hi[j >> 3][j & 7] = hv;
            lo[j >> 3][j & 7] = (Tt)(v - (float)hv);
        }
    *(V8*)orow = hi[0]; *(V8*)(orow + 8) = hi[1];
    *(V8*)(orow + C) = lo[0]; *(V8*)(orow + C + 8) = lo[1];
}

}  // namespace

int mhmr_launch_attention_f32(const float* qkv, void* out, int B, int T, int Tp, int C, int H, int dtype, hipStream_t s) {
    if (B <= 0 || T <= 0 || Tp % 64 || Tp < T || C != 64 * H) return MHMR_ERR_BAD_SHAPE;
    const dim3 grid(Tp / 64, H, B);
    prof_begin(PROF_ATTN, s);
    if (dtype == MHMR_DT_F16) hipLaunchKernelGGL((attn_f32_kernel<MHMR_DT_F16>), grid, dim3(256), 0, s, qkv, out, T, Tp, C);
    else hipLaunchKernelGGL((attn_f32_kernel<MHMR_DT_BF16>), grid, dim3(256), 0, s, qkv, out, T, Tp, C);
    prof_end(PROF_ATTN, s, 4.0 * B * H * (double)T * T * 64);
    MHMR_CHECK_LAUNCH();
    return 0;
}
