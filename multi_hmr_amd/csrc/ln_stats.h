// Row statistics of the LayerNorm fold from the residual epilogues' block sums (GemmArgs::pstats): shared by ln_stats_kernel (vit_misc.hip)
// and the class-row kernel that carries the same work as extra workgroups of its launch (vit_cls.hip, round 6).
#pragma once
#include "mhmr_common.h"

template <int CTRL>
__device__ __forceinline__ float lnst_dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// gid = a global thread index of the statistics role: eight lanes per row, two 64-column blocks (one 16-byte load) per lane; a wave reads
// 8 rows x 128 B as whole lines; the eight partial sums meet on the VALU (quad xor 1, quad xor 2, half-row mirror): a fixed order,
// bit-reproducible.  Row (b, n), n < N, of an image of Tp rows: rowstats[b * Tp + n] = (mean, rsqrt(E[x^2] - mean^2 + eps)).
__device__ __forceinline__ void ln_stats_patch_rows(int gid, const float* __restrict__ pstats, float* __restrict__ rowstats, int B, int N, int Tp,
                                                    int C, int nblk, float eps) {
    const int m = gid >> 3, part = gid & 7;
    const bool live = m < B * N && 2 * part < nblk;
    const int mm = m < B * N ? m : B * N - 1;
    const int b = mm / N, n = mm - b * N;
    const size_t row = (size_t)b * Tp + n;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (live) v = *(const f32x4*)(pstats + row * nblk * 2 + part * 4);
    float s1 = v[0] + v[2], s2 = v[1] + v[3];
    s1 = lnst_dpp_add<0xB1>(s1); s2 = lnst_dpp_add<0xB1>(s2);
    s1 = lnst_dpp_add<0x4E>(s1); s2 = lnst_dpp_add<0x4E>(s2);
    s1 = lnst_dpp_add<0x141>(s1); s2 = lnst_dpp_add<0x141>(s2);
    if (part == 0 && m < B * N) {
        const float mean = s1 * (1.0f / C);
        const float var = fmaxf(s2 * (1.0f / C) - mean * mean, 0.f);
        *(f32x2*)(rowstats + row * 2) = (f32x2){mean, rsqrtf(var + eps)};
    }
}
