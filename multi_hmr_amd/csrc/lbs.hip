// SMPL-X linear blend skinning + camera placement (reference blocks/smpl_layer.py:47-155 -> smplx.SMPLX.forward
// -> lbs), three launches:
//
//  1. lbs_pose_kernel   (one wave per person)  Rodrigues x55, pose feature, joint regression from the
//     pre-contracted regressor (J = J0 + JS.[betas, expr]), kinematic chain, root rotation / recentring /
//     back-projected translation folded into the per-joint skinning transforms, 55 posed joints + projection.  It leaves the two
//     matrix operands of the vertex kernel in MFMA order, each as an f16 pair hi + lo:
//       F16   F[p] = [pose_feature(486) | betas | expr | 0]
//       A16   component c of the folded transform [R0 R_w | R0 (t' - pelvis)] of joint j (joints 55..63 zero)
//     (fragment-major: one 1 KiB block per (person group, part, k step), see the kernel)
//  2. lbs_vertex_kernel (the HBM-bound one; algorithmic bytes = blend basis 64.5 MB + skin weights 2.7 MB once, 214 KB per person out)
//     One workgroup = one 16-vertex tile for ALL persons: the tile's slice of the blend basis D = [posedirs ; shapedirs ; exprdirs]
//     (x 2^10, f16 hi + lo, 98 KB) is DMA'd into LDS once -- HBM sees every basis byte exactly once per launch -- and wave g
//     works on person group g (16 persons; groups beyond the wave count loop).  Both contractions run on the 16-bit matrix pipe at
//     fp32 accuracy (x . y = xh.yh + xl.yh + xh.yl, fp32 accumulate; the dropped xl.yl term is 2^-22 relative):
//       v_posed = v_template + F . D                      16 k-steps x 3 axes x 3 products   (B operand from LDS)
//       T       = sum_j w[v][j] [R | t]_j  (12 numbers)    2 k-steps x 12 components x 3 products: the skinning blend as a GEMM over
//                 the DENSE 64 x V weight matrix -- no per-vertex index list, no gathers of joint transforms (they were 17 us of
//                 the previous kernel's 74 us at 160 persons, 322 MB through the L1)
//     Both land as (vertex = lane & 15, 4 persons per accumulator quad), so the rigid transform, the camera translation (added in
//     fp32, exactly where the reference adds it: smpl_layer.py:139-140) and the pinhole projection are per-lane FMAs, and a lane
//     stores its vertex as one 12-byte and one 8-byte access (16 lanes = 192 / 128 contiguous bytes per person).
//     History at 160 persons: fp32 MFMA 108 us (matrix-bound), split-f16 blend + sparse gathered skinning 74 us (latency-bound K
//     loop at ~2 waves per SIMD, basis re-read per 64-person slab), this kernel: see DESIGN.md section 5.
//  3. lbs_extra_joints_kernel  the 21 vertex-picked joints and 51 barycentric face landmarks.
#include "mhmr_common.h"
#include "mhmr_internal.h"

namespace {

constexpr int NJ = 55;

__device__ __forceinline__ void mat3_mul(const float* a, const float* b, float* c) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) c[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}
__device__ __forceinline__ void mat3_vec(const float* a, const float* v, float* o) {
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = a[i * 3] * v[0] + a[i * 3 + 1] * v[1] + a[i * 3 + 2] * v[2];
}

__device__ __forceinline__ void inv3x3(const float* k, float* o) {
    const float a = k[0], b = k[1], c = k[2], d = k[3], e = k[4], f = k[5], g = k[6], h = k[7], i = k[8];
    const float A = e * i - f * h, B = -(d * i - f * g), Cc = d * h - e * g;
    const float id = 1.0f / (a * A + b * B + c * Cc);
    o[0] = A * id; o[1] = -(b * i - c * h) * id; o[2] = (b * f - c * e) * id;
    o[3] = B * id; o[4] = (a * i - c * g) * id;  o[5] = -(a * f - c * d) * id;
    o[6] = Cc * id; o[7] = -(a * h - b * g) * id; o[8] = (a * e - b * d) * id;
}

// perspective_projection (utils/camera.py:14-27): y = x / x.z ; (K y)[:2]
__device__ __forceinline__ void project(const float* K, const float* x, float* o2) {
    const float yx = x[0] / x[2], yy = x[1] / x[2], yz = x[2] / x[2];
    o2[0] = K[0] * yx + K[1] * yy + K[2] * yz;
    o2[1] = K[3] * yx + K[4] * yy + K[5] * yz;
}

__global__ __launch_bounds__(64) void lbs_pose_kernel(const mhmr_lbs_consts c, const float* __restrict__ rotvec,
                                                      const float* __restrict__ betas, const float* __restrict__ expr,
                                                      const float* __restrict__ loc, const float* __restrict__ dist,
                                                      const float* __restrict__ Kmat, const int* __restrict__ det_b, int P, int Pp,
                                                      _Float16* __restrict__ F16, _Float16* __restrict__ A16, float* __restrict__ xf,
                                                      float* __restrict__ j3d, float* __restrict__ j2d,
                                                      float* __restrict__ transl_out) {
    __shared__ float sR[NJ][9], sJ[NJ][3], sRw[NJ][9], sTw[NJ][3], sX[33];
    const int p = blockIdx.x, j = threadIdx.x;
    // Both operands are stored FRAGMENT-MAJOR: the 64 lanes of a wave read one (person group, part, k step) fragment as 1 KiB of
    // consecutive bytes (16 B per lane, lane = 16 * (k group) + person-in-group), i.e. eight whole 128-byte lines per wave
    // instruction; a row-major [person][k] image makes every fragment load touch 16 lines for 64 useful bytes each, and the TA,
    // not the matrix pipe or HBM, then paces the vertex kernel (93 us at 160 persons).
    //   F16 [group][hi|lo][Kb/32][64 lanes][8]      A16 [12 comps][hi|lo][group][2][64 lanes][8]
    const int ngr = Pp / 16, grp = p >> 4, pin = p & 15, nst = c.Kb / 32;
    auto put_f = [&](int k, float v) {
        const _Float16 h = (_Float16)v;
        _Float16* f = F16 + ((((size_t)grp * 2) * nst + (k >> 5)) * 64 + ((k >> 3) & 3) * 16 + pin) * 8 + (k & 7);
        f[0] = h;
        f[(size_t)nst * 512] = (_Float16)(v - (float)h);
    };
    auto put_a = [&](int comp, int joint, float v) {
        const _Float16 h = (_Float16)v;
        _Float16* a = A16 + (((((size_t)comp * 2) * ngr + grp) * 2 + (joint >> 5)) * 64 + ((joint >> 3) & 3) * 16 + pin) * 8 + (joint & 7);
        a[0] = h;
        a[(size_t)ngr * 1024] = (_Float16)(v - (float)h);
    };
    if (p >= P) {  // padding rows of both operand matrices
        for (int k = j; k < c.Kb; k += 64) put_f(k, 0.f);
        for (int comp = 0; comp < 12; ++comp) put_a(comp, j, 0.f);
        return;
    }
    const int ncoef = c.nb + 10;
    if (j < NJ) {
        // full_pose (55) from the reference's 53-vector (smpl_layer.py:88-101): 0 -> zero (root applied after LBS),
        // 1..21 body, 22 jaw <- 52, 23/24 eyes zero, 25..39 left hand <- 22..36, 40..54 right hand <- 37..51
        int src = -1;
        if (j >= 1 && j <= 21) src = j;
        else if (j == 22) src = 52;
        else if (j >= 25) src = j - 3;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (src >= 0) {
            const float* rv = rotvec + ((size_t)p * 53 + src) * 3;
            v0 = rv[0]; v1 = rv[1]; v2 = rv[2];
        }
        // smplx batch_rodrigues: angle = |v + 1e-8|, R = I + sin K + (1 - cos) K K
        const float a0 = v0 + 1e-8f, a1 = v1 + 1e-8f, a2 = v2 + 1e-8f;
        const float angle = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
        const float rx = v0 / angle, ry = v1 / angle, rz = v2 / angle;
        const float sn = sinf(angle), cs = cosf(angle), omc = 1.f - cs;
        const float Km[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
        float KK[9];
        mat3_mul(Km, Km, KK);
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            const float id = (e == 0 || e == 4 || e == 8) ? 1.f : 0.f;
            const float r = id + sn * Km[e] + omc * KK[e];
            sR[j][e] = r;
            if (j >= 1) put_f((j - 1) * 9 + e, r - id);
        }
        // joints from the pre-contracted regressor
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float s = c.J0[j * 3 + a];
            const float* js = c.JS + (size_t)(j * 3 + a) * ncoef;
            for (int l = 0; l < c.nb; ++l) s += js[l] * betas[(size_t)p * c.nb + l];
            for (int l = 0; l < 10; ++l) s += js[c.nb + l] * expr[(size_t)p * 10 + l];
            sJ[j][a] = s;
        }
    }
    // feature tail: [betas | expr | 0...]   (the template is added in fp32 by the vertex kernel)
    for (int k = 486 + j; k < c.Kb; k += 64) {
        const int t = k - 486;
        float v = 0.f;
        if (t < c.nb) v = betas[(size_t)p * c.nb + t];
        else if (t < ncoef) v = expr[(size_t)p * 10 + (t - c.nb)];
        put_f(k, v);
    }
    __syncthreads();
    // kinematic chain, one tree LEVEL at a time: every joint whose depth equals the level composes its parent's world transform with its
    // own (the 55 joints of SMPL-X sit on 10 levels; a single thread walking the 54 edges took 27 us -- more than the vertex kernel at
    // small person counts)
    int depth = 0;
    if (j < NJ)
        for (int a = c.parents[j]; a >= 0; a = c.parents[a]) ++depth;
    if (j == 0) {
#pragma unroll
        for (int e = 0; e < 9; ++e) sRw[0][e] = sR[0][e];
#pragma unroll
        for (int a = 0; a < 3; ++a) sTw[0][a] = sJ[0][a];
    }
    int maxdepth = depth;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxdepth = max(maxdepth, __shfl_xor(maxdepth, o));
    for (int level = 1; level <= maxdepth; ++level) {
        __syncthreads();
        if (j < NJ && depth == level) {
            const int pa = c.parents[j];
            float Rp[9], Rl[9], Rn[9], rel[3], t[3];
#pragma unroll
            for (int e = 0; e < 9; ++e) { Rp[e] = sRw[pa][e]; Rl[e] = sR[j][e]; }
            mat3_mul(Rp, Rl, Rn);
#pragma unroll
            for (int a = 0; a < 3; ++a) rel[a] = sJ[j][a] - sJ[pa][a];
            mat3_vec(Rp, rel, t);
#pragma unroll
            for (int e = 0; e < 9; ++e) sRw[j][e] = Rn[e];
#pragma unroll
            for (int a = 0; a < 3; ++a) sTw[j][a] = t[a] + sTw[pa][a];
        }
    }
    __syncthreads();
    if (j == 0) {
        // root orientation (roma.rotvec_to_rotmat), translation (inverse_perspective_projection), recentring
        const float* rv = rotvec + (size_t)p * 53 * 3;
        const float th = sqrtf(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
        const float den = fmaxf(th, 1e-6f);
        const float kx = rv[0] / den, ky = rv[1] / den, kz = rv[2] / den;
        const float sn = sinf(th), cs = cosf(th), omc = 1.f - cs;
        const float xs = kx * sn, ys = ky * sn, zs = kz * sn;
        const float xyc = kx * ky * omc, xzc = kx * kz * omc, yzc = ky * kz * omc;
        const float xxc = kx * kx * omc, yyc = ky * ky * omc, zzc = kz * kz * omc;
        float R0[9] = {1.f - yyc - zzc, xyc - zs, xzc + ys, xyc + zs, 1.f - xxc - zzc, -xs + yzc, xzc - ys, xs + yzc, 1.f - xxc - yyc};
        const float* Kp = Kmat + (size_t)det_b[p] * 9;
        float Ki[9];
        inv3x3(Kp, Ki);
        const float lx = loc[2 * p], ly = loc[2 * p + 1], d = dist[p];
        float tr[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) tr[a] = (Ki[a * 3] * lx + Ki[a * 3 + 1] * ly + Ki[a * 3 + 2] * 1.0f) * d;
        // person_center joint given: recentre on it (smpl_layer.py:131-136); None (center_joint < 0): the pelvis is ADDED to the
        // translation instead and nothing is recentred (smpl_layer.py:128-130), i.e. o = tr + pelvis
        float cc[3];
        if (c.center_joint >= 0) {
            float hc[3] = {sTw[c.center_joint][0] - sTw[0][0], sTw[c.center_joint][1] - sTw[0][1], sTw[c.center_joint][2] - sTw[0][2]};
            mat3_vec(R0, hc, cc);
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) cc[a] = -sTw[0][a];
        }
#pragma unroll
        for (int e = 0; e < 9; ++e) sX[e] = R0[e];
#pragma unroll
        for (int a = 0; a < 3; ++a) { sX[9 + a] = sTw[0][a]; sX[12 + a] = tr[a] - cc[a]; transl_out[3 * p + a] = tr[a]; }
#pragma unroll
        for (int e = 0; e < 9; ++e) sX[15 + e] = Kp[e];
    }
    __syncthreads();
    if (j < 24) xf[(size_t)p * 24 + j] = sX[j];
    if (j >= NJ) {                                     // joints 55..63: zero columns of the skinning operand
        for (int comp = 0; comp < 12; ++comp) put_a(comp, j, 0.f);
    }
    if (j < NJ) {
        float R0[9], Rw[9], Rf[9], tp[3], tt[3], u[3], jj[3];
#pragma unroll
        for (int e = 0; e < 9; ++e) { R0[e] = sX[e]; Rw[e] = sRw[j][e]; }
        const float pel[3] = {sX[9], sX[10], sX[11]}, o[3] = {sX[12], sX[13], sX[14]};
        // A'_j = [R_w | t_w - R_w J_j]; folded: [R0 R_w | R0 (t' - pelvis)]; the camera translation o is added per person in fp32 by
        // the vertex kernel (skin weights sum to one; the reference adds transl after the LBS, smpl_layer.py:139-140)
        float Jv[3] = {sJ[j][0], sJ[j][1], sJ[j][2]};
        mat3_vec(Rw, Jv, u);
#pragma unroll
        for (int a = 0; a < 3; ++a) tp[a] = sTw[j][a] - u[a] - pel[a];
        mat3_mul(R0, Rw, Rf);
        mat3_vec(R0, tp, tt);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            put_a(r * 4 + 0, j, Rf[r * 3]); put_a(r * 4 + 1, j, Rf[r * 3 + 1]); put_a(r * 4 + 2, j, Rf[r * 3 + 2]);
            put_a(r * 4 + 3, j, tt[r]);
        }
        // posed joint in camera space + projection
        float dj[3] = {sTw[j][0] - pel[0], sTw[j][1] - pel[1], sTw[j][2] - pel[2]};
        mat3_vec(R0, dj, jj);
#pragma unroll
        for (int a = 0; a < 3; ++a) { jj[a] += o[a]; j3d[((size_t)p * 127 + j) * 3 + a] = jj[a]; }
        float pr[2];
        project(&sX[15], jj, pr);
        j2d[((size_t)p * 127 + j) * 2] = pr[0];
        j2d[((size_t)p * 127 + j) * 2 + 1] = pr[1];
    }
}

// grid = Vp / 16 workgroups (one 16-vertex tile each) of LBS_NW waves; wave w works on person groups w and w + LBS_NW (two per wave
// at 160 persons).  The k range is consumed in LBS_NQ quarters through TWO LDS buffers of one quarter of the tile's basis slice
// ([Kb/32][hi|lo][3][16][8] f16 = 24 KiB each): quarter q + 1 is in flight while quarter q is multiplied, and three workgroups share
// a CU.  (With one buffer and two halves every workgroup of the launch -- all 656 are resident at once -- waited for its DMA at the
// same time and HBM idled during the MFMAs: 20.7 us at ONE person.)  A operands (F16, A16: L2-resident, identical for every vertex
// tile) are ordinary loads issued a quarter ahead; they are requested AFTER the quarter's MFMAs and BEFORE the next barrier, so that
// the one `s_waitcnt vmcnt(0)` per quarter (hipcc waits vmcnt(0) for an ordinary load beside LDS-DMA anyway) only ever waits for
// things the next quarter needs.
struct __attribute__((packed, aligned(4))) Vec3 { float x, y, z; };
struct __attribute__((packed, aligned(4))) Vec2 { float x, y; };
constexpr int LBS_ROW = 16 * 8 * 2;          // bytes of one (k block, part, axis) row of the tile: 16 vertices x 8 k x f16
constexpr int LBS_NW = 5, LBS_MAXG = 2;      // waves per workgroup; person groups a wave keeps accumulators for at once
constexpr int LBS_NQ = 4;                    // k quarters

__global__ __launch_bounds__(64 * LBS_NW, 4) void lbs_vertex_kernel(const mhmr_lbs_consts c, const _Float16* __restrict__ F16,
                                                                    const _Float16* __restrict__ A16, const float* __restrict__ xf, int P,
                                                                    int Pp, int g0, float* __restrict__ v3d, float* __restrict__ v2d) {
    typedef Op<MHMR_DT_F16>::V8 H8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g4 = lane >> 4, l15 = lane & 15;
    const int v0 = blockIdx.x * 16;
    const int ns = c.Kb / 32, nsq = ns / LBS_NQ;                    // k-steps of 32, per quarter (4 at Kb = 512)
    const int rows_q = (c.Kb / 8 / LBS_NQ) * 6;                     // (k block, part, axis) rows of one quarter (96 at Kb = 512)
    const int ngroups = Pp / 16;

    // one quarter of the tile's basis slice -> LDS buffer `buf`: a wave instruction moves 1 KiB (4 rows of 16 vertices x 16 B), source and
    // LDS image both lane-linear (tile-major basis: the tile's slice is one contiguous block)
    const _Float16* bsrc = (const _Float16*)c.basis16 + (size_t)blockIdx.x * (LBS_NQ * rows_q * 128) + lane * 8;
    auto dma_q = [&](int q, int buf) {
        for (int i = w; i < rows_q / 4; i += LBS_NW)
            glds16(bsrc + (size_t)(q * rows_q + 4 * i) * 128, smem + buf * (rows_q * LBS_ROW) + i * (4 * LBS_ROW));
    };
    // this wave's A operands of quarter q (fragment-major: 1 KiB per (group, part, k step))
    H8 ah[LBS_MAXG][4], al[LBS_MAXG][4];
    auto load_f = [&](int q) {
#pragma unroll
        for (int i = 0; i < LBS_MAXG; ++i) {
            const int g = g0 + w + i * LBS_NW;
            if (g < ngroups) {
                const _Float16* fh = F16 + ((((size_t)g * 2) * ns + q * nsq) * 64 + lane) * 8;
                const _Float16* fl = fh + (size_t)ns * 512;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (s < nsq) { ah[i][s] = *(const H8*)(fh + 512 * s); al[i][s] = *(const H8*)(fl + 512 * s); }
                }
            }
        }
    };
    dma_q(0, 0);
    load_f(0);
    // dense skin weights of the tile (B operand of the skinning GEMM): lane (v = l15, k group g4) holds joints 8 (4 t + g4) + 0..7
    H8 wh[2], wl[2];
    {
        const _Float16* wp = (const _Float16*)c.skin16 + (size_t)blockIdx.x * (8 * 2 * 128) + l15 * 8;      // tile-major [8][hi|lo][16][8]
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            wh[t] = *(const H8*)(wp + ((4 * t + g4) * 2 + 0) * 128);
            wl[t] = *(const H8*)(wp + ((4 * t + g4) * 2 + 1) * 128);
        }
    }
    const int v = v0 + l15;
    const bool vok = v < c.V;
    const float vt0 = vok ? c.vtemp[v] : 0.f, vt1 = vok ? c.vtemp[c.Vp + v] : 0.f, vt2 = vok ? c.vtemp[2 * c.Vp + v] : 0.f;   // fp32 template

    f32x4 acc[LBS_MAXG][3];
#pragma unroll
    for (int i = 0; i < LBS_MAXG; ++i)
#pragma unroll
        for (int a = 0; a < 3; ++a) acc[i][a] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- v_posed - v_template = F . D ----
    for (int q = 0; q < LBS_NQ; ++q) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // quarter q's DMA (hipcc does not wait for LDS-DMA by itself) and A operands
        __syncthreads();                                            // ... of every wave; and everyone is done reading the other buffer
#pragma unroll
        for (int i = 0; i < LBS_MAXG; ++i)                          // (pins the compiler's own wait for the A operands in front of the DMA)
#pragma unroll
            for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(ah[i][s]), "+v"(al[i][s]));
        if (q + 1 < LBS_NQ) dma_q(q + 1, (q + 1) & 1);
        const char* brow = smem + (q & 1) * (rows_q * LBS_ROW) + g4 * (6 * LBS_ROW) + l15 * 16;   // + s * 24 rows + (part * 3 + axis) rows
#pragma unroll
        for (int i = 0; i < LBS_MAXG; ++i) {
            if (g0 + w + i * LBS_NW < ngroups) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (s < nsq) {
                        const char* bs = brow + (size_t)s * (24 * LBS_ROW);
#pragma unroll
                        for (int a = 0; a < 3; ++a) {
                            const H8 bh = *(const H8*)(bs + a * LBS_ROW), bl = *(const H8*)(bs + (3 + a) * LBS_ROW);
                            acc[i][a] = Op<MHMR_DT_F16>::mfma16(ah[i][s], bh, acc[i][a]);
                            acc[i][a] = Op<MHMR_DT_F16>::mfma16(al[i][s], bh, acc[i][a]);
                            acc[i][a] = Op<MHMR_DT_F16>::mfma16(ah[i][s], bl, acc[i][a]);
                        }
                    }
                }
            }
        }
        if (q + 1 < LBS_NQ) load_f(q + 1);
    }
    // (no early exit for the padding vertices of the last tile: a lane is a vertex COLUMN of the products but a person ROW of the A
    // operands, so every lane stays active through the MFMAs; only the stores are guarded)
    // ---- T = sum_j w[v][j] [R | t]_j (one accumulator per component), rigid transform, camera translation, projection, stores ----
#pragma unroll
    for (int i = 0; i < LBS_MAXG; ++i) {
        const int g = g0 + w + i * LBS_NW;
        if (g >= ngroups) break;
        const int pg = 16 * g;
        const _Float16* ap = A16 + ((size_t)g * 2 * 64 + lane) * 8;       // + ((2 cmp + part) * ngroups * 2 + t) * 512
        const size_t part = (size_t)ngroups * 1024;
        f32x4 T[12];
#pragma unroll
        for (int cmp = 0; cmp < 12; ++cmp) {
            T[cmp] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const H8 xh = *(const H8*)(ap + (size_t)(2 * cmp) * part + 512 * t), xl = *(const H8*)(ap + (size_t)(2 * cmp + 1) * part + 512 * t);
                T[cmp] = Op<MHMR_DT_F16>::mfma16(xh, wh[t], T[cmp]);
                T[cmp] = Op<MHMR_DT_F16>::mfma16(xl, wh[t], T[cmp]);
                T[cmp] = Op<MHMR_DT_F16>::mfma16(xh, wl[t], T[cmp]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = pg + 4 * g4 + r;
            if (p >= P || !vok) continue;
            const float vx = acc[i][0][r] * (1.0f / 1024.0f) + vt0, vy = acc[i][1][r] * (1.0f / 1024.0f) + vt1,
                        vz = acc[i][2][r] * (1.0f / 1024.0f) + vt2;
            const f32x4* X = (const f32x4*)(xf + (size_t)p * 24 + 12);             // [o (3), K (9)]: three 16-byte loads
            const f32x4 x0 = X[0], x1 = X[1], x2 = X[2];
            float o3[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) o3[a] = T[4 * a][r] * vx + T[4 * a + 1][r] * vy + T[4 * a + 2][r] * vz + T[4 * a + 3][r] + x0[a];
            *(Vec3*)(v3d + ((size_t)p * c.V + v) * 3) = Vec3{o3[0], o3[1], o3[2]};          // one 12-byte store (dword-aligned)
            // perspective_projection (utils/camera.py:14-27) with one reciprocal instead of three divisions (<= 1 ulp apart)
            const float iz = __builtin_amdgcn_rcpf(o3[2]);
            const float yx = o3[0] * iz, yy = o3[1] * iz, yz = o3[2] * iz;
            *(Vec2*)(v2d + ((size_t)p * c.V + v) * 2) = Vec2{x0[3] * yx + x1[0] * yy + x1[1] * yz, x1[2] * yx + x1[3] * yy + x2[0] * yz};
        }
    }
}

// joints 55..75 = vertices picked by id; 76..126 = barycentric face landmarks (smplx vertices2landmarks).
// Both are affine in the vertices, so they are taken from the placed v3d; a landmark whose barycentric
// weights do not sum to exactly one gets the (1 - sum) * (o - R0 pelvis) correction of the affine part.
__global__ __launch_bounds__(128) void lbs_extra_joints_kernel(const mhmr_lbs_consts c, const float* __restrict__ v3d,
                                                               const float* __restrict__ v2d, const float* __restrict__ xf,
                                                               float* __restrict__ j3d, float* __restrict__ j2d) {
    const int p = blockIdx.x, i = threadIdx.x;
    const float* X = xf + (size_t)p * 24;
    if (i < 21) {
        const int vid = c.extra_vid[i];
        const float* s3 = v3d + ((size_t)p * c.V + vid) * 3;
        const float* s2 = v2d + ((size_t)p * c.V + vid) * 2;
        float* d3 = j3d + ((size_t)p * 127 + 55 + i) * 3;
        float* d2 = j2d + ((size_t)p * 127 + 55 + i) * 2;
        d3[0] = s3[0]; d3[1] = s3[1]; d3[2] = s3[2];
        d2[0] = s2[0]; d2[1] = s2[1];
    } else if (i < 72) {
        const int l = i - 21;
        float acc[3] = {0.f, 0.f, 0.f}, bs = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float bk = c.lmk_bary[l * 3 + k];
            const float* s3 = v3d + ((size_t)p * c.V + c.lmk_vidx[l * 3 + k]) * 3;
            acc[0] += bk * s3[0]; acc[1] += bk * s3[1]; acc[2] += bk * s3[2];
            bs += bk;
        }
        float rp[3];
        mat3_vec(X, X + 9, rp);  // R0 . pelvis
#pragma unroll
        for (int a = 0; a < 3; ++a) acc[a] += (1.f - bs) * (X[12 + a] - rp[a]);
        float* d3 = j3d + ((size_t)p * 127 + 76 + l) * 3;
        d3[0] = acc[0]; d3[1] = acc[1]; d3[2] = acc[2];
        float pr[2];
        project(X + 15, acc, pr);
        float* d2 = j2d + ((size_t)p * 127 + 76 + l) * 2;
        d2[0] = pr[0]; d2[1] = pr[1];
    }
}

}  // namespace

extern "C" int mhmr_lbs_forward(const mhmr_lbs_consts* c, const float* rotvec, const float* betas, const float* expr,
                                const float* loc, const float* dist, const float* Kmat, const int* det_b, int P, float* ws_F,
                                float* ws_A, float* ws_xf, float* v3d, float* v2d, float* j3d, float* j2d, float* transl,
                                void* stream) {
    if (!c || P < 0) return MHMR_ERR_BAD_ARG;
    if (P == 0) return 0;
    if (c->Kb % 32 || c->Vp % 64 || c->Kb < 486 + c->nb + 10 || !c->skin16) return MHMR_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int Pp = (P + 15) / 16 * 16;
    if (c->Kb % 128 || c->Kb > 512) return MHMR_ERR_BAD_SHAPE;     // four equal k quarters of at most 4 steps
    const size_t lds = 2 * (size_t)(c->Kb / 32) * 6 * LBS_ROW;     // two buffers of one quarter of a tile's basis slice: 49152 B at Kb = 512
    hipLaunchKernelGGL(lbs_pose_kernel, dim3(Pp), dim3(64), 0, s, *c, rotvec, betas, expr, loc, dist, Kmat, det_b, P, Pp,
                       (_Float16*)ws_F, (_Float16*)ws_A, ws_xf, j3d, j2d, transl);
    MHMR_CHECK_LAUNCH();
    // one launch covers LBS_NW * LBS_MAXG = 10 person groups (160 persons); more persons take further launches over the same tiles
    const int ngroups = Pp / 16;
    prof_begin(PROF_LBS, s);
    for (int g0 = 0; g0 < ngroups; g0 += LBS_NW * LBS_MAXG)
        hipLaunchKernelGGL(lbs_vertex_kernel, dim3(c->Vp / 16), dim3(64 * LBS_NW), lds, s, *c, (const _Float16*)ws_F, (const _Float16*)ws_A, ws_xf,
                           P, Pp, g0, v3d, v2d);
    prof_end(PROF_LBS, s, (double)P);
    MHMR_CHECK_LAUNCH();
    hipLaunchKernelGGL(lbs_extra_joints_kernel, dim3(P), dim3(128), 0, s, *c, v3d, v2d, ws_xf, j3d, j2d);
    MHMR_CHECK_LAUNCH();
    return 0;
}
