// SMPL-X linear blend skinning + camera placement (reference blocks/smpl_layer.py:47-155 -> smplx.SMPLX.forward
// -> lbs), two launches:
//
//  1. lbs_pose_kernel   (one wave per person)  Rodrigues x55, pose feature, joint regression from the
//     pre-contracted regressor (J = J0 + JS.[betas, expr]), kinematic chain, root rotation / recentring /
//     back-projected translation folded into the per-joint skinning transforms, 55 posed joints + projection.  It leaves the two
//     matrix operands of the vertex kernel in MFMA order, each as an f16 pair hi + lo:
//       F16   F[p] = [pose_feature(486) | betas | expr | 0]
//       A16   component c of the folded transform [R0 R_w | R0 (t' - pelvis)] of joint j (joints 55..63 zero)
//     (fragment-major: one 1 KiB block per (person group, part, k step), see the kernel)
//  2. lbs_vertex_kernel (the HBM-bound one; algorithmic bytes = blend basis 64.5 MB + skin weights 2.7 MB once, 214 KB per person out)
//     One workgroup = one 48-vertex tile for ALL persons of the launch: the tile's slice of the blend basis D = [posedirs ; shapedirs ;
//     exprdirs] (x 2^10, f16, 162 KiB) streams through an LDS ring exactly once -- HBM sees every basis byte once per launch --
//     and compute wave g works on person group g (16 persons).  Both contractions run on the 16-bit matrix pipe; the skinning GEMM
//     and the shape / expression part of the blend at fp32 accuracy (f16 pairs, x . y = xh.yh + xl.yh + xh.yl, fp32 accumulate; the
//     dropped xl.yl term is 2^-22 relative), the millimetre-sized pose correctives with one product (LBS_EBYTES_HI below):
//       v_posed = v_template + F . D                      16 k-steps x 3 vertex blocks x 3 axes x (1 | 3) products   (B operand from LDS)
//       T       = sum_j w[v][j] [R | t]_j  (12 numbers)    2 k-steps x 12 components x 3 blocks x 3 products: the skinning blend as a
//                 GEMM over the DENSE 64 x V weight matrix -- no per-vertex index list, no gathers of joint transforms
//     Both land as (vertex = lane & 15, 4 persons per accumulator quad), so the rigid transform, the camera translation (added in
//     fp32, exactly where the reference adds it: smpl_layer.py:139-140) and the pinhole projection are per-lane FMAs, and a lane
//     stores its vertex as one 12-byte and one 8-byte access (16 lanes = 192 / 128 contiguous bytes per person and block).
//     History at 160 persons: fp32 MFMA 108 us (matrix-bound); split-f16 blend + sparse gathered skinning 74 us; 16-vertex tiles with
//     the skinning as a GEMM 54.7 us, with explicit operand prefetch 48.6 us (both bound by the vector-memory path feeding the A
//     operands, see the kernel); this form 31.4 us (matrix pipe ~55 % busy on the 219 CUs that hold a tile).
//     The 72 extra joints (21 vertices picked by id, 51 barycentric face landmarks: joints 55..126) are five more tiles of VIRTUAL
//     vertices behind the real ones (packing.pack_smplx: copies of their corner vertices' operand columns, arranged so that ONE lane ends
//     up with the three posed corners of an extra joint): they cost 5 of 224 workgroups on otherwise idle CUs instead of a third launch
//     (round 2: lbs_extra_joints_kernel, 3.4 us + a launch gap behind the vertex kernel).
#include <stdlib.h>
#include <stddef.h>
#include "mhmr_common.h"
#include "mhmr_internal.h"

// (built without the SLP vectoriser like every file here: mhmr_common.h has the history -- this kernel was the first casualty)

namespace {

constexpr int NJ = 55;
constexpr int LBS_KB_POSE = 512;      // = LBS_KB (declared below with the vertex kernel's constants; static_assert there)

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

// One person = ONE WAVE: its staging lives in a PoseLds of its own and every synchronisation point is wave-local (LDS accesses of a wave
// complete in order: a landing wait is all a wave needs to see its own lanes' writes), so the same code runs as a 64-thread workgroup
// (lbs_pose_kernel) and as one of the twelve waves of a pose-role workgroup of the fused launch (lbs_fused_kernel).
struct __attribute__((aligned(16))) PoseLds {
    float sR[NJ][9], sJ[NJ][3], sRw[NJ][9], sTw[NJ][3], sX[36];
    int sPar[56];
    // the person's two operand rows are collected here and leave as 16-byte chunks (8 consecutive k / joints of one person are 8
    // consecutive f16 of the fragment-major layouts): 2 + 2 x 1.5 wide stores per lane instead of ~50 two-byte ones (round 4)
    __attribute__((aligned(16))) float sF[LBS_KB_POSE];
    __attribute__((aligned(16))) float sA[12][64];
};
static_assert(sizeof(PoseLds) % 16 == 0 && offsetof(PoseLds, sF) % 16 == 0 && offsetof(PoseLds, sA) % 16 == 0, "16-byte chunks");
#define LBS_WAVE_SYNC()                                        \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
        __builtin_amdgcn_wave_barrier();                       \
    } while (0)

__device__ __forceinline__ void lbs_pose_person(const mhmr_lbs_consts& c, const float* __restrict__ rotvec,
                                                const float* __restrict__ betas, const float* __restrict__ expr,
                                                const float* __restrict__ loc, const float* __restrict__ dist,
                                                const float* __restrict__ Kmat, const int* __restrict__ det_b, int P, int Pp,
                                                _Float16* __restrict__ F16, _Float16* __restrict__ A16, float* __restrict__ xf,
                                                float* __restrict__ j3d, float* __restrict__ j2d,
                                                float* __restrict__ transl_out, int p, PoseLds& L) {
    float (&sR)[NJ][9] = L.sR; float (&sJ)[NJ][3] = L.sJ; float (&sRw)[NJ][9] = L.sRw; float (&sTw)[NJ][3] = L.sTw; float (&sX)[36] = L.sX;
    int (&sPar)[56] = L.sPar;
    float (&sF)[LBS_KB_POSE] = L.sF; float (&sA)[12][64] = L.sA;
    const int j = threadIdx.x & 63;
    // Both operands are stored FRAGMENT-MAJOR: the 64 lanes of a wave read one (person group, part, k step) fragment as 1 KiB of
    // consecutive bytes (16 B per lane, lane = 16 * (k group) + person-in-group), i.e. eight whole 128-byte lines per wave
    // instruction; a row-major [person][k] image makes every fragment load touch 16 lines for 64 useful bytes each, and the TA,
    // not the matrix pipe or HBM, then paces the vertex kernel (93 us at 160 persons).
    //   F16 [group][hi|lo][Kb/32][64 lanes][8]      A16 [12 comps][hi|lo][group][2][64 lanes][8]
    const int ngr = Pp / 16, grp = p >> 4, pin = p & 15, nst = c.Kb / 32;
    auto put_f = [&](int k, float v) { sF[k] = v; };
    auto put_a = [&](int comp, int joint, float v) { sA[comp][joint] = v; };
    // chunk kb (k = 8 kb .. 8 kb + 7) of the feature row / chunk (comp, jb) of the transform rows -> hi and lo halves, 16 bytes each
    typedef Op<MHMR_DT_F16>::V8 H8;
    auto flush = [&](bool zero) {
        for (int kb = j; kb < c.Kb / 8; kb += 64) {
            H8 h, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = zero ? 0.f : sF[8 * kb + e];
                h[e] = (_Float16)v;
                l[e] = (_Float16)(v - (float)h[e]);
            }
            _Float16* f = F16 + ((((size_t)grp * 2) * nst + (kb >> 2)) * 64 + (kb & 3) * 16 + pin) * 8;
            *(H8*)f = h;
            *(H8*)(f + (size_t)nst * 512) = l;
        }
        for (int ch = j; ch < 96; ch += 64) {
            const int comp = ch >> 3, jb = ch & 7;
            H8 h, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = zero ? 0.f : sA[comp][8 * jb + e];
                h[e] = (_Float16)v;
                l[e] = (_Float16)(v - (float)h[e]);
            }
            _Float16* a = A16 + (((((size_t)comp * 2) * ngr + grp) * 2 + (jb >> 2)) * 64 + (jb & 3) * 16 + pin) * 8;
            *(H8*)a = h;
            *(H8*)(a + (size_t)ngr * 1024) = l;
        }
    };
    if (p >= P) {  // padding rows of both operand matrices
        flush(true);
        return;
    }
    const int ncoef = c.nb + 10;
    if (j < NJ) sPar[j] = c.parents[j];
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
        float sn, cs;
        sincosf(angle, &sn, &cs);
        const float omc = 1.f - cs;
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
        // joints from the pre-contracted regressor.  The usual 10 betas + 10 expression coefficients: the joint's three rows of JS are
        // 15 independent 16-byte loads (a run-time loop of dependent scalar loads was most of this kernel's 15 us)
        if (ncoef == 20) {
            const f32x4* js = (const f32x4*)(c.JS + (size_t)(j * 3) * 20);
            f32x4 row[15];
#pragma unroll
            for (int i = 0; i < 15; ++i) row[i] = js[i];
            float cf[20];
#pragma unroll
            for (int l = 0; l < 10; ++l) { cf[l] = betas[(size_t)p * 10 + l]; cf[10 + l] = expr[(size_t)p * 10 + l]; }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float s = c.J0[j * 3 + a];
#pragma unroll
                for (int l = 0; l < 20; ++l) s += row[5 * a + (l >> 2)][l & 3] * cf[l];     // (same order of additions as the loop below)
                sJ[j][a] = s;
            }
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float s = c.J0[j * 3 + a];
                const float* js = c.JS + (size_t)(j * 3 + a) * ncoef;
                for (int l = 0; l < c.nb; ++l) s += js[l] * betas[(size_t)p * c.nb + l];
                for (int l = 0; l < 10; ++l) s += js[c.nb + l] * expr[(size_t)p * 10 + l];
                sJ[j][a] = s;
            }
        }
    }
    if (j == 63) {
        // (an otherwise idle lane, beside the Rodrigues / joint-regression work of the others: these loads and the sin / cos no longer
        // sit behind the kinematic chain)  root orientation (roma.rotvec_to_rotmat), translation (inverse_perspective_projection)
        const float* rv = rotvec + (size_t)p * 53 * 3;
        const float th = sqrtf(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
        const float den = fmaxf(th, 1e-6f);
        const float kx = rv[0] / den, ky = rv[1] / den, kz = rv[2] / den;
        float sn, cs;
        sincosf(th, &sn, &cs);
        const float omc = 1.f - cs;
        const float xs = kx * sn, ys = ky * sn, zs = kz * sn;
        const float xyc = kx * ky * omc, xzc = kx * kz * omc, yzc = ky * kz * omc;
        const float xxc = kx * kx * omc, yyc = ky * ky * omc, zzc = kz * kz * omc;
        const float R0[9] = {1.f - yyc - zzc, xyc - zs, xzc + ys, xyc + zs, 1.f - xxc - zzc, -xs + yzc, xzc - ys, xs + yzc, 1.f - xxc - yyc};
        const float* Kp = Kmat + (size_t)det_b[p] * 9;
        float Ki[9];
        inv3x3(Kp, Ki);
        const float lx = loc[2 * p], ly = loc[2 * p + 1], d = dist[p];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float tr = (Ki[a * 3] * lx + Ki[a * 3 + 1] * ly + Ki[a * 3 + 2] * 1.0f) * d;
            sX[24 + a] = tr;
            transl_out[3 * p + a] = tr;
        }
#pragma unroll
        for (int e = 0; e < 9; ++e) { sX[e] = R0[e]; sX[15 + e] = Kp[e]; }
    }
    // feature tail: [betas | expr | 0...]   (the template is added in fp32 by the vertex kernel)
    for (int k = 486 + j; k < c.Kb; k += 64) {
        const int t = k - 486;
        float v = 0.f;
        if (t < c.nb) v = betas[(size_t)p * c.nb + t];
        else if (t < ncoef) v = expr[(size_t)p * 10 + (t - c.nb)];
        put_f(k, v);
    }
    LBS_WAVE_SYNC();
    // kinematic chain, one tree LEVEL at a time: every joint whose depth equals the level composes its parent's world transform with its
    // own (the 55 joints of SMPL-X sit on 10 levels; a single thread walking the 54 edges took 27 us -- more than the vertex kernel at
    // small person counts)
    int depth = 0;
    if (j < NJ)
        for (int a = sPar[j]; a >= 0; a = sPar[a]) ++depth;          // (LDS: a chain of up to ten dependent GLOBAL loads cost 2 us)
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
        LBS_WAVE_SYNC();
        if (j < NJ && depth == level) {
            const int pa = sPar[j];
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
    LBS_WAVE_SYNC();
    if (j == 0) {
        // recentring.  person_center joint given: recentre on it (smpl_layer.py:131-136); None (center_joint < 0): the pelvis is ADDED
        // to the translation instead and nothing is recentred (smpl_layer.py:128-130), i.e. o = tr + pelvis
        float R0[9], cc[3];
#pragma unroll
        for (int e = 0; e < 9; ++e) R0[e] = sX[e];
        if (c.center_joint >= 0) {
            float hc[3] = {sTw[c.center_joint][0] - sTw[0][0], sTw[c.center_joint][1] - sTw[0][1], sTw[c.center_joint][2] - sTw[0][2]};
            mat3_vec(R0, hc, cc);
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) cc[a] = -sTw[0][a];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) { sX[9 + a] = sTw[0][a]; sX[12 + a] = sX[24 + a] - cc[a]; }
    }
    LBS_WAVE_SYNC();
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
    LBS_WAVE_SYNC();
    flush(false);
}

__global__ __launch_bounds__(64) void lbs_pose_kernel(const mhmr_lbs_consts c, const float* __restrict__ rotvec,
                                                      const float* __restrict__ betas, const float* __restrict__ expr,
                                                      const float* __restrict__ loc, const float* __restrict__ dist,
                                                      const float* __restrict__ Kmat, const int* __restrict__ det_b, int P, int Pp,
                                                      _Float16* __restrict__ F16, _Float16* __restrict__ A16, float* __restrict__ xf,
                                                      float* __restrict__ j3d, float* __restrict__ j2d,
                                                      float* __restrict__ transl_out) {
    __shared__ PoseLds L;
    lbs_pose_person(c, rotvec, betas, expr, loc, dist, Kmat, det_b, P, Pp, F16, A16, xf, j3d, j2d, transl_out, (int)blockIdx.x, L);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 6: the same person as FOUR waves (one per SIMD of a CU).  The one-wave form above executes ~3 100 instructions in a row -- the
// static 1 725 with the ten-level loop unrolled in time -- and a lone wave issues one VALU instruction per ~5 cycles: 16 k cycles = 8 us
// of its 10-12 us are ISSUE time, not latency (round 5's review: "2 800 cycles per level").  Here the work is laid across 256 lanes:
//   phase A  four independent roles, one wave each: Rodrigues + pose feature | joint regression | root rotation, K^-1, translation, feature
//            tail, zero columns | topology: depth of every joint, the joints of every tree level as a list (ballot + rank)
//   phase B  the kinematic chain level by level with one lane per (joint of the level, one of its 12 transform elements): a level is
//            ~10 LDS reads, 3-4 FMAs and one store per lane instead of 21 reads, 36 FMAs and 12 stores
//   phase C  recentring (one lane), then three roles: folded skinning rows | posed joints + projection | the person record
//   phase D  the operand rows leave as 16-byte chunks: 160 chunks over 256 lanes, one pass
// Every expression is the one of lbs_pose_person (same operand order, same contraction), so the results are bit-identical to it
// (tests/test_gpu_kernels.py::test_lbs_fused_launch_is_bit_identical... compares against the fused launch, which keeps the one-wave form).
#ifdef MHMR_LBS_STAMPS      // tools/lbs_pose_timeline.py: wall-clock (100 MHz) stamps of the pose kernel, debug build only
__device__ unsigned long long* g_pose_stamps;
#define POSE_STAMP(i)                                                                                                               \
    do {                                                                                                                            \
        if (g_pose_stamps && (threadIdx.x & 63) == 0) g_pose_stamps[(size_t)blockIdx.x * 16 + (i)] = wall_clock64();               \
    } while (0)
#else
#define POSE_STAMP(i)
#endif
constexpr int POSE_MAXLEVEL = 56;          // a tree of NJ = 55 joints has at most 55 levels: every topology fits
struct __attribute__((aligned(16))) Pose4Lds {
    PoseLds L;
    int sDepth[64];
    int sCnt[POSE_MAXLEVEL];
    unsigned char sList[POSE_MAXLEVEL][64];
    int sMaxDepth;
};

__global__ __launch_bounds__(256) void lbs_pose4_kernel(const mhmr_lbs_consts c, const float* __restrict__ rotvec,
                                                       const float* __restrict__ betas, const float* __restrict__ expr,
                                                       const float* __restrict__ loc, const float* __restrict__ dist,
                                                       const float* __restrict__ Kmat, const int* __restrict__ det_b, int P, int Pp,
                                                       _Float16* __restrict__ F16, _Float16* __restrict__ A16, float* __restrict__ xf,
                                                       float* __restrict__ j3d, float* __restrict__ j2d, float* __restrict__ transl_out) {
    __shared__ Pose4Lds S;
    PoseLds& L = S.L;
    float (&sR)[NJ][9] = L.sR; float (&sJ)[NJ][3] = L.sJ; float (&sRw)[NJ][9] = L.sRw; float (&sTw)[NJ][3] = L.sTw; float (&sX)[36] = L.sX;
    int (&sPar)[56] = L.sPar;
    float (&sF)[LBS_KB_POSE] = L.sF; float (&sA)[12][64] = L.sA;
    const int p = (int)blockIdx.x, tid = threadIdx.x, j = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (w == 0) POSE_STAMP(0);
    const int ngr = Pp / 16, grp = p >> 4, pin = p & 15, nst = c.Kb / 32;
    typedef Op<MHMR_DT_F16>::V8 H8;
    // phase D (also the padding rows of both operand matrices: zero = true)
    auto flush = [&](bool zero) {
        const int nf = c.Kb / 8;
        for (int t = tid; t < nf + 96; t += 256) {
            H8 h, l;
            if (t < nf) {
                const int kb = t;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = zero ? 0.f : sF[8 * kb + e];
                    h[e] = (_Float16)v;
                    l[e] = (_Float16)(v - (float)h[e]);
                }
                _Float16* f = F16 + ((((size_t)grp * 2) * nst + (kb >> 2)) * 64 + (kb & 3) * 16 + pin) * 8;
                *(H8*)f = h;
                *(H8*)(f + (size_t)nst * 512) = l;
            } else {
                const int ch = t - nf, comp = ch >> 3, jb = ch & 7;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = zero ? 0.f : sA[comp][8 * jb + e];
                    h[e] = (_Float16)v;
                    l[e] = (_Float16)(v - (float)h[e]);
                }
                _Float16* a = A16 + (((((size_t)comp * 2) * ngr + grp) * 2 + (jb >> 2)) * 64 + (jb & 3) * 16 + pin) * 8;
                *(H8*)a = h;
                *(H8*)(a + (size_t)ngr * 1024) = l;
            }
        }
    };
    if (p >= P) {
        flush(true);
        return;
    }
    const int ncoef = c.nb + 10;
    // the kinematic tree's level schedule from the host (mhmr_lbs_consts::pose_tasks): this lane's task at every level, requested with
    // everything else of phase A (one round trip); without it wave 3 derives the schedule from `parents` (1.5 us more, the timeline says)
    const bool host_tasks = c.pose_tasks != nullptr && c.pose_levels >= 1 && c.pose_levels <= 16;
    int task[16];
#pragma unroll
    for (int level = 0; level < 16; ++level) task[level] = (host_tasks && level < c.pose_levels) ? c.pose_tasks[level * 256 + tid] : -1;
    // ---------------- phase A: four roles ----------------
    if (w == 0) {
        if (j < NJ) {
            // full_pose (55) from the reference's 53-vector (smpl_layer.py:88-101), smplx batch_rodrigues -- as in lbs_pose_person
            int src = -1;
            if (j >= 1 && j <= 21) src = j;
            else if (j == 22) src = 52;
            else if (j >= 25) src = j - 3;
            float v0 = 0.f, v1 = 0.f, v2 = 0.f;
            if (src >= 0) {
                const float* rv = rotvec + ((size_t)p * 53 + src) * 3;
                v0 = rv[0]; v1 = rv[1]; v2 = rv[2];
            }
            const float a0 = v0 + 1e-8f, a1 = v1 + 1e-8f, a2 = v2 + 1e-8f;
            const float angle = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
            const float rx = v0 / angle, ry = v1 / angle, rz = v2 / angle;
            float sn, cs;
            sincosf(angle, &sn, &cs);
            const float omc = 1.f - cs;
            const float Km[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
            float KK[9];
            mat3_mul(Km, Km, KK);
#pragma unroll
            for (int e = 0; e < 9; ++e) {
                const float id = (e == 0 || e == 4 || e == 8) ? 1.f : 0.f;
                const float r = id + sn * Km[e] + omc * KK[e];
                sR[j][e] = r;
                if (j >= 1) sF[(j - 1) * 9 + e] = r - id;
            }
        }
    } else if (w == 1) {
        if (j < NJ) {
            if (ncoef == 20) {
                const f32x4* js = (const f32x4*)(c.JS + (size_t)(j * 3) * 20);
                f32x4 row[15];
#pragma unroll
                for (int i = 0; i < 15; ++i) row[i] = js[i];
                float cf[20];
#pragma unroll
                for (int l = 0; l < 10; ++l) { cf[l] = betas[(size_t)p * 10 + l]; cf[10 + l] = expr[(size_t)p * 10 + l]; }
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    float sm = c.J0[j * 3 + a];
#pragma unroll
                    for (int l = 0; l < 20; ++l) sm += row[5 * a + (l >> 2)][l & 3] * cf[l];
                    sJ[j][a] = sm;
                }
            } else {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    float sm = c.J0[j * 3 + a];
                    const float* js = c.JS + (size_t)(j * 3 + a) * ncoef;
                    for (int l = 0; l < c.nb; ++l) sm += js[l] * betas[(size_t)p * c.nb + l];
                    for (int l = 0; l < 10; ++l) sm += js[c.nb + l] * expr[(size_t)p * 10 + l];
                    sJ[j][a] = sm;
                }
            }
        }
    } else if (w == 2) {
        if (j == 63) {
            // root orientation (roma.rotvec_to_rotmat), translation (inverse_perspective_projection) -- as in lbs_pose_person.  The
            // image index first and the camera matrix right behind it: the second of the two dependent round trips then runs under the
            // sin / cos below instead of behind it
            const int db = det_b[p];
            const float* rvp = rotvec + (size_t)p * 53 * 3;
            const float rv[3] = {rvp[0], rvp[1], rvp[2]};
            const float lx = loc[2 * p], ly = loc[2 * p + 1], d = dist[p];
            const float* Kg = Kmat + (size_t)db * 9;
            float Kp[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) Kp[e] = Kg[e];
            const float th = sqrtf(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
            const float den = fmaxf(th, 1e-6f);
            const float kx = rv[0] / den, ky = rv[1] / den, kz = rv[2] / den;
            float sn, cs;
            sincosf(th, &sn, &cs);
            const float omc = 1.f - cs;
            const float xs = kx * sn, ys = ky * sn, zs = kz * sn;
            const float xyc = kx * ky * omc, xzc = kx * kz * omc, yzc = ky * kz * omc;
            const float xxc = kx * kx * omc, yyc = ky * ky * omc, zzc = kz * kz * omc;
            const float R0[9] = {1.f - yyc - zzc, xyc - zs, xzc + ys, xyc + zs, 1.f - xxc - zzc, -xs + yzc, xzc - ys, xs + yzc, 1.f - xxc - yyc};
            float Ki[9];
            inv3x3(Kp, Ki);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float tr = (Ki[a * 3] * lx + Ki[a * 3 + 1] * ly + Ki[a * 3 + 2] * 1.0f) * d;
                sX[24 + a] = tr;
                transl_out[3 * p + a] = tr;
            }
#pragma unroll
            for (int e = 0; e < 9; ++e) { sX[e] = R0[e]; sX[15 + e] = Kp[e]; }
        }
        // feature tail: [betas | expr | 0...]; joints 55..63: zero columns of the skinning operand
        for (int k = 486 + j; k < c.Kb; k += 64) {
            const int t = k - 486;
            float v = 0.f;
            if (t < c.nb) v = betas[(size_t)p * c.nb + t];
            else if (t < ncoef) v = expr[(size_t)p * 10 + (t - c.nb)];
            sF[k] = v;
        }
        if (j >= NJ && j < 63) {
            for (int comp = 0; comp < 12; ++comp) sA[comp][j] = 0.f;
        }
        if (j == 63) {
            for (int comp = 0; comp < 12; ++comp) sA[comp][63] = 0.f;
        }
    } else if (host_tasks) {
        if (j == 0) S.sMaxDepth = c.pose_levels - 1;          // (the schedule came from the host: nothing to derive)
    } else {
        // topology: parents, depth of every joint, and per tree level the list of its joints (in joint order)
        if (j < NJ) sPar[j] = c.parents[j];
        LBS_WAVE_SYNC();
        int depth = -1;
        if (j < NJ) {
            depth = 0;
            for (int a = sPar[j]; a >= 0; a = sPar[a]) ++depth;
        }
        int maxdepth = depth;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) maxdepth = max(maxdepth, __shfl_xor(maxdepth, o));
        const unsigned long long lt = (1ull << j) - 1ull;
        for (int level = 0; level <= maxdepth; ++level) {
            const unsigned long long m = __ballot(depth == level);
            if (depth == level) S.sList[level][__popcll(m & lt)] = (unsigned char)j;
            if (j == 0) S.sCnt[level] = __popcll(m);
        }
        if (j == 0) S.sMaxDepth = maxdepth;
    }
    POSE_STAMP(1 + w);          // 1..4: the four roles of phase A done (their LDS stores issued)
    __syncthreads();
    if (w == 0) POSE_STAMP(5);
    // ---------------- phase B: the kinematic chain, one tree level at a time, one lane per (joint, element) ----------------
    const int maxdepth = S.sMaxDepth;
    // The timeline (tools/lbs_pose_timeline.py, profiles/r06_session_d*.txt) put 4.3 of the kernel's 8.8 us HERE: per level the loop below is
    // three DEPENDENT rounds of LDS reads (level list -> parent -> transforms) and a barrier, 0.43 us a level.  Common case -- at most 16
    // levels of at most 21 joints (SMPL-X: 10 levels, <= 13 joints): a lane's task at every level is the same (slot, element), so its
    // joint and parent of ALL levels come into registers first (16 independent reads), and a level is ONE round of reads, FMAs, a store.
    bool fastpath = host_tasks || maxdepth < 16;
    if (!host_tasks)
        for (int level = 0; level <= maxdepth && level < 16; ++level) fastpath = fastpath && S.sCnt[level] * 12 <= 256;
    if (fastpath) {
        const int slot = tid / 12, e = tid - slot * 12;
        // task[level]: joint | parent << 8 (parent 0xff = a root), -1 = no task at that level
        if (!host_tasks) {
#pragma unroll
            for (int level = 0; level < 16; ++level) {
                task[level] = -1;
                if (level <= maxdepth && slot < S.sCnt[level]) {
                    const int jj = S.sList[level][slot];
                    task[level] = jj | ((sPar[jj] & 0xff) << 8);
                }
            }
        }
#pragma unroll
        for (int level = 0; level < 16; ++level) {
            if (level <= maxdepth) {
                if (task[level] >= 0) {
                    const int jj = task[level] & 0xff, pa = task[level] >> 8;
                    if (pa == 0xff) {
                        if (e < 9) sRw[jj][e] = sR[jj][e];
                        else sTw[jj][e - 9] = sJ[jj][e - 9];
                    } else if (e < 9) {
                        const int i = e / 3, k = e - i * 3;
                        const float* a = &sRw[pa][0];
                        const float* b = &sR[jj][0];
                        sRw[jj][e] = a[i * 3] * b[k] + a[i * 3 + 1] * b[3 + k] + a[i * 3 + 2] * b[6 + k];
                    } else {
                        const int i = e - 9;
                        const float* a = &sRw[pa][0];
                        const float rel[3] = {sJ[jj][0] - sJ[pa][0], sJ[jj][1] - sJ[pa][1], sJ[jj][2] - sJ[pa][2]};
                        const float tv = a[i * 3] * rel[0] + a[i * 3 + 1] * rel[1] + a[i * 3 + 2] * rel[2];
                        sTw[jj][i] = tv + sTw[pa][i];
                    }
                }
                __syncthreads();
            }
        }
    } else
    for (int level = 0; level <= maxdepth; ++level) {
        const int n12 = S.sCnt[level] * 12;
        for (int t = tid; t < n12; t += 256) {
            const int slot = t / 12, e = t - slot * 12;
            const int jj = S.sList[level][slot];
            const int pa = sPar[jj];
            if (pa < 0) {                               // a root: its own rotation, its rest position
                if (e < 9) sRw[jj][e] = sR[jj][e];
                else sTw[jj][e - 9] = sJ[jj][e - 9];
            } else if (e < 9) {
                const int i = e / 3, k = e - i * 3;
                const float* a = &sRw[pa][0];
                const float* b = &sR[jj][0];
                sRw[jj][e] = a[i * 3] * b[k] + a[i * 3 + 1] * b[3 + k] + a[i * 3 + 2] * b[6 + k];
            } else {
                const int i = e - 9;
                const float* a = &sRw[pa][0];
                const float rel[3] = {sJ[jj][0] - sJ[pa][0], sJ[jj][1] - sJ[pa][1], sJ[jj][2] - sJ[pa][2]};
                const float tv = a[i * 3] * rel[0] + a[i * 3 + 1] * rel[1] + a[i * 3 + 2] * rel[2];
                sTw[jj][i] = tv + sTw[pa][i];
            }
        }
        __syncthreads();
    }
    // ---------------- phase C: recentring, then the per-joint read-outs ----------------
    if (w == 0) POSE_STAMP(6);
    if (tid == 0) {
        float R0[9], cc[3];
#pragma unroll
        for (int e = 0; e < 9; ++e) R0[e] = sX[e];
        if (c.center_joint >= 0) {
            float hc[3] = {sTw[c.center_joint][0] - sTw[0][0], sTw[c.center_joint][1] - sTw[0][1], sTw[c.center_joint][2] - sTw[0][2]};
            mat3_vec(R0, hc, cc);
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) cc[a] = -sTw[0][a];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) { sX[9 + a] = sTw[0][a]; sX[12 + a] = sX[24 + a] - cc[a]; }
    }
    __syncthreads();
    if (w == 0) POSE_STAMP(7);
    if (w == 2 && j < 24) xf[(size_t)p * 24 + j] = sX[j];
    if (w == 0 && j < NJ) {
        float R0[9], Rw[9], Rf[9], tp[3], tt[3], u[3];
#pragma unroll
        for (int e = 0; e < 9; ++e) { R0[e] = sX[e]; Rw[e] = sRw[j][e]; }
        const float pel[3] = {sX[9], sX[10], sX[11]};
        float Jv[3] = {sJ[j][0], sJ[j][1], sJ[j][2]};
        mat3_vec(Rw, Jv, u);
#pragma unroll
        for (int a = 0; a < 3; ++a) tp[a] = sTw[j][a] - u[a] - pel[a];
        mat3_mul(R0, Rw, Rf);
        mat3_vec(R0, tp, tt);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            sA[r * 4 + 0][j] = Rf[r * 3]; sA[r * 4 + 1][j] = Rf[r * 3 + 1]; sA[r * 4 + 2][j] = Rf[r * 3 + 2];
            sA[r * 4 + 3][j] = tt[r];
        }
    }
    if (w == 1 && j < NJ) {
        float R0[9], jj[3];
#pragma unroll
        for (int e = 0; e < 9; ++e) R0[e] = sX[e];
        const float pel[3] = {sX[9], sX[10], sX[11]}, o[3] = {sX[12], sX[13], sX[14]};
        float dj[3] = {sTw[j][0] - pel[0], sTw[j][1] - pel[1], sTw[j][2] - pel[2]};
        mat3_vec(R0, dj, jj);
#pragma unroll
        for (int a = 0; a < 3; ++a) { jj[a] += o[a]; j3d[((size_t)p * 127 + j) * 3 + a] = jj[a]; }
        float pr[2];
        project(&sX[15], jj, pr);
        j2d[((size_t)p * 127 + j) * 2] = pr[0];
        j2d[((size_t)p * 127 + j) * 2 + 1] = pr[1];
    }
    __syncthreads();
    if (w == 0) POSE_STAMP(8);
    flush(false);
    if (w == 0) POSE_STAMP(9);
#ifdef MHMR_LBS_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (w == 0) POSE_STAMP(10);          // this wave's stores have left
#endif
}

#ifdef MHMR_LBS_STAMPS      // tools/lbs_timeline.py: per-workgroup s_memtime stamps of wave 0 (debug build only, never in libmhmr.so)
__device__ unsigned long long* g_lbs_stamps;
#define LBS_STAMP(i)                                                                                           \
    do {                                                                                                       \
        if (g_lbs_stamps && threadIdx.x == 0) g_lbs_stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define LBS_STAMP(i)
#endif
// grid = Vp / 48 workgroups (one 48-vertex tile = three 16-vertex MFMA column blocks, for ALL persons of the launch) of
// LBS_NC compute waves + LBS_NL loader waves; at most one workgroup per CU (126 KB of LDS).
//   * loader waves do nothing but move the tile's slice of the blend basis (162 KiB) into LDS, an eighth of the k range at a time
//     (18 KiB each, 36 KiB the last one) with global_load_lds: four to five eighths are always in flight (LBS_LEAD), so HBM never waits
//     for a barrier; their vmcnt stream holds only these copies, so the landing wait is an exact count.  They also bring the persons' [R0 | translation | K]
//     records into LDS once.
//   * compute wave w owns person group g0 + w (16 persons).  Its A operands (F16, A16: L2-resident, the same for every tile) are
//     ordinary loads one k step / one component ahead of their MFMAs; every loaded A fragment is multiplied against all THREE
//     vertex blocks.  That reuse is the point of the 48-vertex tile: with 16-vertex tiles (three workgroups per CU, the previous
//     form, 48.6 us at 160 persons) each fragment fed 9 MFMAs and the vector-memory path (64 B/clk per CU) moved 57 B/clk in the
//     blend and 2.7x its peak in the skinning products -- tools/lbs_timeline.py showed 8 us per group for 1.9 us of MFMAs no
//     matter how far ahead the loads were issued.  The B operands come from LDS one block ahead.
// The skinning GEMM and the last eighth of the blend keep fp32 accuracy on the 16-bit matrix pipe (x . y = xh.yh + xl.yh + xh.yl).
struct __attribute__((packed, aligned(4))) Vec3 { float x, y, z; };
struct __attribute__((packed, aligned(4))) Vec2 { float x, y; };
constexpr int LBS_TV = 48, LBS_NST = LBS_TV / 16;      // vertices per tile, 16-vertex MFMA column blocks per tile
constexpr int LBS_NC = 10, LBS_NL = 2;                 // compute waves (one person group each), loader waves
constexpr int LBS_KB = 512;                            // padded blend depth (486 pose + betas + 10 expression <= 512)
constexpr int LBS_NX = 72;                             // extra joints 55..126 (virtual vertices in the tiles from c.Vl on)
constexpr int LBS_NS = LBS_KB / 32;                    // k steps of 32
constexpr int LBS_NE = 8;                              // k eighths (2 steps each)
constexpr int LBS_ROW = LBS_TV * 8 * 2;                // bytes of one (k block, part, axis) row of the tile: 48 vertices x 8 k x f16
constexpr int LBS_EBYTES = (LBS_KB / 8 / LBS_NE) * 6 * LBS_ROW;       // one ring slot = one eighth of the tile's slice as an f16 PAIR: 36 KiB
constexpr int LBS_EOPS = LBS_EBYTES / 1024 / LBS_NL;                  // 1-KiB copies per loader wave per eighth (pair form)
// Precision of the blend (round 4): the first LBS_NE - 1 eighths (k < 448: pose correctives only, millimetres) carry the HIGH half of
// the basis alone and are multiplied by the high half of the feature alone -- ONE product per term; BOTH factors are rounded (2^-11 each), so |error| <= 2^-10 sum |F| |D| over the
// 448 terms (worst case tens of micrometres on the synthetic basis; rms 4e-6 m, worst 2.5e-5 m measured: the max-abs gate of
// tests/test_gpu_kernels.py::test_lbs_max_abs_gate_160_persons_x_20_seeds is 5e-5 m = 0.05 mm) -- and the last eighth (k 448..511: the last pose columns and ALL shape / expression
// directions, centimetres times |beta| up to 3) keeps the f16 pair and the three products (5e-7 m).  That is 180 instead of 432 blend
// MFMAs per person group and tile, 162 instead of 288 KiB of basis per tile from HBM and LDS.
constexpr int LBS_EBYTES_HI = LBS_EBYTES / 2;                         // an eighth with the high half only: 18 KiB
constexpr int LBS_EOPS_HI = LBS_EOPS / 2;
constexpr int LBS_TILE_BYTES = (LBS_NE - 1) * LBS_EBYTES_HI + LBS_EBYTES;   // basis bytes of one tile: 162 KiB
constexpr int LBS_WBYTES = 8 * 2 * LBS_TV * 8 * 2;     // the tile's dense skin weights, hi + lo: 12 KiB
constexpr int LBS_XREC = 24 * 4;                       // bytes of one person's record in ws_xf
// LDS image of the tile's slice (round 4: no ring any more).  With the high halves alone the first seven eighths are 18 KiB each and
// ALL fit: every one has a slot of its own, five are requested before the first barrier (two more as the vmcnt budget of 63 allows), so
// at small person counts -- where the kernel is nothing but this stream -- HBM sees up to 90 KiB per workgroup in flight instead of 36.
// The last eighth (pair form, 36 KiB) re-uses the slots of eighths 0 and 1, the skin weights (12 KiB) the slot of eighth 2.
constexpr int LBS_LEAD = 5;                                            // eighths requested before the first barrier
__host__ __device__ constexpr int lbs_eoff(int e) { return e < LBS_NE - 1 ? e * LBS_EBYTES_HI : 0; }
constexpr int LBS_WOFF = 2 * LBS_EBYTES_HI;                            // the skin weights' place
constexpr int LBS_XOFF = (LBS_NE - 1) * LBS_EBYTES_HI;                 // the person records behind the seven slots
constexpr int LBS_LDS = LBS_XOFF + LBS_NC * 16 * LBS_XREC;
static_assert(LBS_EBYTES % (1024 * LBS_NL) == 0 && LBS_EOPS == 18 && LBS_EOPS_HI == 9 && LBS_WBYTES / 1024 / LBS_NL == 6 && LBS_NE == 8 &&
              LBS_LEAD == 5 && LBS_WBYTES <= LBS_EBYTES_HI,
              "the landing waits below are written for 9 / 18 copies per wave and eighth, 6 for the weights, five eighths ahead");
static_assert(LBS_LDS <= 160 * 1024, "LDS");
static_assert(LBS_KB_POSE == LBS_KB, "pose kernel staging");

__device__ __forceinline__ void lbs_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// loader wave lw: the tile's basis slice, eighth by eighth, + the person records
template <bool FUSED>
__device__ __forceinline__ void lbs_loader(const mhmr_lbs_consts& c, const float* __restrict__ xf, int P, int ngroups, int g0, char* smem,
                                           int lw, int tile, int* __restrict__ sync, int ntiles, int Pp) {
    const int lane = threadIdx.x & 63;
    char* xrec = smem + LBS_XOFF;
    // (fused launch: the records are the pose role's OUTPUT -- every compute wave fetches its own group's after its ready wait, and this
    // wave's vmcnt stream holds the basis copies only, which is all the landing waits below count)
    for (int i = FUSED ? LBS_NC : lw; i < LBS_NC; i += LBS_NL) {
        const int g = g0 + i;
        if (g >= ngroups) break;
        const int bytes = min(16, P - 16 * g) * LBS_XREC;
        const char* src = (const char*)(xf + (size_t)g * 16 * 24) + lane * 16;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (k * 1024 + lane * 16 < bytes) glds16(src + k * 1024, xrec + i * (16 * LBS_XREC) + k * 1024);
    }
    // tile-major basis: the tile's slice is one contiguous block, an eighth is 36 consecutive KiB; source and LDS image lane-linear
    const _Float16* bsrc = (const _Float16*)c.basis16 + (size_t)tile * (LBS_TILE_BYTES / 2) + lane * 8;
    auto dma_e = [&](int e) {          // (e is a compile-time value at every call: the loops around are unrolled)
        char* dst = smem + lbs_eoff(e);
        const int nops = e == LBS_NE - 1 ? LBS_EOPS : LBS_EOPS_HI;
#pragma unroll
        for (int k = 0; k < LBS_EOPS; ++k) {
            if (k < nops) {
                const int i = lw * nops + k;
                glds16(bsrc + (size_t)e * (LBS_EBYTES_HI / 2) + i * 512, dst + i * 1024);
            }
        }
    };
#pragma unroll
    for (int e = 0; e < LBS_LEAD; ++e) dma_e(e);
    if constexpr (FUSED) {
        lbs_barrier();          // the ready barrier: compute wave 0 has seen every person row published (lbs_compute)
        // This workgroup reads the ready flags no more.  The LAST vertex workgroup to get here puts the flags and the ticket back to
        // zero -- the workspace leaves the launch as it entered it (no memset launch, no host-side epoch: the call stays capturable).
        if (lw == 0) {
            int t = 0;
            if (lane == 0) t = __hip_atomic_fetch_add(sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t == ntiles - 1) {
                for (int i = lane; i < Pp; i += 64) __hip_atomic_store(sync + 1 + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 0) __hip_atomic_store(sync, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    constexpr int WOPS = LBS_WBYTES / 1024 / LBS_NL;
#pragma unroll
    for (int e = 0; e < LBS_NE; ++e) {
        // eighth e (and everything older: the records) has landed when at most the copies requested AFTER it are outstanding (this
        // wave's vmcnt stream holds nothing but these copies, and loads are counted in order).  Requests: eighths 0..4 up front; after
        // barrier 0 / 1 eighths 5 / 6 (own slots); after barrier 2 the last eighth (18 copies, into the slots of eighths 0 and 1, read
        // for the last time before barriers 1 and 2); after barrier 3 the weights (6 copies, the slot of eighth 2).
        //   e:        0   1   2   3             4                 5             6        7
        //   younger:  36  36  36  27 + 18 = 45  18 + 18 + 6 = 42  9 + 18 + 6    18 + 6   (0: the weights are published with it)
        if (e <= 2) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
        else if (e == 3) asm volatile("s_waitcnt vmcnt(45)" ::: "memory");
        else if (e == 4) asm volatile("s_waitcnt vmcnt(42)" ::: "memory");
        else if (e == 5) asm volatile("s_waitcnt vmcnt(33)" ::: "memory");
        else if (e == 6) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lbs_barrier();                                   // ... and every compute wave is done with eighth e - 1
        if (e + LBS_LEAD < LBS_NE) dma_e(e + LBS_LEAD);
        else if (e == 3) {
            // the tile's dense skin weights (12 KiB, tile-major [8 joint blocks][hi|lo][48][8] = MFMA operand order); the last landing
            // wait (vmcnt(0)) and barrier publish them together with eighth 7
            const _Float16* wsrc = (const _Float16*)c.skin16 + (size_t)tile * (LBS_WBYTES / 2) + lane * 8;
            char* dst = smem + LBS_WOFF;
#pragma unroll
            for (int k = 0; k < WOPS; ++k) {
                const int i = lw * WOPS + k;
                glds16(wsrc + i * 512, dst + i * 1024);
            }
        }
    }
}

// compute wave: person group g (16 persons) against the tile's 48 vertices
template <bool FUSED>
__device__ __forceinline__ void lbs_compute(const mhmr_lbs_consts& c, const _Float16* __restrict__ F16, const _Float16* __restrict__ A16,
                                            int P, int ngroups, int g, int w, float* __restrict__ v3d, float* __restrict__ v2d,
                                            float* __restrict__ j3d, float* __restrict__ j2d, char* smem, int tile,
                                            const float* __restrict__ xf, const int* __restrict__ sync) {
    typedef Op<MHMR_DT_F16>::V8 H8;
    const int lane = threadIdx.x & 63;
    const int g4 = lane >> 4, l15 = lane & 15;
    const int v0 = tile * LBS_TV;
    if constexpr (FUSED) {
        // The pose role of THIS launch writes F16 / A16 / xf.  Compute wave 0 alone polls the ready flags (all person rows of the launch)
        // and takes the acquire fence -- ONE cache invalidation per workgroup: the per-wave form of the first build cost 2 240 of them and
        // ran 30 us SLOWER than two launches -- then the workgroup's ready barrier publishes "poses visible" to the other eleven waves.
        // The loader waves meanwhile stream the basis: by the time the poses exist, five eighths of the tile have landed.
        if (w == 0) {
            int spins = 0;      // (bounded: ~0.5 s of polling means the protocol is broken -- abort the kernel loudly rather than hang the device)
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int r = lane + 64 * i;
                    if (r < 16 * ngroups) ok = ok && __hip_atomic_load(sync + 1 + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1;
                }
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1 << 21)) __builtin_trap();
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        lbs_barrier();                                              // the ready barrier (every wave of the workgroup passes it once)
        float* xr = (float*)(smem + LBS_XOFF + w * (16 * LBS_XREC));
        const int nfl = min(16, P - 16 * g) * 24;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int k = lane + 64 * i;
            if (k < nfl) xr[k] = xf[(size_t)g * (16 * 24) + k];
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    // A operands of k step sg (fragment-major: 1 KiB per (group, part, k step)), [set][hi | lo]
    H8 A[2][2];
    auto load_a = [&](int set, int sg) {          // (the low half only where the pair form of the basis is: the last eighth's two steps)
        const _Float16* fh = F16 + ((((size_t)g * 2) * LBS_NS + sg) * 64 + lane) * 8;
        A[set][0] = *(const H8*)fh;
        if (sg >= LBS_NS - 2) A[set][1] = *(const H8*)(fh + (size_t)LBS_NS * 512);
    };
    // skinning operands of component cmp: [set][xh t0, xh t1, xl t0, xl t1]
    H8 X[2][4];
    const size_t part = (size_t)ngroups * 1024;
    auto load_x = [&](int set, int cmp) {
        const _Float16* ap = A16 + ((size_t)g * 2 * 64 + lane) * 8 + (size_t)(2 * cmp) * part;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            X[set][t] = *(const H8*)(ap + 512 * t);
            X[set][2 + t] = *(const H8*)(ap + part + 512 * t);
        }
    };

    load_a(0, 0);
    f32x4 acc[LBS_NST][3];
#pragma unroll
    for (int st = 0; st < LBS_NST; ++st)
#pragma unroll
        for (int a = 0; a < 3; ++a) acc[st][a] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- v_posed - v_template = F . D ----
#pragma unroll
    for (int e = 0; e < LBS_NE; ++e) {
        lbs_barrier();
        LBS_STAMP(1 + e);
        const bool pair = e == LBS_NE - 1;                   // compile-time after unrolling
        const int rows = pair ? 6 : 3;                       // (part, axis) rows per k block of this eighth
        const char* slot = smem + lbs_eoff(e) + g4 * (rows * LBS_ROW) + l15 * 16;     // + (4 s2) * rows + (part * 3 + axis) rows + st * 256
        H8 B[2][6];
        auto read_b = [&](int set, int j) {           // j = 3 * s2 + st: the (part, axis) fragments of one vertex block of one k step
            const char* b = slot + (j / 3) * (4 * rows * LBS_ROW) + (j % 3) * 256;
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (k < rows) B[set][k] = *(const H8*)(b + k * LBS_ROW);
        };
        read_b(0, 0);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int s2 = j / 3, st = j % 3, sg = 2 * e + s2;
            if (st == 0) {
                if (sg + 1 < LBS_NS) load_a((sg + 1) & 1, sg + 1);
                else load_x(0, 0);                        // the last step: the skinning phase's first operands
            }
            if (j + 1 < 6) read_b((j + 1) & 1, j + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int prod = 0; prod < 3; ++prod)
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    if (prod == 0 || pair)
                        acc[st][a] = Op<MHMR_DT_F16>::mfma16(A[sg & 1][prod == 1 ? 1 : 0], B[j & 1][prod == 2 ? 3 + a : a], acc[st][a]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    LBS_STAMP(9);
    // (no early exit for padding vertices / persons: a lane is a vertex COLUMN of the products but a person ROW of the A operands,
    // so every lane stays active through the MFMAs; only the stores are guarded)
    // ---- T = sum_j w[v][j] [R | t]_j component by component, folded into the rigid transform as it arrives ----
    float o3[LBS_NST][3][4];
#pragma unroll
    for (int st = 0; st < LBS_NST; ++st) {
        const int v = v0 + 16 * st + l15;                                                 // < Vp: the template is padded with zeros
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float vt = c.vtemp[a * c.Vp + v];                                      // fp32 template
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[st][a][r] = acc[st][a][r] * (1.0f / 1024.0f) + vt;
        }
    }
    // B operand of the skinning GEMM = the tile's dense skin weights, from LDS (loader): lane (v = l15, k group g4) of vertex block st
    // holds joints 8 (4 t + g4) + 0..7; [set][wh t0, wh t1, wl t0, wl t1], one block ahead
    const char* wlds = smem + LBS_WOFF + g4 * (2 * LBS_ROW) + l15 * 16;
    H8 W[2][4];
    auto read_w = [&](int set, int st) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            W[set][t] = *(const H8*)(wlds + (4 * t) * (2 * LBS_ROW) + st * 256);
            W[set][2 + t] = *(const H8*)(wlds + (4 * t) * (2 * LBS_ROW) + LBS_ROW + st * 256);
        }
    };
    read_w(0, 0);
#pragma unroll
    for (int cmp = 0; cmp < 12; ++cmp) {
        if (cmp + 1 < 12) load_x((cmp + 1) & 1, cmp + 1);
        const H8* x = X[cmp & 1];
        const int a = cmp >> 2, k = cmp & 3;
#pragma unroll
        for (int st = 0; st < LBS_NST; ++st) {
            const int j = cmp * LBS_NST + st;
            if (j + 1 < 12 * LBS_NST) read_w((j + 1) & 1, (st + 1) % LBS_NST);
            __builtin_amdgcn_sched_barrier(0);
            const H8* wf = W[j & 1];
            f32x4 T = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                T = Op<MHMR_DT_F16>::mfma16(x[t], wf[t], T);
                T = Op<MHMR_DT_F16>::mfma16(x[2 + t], wf[t], T);
                T = Op<MHMR_DT_F16>::mfma16(x[t], wf[2 + t], T);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (k == 0) o3[st][a][r] = T[r] * acc[st][0][r];
                else if (k == 1) o3[st][a][r] += T[r] * acc[st][1][r];
                else if (k == 2) o3[st][a][r] += T[r] * acc[st][2][r];
                else o3[st][a][r] += T[r];
                asm volatile("" : "+v"(o3[st][a][r]));      // fold HERE: LLVM otherwise sinks these FMAs into the guarded store blocks
            }                                               // below and keeps all 36 products alive (340 B of scratch per lane)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    LBS_STAMP(10);
    // ---- camera translation (fp32, exactly where the reference adds it: smpl_layer.py:139-140), projection, stores ----
    const char* xrec = smem + LBS_XOFF + w * (16 * LBS_XREC);
    if (v0 < c.Vl) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = 16 * g + 4 * g4 + r;
            if (p >= P) continue;
            const f32x4* Xr = (const f32x4*)(xrec + (4 * g4 + r) * LBS_XREC + 48);      // [o (3), K (9)]
            const f32x4 x0 = Xr[0], x1 = Xr[1], x2 = Xr[2];
#pragma unroll
            for (int st = 0; st < LBS_NST; ++st) {
                const int v = v0 + 16 * st + l15;
                if (v >= c.V) continue;
                const float ox = o3[st][0][r] + x0[0], oy = o3[st][1][r] + x0[1], oz = o3[st][2][r] + x0[2];
                *(Vec3*)(v3d + ((size_t)p * c.V + v) * 3) = Vec3{ox, oy, oz};          // one 12-byte store (dword-aligned)
                // perspective_projection (utils/camera.py:14-27) with one reciprocal instead of three divisions (<= 1 ulp apart)
                const float iz = __builtin_amdgcn_rcpf(oz);
                const float yx = ox * iz, yy = oy * iz, yz = oz * iz;
                *(Vec2*)(v2d + ((size_t)p * c.V + v) * 2) = Vec2{x0[3] * yx + x1[0] * yy + x1[1] * yz, x1[2] * yx + x1[3] * yy + x2[0] * yz};
            }
        }
    } else {
        // A tile of EXTRA JOINTS (joints 55..126: 21 vertices picked by id, 51 barycentric face landmarks -- smplx vertices2landmarks):
        // extra joint e = 16 t + l15 owns column l15 of the tile's three vertex blocks (copies of its three corner vertices' operand
        // columns, packing.pack_smplx), so this lane holds all three posed corners.  Affine in the corners: sum_k b_k (x_k + o) and, where
        // the weights do not sum to exactly one, the (1 - sum) (o - R0 pelvis) part of the placement; a picked vertex is (1, 0, 0): the
        // sums below then reproduce the vertex path bit for bit (0 * x = 0, x - 0 = x).
        const int e = ((v0 - c.Vl) / LBS_TV) * 16 + l15;
        if (e < LBS_NX) {
            const float b0 = c.xbary[e * 3], b1 = c.xbary[e * 3 + 1], b2 = c.xbary[e * 3 + 2];
            const float rest = 1.f - ((b0 + b1) + b2);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = 16 * g + 4 * g4 + r;
                if (p >= P) continue;
                const float* X = (const float*)(xrec + (4 * g4 + r) * LBS_XREC);          // [R0 (9), pelvis (3), o (3), K (9)]
                const f32x4* Xr = (const f32x4*)(X + 12);
                const f32x4 x0 = Xr[0], x1 = Xr[1], x2 = Xr[2];
                float rp[3];
                mat3_vec(X, X + 9, rp);                                                     // R0 . pelvis
                float q[3];
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    q[a] = (((b0 * o3[0][a][r] + b1 * o3[1][a][r]) + b2 * o3[2][a][r]) + x0[a]) - rest * rp[a];
                *(Vec3*)(j3d + ((size_t)p * 127 + 55 + e) * 3) = Vec3{q[0], q[1], q[2]};
                const float iz = __builtin_amdgcn_rcpf(q[2]);
                const float yx = q[0] * iz, yy = q[1] * iz, yz = q[2] * iz;
                *(Vec2*)(j2d + ((size_t)p * 127 + 55 + e) * 2) = Vec2{x0[3] * yx + x1[0] * yy + x1[1] * yz, x1[2] * yx + x1[3] * yy + x2[0] * yz};
            }
        }
    }
    LBS_STAMP(11);
}

// ONE launch for the whole layer (round 5; opt-in, see lbs_forward_impl for the measurement that keeps it off): workgroups [0, npose) are the pose role -- twelve persons each, one per wave, exactly
// lbs_pose_person above -- and workgroups [npose, npose + tiles) the vertex role.  The two used to be two launches: 10.5 us of pose
// latency chain + a launch gap in front of a vertex kernel whose first 5 us are nothing but the basis stream's ramp.  Here the vertex
// workgroups start their basis DMA at once and only their compute waves wait for the poses (per-person ready flags in `sync`).
// Progress: workgroups are dispatched in index order, so the pose workgroups are resident before any waiting one; the launcher also
// keeps the grid within one workgroup per CU (everything co-resident), and falls back to the two launches otherwise.
// sync [1 + Pp] ints: [0] a ticket, [1 + p] the ready flag of person row p; ZERO on entry, zero again on exit (lbs_loader).
__global__ __launch_bounds__(64 * (LBS_NC + LBS_NL), 3) void lbs_fused_kernel(const mhmr_lbs_consts c, const float* __restrict__ rotvec,
                                                                              const float* __restrict__ betas, const float* __restrict__ expr,
                                                                              const float* __restrict__ loc, const float* __restrict__ dist,
                                                                              const float* __restrict__ Kmat, const int* __restrict__ det_b,
                                                                              int P, int Pp, _Float16* __restrict__ F16,
                                                                              _Float16* __restrict__ A16, float* __restrict__ xf,
                                                                              float* __restrict__ v3d, float* __restrict__ v2d,
                                                                              float* __restrict__ j3d, float* __restrict__ j2d,
                                                                              float* __restrict__ transl, int* __restrict__ sync, int npose) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if ((int)blockIdx.x < npose) {
        const int p = (int)blockIdx.x * (LBS_NC + LBS_NL) + w;
        if (p < Pp) {
            PoseLds& L = *(PoseLds*)(smem + (size_t)w * sizeof(PoseLds));
            lbs_pose_person(c, rotvec, betas, expr, loc, dist, Kmat, det_b, P, Pp, F16, A16, xf, j3d, j2d, transl, p, L);
        }
        // Publish the workgroup's twelve rows with ONE release: every wave's stores have reached the L2 (vmcnt(0)) before the barrier,
        // wave 0 then writes the L2 back once (a release fence per WAVE is a cache write-back per person: 160 of them serialised per
        // XCD in the first build) and raises the flags.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lbs_barrier();
        if (w == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const int lane = threadIdx.x & 63, r = (int)blockIdx.x * (LBS_NC + LBS_NL) + lane;
            if (lane < LBS_NC + LBS_NL && r < Pp) __hip_atomic_store(sync + 1 + r, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const int tile = (int)blockIdx.x - npose, ngroups = Pp / 16, ntiles = (int)gridDim.x - npose;
    if (w >= LBS_NC) {
        lbs_loader<true>(c, xf, P, ngroups, 0, smem, w - LBS_NC, tile, sync, ntiles, Pp);
    } else if (w < ngroups) {
        lbs_compute<true>(c, F16, A16, P, ngroups, w, w, v3d, v2d, j3d, j2d, smem, tile, xf, sync);
    } else {
#pragma unroll
        for (int e = 0; e < LBS_NE + 1; ++e) lbs_barrier();         // the ready barrier + the eight of the blend
    }
}
static_assert((LBS_NC + LBS_NL) * sizeof(PoseLds) <= LBS_LDS, "pose role staging of twelve waves");

__global__ __launch_bounds__(64 * (LBS_NC + LBS_NL), 3) void lbs_vertex_kernel(const mhmr_lbs_consts c, const _Float16* __restrict__ F16,
                                                                               const _Float16* __restrict__ A16, const float* __restrict__ xf,
                                                                               int P, int Pp, int g0, float* __restrict__ v3d,
                                                                               float* __restrict__ v2d, float* __restrict__ j3d,
                                                                               float* __restrict__ j2d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ngroups = Pp / 16;
    LBS_STAMP(0);
#ifdef MHMR_LBS_STAMPS
    if (g_lbs_stamps && threadIdx.x == 0) g_lbs_stamps[(size_t)blockIdx.x * 16 + 12] = wall_clock64();      // (for tools/lbs_pose_timeline.py: the gap behind the pose kernel)
#endif
    // every wave passes the same LBS_NE barriers
    if (w >= LBS_NC) {
        lbs_loader<false>(c, xf, P, ngroups, g0, smem, w - LBS_NC, (int)blockIdx.x, nullptr, 0, Pp);
    } else if (g0 + w < ngroups) {
        lbs_compute<false>(c, F16, A16, P, ngroups, g0 + w, w, v3d, v2d, j3d, j2d, smem, (int)blockIdx.x, nullptr, nullptr);
    } else {
#pragma unroll
        for (int e = 0; e < LBS_NE; ++e) lbs_barrier();
    }
}

}  // namespace

#ifdef MHMR_LBS_STAMPS
extern "C" int mhmr_debug_lbs_stamps(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_lbs_stamps), &p, sizeof(p)); }
extern "C" int mhmr_debug_pose_stamps(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pose_stamps), &p, sizeof(p)); }
#endif

static int lbs_forward_impl(const mhmr_lbs_consts* c, const float* rotvec, const float* betas, const float* expr,
                            const float* loc, const float* dist, const float* Kmat, const int* det_b, int P, float* ws_F,
                            float* ws_A, float* ws_xf, float* v3d, float* v2d, float* j3d, float* j2d, float* transl,
                            int* ws_sync, void* stream) {
    if (!c || P < 0) return MHMR_ERR_BAD_ARG;
    if (P == 0) return 0;
    if (c->Vp % LBS_TV || c->Vl % LBS_TV || c->Vl < c->V || c->Vp != c->Vl + LBS_TV * ((LBS_NX + 15) / 16) || c->Kb != LBS_KB ||
        c->Kb < 486 + c->nb + 10 || !c->skin16 || !c->xbary)
        return MHMR_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int Pp = (P + 15) / 16 * 16;
    {   // 141 KB of dynamic LDS: above the default per-kernel limit.  The attribute is per DEVICE (DeviceOnce, mhmr_internal.h: one
        // driver call per device and process, not one per forward)
        static DeviceOnce once;
        int dev = 0;
        const int need = once.need(&dev);
        if (need == -2) return MHMR_ERR_BAD_ARG;
        if (need >= 0) {
            hipError_t e = hipFuncSetAttribute((const void*)lbs_vertex_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LBS_LDS);
            if (e != hipSuccess) return (int)e;
            e = hipFuncSetAttribute((const void*)lbs_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LBS_LDS);
            if (e != hipSuccess) return (int)e;
            once.mark(dev);
        }
    }
    // The fused launch is CORRECT (bit-identical outputs, tests/test_gpu_kernels.py) and SLOWER than the two launches on this chip
    // (profiles/r05_session_c_lbs_fused_slp.txt: 48.7 vs 40.2 us at 160 persons, 38.3 vs 30.2 at 20, 32.2 vs 26.6 at 1), so it runs only
    // through its own entry point (mhmr_lbs_forward_fused; Model takes it with MHMR_LBS_FUSED=1).  Why it loses: (a) across XCDs the poses become visible only through an L2 write-back on the
    // producer side and an L2 invalidation on the consumer side (one each per workgroup in this form; one per WAVE in the first form: 76 us);
    // (b) the vertex role's time at small person counts is not the basis stream's ramp, as assumed, but its SEQUENCING -- only five of
    // the eight eighths fit in flight before the poses exist (LDS, vmcnt budget), the other three and the skin weights are latency-bound
    // rounds that need the compute waves' barriers, i.e. the poses: what overlaps is ~5 us, what the fences cost is about the same.
    const int npose = (Pp + LBS_NC + LBS_NL - 1) / (LBS_NC + LBS_NL), ntiles = c->Vp / LBS_TV;
    if (ws_sync && Pp / 16 <= LBS_NC && npose + ntiles <= mhmr_cu_count()) {
        prof_begin(PROF_LBS, s);
        hipLaunchKernelGGL(lbs_fused_kernel, dim3(npose + ntiles), dim3(64 * (LBS_NC + LBS_NL)), LBS_LDS, s, *c, rotvec, betas, expr, loc, dist,
                           Kmat, det_b, P, Pp, (_Float16*)ws_F, (_Float16*)ws_A, ws_xf, v3d, v2d, j3d, j2d, transl, ws_sync, npose);
        prof_end(PROF_LBS, s, (double)P);
        MHMR_CHECK_LAUNCH();
        return 0;
    }
    // four waves per person (round 6); MHMR_LBS_POSE1=1: the one-wave form (A/B measurements; bit-identical results)
    static const bool pose1 = getenv("MHMR_LBS_POSE1") && atoi(getenv("MHMR_LBS_POSE1")) != 0;
    if (pose1)
        hipLaunchKernelGGL(lbs_pose_kernel, dim3(Pp), dim3(64), 0, s, *c, rotvec, betas, expr, loc, dist, Kmat, det_b, P, Pp,
                           (_Float16*)ws_F, (_Float16*)ws_A, ws_xf, j3d, j2d, transl);
    else
        hipLaunchKernelGGL(lbs_pose4_kernel, dim3(Pp), dim3(256), 0, s, *c, rotvec, betas, expr, loc, dist, Kmat, det_b, P, Pp,
                           (_Float16*)ws_F, (_Float16*)ws_A, ws_xf, j3d, j2d, transl);
    MHMR_CHECK_LAUNCH();
    // one launch covers LBS_NC person groups (160 persons); more persons take further launches over the same tiles
    const int ngroups = Pp / 16;
    prof_begin(PROF_LBS, s);
    for (int g0 = 0; g0 < ngroups; g0 += LBS_NC)
        hipLaunchKernelGGL(lbs_vertex_kernel, dim3(c->Vp / LBS_TV), dim3(64 * (LBS_NC + LBS_NL)), LBS_LDS, s, *c, (const _Float16*)ws_F,
                           (const _Float16*)ws_A, ws_xf, P, Pp, g0, v3d, v2d, j3d, j2d);
    prof_end(PROF_LBS, s, (double)P);
    MHMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int mhmr_lbs_forward(const mhmr_lbs_consts* c, const float* rotvec, const float* betas, const float* expr,
                                const float* loc, const float* dist, const float* Kmat, const int* det_b, int P, float* ws_F,
                                float* ws_A, float* ws_xf, float* v3d, float* v2d, float* j3d, float* j2d, float* transl,
                                void* stream) {
    return lbs_forward_impl(c, rotvec, betas, expr, loc, dist, Kmat, det_b, P, ws_F, ws_A, ws_xf, v3d, v2d, j3d, j2d, transl, nullptr, stream);
}

extern "C" int mhmr_lbs_forward_fused(const mhmr_lbs_consts* c, const float* rotvec, const float* betas, const float* expr,
                                      const float* loc, const float* dist, const float* Kmat, const int* det_b, int P, float* ws_F,
                                      float* ws_A, float* ws_xf, float* v3d, float* v2d, float* j3d, float* j2d, float* transl,
                                      int* ws_sync, void* stream) {
    if (!ws_sync) return MHMR_ERR_BAD_ARG;
    return lbs_forward_impl(c, rotvec, betas, expr, loc, dist, Kmat, det_b, P, ws_F, ws_A, ws_xf, v3d, v2d, j3d, j2d, transl, ws_sync, stream);
}
