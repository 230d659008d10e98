// SMPL-X linear blend skinning + camera placement (reference blocks/smpl_layer.py:47-155 -> smplx.SMPLX.forward
// -> lbs), three launches:
//
//  1. lbs_pose_kernel   (one wave per person)  Rodrigues x55, pose feature, joint regression from the
//     pre-contracted regressor (J = J0 + JS.[betas, expr]), kinematic chain, root rotation / recentring /
//     back-projected translation folded into the per-joint skinning transforms, 55 posed joints + projection.
//  2. lbs_vertex_kernel (the HBM-bound one)   v_posed = v_template + F . D as ONE GEMM, F[p] = [pose_feature(486) | betas | expr]
//     and D = [posedirs ; shapedirs ; exprdirs], computed to fp32 accuracy on the 16-bit matrix pipe: D (scaled by 2^10 into
//     the f16 normal range) is stored as an f16 pair hi + lo, F is split hi + lo on the fly, and
//     F.D = Fh.Dh + Fl.Dh + Fh.Dl (three v_mfma_f32_16x16x32_f16, fp32 accumulate; the dropped Fl.Dl term is 2^-22 relative).
//     The fp32 MFMA (v_mfma_f32_16x16x4_f32, 1/16 of the 16-bit rate) made this kernel matrix-bound: 108 us for 160 persons,
//     74 us now.  Ablations of the 74 us: K loop 37 us (64 MB of correctives, ~2 waves per SIMD: latency-bound), skinning
//     gathers 17 us, stores 5 us, rest 15 us.  Two re-tilings that raise the wave count were built and measured SLOWER (one
//     16-person group per wave with per-wave basis loads: 82 us, the CU pulls 12x the unique bytes through its L1; the same
//     with the basis tile shared through LDS and a barrier per K step: 94 us; with a 4-slot LDS-DMA ring three steps ahead: 83 us at
//     160 persons although 23 instead of 32 us at 20 -- the per-wave skinning set-up then repeats for every 16 persons).  Accumulators leave the MFMA laid out as
//     (vertex = lane & 15, 4 persons per quad), then the <=K-sparse skinning blend, the folded rigid transform and the pinhole
//     projection run per lane and v3d / v2d are written out.
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
                                                      const float* __restrict__ Kmat, const int* __restrict__ det_b, int P,
                                                      float* __restrict__ F, float* __restrict__ Afold, float* __restrict__ xf,
                                                      float* __restrict__ j3d, float* __restrict__ j2d,
                                                      float* __restrict__ transl_out) {
    __shared__ float sR[NJ][9], sJ[NJ][3], sRw[NJ][9], sTw[NJ][3], sX[33];
    const int p = blockIdx.x, j = threadIdx.x;
    float* Fp = F + (size_t)p * c.Kb;
    if (p >= P) {  // padding rows of the feature matrix
        for (int k = j; k < c.Kb; k += 64) Fp[k] = 0.f;
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
            if (j >= 1) Fp[(j - 1) * 9 + e] = r - id;
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
        Fp[k] = v;
    }
    __syncthreads();
    if (j == 0) {
        // kinematic chain (parents[i] < i), world rotation / translation per joint
#pragma unroll
        for (int e = 0; e < 9; ++e) sRw[0][e] = sR[0][e];
#pragma unroll
        for (int a = 0; a < 3; ++a) sTw[0][a] = sJ[0][a];
        for (int i = 1; i < NJ; ++i) {
            const int pa = c.parents[i];
            float Rp[9], Rl[9], Rn[9], rel[3], t[3];
#pragma unroll
            for (int e = 0; e < 9; ++e) { Rp[e] = sRw[pa][e]; Rl[e] = sR[i][e]; }
            mat3_mul(Rp, Rl, Rn);
#pragma unroll
            for (int a = 0; a < 3; ++a) rel[a] = sJ[i][a] - sJ[pa][a];
            mat3_vec(Rp, rel, t);
#pragma unroll
            for (int e = 0; e < 9; ++e) sRw[i][e] = Rn[e];
#pragma unroll
            for (int a = 0; a < 3; ++a) sTw[i][a] = t[a] + sTw[pa][a];
        }
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
    if (j < NJ) {
        float R0[9], Rw[9], Rf[9], tp[3], tt[3], u[3], jj[3];
#pragma unroll
        for (int e = 0; e < 9; ++e) { R0[e] = sX[e]; Rw[e] = sRw[j][e]; }
        const float pel[3] = {sX[9], sX[10], sX[11]}, o[3] = {sX[12], sX[13], sX[14]};
        // A'_j = [R_w | t_w - R_w J_j]; folded: [R0 R_w | R0 (t' - pelvis) + o]
        float Jv[3] = {sJ[j][0], sJ[j][1], sJ[j][2]};
        mat3_vec(Rw, Jv, u);
#pragma unroll
        for (int a = 0; a < 3; ++a) tp[a] = sTw[j][a] - u[a] - pel[a];
        mat3_mul(R0, Rw, Rf);
        mat3_vec(R0, tp, tt);
        float* ap = Afold + ((size_t)p * NJ + j) * 12;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            ap[r * 4 + 0] = Rf[r * 3]; ap[r * 4 + 1] = Rf[r * 3 + 1]; ap[r * 4 + 2] = Rf[r * 3 + 2];
            ap[r * 4 + 3] = tt[r] + o[r];
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

// grid (ceil(Pp / 64) person slabs [fastest], Vp / 64); 4 waves, each a 16-vertex group; up to 4 person groups of 16 per wave.
__global__ __launch_bounds__(256) void lbs_vertex_kernel(const mhmr_lbs_consts c, const float* __restrict__ F,
                                                         const float* __restrict__ Afold, const float* __restrict__ xf, int P,
                                                         int Pp, float* __restrict__ v3d, float* __restrict__ v2d) {
    typedef Op<MHMR_DT_F16>::V8 H8;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int v0 = blockIdx.y * 64 + w * 16;
    const int p0 = blockIdx.x * 64;
    const int npg = min(4, (Pp - p0) / 16);

    f32x4 acc[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int a = 0; a < 3; ++a) acc[i][a] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // A operand: lane (person p0 + 16 i + l15, k group g) holds F[32 s + 8 g + 0..7]; B operand: lane (vertex v0 + l15, k group g)
    // holds D[32 s + 8 g + 0..7][axis][vertex] = one 16-byte line of basis16 [Kb/8][hi|lo][3][Vp][8]
    const float* fp = F + (size_t)(p0 + l15) * c.Kb + 8 * g;
    const size_t plane = (size_t)c.Vp * 8;                          // halves per (k block, part, axis)
    const _Float16* bp = (const _Float16*)c.basis16 + (size_t)g * 6 * plane + (size_t)(v0 + l15) * 8;
    const int ns = c.Kb / 32;
    for (int s = 0; s < ns; ++s) {
        const _Float16* bs = bp + (size_t)(4 * s) * 6 * plane;
        H8 bh[3], bl[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            bh[a] = *(const H8*)(bs + a * plane);
            bl[a] = *(const H8*)(bs + (3 + a) * plane);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < npg) {
                const float* fr = fp + (size_t)i * 16 * c.Kb + 32 * s;
                const f32x4 f0 = *(const f32x4*)fr, f1 = *(const f32x4*)(fr + 4);
                H8 fh, fl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    fh[e] = (_Float16)f0[e];
                    fl[e] = (_Float16)(f0[e] - (float)fh[e]);
                    fh[4 + e] = (_Float16)f1[e];
                    fl[4 + e] = (_Float16)(f1[e] - (float)fh[4 + e]);
                }
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    acc[i][a] = Op<MHMR_DT_F16>::mfma16(fh, bh[a], acc[i][a]);
                    acc[i][a] = Op<MHMR_DT_F16>::mfma16(fl, bh[a], acc[i][a]);
                    acc[i][a] = Op<MHMR_DT_F16>::mfma16(fh, bl[a], acc[i][a]);
                }
            }
        }
    }

    const int v = v0 + l15;
    if (v >= c.V) return;
    const float vt[3] = {c.vtemp[v], c.vtemp[c.Vp + v], c.vtemp[2 * c.Vp + v]};      // v_template stays fp32
    // skinning influences of this vertex (first 4 in registers)
    int si[4];
    float sw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        si[i] = i < c.Kinf ? c.skin_idx[(size_t)v * c.Kinf + i] : 0;
        sw[i] = i < c.Kinf ? c.skin_w[(size_t)v * c.Kinf + i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i >= npg) break;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = p0 + 16 * i + 4 * g + r;
            if (p >= P) continue;
            const float vx = acc[i][0][r] * (1.0f / 1024.0f) + vt[0], vy = acc[i][1][r] * (1.0f / 1024.0f) + vt[1],
                        vz = acc[i][2][r] * (1.0f / 1024.0f) + vt[2];
            const float* ab = Afold + (size_t)p * NJ * 12;
            f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0, t2 = t0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float* aj = ab + si[q] * 12;
                t0 += sw[q] * *(const f32x4*)(aj);
                t1 += sw[q] * *(const f32x4*)(aj + 4);
                t2 += sw[q] * *(const f32x4*)(aj + 8);
            }
            for (int q = 4; q < c.Kinf; ++q) {
                const float wq = c.skin_w[(size_t)v * c.Kinf + q];
                const float* aj = ab + c.skin_idx[(size_t)v * c.Kinf + q] * 12;
                t0 += wq * *(const f32x4*)(aj);
                t1 += wq * *(const f32x4*)(aj + 4);
                t2 += wq * *(const f32x4*)(aj + 8);
            }
            float o3[3];
            o3[0] = t0[0] * vx + t0[1] * vy + t0[2] * vz + t0[3];
            o3[1] = t1[0] * vx + t1[1] * vy + t1[2] * vz + t1[3];
            o3[2] = t2[0] * vx + t2[1] * vy + t2[2] * vz + t2[3];
            float* vo = v3d + ((size_t)p * c.V + v) * 3;
            vo[0] = o3[0]; vo[1] = o3[1]; vo[2] = o3[2];
            float pr[2];
            project(xf + (size_t)p * 24 + 15, o3, pr);
            float* po = v2d + ((size_t)p * c.V + v) * 2;
            po[0] = pr[0]; po[1] = pr[1];
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
    if (c->Kb % 32 || c->Vp % 64 || c->Kb < 486 + c->nb + 10 || c->Kinf < 1) return MHMR_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int Pp = (P + 15) / 16 * 16;
    hipLaunchKernelGGL(lbs_pose_kernel, dim3(Pp), dim3(64), 0, s, *c, rotvec, betas, expr, loc, dist, Kmat, det_b, P, ws_F, ws_A,
                       ws_xf, j3d, j2d, transl);
    MHMR_CHECK_LAUNCH();
    prof_begin(PROF_LBS, s);
    hipLaunchKernelGGL(lbs_vertex_kernel, dim3((Pp + 63) / 64, c->Vp / 64), dim3(256), 0, s, *c, ws_F, ws_A, ws_xf, P, Pp, v3d, v2d);
    prof_end(PROF_LBS, s, (double)P);
    MHMR_CHECK_LAUNCH();
    hipLaunchKernelGGL(lbs_extra_joints_kernel, dim3(P), dim3(128), 0, s, *c, v3d, v2d, ws_xf, j3d, j2d);
    MHMR_CHECK_LAUNCH();
    return 0;
}
