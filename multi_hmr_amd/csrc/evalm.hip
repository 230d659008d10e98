// Accuracy metrics of the reference's evaluation loop (train.py:372-395, 415-423) for M matched
// (prediction, ground truth) point sets of V points each: the per-vertex error and the Procrustes-aligned per-vertex error
// (roma.rigid_points_registration(pred, gt, compute_scaling=True): similarity transform minimising the squared error).
//
// One workgroup per pair.  Pass 1 streams both point sets once and reduces, in fp64, the raw moments sum x, sum y,
// sum y x^T, sum |x|^2 together with the plain error sum; lane 0 then solves the 3x3 orthogonal Procrustes problem as the
// dominant eigenvector of Horn's symmetric 4x4 quaternion matrix (cyclic Jacobi in fp64: always a proper rotation, which is
// what roma's det-corrected SVD returns) and scale = tr(R^T M) / sum |x - xbar|^2.  Pass 2 re-reads the (L2-resident,
// 2 x 126 KB) pair and reduces the aligned error.  HBM-bound: 2 x V x 12 B per pair.
#include "mhmr_common.h"
#include "mhmr_internal.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// dominant eigenvector of the symmetric 4x4 matrix a (cyclic Jacobi, fp64)
__device__ void jacobi4_max_eigvec(double a[4][4], double q[4]) {
    double v[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 24; ++sweep) {
        double off = 0.0;
        for (int i = 0; i < 4; ++i)
            for (int j = i + 1; j < 4; ++j) off += a[i][j] * a[i][j];
        if (off < 1e-300) break;
        for (int p = 0; p < 3; ++p)
            for (int r = p + 1; r < 4; ++r) {
                if (fabs(a[p][r]) < 1e-300) continue;
                const double theta = (a[r][r] - a[p][p]) / (2.0 * a[p][r]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; ++k) {
                    const double akp = a[k][p], akr = a[k][r];
                    a[k][p] = c * akp - s * akr;
                    a[k][r] = s * akp + c * akr;
                }
                for (int k = 0; k < 4; ++k) {
                    const double apk = a[p][k], ark = a[r][k];
                    a[p][k] = c * apk - s * ark;
                    a[r][k] = s * apk + c * ark;
                }
                for (int k = 0; k < 4; ++k) {
                    const double vkp = v[k][p], vkr = v[k][r];
                    v[k][p] = c * vkp - s * vkr;
                    v[k][r] = s * vkp + c * vkr;
                }
            }
    }
    int best = 0;
    for (int i = 1; i < 4; ++i)
        if (a[i][i] > a[best][best]) best = i;
    for (int k = 0; k < 4; ++k) q[k] = v[k][best];
}

__global__ __launch_bounds__(NT) void mesh_error_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                        const float* __restrict__ pred_c, const float* __restrict__ gt_c, int V,
                                                        float* __restrict__ pve, float* __restrict__ pa_pve,
                                                        float* __restrict__ Rts) {
    __shared__ double sh[4];
    __shared__ double xf[13];     // s*R (9), t (3)
    const int m = blockIdx.x;
    const float* x = pred + (size_t)m * V * 3;
    const float* y = gt + (size_t)m * V * 3;
    float cx[3] = {0, 0, 0}, cy[3] = {0, 0, 0};
    if (pred_c) { cx[0] = pred_c[3 * m]; cx[1] = pred_c[3 * m + 1]; cx[2] = pred_c[3 * m + 2]; }
    if (gt_c) { cy[0] = gt_c[3 * m]; cy[1] = gt_c[3 * m + 1]; cy[2] = gt_c[3 * m + 2]; }

    double sx[3] = {0, 0, 0}, sy[3] = {0, 0, 0}, syx[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, sxx = 0, serr = 0;
    for (int n = threadIdx.x; n < V; n += NT) {
        float a[3], b[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { a[k] = x[3 * n + k] - cx[k]; b[k] = y[3 * n + k] - cy[k]; }   // fp32 centring, as the reference
        const float d0 = b[0] - a[0], d1 = b[1] - a[1], d2 = b[2] - a[2];
        serr += (double)(sqrtf(d0 * d0 + d1 * d1 + d2 * d2) * 1000.f);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            sx[i] += a[i];
            sy[i] += b[i];
            sxx += (double)a[i] * a[i];
#pragma unroll
            for (int j = 0; j < 3; ++j) syx[i][j] += (double)b[i] * a[j];
        }
    }
    double r[17];
    for (int i = 0; i < 3; ++i) { r[i] = block_sum(sx[i], sh); r[3 + i] = block_sum(sy[i], sh); }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r[6 + 3 * i + j] = block_sum(syx[i][j], sh);
    r[15] = block_sum(sxx, sh);
    r[16] = block_sum(serr, sh);

    if (threadIdx.x == 0) {
        const double N = (double)V;
        double xm[3], ym[3], M[3][3];
        for (int i = 0; i < 3; ++i) { xm[i] = r[i] / N; ym[i] = r[3 + i] / N; }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) M[i][j] = r[6 + 3 * i + j] - N * ym[i] * xm[j];      // sum (y - ym)(x - xm)^T
        const double varx = r[15] - N * (xm[0] * xm[0] + xm[1] * xm[1] + xm[2] * xm[2]);
        // Horn: S = M^T (S_ab = sum x_a y_b); rotation x -> y is the top eigenvector (w, qx, qy, qz) of
        const double Sxx = M[0][0], Sxy = M[1][0], Sxz = M[2][0], Syx = M[0][1], Syy = M[1][1], Syz = M[2][1], Szx = M[0][2],
                     Szy = M[1][2], Szz = M[2][2];
        double A[4][4] = {{Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx},
                          {Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz},
                          {Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy},
                          {Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz}};
        double q[4];
        jacobi4_max_eigvec(A, q);
        const double w = q[0], qx = q[1], qy = q[2], qz = q[3];
        double R[3][3] = {{1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - w * qz), 2 * (qx * qz + w * qy)},
                          {2 * (qx * qy + w * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - w * qx)},
                          {2 * (qx * qz - w * qy), 2 * (qy * qz + w * qx), 1 - 2 * (qx * qx + qy * qy)}};
        double tr = 0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) tr += R[i][j] * M[i][j];
        const double s = varx > 0 ? tr / varx : 1.0;
        for (int i = 0; i < 3; ++i) {
            double t = ym[i];
            for (int j = 0; j < 3; ++j) { xf[3 * i + j] = s * R[i][j]; t -= s * R[i][j] * xm[j]; }
            xf[9 + i] = t;
        }
        xf[12] = s;
        pve[m] = (float)(r[16] / N);
        if (Rts) {
            for (int i = 0; i < 9; ++i) Rts[13 * m + i] = (float)R[i / 3][i % 3];
            for (int i = 0; i < 3; ++i) Rts[13 * m + 9 + i] = (float)xf[9 + i];
            Rts[13 * m + 12] = (float)s;
        }
    }
    __syncthreads();
    float A9[9], t3[3];
    for (int i = 0; i < 9; ++i) A9[i] = (float)xf[i];
    for (int i = 0; i < 3; ++i) t3[i] = (float)xf[9 + i];
    double se = 0;
    for (int n = threadIdx.x; n < V; n += NT) {
        float a[3], b[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { a[k] = x[3 * n + k] - cx[k]; b[k] = y[3 * n + k] - cy[k]; }
        float e2 = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float p = A9[3 * i] * a[0] + A9[3 * i + 1] * a[1] + A9[3 * i + 2] * a[2] + t3[i];
            e2 += (b[i] - p) * (b[i] - p);
        }
        se += (double)(sqrtf(e2) * 1000.f);
    }
    se = block_sum(se, sh);
    if (threadIdx.x == 0) pa_pve[m] = (float)(se / V);
}

}  // namespace

extern "C" int mhmr_eval_mesh_errors(const float* pred, const float* gt, const float* pred_center, const float* gt_center, int M, int V,
                                     float* pve, float* pa_pve, float* Rts, void* stream) {
    if (M < 0 || V <= 0) return MHMR_ERR_BAD_SHAPE;
    if (M == 0) return 0;
    if (!pred || !gt || !pve || !pa_pve) return MHMR_ERR_BAD_ARG;
    hipLaunchKernelGGL(mesh_error_kernel, dim3(M), dim3(NT), 0, (hipStream_t)stream, pred, gt, pred_center, gt_center, V, pve, pa_pve, Rts);
    MHMR_CHECK_LAUNCH();
    return 0;
}
