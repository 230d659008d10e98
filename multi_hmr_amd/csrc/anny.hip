// Anny-variant read-outs (multi_hmr_anny/encoder.py:47-56 camera from the class token; multi_hmr_anny/multi_hmr.py:144-177
// person parameters).  Tiny per-image / per-person kernels; all fp32.
#include "mhmr_common.h"
#include "mhmr_internal.h"

namespace {

// encoder.py:48-54: fov = fov_max * sigmoid(logit); focal = (S / 2) / tan(fov / 2); K = [[f,0,S/2],[0,f,S/2],[0,0,1]]
__global__ void anny_camera_kernel(const float* __restrict__ logit, int B, float S, float fov_max, float* __restrict__ fov,
                                   float* __restrict__ K) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float f = fov_max * (1.0f / (1.0f + expf(-logit[b])));
    fov[b] = f;
    const float focal = (S * 0.5f) / tanf(f * 0.5f);
    float* k = K + 9 * b;
    k[0] = focal; k[1] = 0.f; k[2] = S * 0.5f;
    k[3] = 0.f; k[4] = focal; k[5] = S * 0.5f;
    k[6] = 0.f; k[7] = 0.f; k[8] = 1.f;
}

// encoder.py:57-58: scores_logits = mlp_det(feat)[..., 0], scores = sigmoid(logits) -- no clamp, unlike Model.detection.
// hid16 = relu(mlp_det.0(features)) from mhmr_gemm16(..., MHMR_EPI_OP16_RELU); one wave per token.
template <int DT>
__global__ __launch_bounds__(256) void anny_score_kernel(const void* __restrict__ hid_, int ld, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, float* __restrict__ scores,
                                                         float* __restrict__ logits, int rows, int C) {
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V2 V2;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* hp = (const T*)hid_ + (size_t)row * ld;
    float s = 0.f;
    for (int c = lane * 2; c < C; c += 128) {
        const V2 h = *(const V2*)(hp + c);
        s += (float)h[0] * w2[c] + (float)h[1] * w2[c + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    s += b2[0];
    if (lane == 0) {
        logits[row] = s;
        scores[row] = 1.0f / (1.0f + expf(-s));
    }
}

// One workgroup per person, one thread per joint (J <= blockDim.x).
//   rot6d [P][6 J] (mlp_pose output + init_body_pose), read as J (3,2) matrices (multi_hmr.py:151: reshape(-1,3,2)):
//   columns x = (e0,e2,e4), y = (e1,e3,e5) -> roma.special_gramschmidt -> R = [x y x^y]; joints with useful[j] == 0 get I (153-156);
//   rotvec = roma.rotmat_to_rotvec(R).  Thread 0 additionally: loc = ([x,y] + 0.5 + offset) * patch (145-146),
//   dist = focal / clamp(exp(d), 1e-5) (149-152), transl = K^-1 [loc,1] * dist (153, utils/camera.py:30-48, general 3x3 inverse).
//   Threads < nb: shape = sigmoid(logit) (159).
__global__ void anny_decode_kernel(const float* __restrict__ rot6d, const float* __restrict__ useful, int J,
                                   const float* __restrict__ shape_logit, int nb, const float* __restrict__ dist_logit,
                                   const float* __restrict__ offset, const int* __restrict__ det_b, const int* __restrict__ det_y,
                                   const int* __restrict__ det_x, const float* __restrict__ Kmat, float patch,
                                   float* __restrict__ rotmat, float* __restrict__ rotvec, float* __restrict__ shape,
                                   float* __restrict__ loc, float* __restrict__ dist, float* __restrict__ transl) {
    const int p = blockIdx.x, j = threadIdx.x;
    if (j < J) {
        const float* dp = rot6d + ((size_t)p * J + j) * 6;
        float x0 = dp[0], x1 = dp[2], x2 = dp[4];
        float y0 = dp[1], y1 = dp[3], y2 = dp[5];
        const float nx = sqrtf(x0 * x0 + x1 * x1 + x2 * x2);
        x0 /= nx; x1 /= nx; x2 /= nx;
        const float dxy = x0 * y0 + x1 * y1 + x2 * y2;
        y0 -= dxy * x0; y1 -= dxy * x1; y2 -= dxy * x2;
        const float ny = sqrtf(y0 * y0 + y1 * y1 + y2 * y2);
        y0 /= ny; y1 /= ny; y2 /= ny;
        const float z0 = x1 * y2 - x2 * y1, z1 = x2 * y0 - x0 * y2, z2 = x0 * y1 - x1 * y0;
        float R[9] = {x0, y0, z0, x1, y1, z1, x2, y2, z2};
        const float u = useful[j];
#pragma unroll
        for (int e = 0; e < 9; ++e) R[e] = u * R[e] + (1.f - u) * ((e % 4 == 0) ? 1.f : 0.f);
        float* rp = rotmat + ((size_t)p * J + j) * 9;
#pragma unroll
        for (int e = 0; e < 9; ++e) rp[e] = R[e];
        const float tr = R[0] + R[4] + R[8];
        float qx, qy, qz, qw;
        int choice = 0;
        float best = R[0];
        if (R[4] > best) { best = R[4]; choice = 1; }
        if (R[8] > best) { best = R[8]; choice = 2; }
        if (tr > best) { best = tr; choice = 3; }
        if (choice == 3) {
            qx = R[7] - R[5]; qy = R[2] - R[6]; qz = R[3] - R[1]; qw = 1.f + tr;
        } else {
            const int i = choice, jj = (i + 1) % 3, kk = (jj + 1) % 3;
            float qq[3];
            qq[i] = 1.f - tr + 2.f * R[i * 3 + i];
            qq[jj] = R[jj * 3 + i] + R[i * 3 + jj];
            qq[kk] = R[kk * 3 + i] + R[i * 3 + kk];
            qw = R[kk * 3 + jj] - R[jj * 3 + kk];
            qx = qq[0]; qy = qq[1]; qz = qq[2];
        }
        const float qn = sqrtf(qx * qx + qy * qy + qz * qz + qw * qw);
        qx /= qn; qy /= qn; qz /= qn; qw /= qn;
        if (qw < 0.f) { qx = -qx; qy = -qy; qz = -qz; qw = -qw; }
        const float angle = 2.f * atan2f(sqrtf(qx * qx + qy * qy + qz * qz), qw);
        float sc;
        if (fabsf(angle) <= 1e-3f) sc = 2.f + angle * angle / 12.f + 7.f * angle * angle * angle * angle / 2880.f;
        else sc = angle / sinf(angle / 2.f);
        float* vp = rotvec + ((size_t)p * J + j) * 3;
        vp[0] = sc * qx; vp[1] = sc * qy; vp[2] = sc * qz;
    }
    if (j < nb) shape[(size_t)p * nb + j] = 1.0f / (1.0f + expf(-shape_logit[(size_t)p * nb + j]));
    if (j == 0) {
        const float lx = ((float)det_x[p] + 0.5f + offset[2 * p]) * patch;
        const float ly = ((float)det_y[p] + 0.5f + offset[2 * p + 1]) * patch;
        loc[2 * p] = lx;
        loc[2 * p + 1] = ly;
        const float* k = Kmat + 9 * det_b[p];
        const float d = k[0] / fmaxf(expf(dist_logit[p]), 1e-5f);
        dist[p] = d;
        // inverse of the 3x3 K by the adjugate (torch.inverse in the reference)
        const float a = k[0], b = k[1], c = k[2], dd = k[3], e = k[4], f = k[5], g = k[6], h = k[7], i = k[8];
        const float A = e * i - f * h, Bc = -(dd * i - f * g), Cc = dd * h - e * g;
        const float det = a * A + b * Bc + c * Cc;
        const float inv[9] = {A / det, -(b * i - c * h) / det, (b * f - c * e) / det,
                              Bc / det, (a * i - c * g) / det, -(a * f - c * dd) / det,
                              Cc / det, -(a * h - b * g) / det, (a * e - b * dd) / det};
#pragma unroll
        for (int r = 0; r < 3; ++r) transl[3 * p + r] = (inv[3 * r] * lx + inv[3 * r + 1] * ly + inv[3 * r + 2]) * d;
    }
}

}  // namespace

extern "C" int mhmr_anny_camera(const float* fov_logit, int B, int img_size, float fov_max, float* fov, float* K, void* stream) {
    if (B <= 0 || img_size <= 0) return MHMR_ERR_BAD_SHAPE;
    if (!fov_logit || !fov || !K) return MHMR_ERR_BAD_ARG;
    hipLaunchKernelGGL(anny_camera_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, fov_logit, B, (float)img_size, fov_max, fov, K);
    MHMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int mhmr_anny_scores(const void* hid16, int ld, const float* w2, const float* b2, float* scores, float* logits, int rows, int C,
                                int dtype, void* stream) {
    if (rows <= 0 || C % 128) return MHMR_ERR_BAD_SHAPE;
    if (dtype == MHMR_DT_F16)
        hipLaunchKernelGGL((anny_score_kernel<MHMR_DT_F16>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, hid16, ld, w2, b2, scores, logits, rows, C);
    else
        hipLaunchKernelGGL((anny_score_kernel<MHMR_DT_BF16>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, hid16, ld, w2, b2, scores, logits, rows, C);
    MHMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int mhmr_anny_decode(const float* rot6d, const float* useful, int J, const float* shape_logit, int nb,
                                const float* dist_logit, const float* offset, const int* det_b, const int* det_y, const int* det_x,
                                const float* K, int patch, int P, float* rotmat, float* rotvec, float* shape, float* loc, float* dist,
                                float* transl, void* stream) {
    if (P < 0 || J <= 0 || J > 256 || nb < 0 || nb > J) return MHMR_ERR_BAD_SHAPE;
    if (P == 0) return 0;
    hipLaunchKernelGGL(anny_decode_kernel, dim3(P), dim3((J + 63) / 64 * 64), 0, (hipStream_t)stream, rot6d, useful, J, shape_logit, nb,
                       dist_logit, offset, det_b, det_y, det_x, K, (float)patch, rotmat, rotvec, shape, loc, dist, transl);
    MHMR_CHECK_LAUNCH();
    return 0;
}
