// Input preprocessing on the GPU: the step before Model.forward in the reference's demo (demo.py:27-51 open_image =
// PIL ImageOps.contain (bicubic, aspect-preserving) + ImageOps.pad (centre, black) + utils/image.py:12-24 normalize_rgb).
//
// The resample is Pillow's 8-bit two-pass separable convolution restated exactly: horizontal pass to uint8, vertical pass
// to uint8, both in fixed point (coefficients pre-quantised to 22 fractional bits by the host, accumulator seeded with
// 1 << 21, arithmetic shift, clip to 0..255), so the result is bit-identical to PIL's; the ImageNet normalisation is a
// 3 x 256 lookup table the host fills with the reference's own numpy expression (float32 divide, float64 mean/std, cast).
// Byte/integer streaming work: HBM-bound, one thread per output pixel, no LDS needed (taps overlap in L1/L2).
#include "mhmr_common.h"
#include "mhmr_internal.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// tmp[y][ox][c] = clip8(sum_t img[y0 + y][xmin(ox) + t][c] * kh[ox][t])      (rows y0 .. y0 + rows - 1 of the source only)
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ img, int W, int y0, int rows,
                                                         const int* __restrict__ kh, const int* __restrict__ bh, int ksh,
                                                         int ow, uint8_t* __restrict__ tmp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * ow) return;
    const int y = i / ow, ox = i - y * ow;
    const int xmin = bh[2 * ox], xmax = bh[2 * ox + 1];
    const int* k = kh + (size_t)ox * ksh;
    const uint8_t* p = img + ((size_t)(y0 + y) * W + xmin) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int t = 0; t < xmax; ++t) {
        const int kk = k[t];
        s0 += p[3 * t] * kk;
        s1 += p[3 * t + 1] * kk;
        s2 += p[3 * t + 2] * kk;
    }
    uint8_t* o = tmp + (size_t)i * 3;
    o[0] = (uint8_t)clip8(s0);
    o[1] = (uint8_t)clip8(s1);
    o[2] = (uint8_t)clip8(s2);
}

// out[c][Y][X] = lut[c][ inside ? clip8(sum_t tmp[ymin(oy) - y0 + t][ox][c] * kv[oy][t]) : 0 ],  (ox, oy) = (X - pad_x, Y - pad_y)
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const uint8_t* __restrict__ tmp, int y0,
                                                              const int* __restrict__ kv, const int* __restrict__ bv, int ksv,
                                                              int ow, int oh, int S, int pad_x, int pad_y,
                                                              const float* __restrict__ lut, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S * S) return;
    const int Y = i / S, X = i - Y * S;
    const int ox = X - pad_x, oy = Y - pad_y;
    int v0 = 0, v1 = 0, v2 = 0;
    if (ox >= 0 && ox < ow && oy >= 0 && oy < oh) {
        const int ymin = bv[2 * oy], ymax = bv[2 * oy + 1];
        const int* k = kv + (size_t)oy * ksv;
        const uint8_t* p = tmp + ((size_t)(ymin - y0) * ow + ox) * 3;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int t = 0; t < ymax; ++t) {
            const int kk = k[t];
            const uint8_t* q = p + (size_t)t * ow * 3;
            s0 += q[0] * kk;
            s1 += q[1] * kk;
            s2 += q[2] * kk;
        }
        v0 = clip8(s0); v1 = clip8(s1); v2 = clip8(s2);
    }
    out[i] = lut[v0];
    out[(size_t)S * S + i] = lut[256 + v1];
    out[(size_t)2 * S * S + i] = lut[512 + v2];
}

}  // namespace

extern "C" int mhmr_preprocess_u8(const void* img, int H, int W, const int* kh, const int* bh, int ksh, const int* kv,
                                  const int* bv, int ksv, int ow, int oh, int y0, int rows, int S, int pad_x, int pad_y,
                                  const float* lut, void* tmp, float* out, void* stream) {
    if (H <= 0 || W <= 0 || ow <= 0 || oh <= 0 || ow > S || oh > S || ksh <= 0 || ksv <= 0) return MHMR_ERR_BAD_SHAPE;
    if (y0 < 0 || rows <= 0 || y0 + rows > H || pad_x < 0 || pad_y < 0 || pad_x + ow > S || pad_y + oh > S) return MHMR_ERR_BAD_SHAPE;
    if (!img || !kh || !bh || !kv || !bv || !lut || !tmp || !out) return MHMR_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int n1 = rows * ow, n2 = S * S;
    hipLaunchKernelGGL(resample_h_kernel, dim3((n1 + 255) / 256), dim3(256), 0, s, (const uint8_t*)img, W, y0, rows, kh, bh, ksh, ow,
                       (uint8_t*)tmp);
    hipLaunchKernelGGL(resample_v_norm_kernel, dim3((n2 + 255) / 256), dim3(256), 0, s, (const uint8_t*)tmp, y0, kv, bv, ksv, ow, oh, S,
                       pad_x, pad_y, lut, out);
    MHMR_CHECK_LAUNCH();
    return 0;
}
