// HBM-bound helper kernels of the ViT backbone: im2col patchify (fp32 image -> 16-bit GEMM operand), class /
// padding row initialisation, LayerNorm (fp32 residual stream -> 16-bit GEMM operand), and the final LayerNorm
// that emits the patch features both as fp32 and as the 16-bit cross-attention context operand.
#include "mhmr_common.h"
#include "mhmr_internal.h"
#include "ln_stats.h"

namespace {

// a_patch[m = (b*G + gy)*G + gx][k = c*196 + py*14 + px] = x[b][c][gy*14+py][gx*14+px]; k in [588, Kp) = 0.
// One thread = 8 consecutive k of one patch (one 16-byte store).
// PAIR (the f16x3 precision mode): rows of 2 Kp values, [hi = op16(x) | lo = op16(x - hi)]: the three-product A operand (GemmArgs::a_k, K = 3 a_k)
template <int DT, bool PAIR = false>
__global__ void im2col_kernel(const float* __restrict__ x, void* __restrict__ a_, int B, int S, int G, int Kp) {
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V8 V8;
    const int kc = Kp / 8;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * G * G * kc;
    if (gid >= total) return;
    const int c8 = (int)(gid % kc);
    const long long m = gid / kc;
    const int gx = (int)(m % G), gy = (int)((m / G) % G), b = (int)(m / ((long long)G * G));
    V8 v;
    [[maybe_unused]] V8 lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = c8 * 8 + e;
        float f = 0.f;
        if (k < 588) {
            const int c = k / 196, rem = k - c * 196, py = rem / 14, px = rem - py * 14;
            f = x[(((size_t)b * 3 + c) * S + gy * 14 + py) * S + gx * 14 + px];
        }
        v[e] = (T)f;
        if constexpr (PAIR) lo[e] = (T)(f - (float)v[e]);
    }
    if constexpr (PAIR) {
        *(V8*)((T*)a_ + (size_t)m * (2 * Kp) + c8 * 8) = v;
        *(V8*)((T*)a_ + (size_t)m * (2 * Kp) + Kp + c8 * 8) = lo;
    } else {
        *(V8*)((T*)a_ + (size_t)m * Kp + c8 * 8) = v;
    }
}

// Token rows of image b: patch n at row b*Tp + n (n < N = T - 1), the CLASS token LAST at row b*Tp + N, zero padding behind it.
// (Attention is permutation-equivariant over tokens, so the order is free; with the patches first, the 256-row GEMM tiles of the
// patch rows never straddle the class row and B * N rows are a whole number of tile rounds.)
// resid[b*Tp + N][:] = cls_token + pos[0];  resid[b*Tp + t][:] = 0 for t in [T, Tp)
__global__ void init_rows_kernel(float* __restrict__ resid, const float* __restrict__ cls_pos0, int B, int T, int Tp, int C) {
    const int rows_per_img = 1 + (Tp - T);
    const int r = blockIdx.x;  // 0 .. B*rows_per_img
    const int b = r / rows_per_img, i = r - b * rows_per_img;
    float* dst = resid + ((size_t)b * Tp + (T - 1) + i) * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) dst[c] = i == 0 ? cls_pos0[c] : 0.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One wave per row; a lane holds NP vectors of VEC floats (C = NP * 64 * VEC; VEC = 4 -> 16-byte loads / 8-byte stores whenever
// C % 256 == 0, VEC = 2 for C = 384).  Two-pass (mean, then centred variance) in registers.
// PAIR: out16 rows of ld16 >= 2 C values, [hi = op16(y) | lo = op16(y - hi)] (the f16x3 precision mode)
template <int DT, int NP, int VEC, bool FINAL, bool PAIR = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ in, const float* __restrict__ gw,
                                                        const float* __restrict__ gb, void* __restrict__ out16_, int ld16,
                                                        float* __restrict__ out32, int rows, int Np, int Tp, float eps, int o8) {
    // o8 > 0 (VEC == 4 only): byte offset inside an out16 row where the bf8 (e5m2) copy of the row goes (the fp8 low-half range of the next linear)
    typedef typename Op<DT>::T T;
    typedef float fv __attribute__((ext_vector_type(VEC)));
    typedef T hv __attribute__((ext_vector_type(VEC)));
    constexpr int C = NP * 64 * VEC;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    size_t in_row = row, out_row = row;
    if constexpr (FINAL) {  // row = b*Np + n  reads patch token n of image b (row b*Tp + n: the class token is last)
        const int b = row / Np, n = row - b * Np;
        in_row = (size_t)b * Tp + n;
    }
    const float* ip = in + in_row * C;
    fv v[NP];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        v[i] = *(const fv*)(ip + (i * 64 + lane) * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) s += v[i][e];
    }
    const float mean = wave_sum(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            v[i][e] -= mean;
            q += v[i][e] * v[i][e];
        }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / C) + eps);
    T* o16 = (T*)out16_ + out_row * ld16;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c = (i * 64 + lane) * VEC;
        const fv w = *(const fv*)(gw + c), bb = *(const fv*)(gb + c);
        fv y;
        hv h;
        [[maybe_unused]] hv hl;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            y[e] = v[i][e] * rstd * w[e] + bb[e];
            h[e] = (T)y[e];
            if constexpr (PAIR) hl[e] = (T)(y[e] - (float)h[e]);
        }
        *(hv*)(o16 + c) = h;
        if constexpr (PAIR) *(hv*)(o16 + C + c) = hl;
        if constexpr (VEC == 4 && !PAIR && !FINAL) {
            if (o8 > 0) *(uint32_t*)((char*)o16 + o8 + c) = pack_bf8x4(y[0], y[1], y[2], y[3]);
        }
        if constexpr (FINAL) *(fv*)(out32 + out_row * C + c) = y;
    }
}

// LayerNorm fold (GemmArgs::pstats): row statistics of the residual stream for the consumers of a folded LayerNorm.
//   patch row (b, n):  (sum, sum of squares) of its nblk 64-column blocks, written by the residual epilogue of the producing GEMM, are
//                      added in block order (bit-reproducible) -> mean, rstd = rsqrt(E[x^2] - mean^2 + eps)
//   class row (b, N):  its residual row is updated by the skinny kernel (csrc/vit_cls.hip), which leaves no block sums: one wave reads
//                      the fp32 row and takes the centred variance, like layernorm_kernel
// rowstats[b * Tp + t] = (mean, rstd).  grid = ceil(B * N / 32) + ceil(B / 4) blocks of 256 threads.
__global__ __launch_bounds__(256) void ln_stats_kernel(const float* __restrict__ pstats, const float* __restrict__ resid,
                                                       float* __restrict__ rowstats, int B, int N, int Tp, int C, int nblk, float eps,
                                                       int patch_blocks) {
    if ((int)blockIdx.x < patch_blocks) {
        ln_stats_patch_rows((int)(blockIdx.x * 256 + threadIdx.x), pstats, rowstats, B, N, Tp, C, nblk, eps);      // (ln_stats.h)
    } else {
        const int b = (blockIdx.x - patch_blocks) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (b >= B) return;
        const size_t row = (size_t)b * Tp + N;
        const float* ip = resid + row * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += ip[c];
        const float mean = wave_sum(s) * (1.0f / C);
        float q = 0.f;
        for (int c = lane; c < C; c += 64) { const float d = ip[c] - mean; q += d * d; }
        const float rstd = rsqrtf(wave_sum(q) * (1.0f / C) + eps);
        if (lane == 0) *(f32x2*)(rowstats + row * 2) = (f32x2){mean, rstd};
    }
}

// Second half of a split-k residual linear (csrc/gemm256.hip SPLITK; a batch of one: DESIGN.md section 12).  part = [nslices][rows][C] fp32
// partial products.  One wave per row:  v = sum over the slices IN SLICE ORDER (deterministic) + bias;  r = resid + gamma * v  -> resid (fp32,
// in place), x16 = the 16-bit copy of the RAW row (the next linear's A operand under the LayerNorm fold) and rowstats[row] = (mean, rstd) of r
// with the centred variance, two passes in registers like layernorm_kernel -- so no ln_stats launch follows this linear.
template <int DT, int NP>
__global__ __launch_bounds__(256) void splitk_resid_kernel(const float* __restrict__ part, int nslices, size_t slab, const float* __restrict__ bias,
                                                           const float* __restrict__ gamma, float* __restrict__ resid, void* __restrict__ x16_,
                                                           int ldx, float* __restrict__ rowstats, int rows, float eps) {
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V4 V4;
    constexpr int C = NP * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* pp = part + (size_t)row * C;
    float* rp = resid + (size_t)row * C;
    f32x4 v[NP], r[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = *(const f32x4*)(pp + c);
        r[i] = *(const f32x4*)(rp + c);
    }
    for (int sidx = 1; sidx < nslices; ++sidx) {
#pragma unroll
        for (int i = 0; i < NP; ++i) v[i] += *(const f32x4*)(pp + (size_t)sidx * slab + (i * 64 + lane) * 4);
    }
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c = (i * 64 + lane) * 4;
        const f32x4 b4 = bias ? *(const f32x4*)(bias + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 g4 = gamma ? *(const f32x4*)(gamma + c) : (f32x4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            r[i][e] = r[i][e] + g4[e] * (v[i][e] + b4[e]);
            s1 += r[i][e];
        }
        *(f32x4*)(rp + c) = r[i];
        if (x16_) {
            V4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (T)r[i][e];
            *(V4*)((T*)x16_ + (size_t)row * ldx + c) = o;
        }
    }
    if (rowstats) {
        const float mean = wave_sum(s1) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float dlt = r[i][e] - mean; q += dlt * dlt; }
        const float rstd = rsqrtf(wave_sum(q) * (1.0f / C) + eps);
        if (lane == 0) *(f32x2*)(rowstats + (size_t)row * 2) = (f32x2){mean, rstd};
    }
}

// V rows [B * Tp, ldv] (row-major, head h at columns 64 h) -> the attention kernel's V^T [B][H][64][Tp] with the key permutation of the V^T
// GEMM epilogue (bits 2 <-> 3 of the token index swapped inside every 16-group).  Behind the merged qkv linear of a short batch (gemm256.hip
// QKV).  One workgroup = one (64 tokens, head, image) tile through LDS; 16-byte loads and stores.
template <int DT>
__global__ __launch_bounds__(256) void vt_transpose_kernel(const void* __restrict__ v_, int ldv, void* __restrict__ vt_, int Tp, int H) {
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V8 V8;
    __shared__ T tile[64][72];                 // [token][d], 144-byte rows
    const int t0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const T* v = (const T*)v_;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i, t = idx >> 3, ch = idx & 7;
        *(V8*)&tile[t][ch * 8] = *(const V8*)(v + ((size_t)b * Tp + t0 + t) * ldv + h * 64 + ch * 8);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i, d = idx >> 3, ch = idx & 7;
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int tp = ch * 8 + e;                                        // position in the permuted row
            const int ts = (tp & ~12) | ((tp & 4) << 1) | ((tp & 8) >> 1);    // the token it holds (the swap is its own inverse)
            o[e] = tile[ts][d];
        }
        *(V8*)((T*)vt_ + ((size_t)(b * H + h) * 64 + d) * Tp + t0 + ch * 8) = o;
    }
}

// exact-erf GELU of an fp32 matrix [M, N] -> the op16 pair [M, 2 N] = [hi | lo] (the f16x3 mode's fc2 operand); 4 values per thread
template <int DT>
__global__ __launch_bounds__(256) void gelu_pair_kernel(const float* __restrict__ in, void* __restrict__ out_, long long total4, int N) {
    typedef typename Op<DT>::T T;
    typedef typename Op<DT>::V4 V4;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total4) return;
    const int n4 = N >> 2;
    const long long m = gid / n4;
    const int c = (int)(gid - m * n4) * 4;
    const f32x4 v = *(const f32x4*)(in + m * N + c);
    V4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float y = gelu_erf(v[e]);
        hi[e] = (T)y;
        lo[e] = (T)(y - (float)hi[e]);
    }
    T* o = (T*)out_ + m * (2LL * N) + c;
    *(V4*)o = hi;
    *(V4*)(o + N) = lo;
}

template <int DT, bool FINAL, bool PAIR = false>
int launch_ln(const float* in, const float* w, const float* b, void* out16, int ld16, float* out32, int rows, int C,
              int Np, int Tp, float eps, hipStream_t s, int o8 = 0) {
    if (o8 > 0 && (C % 256 || o8 < 2 * C || o8 + C > 2 * ld16)) return MHMR_ERR_BAD_ARG;
    const int grid = (rows + 3) / 4;
#define LN_CASE(CV, NPV, VECV)                                                                                           \
    case CV:                                                                                                             \
        hipLaunchKernelGGL((layernorm_kernel<DT, NPV, VECV, FINAL, PAIR>), dim3(grid), dim3(256), 0, s, in, w, b, out16, ld16, out32, \
                           rows, Np, Tp, eps, o8);                                                                       \
        break;
    switch (C) {
        LN_CASE(384, 3, 2)
        LN_CASE(768, 3, 4)
        LN_CASE(1024, 4, 4)
        default:
            return MHMR_ERR_BAD_SHAPE;
    }
#undef LN_CASE
    MHMR_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int mhmr_launch_im2col(const float* x, void* a, int B, int S, int G, int Kp, int dtype, hipStream_t s) {
    const long long total = (long long)B * G * G * (Kp / 8);
    const int grid = (int)((total + 255) / 256);
    if (dtype == MHMR_DT_F16)
        hipLaunchKernelGGL((im2col_kernel<MHMR_DT_F16>), dim3(grid), dim3(256), 0, s, x, a, B, S, G, Kp);
    else
        hipLaunchKernelGGL((im2col_kernel<MHMR_DT_BF16>), dim3(grid), dim3(256), 0, s, x, a, B, S, G, Kp);
    MHMR_CHECK_LAUNCH();
    return 0;
}

int mhmr_launch_im2col_pair(const float* x, void* a, int B, int S, int G, int Kp, int dtype, hipStream_t s) {
    const long long total = (long long)B * G * G * (Kp / 8);
    const int grid = (int)((total + 255) / 256);
    if (dtype == MHMR_DT_F16)
        hipLaunchKernelGGL((im2col_kernel<MHMR_DT_F16, true>), dim3(grid), dim3(256), 0, s, x, a, B, S, G, Kp);
    else
        hipLaunchKernelGGL((im2col_kernel<MHMR_DT_BF16, true>), dim3(grid), dim3(256), 0, s, x, a, B, S, G, Kp);
    MHMR_CHECK_LAUNCH();
    return 0;
}

int mhmr_launch_layernorm_pair(const float* in, const float* w, const float* b, void* out16, int rows, int C, float eps, int dtype,
                               hipStream_t s) {
    return dtype == MHMR_DT_F16 ? launch_ln<MHMR_DT_F16, false, true>(in, w, b, out16, 2 * C, nullptr, rows, C, 0, 0, eps, s)
                                : launch_ln<MHMR_DT_BF16, false, true>(in, w, b, out16, 2 * C, nullptr, rows, C, 0, 0, eps, s);
}

int mhmr_launch_gelu_pair(const float* in, void* out, long long M, int N, int dtype, hipStream_t s) {
    if (N % 4) return MHMR_ERR_BAD_SHAPE;
    const long long total4 = M * (N / 4);
    const int grid = (int)((total4 + 255) / 256);
    if (dtype == MHMR_DT_F16) hipLaunchKernelGGL((gelu_pair_kernel<MHMR_DT_F16>), dim3(grid), dim3(256), 0, s, in, out, total4, N);
    else hipLaunchKernelGGL((gelu_pair_kernel<MHMR_DT_BF16>), dim3(grid), dim3(256), 0, s, in, out, total4, N);
    MHMR_CHECK_LAUNCH();
    return 0;
}

int mhmr_launch_vt_transpose(const void* v, int ldv, void* vt, int B, int Tp, int H, int dtype, hipStream_t s) {
    if (Tp % 64 || ldv < 64 * H || ldv % 8) return MHMR_ERR_BAD_SHAPE;
    const dim3 grid(Tp / 64, H, B);
    if (dtype == MHMR_DT_F16) hipLaunchKernelGGL((vt_transpose_kernel<MHMR_DT_F16>), grid, dim3(256), 0, s, v, ldv, vt, Tp, H);
    else hipLaunchKernelGGL((vt_transpose_kernel<MHMR_DT_BF16>), grid, dim3(256), 0, s, v, ldv, vt, Tp, H);
    MHMR_CHECK_LAUNCH();
    return 0;
}

int mhmr_launch_splitk_resid(const float* part, int nslices, int rows, int C, const float* bias, const float* gamma, float* resid, void* x16,
                             int ldx, float* rowstats, float eps, int dtype, hipStream_t s) {
    if (nslices < 1 || rows <= 0 || (x16 && ldx < C)) return MHMR_ERR_BAD_ARG;
    const size_t slab = (size_t)rows * C;
    const int grid = (rows + 3) / 4;
#define SK_CASE(CV, NPV)                                                                                                                   \
    case CV:                                                                                                                               \
        if (dtype == MHMR_DT_F16)                                                                                                          \
            hipLaunchKernelGGL((splitk_resid_kernel<MHMR_DT_F16, NPV>), dim3(grid), dim3(256), 0, s, part, nslices, slab, bias, gamma, resid, x16, \
                               ldx, rowstats, rows, eps);                                                                                  \
        else                                                                                                                               \
            hipLaunchKernelGGL((splitk_resid_kernel<MHMR_DT_BF16, NPV>), dim3(grid), dim3(256), 0, s, part, nslices, slab, bias, gamma, resid, x16, \
                               ldx, rowstats, rows, eps);                                                                                  \
        break;
    switch (C) {
        SK_CASE(256, 1)
        SK_CASE(512, 2)
        SK_CASE(768, 3)
        SK_CASE(1024, 4)
        default:
            return MHMR_ERR_BAD_SHAPE;
    }
#undef SK_CASE
    MHMR_CHECK_LAUNCH();
    return 0;
}

int mhmr_launch_init_rows(float* resid, const float* cls_pos0, int B, int T, int Tp, int C, hipStream_t s) {
    hipLaunchKernelGGL(init_rows_kernel, dim3(B * (1 + Tp - T)), dim3(256), 0, s, resid, cls_pos0, B, T, Tp, C);
    MHMR_CHECK_LAUNCH();
    return 0;
}

int mhmr_launch_layernorm(const float* in, const float* w, const float* b, void* out16, int rows, int C, float eps,
                          int dtype, hipStream_t s) {
    return dtype == MHMR_DT_F16 ? launch_ln<MHMR_DT_F16, false>(in, w, b, out16, C, nullptr, rows, C, 0, 0, eps, s)
                                : launch_ln<MHMR_DT_BF16, false>(in, w, b, out16, C, nullptr, rows, C, 0, 0, eps, s);
}

// the same into rows of pitch ld16 (elements), optionally with the bf8 copy of each row at byte offset o8 (0 = none)
int mhmr_launch_layernorm_pitch(const float* in, const float* w, const float* b, void* out16, int ld16, int o8, int rows, int C, float eps,
                                int dtype, hipStream_t s) {
    if (ld16 < C) return MHMR_ERR_BAD_ARG;
    return dtype == MHMR_DT_F16 ? launch_ln<MHMR_DT_F16, false>(in, w, b, out16, ld16, nullptr, rows, C, 0, 0, eps, s, o8)
                                : launch_ln<MHMR_DT_BF16, false>(in, w, b, out16, ld16, nullptr, rows, C, 0, 0, eps, s, o8);
}

int mhmr_launch_ln_stats(const float* pstats, const float* resid, float* rowstats, int B, int N, int Tp, int C, float eps, hipStream_t s) {
    if (C % 128 || C > 1024) return MHMR_ERR_BAD_SHAPE;       // nblk = C / 64 even, <= 16 (eight lanes x two blocks)
    const int patch_blocks = (B * N + 31) / 32;
    // N == Tp: every row of an image has block sums (the GEMMs covered all B * Tp rows: no class rows of their own to finish)
    const int cls_blocks = N < Tp ? (B + 3) / 4 : 0;
    hipLaunchKernelGGL(ln_stats_kernel, dim3(patch_blocks + cls_blocks), dim3(256), 0, s, pstats, resid, rowstats, B, N, Tp, C, C / 64, eps,
                       patch_blocks);
    MHMR_CHECK_LAUNCH();
    return 0;
}

int mhmr_launch_final_norm(const float* resid, const float* w, const float* b, void* ctx16, int ldctx, float* feat32,
                           int B, int Np, int Tp, int C, float eps, int dtype, hipStream_t s) {
    return dtype == MHMR_DT_F16
               ? launch_ln<MHMR_DT_F16, true>(resid, w, b, ctx16, ldctx, feat32, B * Np, C, Np, Tp, eps, s)
               : launch_ln<MHMR_DT_BF16, true>(resid, w, b, ctx16, ldctx, feat32, B * Np, C, Np, Tp, eps, s);
}
