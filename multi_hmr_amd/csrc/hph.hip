// fp32 kernels of the detection heads and the Human Perception Head (HPH) cross-attention decoder
// (reference model.py:133-158, 479-593; blocks/cross_attn_transformer.py).  Everything downstream of the
// backbone features is <1 % of the FLOPs, so it stays in exact fp32: linears run on the fp32-input MFMA
// (v_mfma_f32_16x16x4_f32 == an fmaf chain), attention over the ragged per-image query groups is done with
// lanes = queries (self-attention) or lanes = 8 queries x 8 key slices (cross-attention over the N patch
// tokens) and an online softmax; padded queries of the reference never exist here.
#include "mhmr_common.h"
#include "mhmr_internal.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------------------------------------------------
// Y[m][n] = act( sum_k X[row(m)][k] * W[n][k] + bias[n] ) (+ R[m][n]);  K % 16 == 0 (callers pad with zeros).
// One wave = 16 (m) x 32 (n); block = 4 waves along n = 16 x 128.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ X, int ldx, const int* __restrict__ row_idx,
                                                         const float* __restrict__ W, int ldw, const float* __restrict__ bias,
                                                         const float* R, int ldr, float* Y, int ldy, int M, int N, int K,
                                                         int act) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 128 + w * 32;
    if (n0 >= N) return;
    int mrow = min(m0 + l15, M - 1);
    if (row_idx) mrow = row_idx[mrow];
    const float* xp = X + (size_t)mrow * ldx + 4 * g;
    const float* wp0 = W + (size_t)min(n0 + l15, N - 1) * ldw + 4 * g;
    const float* wp1 = W + (size_t)min(n0 + 16 + l15, N - 1) * ldw + 4 * g;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k = 0; k < K; k += 16) {
        const f32x4 xa = *(const f32x4*)(xp + k);
        const f32x4 w0 = *(const f32x4*)(wp0 + k);
        const f32x4 w1 = *(const f32x4*)(wp1 + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[e], w0[e], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[e], w1[e], acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = n0 + 16 * t + l15;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 4 * g + r;
            if (m >= M) continue;
            float v = (t == 0 ? acc0[r] : acc1[r]) + bv;
            if (act == MHMR_ACT_RELU) v = fmaxf(v, 0.f);
            else if (act == MHMR_ACT_GELU) v = gelu_erf(v);
            if (R) v += R[(size_t)m * ldr + n];
            Y[(size_t)m * ldy + n] = v;
        }
    }
}

// The same product with the k range split over the four waves of a workgroup (one 16 x 32 tile per workgroup, partial sums added
// in wave order through LDS: deterministic).  At the HPH shapes (M = persons <= a few hundred, K = N = 1024) the form above runs
// two waves per CU through 64 dependent load -> MFMA iterations (30 us per linear, 16 linears per forward); here each wave has
// 16 iterations and a CU holds eight waves.
__global__ __launch_bounds__(256) void linear_f32_splitk_kernel(const float* __restrict__ X, int ldx, const int* __restrict__ row_idx,
                                                                const float* __restrict__ W, int ldw, const float* __restrict__ bias,
                                                                const float* R, int ldr, float* Y, int ldy, int M, int N, int K,
                                                                int act) {
    __shared__ f32x4 part[3][2][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 32;
    int mrow = min(m0 + l15, M - 1);
    if (row_idx) mrow = row_idx[mrow];
    const int nch = K >> 4, cb = nch >> 2, cr = nch & 3;          // 16-wide k chunks, split as evenly as they go
    const int kq = 16 * (cb + (w < cr ? 1 : 0)), k0 = 16 * (w * cb + min(w, cr));
    const float* xp = X + (size_t)mrow * ldx + 4 * g + k0;
    const float* wp0 = W + (size_t)min(n0 + l15, N - 1) * ldw + 4 * g + k0;
    const float* wp1 = W + (size_t)min(n0 + 16 + l15, N - 1) * ldw + 4 * g + k0;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k = 0; k < kq; k += 16) {
        const f32x4 xa = *(const f32x4*)(xp + k);
        const f32x4 w0 = *(const f32x4*)(wp0 + k);
        const f32x4 w1 = *(const f32x4*)(wp1 + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[e], w0[e], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[e], w1[e], acc1, 0, 0, 0);
        }
    }
    if (w > 0) { part[w - 1][0][lane] = acc0; part[w - 1][1][lane] = acc1; }
    __syncthreads();
    if (w > 0) return;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const f32x4 p0 = part[q][0][lane], p1 = part[q][1][lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc0[r] += p0[r]; acc1[r] += p1[r]; }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = n0 + 16 * t + l15;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 4 * g + r;
            if (m >= M) continue;
            float v = (t == 0 ? acc0[r] : acc1[r]) + bv;
            if (act == MHMR_ACT_RELU) v = fmaxf(v, 0.f);
            else if (act == MHMR_ACT_GELU) v = gelu_erf(v);
            if (R) v += R[(size_t)m * ldr + n];
            Y[(size_t)m * ldy + n] = v;
        }
    }
}

// One wave per row, fp32 in / fp32 out, C % 64 == 0, C <= 2048.
__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* __restrict__ in, const float* __restrict__ gw,
                                                            const float* __restrict__ gb, float* __restrict__ out, int rows,
                                                            int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* ip = in + (size_t)row * C;
    float v[32];
    const int n = C / 64;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i)
        if (i < n) { v[i] = ip[i * 64 + lane]; s += v[i]; }
    const float mean = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i)
        if (i < n) { v[i] -= mean; q += v[i] * v[i]; }
    const float rstd = rsqrtf(wave_sum(q) / C + eps);
#pragma unroll
    for (int i = 0; i < 32; ++i)
        if (i < n) out[(size_t)row * C + i * 64 + lane] = v[i] * rstd * gw[i * 64 + lane] + gb[i * 64 + lane];
}

// ------------------------------------------------------------------------------------------------------------
// Detection: score[m] = clamp(sigmoid(hidden16[m] . w2 + b2), 1e-4, 1-1e-4)   (model.py:135, 641-643)
// ------------------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void score_kernel(const void* __restrict__ hid_, int ld, const float* __restrict__ w2,
                                                    const float* __restrict__ b2, float* __restrict__ scores, int rows, int C) {
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
    s = wave_sum(s) + b2[0];
    if (lane == 0) scores[row] = fminf(fmaxf(1.0f / (1.0f + expf(-s)), 1e-4f), 1.0f - 1e-4f);
}

// NMS (model.py:620-638) + threshold (612-617).  hmax over the k x k window anchored at (y - pad, x - pad),
// out-of-image = -inf, exactly max_pool2d(stride 1, padding pad) cropped to G x G.  score' = heat * (hmax == heat).
__device__ __forceinline__ float nms_score(const float* __restrict__ heat, int G, int y, int x, int k, int pad) {
    const float c = heat[y * G + x];
    if (k <= 1) return c;
    float mx = -INFINITY;
    for (int dy = 0; dy < k; ++dy) {
        const int yy = y - pad + dy;
        if (yy < 0 || yy >= G) continue;
        for (int dx = 0; dx < k; ++dx) {
            const int xx = x - pad + dx;
            if (xx < 0 || xx >= G) continue;
            mx = fmaxf(mx, heat[yy * G + xx]);
        }
    }
    return mx == c ? c : 0.f;
}

// pass 1: one block per image -> counts[b]
__global__ __launch_bounds__(256) void detect_count_kernel(const float* __restrict__ scores, int G, int k, int pad, float thr,
                                                           int* __restrict__ counts) {
    __shared__ int red[4];
    const int b = blockIdx.x, N = G * G;
    const float* heat = scores + (size_t)b * N;
    int c = 0;
    for (int n = threadIdx.x; n < N; n += 256) c += nms_score(heat, G, n / G, n % G, k, pad) >= thr ? 1 : 0;
    c = (int)wave_sum((float)c);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[b] = red[0] + red[1] + red[2] + red[3];
}

// pass 2: one block per image; ordered ((b, y, x) ascending == torch.where order, model.py:146-149) compaction.
// base[b] = exclusive prefix of counts (host).  Each thread owns a contiguous run of cells.
__global__ __launch_bounds__(256) void detect_write_kernel(const float* __restrict__ scores, int G, int k, int pad, float thr,
                                                           const int* __restrict__ base, int* __restrict__ det_b,
                                                           int* __restrict__ det_y, int* __restrict__ det_x,
                                                           float* __restrict__ det_score, int cap) {
    __shared__ int cnt[256];
    const int b = blockIdx.x, N = G * G, t = threadIdx.x;
    const float* heat = scores + (size_t)b * N;
    const int per = (N + 255) / 256, n_lo = min(t * per, N), n_hi = min(n_lo + per, N);
    int c = 0;
    for (int n = n_lo; n < n_hi; ++n) c += nms_score(heat, G, n / G, n % G, k, pad) >= thr ? 1 : 0;
    cnt[t] = c;
    __syncthreads();
    int off = base[b];
    for (int i = 0; i < t; ++i) off += cnt[i];
    for (int n = n_lo; n < n_hi; ++n) {
        const float s = nms_score(heat, G, n / G, n % G, k, pad);
        if (s >= thr) {
            if (off < cap) { det_b[off] = b; det_y[off] = n / G; det_x[off] = n % G; det_score[off] = s; }      // (fixed-capacity callers)
            ++off;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Device-side bookkeeping of the person set: what the reference derives on the HOST from torch.where / the caller's idx
// (model.py:146-151; rebatch / pad_to_max, utils/tensor_manip.py:7-45) -- per-image person counts, their exclusive prefix sums, the
// ragged query groups of the decoder's self-attention and the <= 8-query work items of its cross-attention -- without a host round trip.
//   counts != null (inference): per-image counts from detect_count_kernel;  counts == null (training hook): histogram of det_b[0..P)
//   (persons sorted by image, as torch.where leaves them).  Persons beyond `cap` are dropped (the host notices total > cap and re-runs).
//   base[b]            exclusive prefix sum of the counts (detect_write_kernel's write offsets)            [B]     (nullable)
//   gstart[0..ngcap]   person offsets of the non-empty images; entries past the last group repeat the end (empty groups)
//   chunks[3 * nccap]  (image, first person, count <= 8); entries past the last chunk are (0, 0, 0): their workgroups return
//   info[4]            {persons kept = min(total, cap), groups, chunks, total}
// One workgroup; the serial part is O(B + P / 8) on one lane (B = 32, P = 256: a few microseconds, off the critical path's host).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void person_groups_kernel(const int* __restrict__ counts, const int* __restrict__ det_b, int P, int B,
                                                            int cap, int* __restrict__ base, int* __restrict__ gstart, int ngcap,
                                                            int* __restrict__ chunks, int nccap, int* __restrict__ info) {
    extern __shared__ int cnt[];      // [B]
    const int t = threadIdx.x;
    for (int b = t; b < B; b += 256) cnt[b] = counts ? counts[b] : 0;
    __syncthreads();
    if (!counts) {
        for (int p = t; p < P; p += 256) {
            const int b = det_b[p];
            if (b >= 0 && b < B) atomicAdd(&cnt[b], 1);
        }
        __syncthreads();
    }
    // the serial part runs on one lane against LDS (a chain of dependent GLOBAL stores made this 10 us); everybody writes the tables out
    int* sbase = cnt + B;                 // [B]
    int* sg = sbase + B;                  // [ngcap + 1]
    int* sc = sg + ngcap + 1;             // [3 * nccap]
    __shared__ int sinfo[4];
    if (t == 0) {
        int start = 0, total = 0, ng = 0, nc = 0;
        sg[0] = 0;
        for (int b = 0; b < B; ++b) {
            const int c = cnt[b];
            sbase[b] = total;
            total += c;
            const int kept = min(c, max(cap - start, 0));
            if (kept > 0) {
                for (int q0 = 0; q0 < kept; q0 += 8) {
                    if (nc < nccap) { sc[3 * nc] = b; sc[3 * nc + 1] = start + q0; sc[3 * nc + 2] = min(8, kept - q0); }
                    ++nc;
                }
                start += kept;
                if (ng < ngcap) sg[ng + 1] = start;
                ++ng;
            }
        }
        for (int g = min(ng, ngcap); g < ngcap; ++g) sg[g + 1] = start;
        for (int c = min(nc, nccap); c < nccap; ++c) { sc[3 * c] = 0; sc[3 * c + 1] = 0; sc[3 * c + 2] = 0; }
        sinfo[0] = start; sinfo[1] = min(ng, ngcap); sinfo[2] = min(nc, nccap); sinfo[3] = total;
    }
    __syncthreads();
    if (base) for (int b = t; b < B; b += 256) base[b] = sbase[b];
    for (int g = t; g <= ngcap; g += 256) gstart[g] = sg[g];
    for (int c = t; c < 3 * nccap; c += 256) chunks[c] = sc[c];
    if (t < 4) info[t] = sinfo[t];
}

// ------------------------------------------------------------------------------------------------------------
// Camera embedding (model.py:160-187, utils/camera.py:30-48, blocks/camera_embed.py:39-58).
// ray = K^-1 [i*14+7, j*14+7, 1] with i = ROW index fed as pixel-x, j = COLUMN as pixel-y (reproduced verbatim);
// z_K = [ray(3), sin(pi*ray_a*f_k) (a*16+k), cos(...)] = 99 channels.  Also writes the 16-bit copy into the
// cross-attention context operand ctx16[:, C : C+99] and zeros ctx16[:, C+99 : Kc].
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void inv3x3(const float* k, float* o) {
    const float a = k[0], b = k[1], c = k[2], d = k[3], e = k[4], f = k[5], g = k[6], h = k[7], i = k[8];
    const float A = e * i - f * h, B = -(d * i - f * g), Cc = d * h - e * g;
    const float det = a * A + b * B + c * Cc;
    const float id = 1.0f / det;
    o[0] = A * id; o[1] = -(b * i - c * h) * id; o[2] = (b * f - c * e) * id;
    o[3] = B * id; o[4] = (a * i - c * g) * id;  o[5] = -(a * f - c * d) * id;
    o[6] = Cc * id; o[7] = -(a * h - b * g) * id; o[8] = (a * e - b * d) * id;
}

template <int DT>
__global__ __launch_bounds__(128) void camera_embed_kernel(const float* __restrict__ Kmat, const float* __restrict__ freq,
                                                           int G, int patch, float* __restrict__ zK, void* __restrict__ ctx16_,
                                                           int Kc, int C, int nbands) {
    typedef typename Op<DT>::T T;
    const int N = G * G;
    const int E = 3 + 6 * nbands;            // FourierPositionEncoding.channels (blocks/camera_embed.py:19-29): 99 for 16 bands
    const int row = blockIdx.x;  // b*N + n
    const int b = row / N, n = row - b * N;
    const int i = n / G, j = n - i * G;
    float Ki[9];
    inv3x3(Kmat + b * 9, Ki);
    const float px = (float)(i * patch + patch / 2), py = (float)(j * patch + patch / 2);
    float ray[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) ray[a] = (Ki[a * 3 + 0] * px + Ki[a * 3 + 1] * py + Ki[a * 3 + 2] * 1.0f) * 1.0f;
    const int t = threadIdx.x;
    T* cp = (T*)ctx16_ + (size_t)row * Kc + C;
    const float PI_F = 3.14159265358979323846f;
    if (t < E) {
        float v;
        if (t < 3) v = ray[t];
        else {
            const int u = (t - 3) % (3 * nbands), a = u / nbands, kb = u % nbands;
            const float arg = PI_F * (ray[a] * freq[a * nbands + kb]);
            v = (t - 3) < 3 * nbands ? sinf(arg) : cosf(arg);
        }
        zK[(size_t)row * E + t] = v;
        cp[t] = (T)v;
    } else if (C + t < Kc) {
        cp[t] = (T)0.f;
    }
    // (Kc - C) <= 128 is asserted by the launcher
}

// ------------------------------------------------------------------------------------------------------------
// HPH inputs (model.py:255, 263-265, 500-504, 514-517, 541-552)
//   zc[p]    = feat32[row_p]                                            (mlp_offset input)
//   token[p] = [ feat32[row_p] | zK[row_p] ] + cq_x[y_p] + cq_y[x_p]  |  init_pose | init_betas | init_cam | 0-pad
//   ctx16[row_p][0:Cc] = 16-bit( [feat32 | zK][row_p] + cv_x[y_p] + cv_y[x_p] )
// NB the *_x tables are indexed by the ROW y and *_y by the COLUMN x, as the reference does.
// ------------------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void hph_inputs_kernel(const float* __restrict__ feat32, const float* __restrict__ zK,
                                                         const int* __restrict__ det_b, const int* __restrict__ det_y,
                                                         const int* __restrict__ det_x, const float* __restrict__ cq_x,
                                                         const float* __restrict__ cq_y, const float* __restrict__ cv_x,
                                                         const float* __restrict__ cv_y, const float* __restrict__ init_tail,
                                                         int ntail, float* __restrict__ zc, float* __restrict__ token, int Ktok,
                                                         void* __restrict__ ctx16_, int Kc, int* __restrict__ det_row, int G,
                                                         int C, const int* __restrict__ nvalid, int E) {
    typedef typename Op<DT>::T T;
    const int p = blockIdx.x, Cc = C + E;        // E = camera embedding channels (99 for 16 bands)
    // fixed-capacity callers: rows >= *nvalid are padding (detection (0, 0, 0)); they are computed like persons and sliced off by the
    // host, but must not touch the context row of a cell nobody detected
    const bool real = nvalid == nullptr || p < *nvalid;
    const int b = det_b[p], y = det_y[p], x = det_x[p];
    const size_t row = (size_t)b * G * G + (size_t)y * G + x;
    if (threadIdx.x == 0) det_row[p] = (int)row;
    T* cp = (T*)ctx16_ + row * Kc;
    for (int c = threadIdx.x; c < Ktok; c += 256) {
        float tv = 0.f;
        if (c < Cc) {
            const float f = c < C ? feat32[row * C + c] : zK[row * E + (c - C)];
            if (c < C) zc[(size_t)p * C + c] = f;
            tv = f + (cq_x[(size_t)y * Cc + c] + cq_y[(size_t)x * Cc + c]);
            if (real) cp[c] = (T)(f + (cv_x[(size_t)y * Cc + c] + cv_y[(size_t)x * Cc + c]));
        } else if (c < Cc + ntail) {
            tv = init_tail[c - Cc];
        }
        token[(size_t)p * Ktok + c] = tv;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Self-attention over the queries of one image (Attention.forward :129-159, ragged, unmasked).
// qkv: [P, 3*inner] (q | k | v), head h = columns h*32..h*32+31.  grid (groups, heads, ceil(nmax/64)).
// lane = one query; keys streamed (wave-uniform addresses), online softmax.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void hph_self_attn_kernel(const float* __restrict__ qkv, const int* __restrict__ gstart,
                                                           float* __restrict__ out, int inner, float scale) {
    const int g = blockIdx.x, h = blockIdx.y;
    const int s0 = gstart[g], n = gstart[g + 1] - s0;
    const int qi = blockIdx.z * 64 + threadIdx.x;
    if (blockIdx.z * 64 >= n) return;
    const bool active = qi < n;
    const int ld = 3 * inner;
    const float* qp = qkv + (size_t)(s0 + (active ? qi : 0)) * ld + h * 32;
    float q[32], o[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) { q[d] = qp[d] * scale; o[d] = 0.f; }
    float m = -INFINITY, l = 0.f;
    for (int j = 0; j < n; ++j) {
        const float* kp = qkv + (size_t)(s0 + j) * ld + inner + h * 32;
        const float* vp = kp + inner;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) s += q[d] * kp[d];
        const float mn = fmaxf(m, s);
        const float a = expf(m - mn), pj = expf(s - mn);
        l = l * a + pj;
#pragma unroll
        for (int d = 0; d < 32; ++d) o[d] = o[d] * a + pj * vp[d];
        m = mn;
    }
    if (active) {
        const float inv = 1.0f / l;
        float* op = out + (size_t)(s0 + qi) * inner + h * 32;
#pragma unroll
        for (int d = 0; d < 32; ++d) op[d] = o[d] * inv;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Cross-attention of the queries of image b over its N context tokens (CrossAttention.forward :185-205).
// q: [P, inner]; kv: [B*N, 2*inner] (k | v) fp32.  One workgroup of CA_WAVES waves per (chunk of <= 8 queries, head):
// lane = slice*8 + qi, wave w's slice s handles keys j = s + 8 w (mod 8 CA_WAVES); the 8 partial (m, l, o) per query of a
// wave are merged with 3 xor-shuffle rounds, the CA_WAVES wave results through LDS in wave order (deterministic).  The loop
// is latency-bound (one 256-byte K|V row pair per lane-slice per trip): a single wave per (chunk, head) walked 512 trips at
// N = 4096 (0.56 ms per layer); 8 waves walk 64 each.
// chunks: (image b, first query, count) int triples (person_groups_kernel, or the host); count 0 = padding of the work list.
// ------------------------------------------------------------------------------------------------------------
constexpr int CA_WAVES = 8;
__global__ __launch_bounds__(64 * CA_WAVES) void hph_cross_attn_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                            const int* __restrict__ chunks, int ncap, float* __restrict__ out, int inner,
                                                            int N, float scale) {
    // 1-D grid of ncap x heads workgroups over a work list of ncap entries whose tail may be padding (count 0: person_groups_kernel
    // pads up to the launch's upper bound).  The real work is the FIRST nc x heads workgroups: the dispatcher hands out workgroups in
    // index order, two per CU -- with the padding interleaved (a 2-D grid, real chunks 0..31 of 64 in every row) half of the CUs
    // received two real workgroups and the other half two that return at once: 221 instead of 118 us per layer.
    const int nc = __syncthreads_count(threadIdx.x < ncap && chunks[3 * threadIdx.x + 2] > 0);      // (ncap <= 512: the launcher)
    const int heads = inner >> 5;
    if ((int)blockIdx.x >= nc * heads) return;
    const int ch = blockIdx.x % nc, h = blockIdx.x / nc;
    const int b = chunks[3 * ch], q0 = chunks[3 * ch + 1], nq = chunks[3 * ch + 2];
    __shared__ float part[CA_WAVES][8][34];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, qi = lane & 7, sl = lane >> 3;
    const bool active = qi < nq;
    const float* qp = q + (size_t)(q0 + (active ? qi : 0)) * inner + h * 32;
    float qv[32], o[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) { qv[d] = qp[d] * scale; o[d] = 0.f; }
    float m = -INFINITY, l = 0.f;
    const int ld = 2 * inner;
    const float* kbase = kv + (size_t)b * N * ld + h * 32;
    for (int j = sl + 8 * wv; j < N; j += 8 * CA_WAVES) {
        const float* kp = kbase + (size_t)j * ld;
        const float* vp = kp + inner;
        float kk[32];
#pragma unroll
        for (int d = 0; d < 32; d += 4) *(f32x4*)(kk + d) = *(const f32x4*)(kp + d);
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) s += qv[d] * kk[d];
        if (s > m) {  // rare after the first few keys
            const float a = expf(m - s);
            l *= a;
#pragma unroll
            for (int d = 0; d < 32; ++d) o[d] *= a;
            m = s;
        }
        const float pj = expf(s - m);
        l += pj;
#pragma unroll
        for (int d = 0; d < 32; d += 4) {
            const f32x4 vv = *(const f32x4*)(vp + d);
            o[d] += pj * vv[0]; o[d + 1] += pj * vv[1]; o[d + 2] += pj * vv[2]; o[d + 3] += pj * vv[3];
        }
    }
    // merge the 8 key slices (lanes differing in bits 3..5)
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(m, off), l2 = __shfl_xor(l, off);
        const float mn = fmaxf(m, m2);
        const float a1 = (m == -INFINITY) ? 0.f : expf(m - mn), a2 = (m2 == -INFINITY) ? 0.f : expf(m2 - mn);
        l = l * a1 + l2 * a2;
#pragma unroll
        for (int d = 0; d < 32; ++d) o[d] = o[d] * a1 + __shfl_xor(o[d], off) * a2;
        m = mn;
    }
    // merge the waves: lanes 0..7 of every wave hold (m, l, o) of query qi over that wave's keys
    if (sl == 0) {
        part[wv][qi][32] = m;
        part[wv][qi][33] = l;
#pragma unroll
        for (int d = 0; d < 32; ++d) part[wv][qi][d] = o[d];
    }
    __syncthreads();
    if (wv == 0 && active && sl == 0) {
        float mt = part[0][qi][32];
#pragma unroll
        for (int w2 = 1; w2 < CA_WAVES; ++w2) mt = fmaxf(mt, part[w2][qi][32]);
        float lt = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) o[d] = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < CA_WAVES; ++w2) {
            const float mw = part[w2][qi][32];
            const float a = (mw == -INFINITY) ? 0.f : expf(mw - mt);
            lt += part[w2][qi][33] * a;
#pragma unroll
            for (int d = 0; d < 32; ++d) o[d] += part[w2][qi][d] * a;
        }
        const float inv = 1.0f / lt;
        float* op = out + (size_t)(q0 + qi) * inner + h * 32;
#pragma unroll
        for (int d = 0; d < 32; ++d) op[d] = o[d] * inv;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Read-out decode (model.py:571-583, 287-298; utils/humans.py:12-22; roma special_gramschmidt / rotmat_to_rotvec;
// utils/camera.py:71-90).  dec[p] = [pose6d(318) | betas(nb) | cam(3) | expr(10)] (decoder + init already added).
// One thread per (person, joint); thread joint 0 also post-processes the distance.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void hph_decode_kernel(const float* __restrict__ dec, int ldd, int nb,
                                                        const float* __restrict__ Kmat, const int* __restrict__ det_b,
                                                        float fn, int nearness, float* __restrict__ rotmat,
                                                        float* __restrict__ rotvec, float* __restrict__ betas,
                                                        float* __restrict__ expr, float* __restrict__ dist_pp,
                                                        float* __restrict__ dist) {
    const int p = blockIdx.x, j = threadIdx.x;
    const float* dp = dec + (size_t)p * ldd;
    if (j < 53) {
        // 6D -> (2,3) -> transpose: first three numbers = column 0, next three = column 1
        float x0 = dp[6 * j], x1 = dp[6 * j + 1], x2 = dp[6 * j + 2];
        float y0 = dp[6 * j + 3], y1 = dp[6 * j + 4], y2 = dp[6 * j + 5];
        const float nx = sqrtf(x0 * x0 + x1 * x1 + x2 * x2);
        x0 /= nx; x1 /= nx; x2 /= nx;
        const float dxy = x0 * y0 + x1 * y1 + x2 * y2;
        y0 -= dxy * x0; y1 -= dxy * x1; y2 -= dxy * x2;
        const float ny = sqrtf(y0 * y0 + y1 * y1 + y2 * y2);
        y0 /= ny; y1 /= ny; y2 /= ny;
        const float z0 = x1 * y2 - x2 * y1, z1 = x2 * y0 - x0 * y2, z2 = x0 * y1 - x1 * y0;
        float R[9] = {x0, y0, z0, x1, y1, z1, x2, y2, z2};  // columns [x y z]
        float* rp = rotmat + ((size_t)p * 53 + j) * 9;
#pragma unroll
        for (int e = 0; e < 9; ++e) rp[e] = R[e];
        // rotmat -> unit quaternion (XYZW), branch on the largest of (R00, R11, R22, trace)
        const float tr = R[0] + R[4] + R[8];
        float qx, qy, qz, qw;
        int choice = 0;  // argmax over (R00, R11, R22, trace), first maximal index wins
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
        float* vp = rotvec + ((size_t)p * 53 + j) * 3;
        vp[0] = sc * qx; vp[1] = sc * qy; vp[2] = sc * qz;
    }
    if (j < nb) betas[(size_t)p * nb + j] = dp[318 + j];
    if (j < 10) expr[(size_t)p * 10 + j] = dp[318 + nb + 3 + j];
    if (j == 0) {
        const float d0 = dp[318 + nb];
        dist_pp[p] = d0;
        const float focal = Kmat[det_b[p] * 9 + 0];
        float d = d0 * (focal / fn);
        if (nearness) d = expf(d) - 1e-10f;
        dist[p] = fminf(fmaxf(d, 0.f), 50.f);
    }
}

// offsets -> loc:  loc = ([x, y] + 0.5 + offset) * patch     (model.py:272-275)
__global__ void loc_kernel(const float* __restrict__ offset, const int* __restrict__ det_y, const int* __restrict__ det_x,
                           float patch, float* __restrict__ loc, int P) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    loc[2 * p] = ((float)det_x[p] + 0.5f + offset[2 * p]) * patch;
    loc[2 * p + 1] = ((float)det_y[p] + 0.5f + offset[2 * p + 1]) * patch;
}

}  // namespace

// ---------------------------------------------------------------- launchers
int mhmr_launch_linear_f32(const float* X, int ldx, const int* row_idx, const float* W, int ldw, const float* bias,
                           const float* R, int ldr, float* Y, int ldy, int M, int N, int K, int act, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 16 || ldx % 4 || ldw % 4) return MHMR_ERR_BAD_SHAPE;
    if (K >= 256 && (long long)((N + 127) / 128) * ((M + 15) / 16) < 1024)       // few tiles, long k: split k over the waves
        hipLaunchKernelGGL(linear_f32_splitk_kernel, dim3((N + 31) / 32, (M + 15) / 16), dim3(256), 0, s, X, ldx, row_idx, W, ldw,
                           bias, R, ldr, Y, ldy, M, N, K, act);
    else
        hipLaunchKernelGGL(linear_f32_kernel, dim3((N + 127) / 128, (M + 15) / 16), dim3(256), 0, s, X, ldx, row_idx, W, ldw,
                           bias, R, ldr, Y, ldy, M, N, K, act);
    MHMR_CHECK_LAUNCH();
    return 0;
}

int mhmr_launch_layernorm_f32(const float* in, const float* w, const float* b, float* out, int rows, int C, float eps,
                              hipStream_t s) {
    if (C % 64 || C > 2048 || rows <= 0) return MHMR_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(layernorm_f32_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, in, w, b, out, rows, C, eps);
    MHMR_CHECK_LAUNCH();
    return 0;
}

int mhmr_launch_scores(const void* hid, int ld, const float* w2, const float* b2, float* scores, int rows, int C, int dtype,
                       hipStream_t s) {
    if (C % 128) return MHMR_ERR_BAD_SHAPE;
    if (dtype == MHMR_DT_F16)
        hipLaunchKernelGGL((score_kernel<MHMR_DT_F16>), dim3((rows + 3) / 4), dim3(256), 0, s, hid, ld, w2, b2, scores, rows, C);
    else
        hipLaunchKernelGGL((score_kernel<MHMR_DT_BF16>), dim3((rows + 3) / 4), dim3(256), 0, s, hid, ld, w2, b2, scores, rows, C);
    MHMR_CHECK_LAUNCH();
    return 0;
}

static inline int nms_pad(int k) { return (k == 2) ? 1 : (k == 4) ? 2 : (k - 1) / 2; }

int mhmr_launch_detect_count(const float* scores, int B, int G, int nms_kernel, float thr, int* counts, hipStream_t s) {
    hipLaunchKernelGGL(detect_count_kernel, dim3(B), dim3(256), 0, s, scores, G, nms_kernel, nms_pad(nms_kernel), thr, counts);
    MHMR_CHECK_LAUNCH();
    return 0;
}

int mhmr_launch_detect_write(const float* scores, int B, int G, int nms_kernel, float thr, const int* base, int* det_b,
                             int* det_y, int* det_x, float* det_score, int cap, hipStream_t s) {
    hipLaunchKernelGGL(detect_write_kernel, dim3(B), dim3(256), 0, s, scores, G, nms_kernel, nms_pad(nms_kernel), thr, base,
                       det_b, det_y, det_x, det_score, cap);
    MHMR_CHECK_LAUNCH();
    return 0;
}

int mhmr_launch_person_groups(const int* counts, const int* det_b, int P, int B, int cap, int* base, int* gstart, int ngcap, int* chunks,
                              int nccap, int* info, hipStream_t s) {
    if (B <= 0 || B > 8192 || P < 0 || cap < 0 || ngcap < 0 || nccap < 0 || !gstart || !info || (nccap > 0 && !chunks)) return MHMR_ERR_BAD_ARG;
    if (!counts && P > 0 && !det_b) return MHMR_ERR_BAD_ARG;
    const size_t lds = ((size_t)2 * B + (size_t)ngcap + 1 + (size_t)3 * nccap) * sizeof(int);
    if (lds > 60 * 1024) return MHMR_ERR_BAD_SHAPE;          // (B = 8192 with the sufficient bounds is 160 KB: far beyond any batch)
    hipLaunchKernelGGL(person_groups_kernel, dim3(1), dim3(256), lds, s, counts, det_b, P, B, cap, base, gstart, ngcap, chunks, nccap, info);
    MHMR_CHECK_LAUNCH();
    return 0;
}

int mhmr_launch_camera_embed(const float* Kmat, const float* freq, int B, int G, int patch, float* zK, void* ctx16, int Kc,
                             int C, int dtype, int nbands, hipStream_t s) {
    if (nbands < 1 || Kc - C > 128 || Kc - C < 3 + 6 * nbands) return MHMR_ERR_BAD_SHAPE;      // one thread per camera column: <= 20 bands
    if (dtype == MHMR_DT_F16)
        hipLaunchKernelGGL((camera_embed_kernel<MHMR_DT_F16>), dim3(B * G * G), dim3(128), 0, s, Kmat, freq, G, patch, zK, ctx16, Kc, C, nbands);
    else
        hipLaunchKernelGGL((camera_embed_kernel<MHMR_DT_BF16>), dim3(B * G * G), dim3(128), 0, s, Kmat, freq, G, patch, zK, ctx16, Kc, C, nbands);
    MHMR_CHECK_LAUNCH();
    return 0;
}

int mhmr_launch_hph_inputs(const float* feat32, const float* zK, const int* det_b, const int* det_y, const int* det_x,
                           const float* cq_x, const float* cq_y, const float* cv_x, const float* cv_y, const float* init_tail,
                           int ntail, float* zc, float* token, int Ktok, void* ctx16, int Kc, int* det_row, int P, int G, int C,
                           int dtype, const int* nvalid, int cam_dim, hipStream_t s) {
    if (P <= 0) return 0;
    if (cam_dim < 3 || C + cam_dim > Kc) return MHMR_ERR_BAD_SHAPE;
    if (dtype == MHMR_DT_F16)
        hipLaunchKernelGGL((hph_inputs_kernel<MHMR_DT_F16>), dim3(P), dim3(256), 0, s, feat32, zK, det_b, det_y, det_x, cq_x, cq_y,
                           cv_x, cv_y, init_tail, ntail, zc, token, Ktok, ctx16, Kc, det_row, G, C, nvalid, cam_dim);
    else
        hipLaunchKernelGGL((hph_inputs_kernel<MHMR_DT_BF16>), dim3(P), dim3(256), 0, s, feat32, zK, det_b, det_y, det_x, cq_x, cq_y,
                           cv_x, cv_y, init_tail, ntail, zc, token, Ktok, ctx16, Kc, det_row, G, C, nvalid, cam_dim);
    MHMR_CHECK_LAUNCH();
    return 0;
}

int mhmr_launch_hph_self_attn(const float* qkv, const int* gstart, float* out, int ngroups, int nmax, int heads,
                              hipStream_t s) {
    if (ngroups <= 0) return 0;
    const int inner = heads * 32;
    hipLaunchKernelGGL(hph_self_attn_kernel, dim3(ngroups, heads, (nmax + 63) / 64), dim3(64), 0, s, qkv, gstart, out, inner,
                       0.17677669529663688110f);
    MHMR_CHECK_LAUNCH();
    return 0;
}

int mhmr_launch_hph_cross_attn(const float* q, const float* kv, const int* chunks, int nchunks, float* out, int heads, int N,
                               hipStream_t s) {
    if (nchunks <= 0) return 0;
    // work lists longer than one workgroup can count (512 entries = 4096 persons in one batch) take one launch per 512 entries
    for (int c0 = 0; c0 < nchunks; c0 += 64 * CA_WAVES) {
        const int n = nchunks - c0 < 64 * CA_WAVES ? nchunks - c0 : 64 * CA_WAVES;
        hipLaunchKernelGGL(hph_cross_attn_kernel, dim3(n * heads), dim3(64 * CA_WAVES), 0, s, q, kv, chunks + 3 * c0, n, out, heads * 32, N,
                           0.17677669529663688110f);
    }
    MHMR_CHECK_LAUNCH();
    return 0;
}

int mhmr_launch_hph_decode(const float* dec, int ldd, int nb, const float* Kmat, const int* det_b, float fn, int nearness,
                           float* rotmat, float* rotvec, float* betas, float* expr, float* dist_pp, float* dist, int P,
                           hipStream_t s) {
    if (P <= 0) return 0;
    hipLaunchKernelGGL(hph_decode_kernel, dim3(P), dim3(64), 0, s, dec, ldd, nb, Kmat, det_b, fn, nearness, rotmat, rotvec,
                       betas, expr, dist_pp, dist);
    MHMR_CHECK_LAUNCH();
    return 0;
}

int mhmr_launch_loc(const float* offset, const int* det_y, const int* det_x, int patch, float* loc, int P, hipStream_t s) {
    if (P <= 0) return 0;
    hipLaunchKernelGGL(loc_kernel, dim3((P + 127) / 128), dim3(128), 0, s, offset, det_y, det_x, (float)patch, loc, P);
    MHMR_CHECK_LAUNCH();
    return 0;
}
