/* libmhmr.so -- C ABI of the MI355X-native Multi-HMR batched-inference path.
 *
 * The reference (naver/multi-hmr) has no FFI / operator layer: its only seam is the Python class
 * model.Model (reference model.py:30-349) whose forward dispatches PyTorch library kernels.  This header is
 * the boundary a maintainer would bind instead; every entry point names the reference code it replaces.
 *
 * Conventions: extern "C"; raw DEVICE pointers (tensor.data_ptr()) and explicit int shapes; caller-allocated
 * outputs and workspaces; `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 * no allocation, no hidden synchronisation, no exceptions: every call returns 0 on success, a negative
 * MHMR_ERR_* for argument errors or a positive hipError_t.  All matrices are row-major.  "op16" means the 16-bit
 * MFMA operand type selected by `dtype` (MHMR_DT_BF16 or MHMR_DT_F16; fp32 accumulation either way).
 */
#ifndef MHMR_H
#define MHMR_H

#ifdef __cplusplus
extern "C" {
#endif

#define MHMR_VERSION 106   /* 106: mhmr_attention16_ex variant 10 (class query on workgroups of its own; opt-in); the fc1 epilogue's GELU is max(x,0) - |x| exp2(P5(|x|)) (6.4e-7 absolute; was Abramowitz-Stegun 7.1.25, 2.6e-5); 105: mhmr_vit_desc.cls_pstats (row statistics inside the class-row launches); mhmr_vit_desc.v16 (merged qkv launch of a short batch); mhmr_vit_desc.cpad (ViT-S on the 256x256 kernel: C-wide linears as N = 512 with masked columns); mhmr_vit_desc.{splitk, splitk_bytes}, mhmr_splitk_workspace_bytes, mhmr_gemm16_splitk_resid: split-k residual linears for launches that fill less than half the chip (a batch of one); 104: mhmr_vit_desc.{x3, qkv32, hid32}: the f16x3 precision mode (three 16-bit products per term in every backbone linear, fp32 attention); mhmr_gemm16_ex a_k with K = 3 a_k; mhmr_attention_f32; 103: mhmr_attention16_ex variant 6 (the default of mhmr_vit_forward); mhmr_camera_embed(num_bands), mhmr_hph_desc.cam_dim; mhmr_lbs_consts.basis16 layout (high halves for k < Kb - 64); mhmr_person_groups, mhmr_detect_write_cap, mhmr_hph_desc.nvalid (no host round trip for the person set; group / chunk counts of mhmr_hph_forward are upper bounds); 102: mhmr_lbs_consts: extra joints as virtual vertex tiles (Vl, xbary); 101: class token LAST in the token rows, mhmr_vit_block.{v_w2,proj_w2}, mhmr_gemm16_ex, mhmr_cls_linear16, mhmr_attention16_ex variants 4 / 5 */

#define MHMR_OK 0
#define MHMR_ERR_BAD_ARG (-1)
#define MHMR_ERR_BAD_SHAPE (-2)

#define MHMR_DT_BF16 0
#define MHMR_DT_F16 1

#define MHMR_ACT_NONE 0
#define MHMR_ACT_RELU 1
#define MHMR_ACT_GELU 2

/* GEMM epilogues of mhmr_gemm16 */
#define MHMR_EPI_OP16 0      /* out16 = acc + bias                                           */
#define MHMR_EPI_OP16_GELU 1 /* out16 = gelu(acc + bias), erf form, 3-term A&S 7.1.25 erfc: |abs err| < 2.6e-5 (Mlp fc1) */
#define MHMR_EPI_OP16_RELU 2 /* out16 = relu(acc + bias)                (regression_mlp, model.py:596-609) */
#define MHMR_EPI_RESID 3     /* out32 += gamma * (acc + bias)           (LayerScale + residual) */
#define MHMR_EPI_PATCH 4     /* patch-embed: + bias + pos-embed, scattered to token rows     */
#define MHMR_EPI_F32 5       /* out32 = acc (+ bias)                    (HPH to_kv)          */
#define MHMR_EPI_VT 6        /* V^T[b][h][d][swap23(t)] = acc + bias    (attention V operand) */
#define MHMR_EPI_OP16_QK 7   /* out16 = (acc + bias) * (n < N/2 ? MHMR_ATTN_QSCALE : 1): the Q | K projection (N = 2C) with the
                                softmax scale and the exp -> exp2 change of base folded into Q in fp32, before the one rounding */
/* head_dim^-0.5 * log2(e) for head_dim = 64 (DINOv2 Attention: softmax((q * scale) k^T)) */
#define MHMR_ATTN_QSCALE 0.18033688011112042f

int mhmr_version(void);
/* sha256 prefix (16 hex digits) of the sources + compile flags this library was built from (multi_hmr_amd/_lib.py::source_hash);
 * "unknown" for a build that did not pass it.  Evidence files (profiles/) are keyed by it. */
const char* mhmr_source_hash(void);

/* ------------------------------------------------------------------------------------------------------------
 * ViT backbone.  Replaces blocks/dinov2.py:16-26 -> torch.hub DinoVisionTransformer.get_intermediate_layers
 * (patch embed + cls + interpolated pos-embed, L pre-norm blocks with LayerScale, final LayerNorm, cls dropped).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    const float *ln1_w, *ln1_b;   /* [C]                                  blocks.i.norm1            */
    const void* qkv_w;            /* op16 [3C, C]                         blocks.i.attn.qkv.weight  */
    const float* qkv_b;           /* [3C]                                                            */
    const void* proj_w;           /* op16 [C, C]                          blocks.i.attn.proj        */
    const float *proj_b, *ls1;    /* [C], [C] (ls1.gamma)                                           */
    const float *ln2_w, *ln2_b;   /* [C]                                  blocks.i.norm2            */
    const void* fc1_w;            /* op16 [4C, C]                         blocks.i.mlp.fc1          */
    const float* fc1_b;           /* [4C]                                                            */
    const void* fc2_w;            /* op16 [C, 4C]                         blocks.i.mlp.fc2          */
    const float *fc2_b, *ls2;     /* [C], [C] (ls2.gamma)                                           */
    /* Optional low halves of the two projections whose one-time weight rounding dominates the 1e-3 parity budget:
     * op16 [C, 2C] = [W_hi | W_lo] along k with W_hi = op16(W), W_lo = op16(W - W_hi); NULL = single pass over qkv_w[2C:3C] / proj_w. */
    const void* v_w2;             /* blocks.i.attn.qkv.weight[2C:3C] as hi | lo, or NULL              */
    const void* proj_w2;          /* blocks.i.attn.proj.weight as hi | lo, or NULL                    */
    /* LayerNorm folded into the consuming linear (used when mhmr_vit_desc.pstats / rowstats are given and the token-row map is on):
     * flags bit 0: norm1 -> qkv (never for blocks[0]: the patch embedding leaves no row statistics, MHMR_ERR_BAD_ARG), bit 1: norm2 -> fc1;
     * a set bit needs its *_colsum (MHMR_ERR_BAD_ARG otherwise).  A folded linear's weight is W diag(w_ln) (v_w2 likewise), its bias is
     * b + W b_ln, and *_colsum[n] = sum_k of the ROUNDED folded weight row (hi + lo where there is a low half), fp32. */
    int flags;
    const float* qkv_colsum;      /* [3C] or NULL */
    const float* fc1_colsum;      /* [4C] or NULL */
    /* The same low halves for the fp8 low-half range of the big GEMMs (mhmr_vit_desc.lo8; round 5): rows of 3C BYTES = [W_hi: C op16 values |
     * e4m3(W_lo * 2^-e): C bytes], *_w8_scale = 127 + e (the E8M0 scale byte).  The weight's low half corrects the high half's 2^-12
     * rounding: it needs three significant bits, not eleven, and the fp8 matrix pipe runs at twice the 16-bit rate.  NULL = that linear
     * keeps the op16 low half (v_w2 / proj_w2, which the class-row kernel uses in either case).  For a folded V the rows hold W diag(w_ln). */
    const void* v_w8;
    const void* proj_w8;
    int v_w8_scale, proj_w8_scale;
} mhmr_vit_block;

typedef struct {
    int dtype;              /* MHMR_DT_*                                                                  */
    int B, S, C, H, L;      /* images, image size (S % 14 == 0), embed dim (= 64 H), heads, depth         */
    int G, N, T, Tp;        /* S/14, G*G, N+1, rows per image: T rounded up to a multiple of 64 (128 when a linear
                               of the encoder runs on the 128x128 kernel: B*Tp or C not a multiple of 256).
                               Token rows of image b: patch n (= y*G + x) at row b*Tp + n, the CLASS token at row b*Tp + N
                               (last: attention is permutation-equivariant), zeros behind it */
    int Kp;                 /* 588 rounded up to a multiple of 64 (= 640)                                 */
    const void* patch_w;    /* op16 [C, Kp]      patch_embed.proj.weight flattened (c,py,px), zero padded */
    const float* patch_b;   /* [C]                                                                        */
    const float* cls_pos0;  /* [C]               cls_token + pos[0]                                       */
    const float* pos;       /* [1+N, C]          pos_embed bicubically interpolated to G x G (host, once) */
    const mhmr_vit_block* blocks; /* HOST array of L block descriptors                                    */
    const float *norm_w, *norm_b; /* [C]          final norm                                              */
    /* workspaces (device) */
    void* a_patch;          /* op16 [roundup(B*N,128), Kp]   rows >= B*N must be zero                     */
    float* resid;           /* [B*Tp, C]        fp32 residual stream                                      */
    void* xn;               /* op16 [B*Tp, C]                                                             */
    void* qk;               /* op16 [B*Tp, 2C]  (Q | K)                                                   */
    void* vt;               /* op16 [B, H, 64, Tp]                                                        */
    void* att;              /* op16 [B*Tp, C]                                                             */
    void* hid;              /* op16 [B*Tp, 4C]                                                            */
    int* attn_flags;        /* [mhmr_attention_flag_count(B, Tp, H)] or NULL (then the self-contained attention form runs) */
    /* LayerNorm fold workspaces, or NULL (then every LayerNorm is a pass of its own and no block may have flags set): the residual
     * epilogues of proj / fc2 leave the 16-bit copy of the raw residual rows in `xn` and per-row block sums in `pstats`; row statistics
     * are finished by a small kernel into `rowstats`; the consuming linears normalise in their epilogues. */
    float* pstats;          /* [B*Tp, C/64, 2]  (sum, sum of squares) of every 64-column block of a residual row               */
    float* rowstats;        /* [B*Tp, 2]        (mean, rstd)                                                                  */
    /* x3 != 0: the "f16x3" precision mode -- for checkpoints whose statistics a single 16-bit rounding per operand does not survive
     * (multi_hmr_amd/vit.py logit_gain).  Every operand of every backbone linear is an op16 PAIR (hi = op16(v), lo =
     * op16(v - hi): 22 significant bits for f16) and every term is three products (a_hi w_hi + a_hi w_lo + a_lo w_hi) in one fp32
     * accumulator chain of the same MFMA kernels (mhmr_gemm16_ex: K = 3 a_k); LayerNorm, GELU (exact erf), the residual stream and the
     * WHOLE attention (mhmr_attention_f32) are fp32.  Layout changes against the fields above:
     *   patch_w op16 [C, 3 Kp] and every block weight op16 [N, 3 K] = [W_hi | W_lo | W_hi] along k (qkv_w [3C, 3C], proj_w [C, 3C],
     *   fc1_w [4C, 3C], fc2_w [C, 12C]); v_w2 / proj_w2 NULL, flags 0;  Tp % 128 == 0 (Tp % 256 for C % 256 == 0: every linear on the
     *   256x256 kernel); a_patch op16 [roundup(B*N,128), 2 Kp], xn / att op16 [B*Tp, 2C], hid op16 [B*Tp, 8C] = [hi | lo];
     *   qk, vt, attn_flags, pstats, rowstats unused (may be NULL). */
    /* lo8 != 0: `xn` and `att` are op16 [B*Tp, 3C/2] -- every row = [C op16 values | C bytes: the bf8 (e5m2) copy of the same values],
     * written by their producers (LayerNorm kernel, residual epilogue, attention) where the next linear has an fp8 low-half range
     * (blocks[i].v_w8 / proj_w8); C % 256 == 0.  lo8 == 0: rows of C op16 values, v_w8 / proj_w8 ignored. */
    int lo8;
    int x3;
    float* qkv32;           /* x3: [B*Tp, 3C] fp32  (Q | K | V), bias included                                                 */
    float* hid32;           /* x3: [B*Tp, 4C] fp32  fc1 output before the GELU                                                 */
    /* Split-k workspace, or NULL.  When every row goes through the 256x256 kernel (Tp % 256 == 0 with N % 256 == 0: the "all rows" form a
     * caller picks for tiny batches) and a residual linear (proj / fc2) has at most half as many 256x256 tiles as the device has CUs, its k
     * range is cut into slices whose fp32 partial tiles go here -- mhmr_splitk_workspace_bytes(B*Tp, C, 4C) bytes cover every linear -- and
     * a row-wise kernel sums them in slice order, applies bias / LayerScale / residual and leaves xn and rowstats (no ln_stats launch).
     * Deterministic; the summation order differs from the unsplit linear's, i.e. from the same image inside a large batch, at the 16-bit
     * noise level -- as the all-rows form itself already does. */
    float* splitk;
    long long splitk_bytes;
    /* C = 384 (ViT-S) on the 256x256 kernel.  cpad = 512 says: Tp % 256 == 0, pstats / rowstats given, and every array indexed by the
     * output channel of the three C-wide linears is zero-padded to 512 entries -- qkv_w [2C + 512, C] (the V rows are its last C rows + 128
     * zero rows), qkv_b and qkv_colsum [2C + 512], v_w2 [512, 2C], proj_w / proj_w2 [512, C | 2C], fc2_w [512, 4C], proj_b, ls1, fc2_b,
     * ls2 [512].  Those linears then run as N = 512 with the last 128 output columns masked (GemmArgs::n_valid), every block linear is
     * on the 256x256 kernel and the LayerNorm fold applies.  0 = C-wide linears of such a model run on the 128x128 kernel, no fold. */
    int cpad;
    /* op16 [B*Tp, C] or NULL.  Given (with Tp % 256 == 0, all rows through the 256x256 kernel) and the whole qkv linear at most one round
     * of 256x256 tiles (B*Tp/256 * 3C/256 <= CUs): blocks whose V has no low half run Q | K | V as ONE launch -- V row-major into this
     * buffer -- followed by a transpose into `vt`, instead of a Q | K and a V launch of half a round each. */
    void* v16;
    /* [B, C/16, 2] fp32 or NULL.  Given (with pstats / rowstats, under the token-row map): the class-row launches of proj / fc2 also run the
     * patch rows' row statistics (extra workgroups of the same launch) and leave block sums of the class rows here, which the class-row
     * launches of qkv / fc1 turn into (mean, rstd) themselves -- the forward then issues no statistics launch of its own (47 fewer
     * launches per ViT-L forward).  NULL = mhmr_ln_stats-style launches as before. */
    float* cls_pstats;
} mhmr_vit_desc;

/* x: [B,3,S,S] fp32 (ImageNet-normalised).  feat32: [B*N, C] fp32 patch features (token n = y*G + x).
 * ctx16: op16 [>= B*N rows, ldctx]; columns [0, C) receive the 16-bit copy of feat32.                       */
int mhmr_vit_forward(const mhmr_vit_desc* d, const float* x, float* feat32, void* ctx16, int ldctx, void* stream);

/* Building blocks, exported for unit tests and bisecting. */
int mhmr_gemm16(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* bias,
                const float* gamma, void* out, int ldo, const float* pos, int Np, int Tp, int H, int Mvalid, int epi,
                int dtype, void* stream);
/* The same with (a) a token-row map: logical activation / output row m = b * img_rows + n lives at physical row b * img_stride + n
 * (img_rows % 256 == 0; 0 = rows are physical), so a GEMM can cover the patch rows of every image and skip its class / padding rows;
 * (b) a low-half weight pass: W = [W_hi | W_lo] along k, K = 2 * a_k, the activation's k index wraps at a_k (0 = off);
 * (c) K = 3 * a_k: W = [W_hi | W_lo | W_hi], A = [A_hi | A_lo] (lda >= 2 a_k): the third k range reads the activation's second a_k
 *     columns -- three 16-bit products per term (the f16x3 mode).                                                                   */
int mhmr_gemm16_ex(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* bias,
                   const float* gamma, void* out, int ldo, const float* pos, int Np, int Tp, int H, int Mvalid, int epi,
                   int dtype, int img_rows, int img_stride, int a_k, void* stream);
/* mhmr_gemm16_ex with the LayerNorm fold (csrc/gemm256.hip; M, N % 256 == 0, K % 128 == 0 only).  Producer (epi = MHMR_EPI_RESID): x16 !=
 * NULL receives the op16 copy of the updated residual rows [rows, N] and pstats [rows, N/64, 2] the (sum, sum of squares) of each
 * 64-column block.  Consumer (epi = MHMR_EPI_OP16_QK / _VT / _OP16_GELU, bias = NULL): out = rstd_m (acc - mean_m colsum_n) + fbias_n
 * with rowstats [rows, 2] = (mean, rstd) as mhmr_ln_stats leaves them.                                                              */
int mhmr_gemm16_ln(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* bias,
                   const float* gamma, void* out, int ldo, int Tp, int H, int epi, int dtype, int img_rows, int img_stride,
                   int a_k, void* x16, float* pstats, const float* rowstats, const float* colsum, const float* fbias, void* stream);
/* mhmr_gemm16_ln for an output width that is a multiple of 128 but not of 256 (ViT-S: 384), on the 256x256 kernel: N = n_valid + 128, W
 * [N, K] and every per-column vector (bias, gamma, colsum, fbias) zero-padded to N entries by the caller; the last 128 columns are
 * computed and not stored.  epi = MHMR_EPI_RESID (out32 [M, ldo >= n_valid], x16 / pstats [M, n_valid / 64, 2] optional) or MHMR_EPI_VT
 * (n_valid / 64 heads; rowstats / colsum / fbias optional: the LayerNorm-fold consumer).  M % 256 == 0, K % 128 == 0. */
int mhmr_gemm16_masked(const void* A, int lda, const void* W, int ldw, int M, int N, int n_valid, int K, int a_k, const float* bias,
                       const float* gamma, void* out, int ldo, int Tp, int H, int epi, int dtype, void* x16, float* pstats,
                       const float* rowstats, const float* colsum, const float* fbias, void* stream);
/* The qkv linear of a SHORT batch as one launch + a transpose (what mhmr_vit_forward runs when mhmr_vit_desc.v16 is given): W [3C, K = C],
 * qk [B*Tp, 2C] = (Q * MHMR_ATTN_QSCALE | K), vt [B][H][64][Tp] with the key permutation of MHMR_EPI_VT, v16 [B*Tp, C] scratch.
 * B*Tp % 256 == 0, C % 256 == 0, Tp % 64 == 0.  rowstats / colsum [3C] / fbias [3C]: the LayerNorm-fold consumer form (then bias = NULL). */
int mhmr_qkv16(const void* A, int lda, const void* W, int ldw, int B, int Tp, int C, int H, const float* bias, void* qk, void* v16, void* vt,
               int dtype, const float* rowstats, const float* colsum, const float* fbias, void* stream);
/* Split-k residual linear (csrc/gemm256.hip SPLITK + csrc/vit_misc.hip splitk_resid_kernel): out32 += gamma * (A . W^T + bias) for a launch
 * that would otherwise occupy at most half of the CUs.  mhmr_splitk_workspace_bytes: bytes of fp32 partial tiles the pair needs for an
 * [M, N] output over K (0 = such a problem is not split: M, N % 256, K % 128, tiles <= CUs / 2, K >= 512).  a_k as in mhmr_gemm16_ex.
 * x16 (row pitch ldx16 elements, 0 = N) receives the op16 copy of the updated rows and rowstats [M, 2] their (mean, rstd) with the centred
 * variance (eps inside the square root); either may be NULL.  N must be 256, 512, 768 or 1024 (one wave per row in the reduction). */
long long mhmr_splitk_workspace_bytes(int M, int N, int K);
int mhmr_gemm16_splitk_resid(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int a_k, const float* bias,
                             const float* gamma, float* out32, void* x16, int ldx16, float* rowstats, float eps, float* ws,
                             long long ws_bytes, int dtype, void* stream);
/* rowstats[b*Tp + n] = (mean, rstd) of residual row (b, n): n < N from the block sums pstats[b*Tp + n][C/64][2], n == N (the class row)
 * from the fp32 row resid[b*Tp + N][C] itself.                                                                                       */
/* mhmr_gemm16_ln with an fp8 low-half range (csrc/gemm256.hip, GemmArgs::lo8; epi = MHMR_EPI_VT or MHMR_EPI_RESID only): A rows = [a_k op16 |
 * a_k bytes bf8 (e5m2) of the same values] (lda >= 3 a_k / 2), W rows = [a_k op16 | a_k bytes e4m3 of W_lo * 2^-e], w8_scale = 127 + e;
 * K is implied (a_k + a_k / 2 in 16-bit units), a_k % 256 == 0.  Producer side (MHMR_EPI_RESID, x16 != NULL): ldx16 = row pitch of x16 in
 * elements (0 = ldo) and x8_off > 0 = byte offset inside an x16 row for the bf8 copy of the new residual values.  lo8 = 0 makes it
 * mhmr_gemm16_ln with the producer's pitch options (then K = a_k, one pass). */
int mhmr_gemm16_lo8(const void* A, int lda, const void* W, int ldw, int M, int N, int a_k, int lo8, int w8_scale, const float* bias,
                    const float* gamma, void* out, int ldo, int Tp, int H, int epi, int dtype, int img_rows, int img_stride, void* x16,
                    int ldx16, int x8_off, float* pstats, const float* rowstats, const float* colsum, const float* fbias, void* stream);
int mhmr_ln_stats(const float* pstats, const float* resid, float* rowstats, int B, int N, int Tp, int C, float eps, void* stream);
/* The class-token rows of a block linear (csrc/vit_cls.hip): B rows, a_stride / o_stride elements apart.  epi 0: Q | K | V projection
 * (columns n_base + [0, N) of [Q * MHMR_ATTN_QSCALE | K | V]; Q, K -> out16 row, V -> column vcol of vt [B,H,64,Tp]); epi 1: out32 +=
 * gamma * (acc + bias); epi 2: out16 = gelu(acc + bias).  N % 16 == 0, K % 128 == 0, a_k as in mhmr_gemm16_ex.                     */
int mhmr_cls_linear16(const void* A, long long a_stride, const void* W, int ldw, int B, int N, int K, int a_k, const float* bias,
                      const float* gamma, void* out, long long o_stride, int n_base, int C, void* vt, int H, int Tp, int vcol,
                      int epi, int dtype, void* stream);
/* qk: op16 [B*Tp, 2C] = (Q * MHMR_ATTN_QSCALE | K), head h at columns h*64; vt: op16 [B,H,64,Tp] key-permuted V^T
 * (MHMR_EPI_VT); out: op16 [B*Tp, C] = softmax_2(Q K^T) V over the T real keys of each image.
 * `out` (here, in mhmr_attention16_ex and as mhmr_vit_desc.att) must be ZERO-INITIALISED ONCE by the caller: a 128-query workgroup whose
 * rows are all padding (rows >= T of an image) returns without storing, so those rows keep what was allocated; the linears behind read
 * them (row-local: real rows never depend on them) and NaN / Inf bit patterns there would be carried along.                       */
int mhmr_attention16(const void* qk, const void* vt, void* out, int B, int T, int Tp, int C, int H, int dtype,
                     void* stream);
/* The attention kernel forms (csrc/attention.hip), for tests and A/B measurements.  Every form subtracts a per-query reference
 * level from the scores inside the matrix pipe; they differ in how the level follows the row maximum:
 *   variant 6  (what mhmr_vit_forward runs since round 4) the arithmetic and flag protocol of variant 0 on v_mfma_f32_16x16x32 (a query's
 *              keys spread over four lanes; the 32x32x16 shape of the other forms costs 5-8 % more power per flop on this chip).
 *   variant 0  level = exact row maximum of key tile 0, no maximum afterwards; a workgroup in which
 *              a lane's tile sum of exp2(score - level) exceeded 2^limit_log2 (0 <= limit_log2 <= 15; 15 = "would leave the
 *              16-bit range", 0 = nearly every workgroup) sets its entries of `flags` and is recomputed by variant 1, launched
 *              right behind it on the same stream.  flags: int workspace of mhmr_attention_flag_count(B, Tp, H) entries (four
 *              per 128-query workgroup); the call writes every entry, the caller need not clear them.  T = 64 n + 1: the lone
 *              key of the last tile is folded in as a rank-1 update.
 *   variant 1  textbook online softmax (running maximum + subtract every tile).        flags unused (NULL)
 *   variant 2  level moves when the running maximum leaves a +-8 band (what mhmr_attention16 runs).   flags unused (NULL)
 *   variant 3  variant 2 with 8-wave workgroups.                                        flags unused (NULL)
 *   variant 4 / 5  the arithmetic of variant 0 with 64 queries per wave (two 32-query blocks: a wave's softmax of one block issues in
 *              the shadow of its MFMAs on the other; 256-query workgroups, 3- / 2-slot K/V ring); flags as for variant 0.
 *   variant 7 / 8 / 9  round-6 experiment forms of variant 6 (the next tile's copies in front of the score MFMAs, a three-slot K / V^T
 *              ring, both): same results, none faster inside the forward.
 *   variant 10 variant 6 with the lone query of T = 128 n + 1 (the class token, the last token row) on workgroups of its own -- vector
 *              ALU, exact online softmax in fp32 -- instead of a 128-query workgroup with one real row; any other T runs variant 6's
 *              launch.  Measured slower than variant 6 (round 6): kept for tests and A/B measurements.                                   */
int mhmr_attention16_ex(const void* qk, const void* vt, void* out, int B, int T, int Tp, int C, int H, int dtype,
                        float limit_log2, int variant, int* flags, void* stream);
int mhmr_attention_flag_count(int B, int Tp, int H);
/* variant 6 of mhmr_attention16_ex into rows of pitch ldo elements (>= C), with the bf8 (e5m2) copy of every output row at byte offset
 * o8 of the row (0 = none; 2C <= o8, o8 + C <= 2 ldo): the A operand of an output projection with an fp8 low-half range. */
int mhmr_attention16_pitch(const void* qk, const void* vt, void* out, int B, int T, int Tp, int C, int H, int dtype, int* flags, int ldo,
                           int o8, void* stream);
/* mhmr_layernorm16 into rows of pitch ld16 elements, with the bf8 copy at byte offset o8 (0 = none). */
int mhmr_layernorm16_pitch(const float* in, const float* w, const float* b, void* out16, int ld16, int o8, int rows, int C, float eps,
                           int dtype, void* stream);
/* The attention of the f16x3 mode (csrc/attention_f32.hip): qkv fp32 [B*Tp, 3C] = (Q | K | V) un-scaled, head h at columns h*64;
 * out op16 PAIR [B*Tp, 2C] = [hi | lo] of softmax(Q K^T / 8) V over the T real keys; every product on v_mfma_f32_16x16x4_f32 (exact fp32).
 * Rows >= T of an image are written as zeros or as the (finite) attention of a padding row: never left unwritten.   Tp % 64 == 0. */
int mhmr_attention_f32(const float* qkv, void* out, int B, int T, int Tp, int C, int H, int dtype, void* stream);
/* Producers of operand PAIRS in the f16x3 mode: out16 rows of 2 C (2 N) values = [hi = op16(y) | lo = op16(y - hi)].
 * mhmr_layernorm16_pair: y = LayerNorm(in row) (C in {384, 768, 1024}); mhmr_gelu16_pair: y = gelu_erf(in[m][n]), in fp32 [M, N], N % 4 == 0. */
int mhmr_layernorm16_pair(const float* in, const float* w, const float* b, void* out16, int rows, int C, float eps, int dtype,
                          void* stream);
int mhmr_gelu16_pair(const float* in, void* out16, long long M, int N, int dtype, void* stream);
int mhmr_layernorm16(const float* in, const float* w, const float* b, void* out16, int rows, int C, float eps,
                     int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Detection head.  Replaces Model.detection (model.py:133-158): mlp_classif -> sigmoid clamp (641-643) ->
 * max-pool NMS (620-638) -> threshold (612-617) -> (b, y, x) coordinates in torch.where order.
 * ---------------------------------------------------------------------------------------------------------- */
/* scores[m] = clamp(sigmoid(hid16[m] . w2 + b2), 1e-4, 1 - 1e-4); hid16 = relu(mlp_classif.0(features)) from
 * mhmr_gemm16(..., MHMR_EPI_OP16_RELU).                                                                      */
int mhmr_detect_scores(const void* hid16, int ld, const float* w2, const float* b2, float* scores, int rows, int C,
                       int dtype, void* stream);
/* pass 1: counts[b] = number of cells with nms(score) >= thr.  pass 2 (after the host prefix sum `base`):
 * ordered compaction into det_b/det_y/det_x/det_score.                                                       */
int mhmr_detect_count(const float* scores, int B, int G, int nms_kernel, float thr, int* counts, void* stream);
int mhmr_detect_write(const float* scores, int B, int G, int nms_kernel, float thr, const int* base, int* det_b,
                      int* det_y, int* det_x, float* det_score, void* stream);
/* The same into buffers of `cap` entries: detections whose position is >= cap are dropped (the caller compares info[3] of
 * mhmr_person_groups with cap afterwards). */
int mhmr_detect_write_cap(const float* scores, int B, int G, int nms_kernel, float thr, const int* base, int* det_b,
                          int* det_y, int* det_x, float* det_score, int cap, void* stream);
/* The person set's bookkeeping ON THE DEVICE -- what the reference does on the host after torch.where (model.py:146-151) and in
 * rebatch / pad_to_max (utils/tensor_manip.py:7-45): per-image counts -> base[b] (exclusive prefix sums, nullable), gstart
 * [ngroups_cap + 1], chunks [3 * nchunks_cap] as mhmr_hph_forward reads them (unused tail entries = empty groups / count-0 items) and
 * info[4] = {persons kept = min(total, cap), groups, chunks, total}.  counts [B] from mhmr_detect_count, or NULL: the counts are the
 * histogram of det_b[0..P) (training hook: the caller's idx, sorted by image).  Sufficient bounds: ngroups_cap = min(B, cap),
 * nchunks_cap = cap / 8 + min(B, cap).  One workgroup builds the tables in LDS: (2 B + ngroups_cap + 1 + 3 nchunks_cap) ints must fit
 * 60 KB (B <= 8192 and, e.g., 2048 images with 16 persons each), MHMR_ERR_BAD_SHAPE otherwise. */
int mhmr_person_groups(const int* counts, const int* det_b, int P, int B, int cap, int* base, int* gstart, int ngroups_cap,
                       int* chunks, int nchunks_cap, int* info, void* stream);

/* Camera embedding.  Replaces Model.embedd_camera (model.py:160-187) + inverse_perspective_projection
 * (utils/camera.py:30-48) + FourierPositionEncoding (blocks/camera_embed.py:9-58) with num_bands frequency bands per ray component
 * (model.py:39 camera_embedding_num_bands; <= 20): E = 3 + 6 num_bands channels (99 for the released checkpoints' 16).  zK: [B*N, E]
 * fp32; also writes op16 copies to ctx16[:, C:C+E] and zeros ctx16[:, C+E:ldctx] (ldctx - C <= 128).  freq: [3*num_bands] =
 * linspace(1, max_resolution / 2, num_bands) x3.                                                                     */
int mhmr_camera_embed(const float* K, const float* freq, int B, int G, int patch, float* zK, void* ctx16, int ldctx,
                      int C, int dtype, int num_bands, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Human Perception Head.  Replaces HPH.cross_attn_inputs / HPH.forward (model.py:479-593), TransformerDecoder
 * (blocks/cross_attn_transformer.py:302-359), rot6d_to_rotmat (utils/humans.py:12-22), roma.rotmat_to_rotvec
 * (model.py:291), Model.to_euclidean_dist (model.py:189-203), mlp_offset + loc (model.py:258, 272-275).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    const float *ln_sa_w, *ln_sa_b;  /* layers.l.0.norm                                  */
    const float* to_qkv;             /* [3*inner, dim]      layers.l.0.fn.to_qkv.weight  */
    const float *sa_out_w, *sa_out_b;/* [dim, inner],[dim]  layers.l.0.fn.to_out.0       */
    const float *ln_ca_w, *ln_ca_b;  /* layers.l.1.norm                                  */
    const void* to_kv16;             /* op16 [2*inner, Kc]  layers.l.1.fn.to_kv.weight, zero-padded columns */
    const float* to_q;               /* [inner, dim]        layers.l.1.fn.to_q.weight    */
    const float *ca_out_w, *ca_out_b;/* [dim, inner],[dim]  layers.l.1.fn.to_out.0       */
    const float *ln_ff_w, *ln_ff_b;  /* layers.l.2.norm                                  */
    const float *ff1_w, *ff1_b;      /* [mlp, dim],[mlp]    layers.l.2.fn.net.0          */
    const float *ff2_w, *ff2_b;      /* [dim, mlp],[dim]    layers.l.2.fn.net.3          */
} mhmr_hph_layer;

typedef struct {
    int dtype;
    int C, G, N;                 /* backbone dim, grid, tokens per image                                      */
    int Kc;                      /* context operand width: C + E (E = cam_dim) rounded up to a multiple of 64 */
    int dim, heads, mlp, depth;  /* 1024, xat_num_heads, 1024, xat_depth (model.py:122-126)                   */
    int nb;                      /* num_betas                                                                */
    int Ktok;                    /* token width C + E + 318 + nb + 3 rounded up to a multiple of 16          */
    int Ndec;                    /* 318 + nb + 3 + 10                                                        */
    int patch;                   /* 14                                                                       */
    int nearness;                /* model.py:196                                                             */
    float fn;                    /* S / (2 tan(30 deg)): focal length of the normalising 60-degree camera (utils/camera.py:71-77) */
    const float *off1_w, *off1_b, *off2_w, *off2_b; /* mlp_offset.{0,2}: [C,C],[C],[2,C],[2]                  */
    const float *cq_x, *cq_y, *cv_x, *cv_y;         /* cross_{queries,values}_{x,y}: [G, C+E]                 */
    const float* init_tail;      /* [318 + nb + 3] = init_body_pose | init_betas | init_cam                   */
    const float *tok_w, *tok_b;  /* to_token_embedding: [dim, Ktok] (zero padded), [dim] (+ pos_embedding[:,0]) */
    const mhmr_hph_layer* layers;/* HOST array of `depth`                                                    */
    const float *dec_w, *dec_b;  /* [Ndec, dim], [Ndec]: decpose|decshape|deccam|decexpression stacked, bias has the init_* added */
    /* workspaces for P persons (device, fp32 unless noted) */
    float* zc;      /* [P, C]        */
    float* token;   /* [P, Ktok]     */
    float* x;       /* [P, dim]      */
    float* xn;      /* [P, dim]      */
    float* t1;      /* [P, max(3*inner, mlp, C)] */
    float* t2;      /* [P, inner]    */
    float* kv;      /* [Mctx, 2*inner], Mctx = roundup(B*N, 128) */
    float* dec;     /* [P, Ndec]     */
    int* det_row;   /* [P]           */
    /* fixed-capacity callers (P = a capacity, the person count known on the device only): DEVICE pointer to the number of real persons
     * (mhmr_person_groups' info[0]); rows behind it are padding -- computed like persons, never allowed to touch the context operand.
     * NULL = all P rows are persons. */
    const int* nvalid;
    int cam_dim;    /* camera embedding channels E = 3 + 6 num_bands (0 = 99): context width C + E <= Kc, zK rows of E floats */
} mhmr_hph_desc;

/* Inputs: feat32 [B*N, C], zK [B*N, E], ctx16 op16 [Mctx, Kc] (features | camera | 0), detections det_{b,y,x}
 * [P] (sorted by (b, y, x)), gstart [ngroups+1] = person offsets of the non-empty images, chunks [nchunks*3] =
 * (image b, first person, count <= 8) cross-attention work items, K [B,3,3].  ngroups / nmax / nchunks size the launches and may be
 * UPPER BOUNDS when the tables come from mhmr_person_groups (empty groups and count-0 work items return at once).
 * Outputs: offset [P,2], loc [P,2], rotmat [P,53,3,3], rotvec [P,53,3], betas [P,nb], expr [P,10],
 * dist_pp [P] (raw), dist [P] (post-processed).                                                              */
int mhmr_hph_forward(const mhmr_hph_desc* d, const float* feat32, const float* zK, void* ctx16, const int* det_b,
                     const int* det_y, const int* det_x, int P, const int* gstart, int ngroups, int nmax,
                     const int* chunks, int nchunks, const float* K, int B, float* offset, float* loc, float* rotmat,
                     float* rotvec, float* betas, float* expr, float* dist_pp, float* dist, void* stream);

/* The decoder layer stack alone: `depth` x (self-attention among the queries of one image, cross-attention over the
 * image's N context tokens, GELU feed-forward), pre-norm, residual.  Replaces TransformerCrossAttn.forward of BOTH
 * blocks/cross_attn_transformer.py:239-261 (dim 1024 / 8 heads / mlp 1024 / depth 2) and the Anny variant
 * multi_hmr_anny/hph.py:114-151 (dim 512 / 16 heads / mlp 2048 / depth 8, context_dim = dim).  x [P, dim] is updated in
 * place; ctx16 op16 [roundup(B*N,128), Kc]; workspaces xn [P,dim], t1 [P, max(3*32*heads, mlp)], t2 [P, 32*heads],
 * kv [roundup(B*N,128), 64*heads] fp32; gstart / chunks as in mhmr_hph_forward.                                   */
int mhmr_xattn_layers_forward(const mhmr_hph_layer* layers, int depth, int dim, int heads, int mlp, int Kc, int N, int B,
                              int dtype, float* x, float* xn, float* t1, float* t2, float* kv, const void* ctx16,
                              const int* gstart, int ngroups, int nmax, const int* chunks, int nchunks, int P, void* stream);

/* Building blocks (unit tests). */
int mhmr_linear_f32(const float* X, int ldx, const int* row_idx, const float* W, int ldw, const float* bias,
                    const float* R, int ldr, float* Y, int ldy, int M, int N, int K, int act, void* stream);
int mhmr_layernorm_f32(const float* in, const float* w, const float* b, float* out, int rows, int C, float eps,
                       void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * SMPL-X layer.  Replaces SMPL_Layer.forward (blocks/smpl_layer.py:47-155) -> smplx.SMPLX.forward / lbs,
 * roma.rotvec_to_rotmat (:107), inverse_perspective_projection (:117-123), perspective_projection (:143-144).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    int V, Vp, Vl;        /* 10475; columns of the vertex kernel's operands: V rounded up to a multiple of 48 (= Vl, the kernel's tile) + the
                             tiles of the 72 extra joints (21 picked vertices + 51 barycentric landmarks) as VIRTUAL vertices: extra
                             joint e = 16 t + i owns column Vl + 48 t + 16 k + i, k = 0..2 = copies of its three corner vertices' columns
                             (a picked vertex: three copies of itself), Vp = Vl + 48 * 5                                       */
    int Kb;               /* 486 + nb + 10 rounded up to a multiple of 32 (width of the feature rows F); the vertex kernel is built for
                             Kb == 512, i.e. num_betas <= 16 (MHMR_ERR_BAD_SHAPE otherwise)                     */
    int nb, Kinf;         /* num_betas, max skinning influences per vertex                                   */
    int center_joint;     /* JOINT_NAMES.index(person_center) = 15 ('head'); < 0 = person_center None: nothing is
                             recentred and the pelvis is added to the translation (smpl_layer.py:128-130)       */
    const void* basis16;  /* f16, 1024 x [posedirs(486) | shapedirs(nb) | exprdirs(10) | 0], tile-major (the slice of a 48-vertex tile is
                             one contiguous block of 82944 values): [Vp/48]{ [Kb/8 - 8][3][48][8] the HIGH halves of k < Kb - 64 (pose
                             correctives: one product per term in the kernel), then [8][2][3][48][8] hi + lo of the last 64 k (the last
                             pose columns, every shape / expression direction: fp32 accuracy) }                              */
    const float* vtemp;   /* [3][Vp]           v_template, fp32                                                */
    const float* J0;      /* [55*3]            J_regressor . v_template                                      */
    const float* JS;      /* [55*3][nb+10]     J_regressor . [shapedirs | exprdirs]                          */
    const int* parents;   /* [55]                                                                            */
    const int* skin_idx;  /* [V][Kinf]         K-sparse skinning list (kept for tools; the kernel reads skin16)        */
    const float* skin_w;  /* [V][Kinf]                                                                       */
    const void* skin16;   /* f16 [Vp/48][8][2][48][8]: the DENSE skinning weights w[v][j] (joints 55..63 zero), hi + lo,
                             j = 8 * block + lane-local index: the B operand of the skinning GEMM            */
    const float* xbary;   /* [72*3]            corner weights of the extra joints: (1, 0, 0) for joints 55..75 (vertices picked by id),
                             lmk_bary_coords for 76..126 (faces[lmk_faces_idx] are the corners)                 */
    /* (105) The kinematic tree's level schedule, or NULL (then the pose kernel derives it from `parents` at run time).  int32 [16][256]:
     * pose_tasks[level][lane] = joint | parent << 8 (parent 0xff = a root) for lane = 12 * (position of the joint in its level's list, in
     * joint order) + element (0..8 of the rotation, 9..11 of the translation), -1 where the level has no such lane; pose_levels = number of
     * tree levels.  Only for trees of at most 16 levels with at most 21 joints per level (SMPL-X: 10 levels, <= 13 joints). */
    const int* pose_tasks;
    int pose_levels;
} mhmr_lbs_consts;

/* rotvec [P,53,3], betas [P,nb], expr [P,10], loc [P,2], dist [P], K [B,3,3], det_b [P] (image of each person).
 * Workspaces (fp32-sized, contents are f16 operand matrices): ws_F [roundup(P,16), Kb], ws_A [roundup(P,16), 768], ws_xf [P,24].
 * Outputs: v3d [P,V,3], v2d [P,V,2], j3d [P,127,3], j2d [P,127,2], transl [P,3]  (transl_pelvis = j3d[:,0]). */
int mhmr_lbs_forward(const mhmr_lbs_consts* c, const float* rotvec, const float* betas, const float* expr,
                     const float* loc, const float* dist, const float* K, const int* det_b, int P, float* ws_F,
                     float* ws_A, float* ws_xf, float* v3d, float* v2d, float* j3d, float* j2d, float* transl,
                     void* stream);

/* The same layer as ONE launch (round 5) -- built, tested bit-identical, and measured SLOWER than the two launches on MI355X (48.7 vs 40.2 us
 * at 160 persons: cross-XCD visibility costs an L2 write-back + invalidation, csrc/lbs.hip), so multi_hmr_amd.Model calls
 * mhmr_lbs_forward unless the environment says MHMR_LBS_FUSED=1.  The fused form: the pose work (Rodrigues, joint regression, kinematic chain: a 10 us latency chain) runs as the
 * leading workgroups of the vertex grid while the other workgroups already stream the blend basis; per-person ready flags in ws_sync
 * order the two.  ws_sync: [1 + roundup(P,16)] ints, ZERO before the first call; every call leaves it zero again (so a workspace can be
 * allocated and cleared once and reused by every later call on the same stream; two calls in flight need two workspaces).  Results are
 * bit-identical to mhmr_lbs_forward.  Falls back to mhmr_lbs_forward's two launches (ws_sync untouched) for P > 160 or when the device
 * has fewer CUs than the fused grid has workgroups (224 + 5 vertex tiles + roundup(P,16) / 12 pose workgroups). */
int mhmr_lbs_forward_fused(const mhmr_lbs_consts* c, const float* rotvec, const float* betas, const float* expr,
                           const float* loc, const float* dist, const float* K, const int* det_b, int P, float* ws_F,
                           float* ws_A, float* ws_xf, float* v3d, float* v2d, float* j3d, float* j2d, float* transl,
                           int* ws_sync, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Anny variant read-outs; the backbone, the detection head and the decoder stack reuse
 * mhmr_vit_forward / mhmr_detect_* / mhmr_xattn_layers_forward.
 *   mhmr_anny_camera: multi_hmr_anny/encoder.py:47-56 -- fov = fov_max * sigmoid(logit), focal = (S/2) / tan(fov/2),
 *                     K [B][3][3] with the principal point at S/2.
 *   mhmr_anny_decode: multi_hmr_anny/multi_hmr.py:144-177 -- per person: loc, dist = focal / clamp(exp(d), 1e-5),
 *                     transl = K^-1 [loc,1] dist (utils/camera.py:30-48), J 6D rotations (rows of (3,2)) ->
 *                     roma.special_gramschmidt -> identity where useful[j] == 0 -> roma.rotmat_to_rotvec,
 *                     shape = sigmoid(shape_logit).  The body model (anny package) is outside this library.
 * ---------------------------------------------------------------------------------------------------------- */
/* encoder.py:57-58: logits = hid16 . w2 + b2, scores = sigmoid(logits) (no clamp); hid16 as for mhmr_detect_scores. */
int mhmr_anny_scores(const void* hid16, int ld, const float* w2, const float* b2, float* scores, float* logits, int rows,
                     int C, int dtype, void* stream);
int mhmr_anny_camera(const float* fov_logit, int B, int img_size, float fov_max, float* fov, float* K, void* stream);
int mhmr_anny_decode(const float* rot6d, const float* useful, int J, const float* shape_logit, int nb,
                     const float* dist_logit, const float* offset, const int* det_b, const int* det_y, const int* det_x,
                     const float* K, int patch, int P, float* rotmat, float* rotvec, float* shape, float* loc, float* dist,
                     float* transl, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Input preprocessing, the step before Model.forward:
 *   demo.py:27-51 open_image  = PIL ImageOps.contain(img, (S,S)) [bicubic, aspect kept] + ImageOps.pad(.., (S,S))
 *                               [centred, black] + utils/image.py:12-24 normalize_rgb.
 * img: decoded uint8 RGB [H][W][3] on the device.  The resample is Pillow's 8-bit two-pass convolution in its
 * own fixed point (22 fractional bits), bit-identical to PIL: the host passes Pillow's coefficient tables
 * (multi_hmr_amd/preprocess.py: kh [ow][ksh] / kv [oh][ksv] int32, bounds bh [ow][2] / bv [oh][2] = (first tap,
 * tap count)) and the 3 x 256 normalisation table lut (the reference's numpy expression evaluated on 0..255).
 * Only source rows y0 .. y0+rows-1 (those the vertical taps touch) are resampled horizontally into tmp
 * [rows][ow][3] uint8.  out: [3][S][S] fp32, the resized image at (pad_x, pad_y), lut[c][0] elsewhere.
 * ---------------------------------------------------------------------------------------------------------- */
int mhmr_preprocess_u8(const void* img, int H, int W, const int* kh, const int* bh, int ksh, const int* kv,
                       const int* bv, int ksv, int ow, int oh, int y0, int rows, int S, int pad_x, int pad_y,
                       const float* lut, void* tmp, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Accuracy metrics of the reference's evaluation loop, train.py:372-395 (PVE, PA-PVE) and
 * 415-423 (MPJPE, PA-MPJPE): for M matched pairs of V points, pred / gt [M][V][3] fp32, optionally recentred by
 * pred_center / gt_center [M][3] (the pelvis translations; NULL = none):
 *   pve[m]    = mean_n |gt_n - pred_n| * 1000
 *   pa_pve[m] = mean_n |gt_n - (s R pred_n + t)| * 1000,  (R, t, s) = roma.rigid_points_registration(pred, gt,
 *               compute_scaling=True)  (proper rotation; scale = tr(R^T M) / sum |pred - mean|^2)
 * Rts (nullable) [M][13] receives R (row-major 9), t (3), s.
 * ---------------------------------------------------------------------------------------------------------- */
int mhmr_eval_mesh_errors(const float* pred, const float* gt, const float* pred_center, const float* gt_center, int M,
                          int V, float* pve, float* pa_pve, float* Rts, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Measurement: hipEvent brackets around every launch of one kernel family (0 = GEMM, 1 = attention, 2 = LBS
 * vertex kernel), recorded on the launch stream.  enable(kind >= 0) starts a fresh window, enable(-1) stops;
 * collect() synchronises the recorded events and returns launches, summed milliseconds and summed work
 * (FLOPs for 0/1, persons for 2).
 * ---------------------------------------------------------------------------------------------------------- */
int mhmr_prof_enable(int kind);
int mhmr_prof_collect(int* launches, double* total_ms, double* total_work);

#ifdef __cplusplus
}
#endif
#endif /* MHMR_H */
