/* wf3d.h — C ABI of libwf3d.so: the MI355X (gfx950) kernels behind the
 * PointNetEncoder -> VertexPredictor -> EdgePredictor forward/backward path.
 *
 * The reference (cansdev/wireframe-3d-prediction) has no FFI/operator layer: its
 * path is torch.nn modules calling ATen.  Each entry point
 * below therefore replaces the ATen op sequence of the cited reference lines;
 * the Python host (wireframe-3d-prediction_amd/wf3d/ops.py) binds them with
 * ctypes and the drop-in classes in wireframe-3d-prediction_amd/models/ call
 * them from torch.autograd.Function.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; every pointer is a DEVICE
 *     pointer borrowed for the call (no ownership taken, nothing allocated).
 *   - all tensors fp32, row-major, unless a name says otherwise
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*),
 *     never synchronises, never throws; returns WF3D_OK or a negative code and
 *     leaves a message for wf3d_last_error() (thread-local).
 *   - scratch comes from the caller: `ws` of at least the matching
 *     *_ws_bytes(...) bytes, 256-byte aligned.
 */
#ifndef WF3D_H
#define WF3D_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WF3D_VERSION 105 /* 101: wf3d_gemm_t gained `x3`; 102: `lr_u`, `lr_v`, `lr_k`, `ld_lr_u`, `ld_lr_v` (all appended); 103: wf3d_set_option; 104: wf3d_edge_pair_ln_bwd, wf3d_edge_pair_fwd_ln accepts pre = NULL; 105: `vd` (coordinates per vertex, 1..8) in the edge-head entry points, wf3d_meter_update, wf3d_parse_table */

#define WF3D_OK 0
#define WF3D_ERR_ARG (-1)
#define WF3D_ERR_LAUNCH (-2)
#define WF3D_ERR_WS (-3)
#define WF3D_ERR_UNSUPPORTED (-4)

/* activations fused into prologues / elementwise kernels */
#define WF3D_ACT_NONE 0
#define WF3D_ACT_RELU 1 /* nn.ReLU   — PointNetEncoder.py:38, VertexPredictor.py:30 */
#define WF3D_ACT_GELU 2 /* nn.GELU (erf) — EdgePredictor.py:34,59,63,66 */

/* GEMM operand layouts (k = reduction index) */
#define WF3D_NT 0 /* C[M,N] = A[M,K] · B[N,K]^T  — nn.Linear forward  (aten::addmm) */
#define WF3D_NN 1 /* C[M,N] = A[M,K] · B[K,N]    — Linear dgrad  dX = dY·W           */
#define WF3D_TN 2 /* C[M,N] = A[K,M]^T · B[K,N]  — Linear wgrad  dW = dY^T·X         */

int wf3d_version(void);
const char* wf3d_last_error(void);

/* Process-wide run-time switches (everything else is per call).
 *   "tn_rounds" (1..8, default 1 or $WF3D_TN_ROUNDS): workgroups per CU that the large weight-gradient launches of
 *       wf3d_gemm_split_tn are cut into.  1 is fastest on an idle chip; with a collective's kernels holding CUs beside
 *       the backward pass (data parallel) a launch of one workgroup per CU waits a whole extra
 *       round for its last workgroup (+67 %), 2 / 4 rounds lose 15 % / 0 % there and cost 2.5 % / 4 % otherwise.
 *   "gemm_cus" (0 = one workgroup per CU, the default; or 8..1024, rounded down to a multiple of 8): workgroups the
 *       persistent kernel of wf3d_gemm_split is launched with.  Its tiles are claimed, so fewer workgroups only leave
 *       CUs to whatever else the caller runs beside it (224 of 256 CUs: +9 % time; scripts/bench_overlap.py).
 * Returns WF3D_OK, or WF3D_ERR_ARG for an unknown name or a value out of range. */
int wf3d_set_option(const char* name, int value);

/* ------------------------------------------------------------------------
 * wf3d_gemm — fp32 MFMA (v_mfma_f32_32x32x2_f32) GEMM with a fused
 * LayerNorm-affine + activation (+dropout) PROLOGUE on the activation operand
 * and a bias / residual / accumulate EPILOGUE.
 *
 * Replaces, per reference Linear: aten::addmm (+ the native_layer_norm apply,
 * relu_/gelu and dropout that precede it), e.g. PointNetEncoder.py:35-45,94;
 * VertexPredictor.py:27-61,105-117; EdgePredictor.py:31-38,56-68,106,137;
 * and in backward the two mm's autograd derives per Linear.
 *
 * Prologue (applied while the operand tile is staged into LDS), on operand A
 * for WF3D_NT and on operand B for WF3D_TN, over that operand's own [rows, cols]
 * global shape:   v' = drop( act( (v - mu[row]) * rs[row] * gamma[col] + beta[col] ) )
 * pro_mu == NULL skips the LayerNorm part (v' = drop(act(v))).
 * ------------------------------------------------------------------------ */
typedef struct wf3d_gemm_t {
    const float* A;
    const float* B;
    float* C;
    const float* bias;   /* [N] or NULL                                         */
    const float* addend; /* [M, ld_addend] added in the epilogue, or NULL       */
    int M, N, K;
    int lda, ldb, ldc, ld_addend;
    int layout;          /* WF3D_NT / WF3D_NN / WF3D_TN                        */
    int pro_act;         /* WF3D_ACT_*; prologue enabled iff pro_enable != 0    */
    int pro_enable;
    const float* pro_mu;    /* [rows] or NULL */
    const float* pro_rs;    /* [rows]         */
    const float* pro_gamma; /* [cols] or NULL (NULL with pro_mu == NULL)        */
    const float* pro_beta;  /* [cols]         */
    float drop_p;           /* 0 = no dropout; else keep-prob 1-p, scaled 1/(1-p) */
    uint32_t drop_seed;
    int accumulate;      /* C += result instead of C = result                   */
    void* ws;            /* split-K slabs, wf3d_gemm_ws_bytes() bytes, or NULL  */
    size_t ws_bytes;
    int x3;              /* 0: exact fp32 MFMA.  1: bf16x3 arithmetic — the fp32 operands (after the prologue) are
                            split into bf16 (hi, lo) while they are staged into LDS and multiplied as
                            hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (fp32 accumulate; relative error
                            ~2^-16 per product).  Ignored for M <= 64 (32-row tile).                       */
    const float* lr_u;   /* epilogue low-rank term  C += U[M, lr_k] . V[N, lr_k]^T  (exact fp32), lr_k = 0..4, 0 = none:  */
    const float* lr_v;   /* the coordinate columns of the first edge Linear — Pa = F.Wa^T + c.Wc^T, EdgePredictor.py:     */
    int lr_k;            /* 130-137 — ride on the GEMM that already writes Pa instead of a K = 3 GEMM with accumulate      */
    int ld_lr_u, ld_lr_v;
} wf3d_gemm_t;

size_t wf3d_gemm_ws_bytes(int M, int N, int K, int layout);
int wf3d_gemm(const wf3d_gemm_t* desc, void* stream);

/* ------------------------------------------------------------------------
 * bf16x3 split-precision path for the large Linear layers (same results as the
 * fp32 path to ~1e-5, 3 bf16 MFMAs per product instead of one fp32 MFMA at 1/16
 * of the bf16 rate).  Operands are in the "sx8" format: a logical fp32 [R, C]
 * matrix (C % 8 == 0) with the same bytes/pitch as fp32, each 32-byte group of
 * 8 consecutive columns holding [8 x bf16 high | 8 x bf16 low], v ~= hi + lo.
 * ------------------------------------------------------------------------ */
/* C[M,N] (+)= A·B^T + bias, A = sx8[M,K], B = sx8[N,K] (nn.Linear forward with
 * B = W; its dgrad with A = dY, B = W^T in sx8) */
size_t wf3d_gemm_split_ws_bytes(int M, int N, int K);
int wf3d_gemm_split(const void* A_sx8, const void* B_sx8, float* C, const float* bias, int M, int N, int K,
                    int lda, int ldb, int ldc, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* C[Mo,No] (+)= A^T·B with A = sx8[K,Mo], B = sx8[K,No]: the Linear wgrad dW = dY^T·X on the
 * operands exactly as backward/forward leave them (no transposed copies; fragments are gathered
 * with ds_read_b64_tr_b16).  Supported when wf3d_gemm_split_tn_ok() (Mo % 256, No % 128, K % 32). */
int wf3d_gemm_split_tn_ok(int Mo, int No, int K, int lda, int ldb);
size_t wf3d_gemm_split_tn_ws_bytes(int Mo, int No, int K);
int wf3d_gemm_split_tn(const void* A_sx8, const void* B_sx8, float* C, int Mo, int No, int K, int lda, int ldb,
                       int ldc, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* out_sx8[r, c] = split(in[r*row_stride + c*col_stride])  (col_stride 1: convert;
 * row_stride 1: transpose-convert, used for W^T) */
int wf3d_split_rows(const float* in, long row_stride, long col_stride, int R, int C, void* out_sx8, void* stream);
/* the same for up to 8 matrices in one launch (host arrays of length njobs): a stage's weights, each as W and as W^T */
int wf3d_split_rows_multi(const float* const* in, const long* row_stride, const long* col_stride, const int* R, const int* C,
                          void* const* out_sx8, int njobs, void* stream);
/* out_sx8[C, R] = split(drop(act(LN-affine(in[R, C]))))^T — the wgrad operands (reduction
 * index = row index made contiguous): dW = dY^T·X becomes the NT-form
 * wf3d_gemm_split(dY^T_sx8, X^T_sx8).  mu/gamma NULL skip the respective part;
 * in_sx8 != 0: `in` is itself an sx8 matrix (no prologue), transposed plane-wise. */
int wf3d_split_transpose(const float* in, int R, int C, int ld, const float* mu, const float* rs,
                         const float* gamma, const float* beta, int act, float drop_p, uint32_t drop_seed,
                         int in_sx8, void* out_sx8, void* stream);
/* LayerNorm statistics of z[R,D] AND h = drop(act(LN(z))) in sx8, one read of z:
 * the operand producer between two split GEMMs (replaces native_layer_norm +
 * relu_ of PointNetEncoder.py:36-38 in the split path). */
int wf3d_ln_prep(const float* z, int R, int D, const float* gamma, const float* beta, int act, float eps,
                 float drop_p, uint32_t drop_seed, float* mu, float* rs, void* h_sx8, void* stream);
/* First Linear of the per-point MLP (in_features K <= 8) fused with its LayerNorm / activation / split
 * (models/PointNetEncoder.py:35-45, layer 0): writes z = x·W^T + b [R, D] (kept for backward), (mu, rstd) and
 * h = act(LN(z)) in sx8 — one pass instead of a GEMM that writes z plus a normalisation pass that reads it back. */
int wf3d_first_layer_fwd(const float* x, int R, int K, int ldx, const float* W, int ldw, const float* bias, int D,
                         const float* gamma, const float* beta, int act, float eps, float* z, float* mu, float* rs,
                         void* h_sx8, void* stream);
/* ... and its backward: LayerNorm/activation backward of layer 0 that also accumulates that layer's weight
 * gradient dW[D, K] = dz^T·x (K more weighted column sums) and never writes dz (the input takes no gradient).
 * dgamma, dbeta, dbias must be one contiguous [3][D] buffer. */
size_t wf3d_ln_act_bwd_first_ws_bytes(int R, int D);
int wf3d_ln_act_bwd_first(const float* dh, const float* z, const float* x, int R, int D, int K, int ldx,
                          const float* mu, const float* rs, const float* gamma, const float* beta, int act,
                          float* dgamma, float* dbeta, float* dbias, float* dW, void* ws, size_t ws_bytes, void* stream);
/* LayerNorm / activation backward that also returns one weighted column sum of the dz it produces,
 * wsum[c] = sum_r dz[r, c] * wrow[r] — for the first edge layer (models/EdgePredictor.py:130-137) the gradient of the
 * weight column that multiplies the pair distance — instead of a second pass over dz.  dgamma, dbeta: one [2][D] buffer. */
size_t wf3d_ln_act_bwd_wsum_ws_bytes(int R, int D);
int wf3d_ln_act_bwd_wsum(const float* dh, const float* z, const float* wrow, int R, int D, const float* mu, const float* rs,
                         const float* gamma, const float* beta, int act, float drop_p, uint32_t drop_seed, float* dz,
                         void* dz_sx8, float* dgamma, float* dbeta, float* wsum, void* ws, size_t ws_bytes, void* stream);


/* ------------------------------------------------------------------------
 * LayerNorm pieces (nn.LayerNorm, eps 1e-5, biased variance — SURVEY App. A)
 * ------------------------------------------------------------------------ */
/* mu[r], rs[r] = mean, 1/sqrt(var+eps) of row r of z[R, D] (row stride ld).
 * Replaces the statistics half of aten::native_layer_norm. */
int wf3d_row_stats(const float* z, int R, int D, int ld, float eps, float* mu, float* rs, void* stream);

/* out[r,c] = drop(act(LN(z)))[r,c] + addend[r,c]  (materialising form, used
 * where the value has several consumers: VertexPredictor.py:110,114 residual
 * sums, EdgePredictor.py:106,114).  mu == NULL skips the LayerNorm. */
int wf3d_ln_act_apply(const float* z, int R, int D, const float* mu, const float* rs,
                      const float* gamma, const float* beta, int act, const float* addend,
                      float drop_p, uint32_t drop_seed, float* out, void* stream);

/* Backward of h = drop(act(LN(z))) given dh: writes dz (may alias dh) and the
 * column reductions dgamma[D], dbeta[D] (LayerNorm affine grads) and
 * dbias[D] = sum_r dz[r,:] (the preceding Linear's bias grad).  Any of the
 * three outputs may be NULL; dz_sx8 (optional) receives dz in the sx8 split format
 * for wf3d_gemm_split, and then dz itself may be NULL.  Replaces native_layer_norm_backward +
 * threshold_backward / gelu_backward + the bias sum. */
size_t wf3d_ln_act_bwd_ws_bytes(int R, int D);
int wf3d_ln_act_bwd(const float* dh, const float* z, int R, int D, const float* mu, const float* rs,
                    const float* gamma, const float* beta, int act, float drop_p, uint32_t drop_seed,
                    float* dz, void* dz_sx8, float* dgamma, float* dbeta, float* dbias, void* ws,
                    size_t ws_bytes, void* stream);

/* out[c] = sum_r act(x[r,c]) * (w ? w[r] : 1)   — bias grads of Linears with no LN
 * behind them, the distance-weight grad of the split edge layer, and the weight
 * grad of a 1-output Linear (edge_mlp.10: dW[c] = sum_r dlogit[r] * gelu(z[r,c])). */
size_t wf3d_colsum_ws_bytes(int R, int D);
int wf3d_colsum(const float* x, int R, int D, int ld, const float* w, int act, float* out, void* ws,
                size_t ws_bytes, void* stream);

/* One-output Linear on act(z) — the logits layer edge_mlp[9..10] (models/EdgePredictor.py:66-67) — as a row dot
 * product, D = 8 * 2^k <= 512:  out[r] = bias[0] + sum_c act(z[r,c]) * w[c].
 * Backward, fused with the activation backward below it: dz[r,c] = dlogit[r] * w[c] * act'(z[r,c]) (fp32 and / or
 * sx8), dw[c] = sum_r dlogit[r] * act(z[r,c]), dbias_z[c] = sum_r dz[r,c]; dw and dbias_z are one [2][D] buffer. */
int wf3d_rowdot_act_ok(int D);
int wf3d_rowdot_act(const float* z, int R, int D, const float* w, const float* bias, int act, float* out, void* stream);
size_t wf3d_rowdot_act_bwd_ws_bytes(int R, int D);
int wf3d_rowdot_act_bwd(const float* z, const float* dlogit, int R, int D, const float* w, int act, float* dz,
                        void* dz_sx8, float* dw, float* dbias_z, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * "Skinny" Linear layers: M <= 32 rows (one per cloud of the batch) — the feature-fusion MLP
 * (models/PointNetEncoder.py:57-65,116) and the vertex head (models/VertexPredictor.py:94-117),
 * forward and backward.  Replaces, per nn.Linear: aten::addmm, the native_layer_norm / relu_ in
 * front of it and the residual add behind it; in backward the two mm's, native_layer_norm_backward,
 * threshold_backward and the bias sum.  Pure weight streaming (csrc/skinny.hip): LayerNorm
 * statistics travel as per-16-column (mean, M2) partials from the producing launch to the
 * consuming one, the LayerNorm backward's row sums as per-64-column partials, so neither pass
 * exists as a kernel.  Up to WF3D_SKINNY_MAX_PROBLEMS Linears per launch.
 * ------------------------------------------------------------------------ */
#define WF3D_SKINNY_MAX_ROWS 32
#define WF3D_SKINNY_MAX_PROBLEMS 4

/* Y[M,N] = x'·W^T + bias + out_addend,  x' = act(LayerNorm(X; gamma, beta)) + in_addend  (gamma NULL: x' = X + in_addend).
 * LayerNorm statistics of X: merged from in_part [in_nblk = K/16][M][2] (what the launch that wrote X left in its
 * stat_part) and, when mu_out != NULL, stored to mu_out/rs_out [M] for backward; or given as mu/rs [M]. */
typedef struct wf3d_skinny_fwd_t {
    const float* X; int ldx;
    const float* W; int ldw;            /* [N, K] (nn.Linear weight) */
    const float* bias;                  /* [N] or NULL */
    float* Y; int ldy;
    int N, K;
    const float* gamma; const float* beta; int act;      /* input prologue (WF3D_ACT_*) */
    const float* in_part; int in_nblk;  /* statistics partials of X, or NULL */
    const float* mu; const float* rs;   /* ... or final statistics */
    float* mu_out; float* rs_out;       /* [M] or NULL */
    const float* in_addend; int ld_in_addend;
    const float* out_addend; int ld_out_addend;
    float* stat_part;                   /* [N/16][M][2] (mean, M2) of Y per 16-column block, or NULL (needs N % 16 == 0) */
} wf3d_skinny_fwd_t;
int wf3d_skinny_ok(int M, int K);
int wf3d_skinny_fwd(const wf3d_skinny_fwd_t* probs, int nprob, int M, float eps, void* stream);

/* Backward of one Linear y = x'·W^T + b given the gradient of its output:
 *   plain form (z == NULL):  dz = dY
 *   LayerNorm form:          dY holds G = dh * act'(LN(z)) * gamma (wf3d_skinny_reduce), z/mu/rs this Linear's own
 *                            pre-LayerNorm output and statistics, rowpart [rowpart_nblk][M][2] the partial row sums of
 *                            (G, G*xhat):  dz = rs * (G - mean(G) - xhat * mean(G*xhat))
 * Outputs: dW[N,K] = dz^T·x' and db[N] = colsum(dz) (x' rebuilt from X with the forward's prologue; dW NULL skips
 * both), and slabs [ceil(N/nc)][M][K]: partial dgrad dX_s = dz[:, s*nc:(s+1)*nc]·W[s*nc:(s+1)*nc, :] (NULL skips). */
typedef struct wf3d_skinny_bwd_t {
    const float* dY; int lddy;
    const float* z; int ldz; const float* mu; const float* rs; const float* rowpart; int rowpart_nblk;
    const float* W; int ldw; int N, K;
    const float* X; int ldx;
    const float* xmu; const float* xrs; const float* xgamma; const float* xbeta; int xact;
    const float* xadd; int ldxadd;
    float* dW; int lddw; float* db;
    float* slabs; int nc;
} wf3d_skinny_bwd_t;
size_t wf3d_skinny_slab_floats(int M, int N, int K, int nc);
int wf3d_skinny_bwd(const wf3d_skinny_bwd_t* probs, int nprob, int M, void* stream);

/* v[M,K] = sum of up to three slab sets (fixed order: deterministic) + extra; optionally stored to dh.  With z != NULL
 * v is the gradient of h = act(LayerNorm(z)) and the elementwise half of that backward is applied in the same pass:
 * G = v*act'*gamma -> G[M,K], dgamma[K], dbeta[K], rowpart [ceil(K/64)][M][2] = per-block row sums of (G, G*xhat). */
typedef struct wf3d_skinny_red_t {
    const float* slabs[3]; int nslab[3];
    const float* extra; int ldextra;
    int K;
    float* dh; int lddh;
    const float* z; int ldz; const float* mu; const float* rs; const float* gamma; const float* beta; int act;
    float* G; int ldg; float* dgamma; float* dbeta; float* rowpart;
} wf3d_skinny_red_t;
int wf3d_skinny_reduce(const wf3d_skinny_red_t* prob, int M, void* stream);

/* ------------------------------------------------------------------------
 * Pools (PointNetEncoder.py:85-86,103-111 and VertexPredictor.py:86-88)
 * ------------------------------------------------------------------------ */
/* valid[m] = (sum_k |x[m,k]| > 1e-9) ? 1 : 0 */
int wf3d_point_valid(const float* x, int M, int in_dim, float* valid, void* stream);

/* One pass over point_features pf[B,N,C] producing all four reductions the two
 * heads need: masked max (0 where a cloud has no valid point) + its first-max
 * index, masked mean over max(count,1), unmasked mean, unmasked max + index,
 * and cnt[b] = max(#valid, 1). */
size_t wf3d_pool4_ws_bytes(int B, int N, int C);
/* The four pooled [B, C] outputs are written with row stride ldo (>= C), so that [max | avg] and [mean | max] can be
 * the two halves of the [B, 2C] vectors the fusion MLP / the vertex head consume (the reference's torch.cat,
 * PointNetEncoder.py:115, VertexPredictor.py:88); nvalid[b] (optional) = the unclamped number of valid points. */
int wf3d_pool4_fwd(const float* pf, const float* valid, int B, int N, int C, float* mmax, float* mavg,
                   float* umean, float* umax, int ldo, int32_t* arg_m, int32_t* arg_u, float* cnt, float* nvalid,
                   void* ws, size_t ws_bytes, void* stream);
/* dpf[b,n,c] = valid*dmavg/cnt + dumean/N + [n==arg_m]*dmmax + [n==arg_u]*dumax (+ dpf_direct);
 * the masked / unmasked cotangent pairs have row strides ldm / ldu (halves of [B, 2C] gradients). */
int wf3d_pool4_bwd(const float* valid, const float* cnt, const int32_t* arg_m, const int32_t* arg_u,
                   const float* dmmax, const float* dmavg, int ldm, const float* dumean, const float* dumax, int ldu,
                   const float* dpf_direct, int B, int N, int C, float* dpf, void* stream);
/* dbias[c] = sum over all B*N rows of the dpf above without dpf_direct — the bias gradient of the Linear that
 * produced point_features (encoder.mlp.16) — from the [B, C] cotangents alone. */
int wf3d_pool4_bwd_bias(const float* cnt, const float* nvalid, const int32_t* arg_m, const int32_t* arg_u,
                        const float* dmmax, const float* dmavg, int ldm, const float* dumean, const float* dumax, int ldu,
                        int B, int C, float* dbias, void* stream);
/* Same values written as the sx8 split operand of the output Linear's dgrad / wgrad GEMMs (C % 8 == 0):
 * saves the fp32 round trip through wf3d_split_rows. */
int wf3d_pool4_bwd_sx8(const float* valid, const float* cnt, const int32_t* arg_m, const int32_t* arg_u,
                       const float* dmmax, const float* dmavg, int ldm, const float* dumean, const float* dumax, int ldu,
                       const float* dpf_direct, int B, int N, int C, float* dpf_sx8, void* stream);

/* ------------------------------------------------------------------------
 * Vertex head tail (VertexPredictor.py:118-127): existence = sigmoid(o[:,:,3]),
 * actual_vertex_counts = sum(existence > 0.5) (int64), o = final_layer output
 * [B, V, vertex_dim >= 4]; backward: d_o[.., k] = d_o_in[.., k] for k < in_dim (d_o_in [B, V, in_dim] contiguous:
 * in_dim = 3 is the gradient of the `vertices` view o[:, :, :3] itself, in_dim = vertex_dim a gradient of o; may be
 * NULL), 0 elsewhere, with dexist*p*(1-p) added into channel 3.
 * ------------------------------------------------------------------------ */
int wf3d_vertex_finalize_fwd(const float* o, int B, int V, int vertex_dim, float* exist, int64_t* counts,
                             void* stream);
int wf3d_vertex_finalize_bwd(const float* exist, const float* dexist, const float* d_o_in, int in_dim, int B, int V,
                             int vertex_dim, float* d_o, void* stream);

/* ------------------------------------------------------------------------
 * Edge head (EdgePredictor.py:91-140, PointCloudToWireframe.py:77-112), batched
 * over samples with COMPACT ragged rows instead of the reference's serial
 * batch-1 loop: sample s owns vertex rows voff[s]..voff[s+1]-1 (its first
 * counts[s] predicted vertices) and edge rows eoff[s]..eoff[s+1]-1 (all pairs
 * i<j in lexicographic order); vsample/esample map a row back to its sample.
 * ------------------------------------------------------------------------ */
/* vd = coordinates per vertex, EdgePredictor(vertex_dim=...), 1..8 (the model uses 3: EdgePredictor.py:19,
 * PointCloudToWireframe.py:38).
 * cv[r,0:vd] = verts[s, r-voff[s], 0:vd]; strides in floats (vertices is a view of [B,V,4]) */
int wf3d_edge_gather_verts(const float* verts, long sample_stride, long vertex_stride, const int32_t* voff,
                           const int32_t* vsample, int Rv, int vd, float* cv, void* stream);
/* dverts[B,V,vd] = scatter of dcv rows, zero for v >= counts[s] */
int wf3d_edge_scatter_dverts(const float* dcv, const int32_t* voff, int B, int V, int vd, float* dverts, void* stream);

/* 8-head self-attention of nn.MultiheadAttention (EdgePredictor.py:41-46,109):
 * qkv[Rv, 3E] packed in_proj output; one workgroup per (head, sample); writes
 * ctx[Rv, E] (heads concatenated, before out_proj) and lse[Rv, heads].  head_dim 64
 * runs QK^T and PV (and the five backward contractions) on fp32 MFMA; backward takes
 * the forward's ctx (delta_i = dctx_i · ctx_i). */
int wf3d_attn_fwd(const float* qkv, const int32_t* voff, int S, int vmax, int E, int heads, float drop_p,
                  uint32_t drop_seed, float* ctx, float* lse, void* stream);
int wf3d_attn_bwd(const float* qkv, const float* dctx, const float* ctx, const float* lse, const int32_t* voff,
                  int S, int vmax, int E, int heads, float drop_p, uint32_t drop_seed, float* dqkv, void* stream);

/* Split first layer of edge_mlp (EdgePredictor.py:122-137):
 * pre[e,:] = Pa[i,:] + Pb[j,:] + |c_i - c_j| * wdelta  for edge e = (i, j), plus
 * the row's LayerNorm statistics and delta[e] = |c_i - c_j| over the vd coordinates of cv [Rv, vd].  wdelta is column
 * 2H+2vd of edge_mlp.0.weight, read with stride wdelta_stride (= 2H+2vd+1, or 1 for a gathered copy). */
int wf3d_edge_pair_fwd(const float* Pa, const float* Pb, const float* cv, const float* wdelta, int wdelta_stride,
                       const int32_t* voff, const int32_t* eoff, const int32_t* esample, int Re, int H, int vd, float eps,
                       float* pre, float* mu, float* rs, float* delta, void* stream);
/* Same, and also h = drop(act(LayerNorm(pre))) as the sx8 operand of the next Linear (edge_mlp[1..4] of
 * EdgePredictor.py:57-60) from the row the wave still holds: saves wf3d_ln_prep's second read of pre.
 * pre may be NULL: the pre-activation is then not stored at all (2 KB per edge row at hidden 512) and the backward
 * rebuilds it from Pa / Pb with wf3d_edge_pair_ln_bwd below. */
int wf3d_edge_pair_fwd_ln(const float* Pa, const float* Pb, const float* cv, const float* wdelta, int wdelta_stride,
                          const int32_t* voff, const int32_t* eoff, const int32_t* esample, int Re, int H, int vd, float eps,
                          float* pre, float* mu, float* rs, float* delta, const float* gamma, const float* beta, int act,
                          float drop_p, uint32_t drop_seed, void* h_sx8, void* stream);
/* LayerNorm / activation backward of that first edge layer with its pre-activation REBUILT, not read:
 * pre[e, :] = Pa[i, :] + Pb[j, :] + delta[e] * wdelta (the expression the forward kernel evaluated) from the per-vertex
 * tables, which stay L2-resident (V x H per sample).  dh [Re, H] -> dz (may alias dh), dgamma / dbeta (one [2][H]
 * buffer) and wsum[c] = sum_e dz[e, c] * delta[e], the gradient of the distance column of edge_mlp[0].weight
 * (EdgePredictor.py:130-137).  Same arithmetic as wf3d_ln_act_bwd_wsum on a stored pre. */
size_t wf3d_edge_pair_ln_bwd_ws_bytes(int Re, int H);
int wf3d_edge_pair_ln_bwd(const float* dh, const float* Pa, const float* Pb, const float* delta, const float* wdelta,
                          int wdelta_stride, const int32_t* voff, const int32_t* eoff, const int32_t* esample, int Re, int H,
                          const float* mu, const float* rs, const float* gamma, const float* beta, int act, float drop_p,
                          uint32_t drop_seed, float* dz, float* dgamma, float* dbeta, float* wsum, void* ws, size_t ws_bytes,
                          void* stream);
/* dPa[v] / dPb[v] = segmented sums of dpre over the edges where v is i / j, and
 * dcv[v] = sum over incident edges of (dpre[e]·wdelta)(c_v - c_other)/delta[e]
 *          (+ dPa[v]·Wc + dPb[v]·Wd when wcoord != NULL: wcoord[c * wcoord_stride + 0..2vd) = [Wc[c, :] | Wd[c, :]], the
 *          coordinate columns 2H..2H+2vd-1 of edge_mlp[0].weight, EdgePredictor.py:130-137).  dcv is [Rv, vd]. */
int wf3d_edge_pair_bwd(const float* dpre, const float* delta, const float* cv, const float* wdelta,
                       int wdelta_stride, const int32_t* voff, const int32_t* eoff, const int32_t* vsample, int Rv,
                       int H, int vd, float* dPa, float* dPb, float* dcv, const float* wcoord, int wcoord_stride, void* stream);

/* probs[s, j] = sigmoid(logit[eoff[s] + j]) for j < E_s and exactly 0.0 for the padding j >= E_s: the whole padded
 * [B, max_e] output of PointCloudToWireframe.py:103-112 in one pass (no zero-fill first), and its backward. */
int wf3d_edge_prob_fwd(const float* logit, const int32_t* eoff, int B, int max_e, float* probs, void* stream);
int wf3d_edge_prob_bwd(const float* probs, const float* dprobs, const int32_t* eoff, const int32_t* esample, int Re,
                       int max_e, float* dlogit, void* stream);

/* ------------------------------------------------------------------------
 * Row f-1 (next row after the hot path): losses/WireframeLoss.py — the
 * caller of backward.  cost: all B Hungarian cost matrices in one launch
 * (WireframeLoss.py:142-219; the assignment itself stays scipy on the host, :236);
 * terms: SmoothL1 over matched pairs + BCE(existence) + BCE(edges over the common
 * width) and their gradients w.r.t. vertices / existence / edge_probs (:60-104,241-283).
 * losses[4] = {vertex, existence, edge, weighted total}.
 * ------------------------------------------------------------------------ */
int wf3d_loss_cost_matrix(const float* verts, long sample_stride, long vertex_stride, const float* exist,
                          const float* tverts, int Vt, const int64_t* counts, int B, int V, float* cost,
                          void* stream);
/* Hungarian assignment on the device (Jonker-Volgenant shortest augmenting paths, fp64 duals,
 * one wave per sample) — replaces scipy.optimize.linear_sum_assignment + the per-sample
 * .cpu().numpy() sync of WireframeLoss.py:235-236.  col4row[b,p] = column of prediction p;
 * p is matched to a real target iff col4row[b,p] < counts[b]. */
int wf3d_loss_assign(const float* cost, int B, int V, int32_t* col4row, void* stream);
/* Same result for the wireframe cost (dummy columns >= counts[b] identical per row): solves the count x V rectangular
 * problem "real targets -> distinct predictions, cost c[p,t] - e_p" instead of the padded square one (V <= 64). */
int wf3d_loss_assign_counts(const float* cost, const int64_t* counts, int B, int V, int32_t* col4row, void* stream);
int wf3d_loss_terms_assigned(const float* verts, long sample_stride, long vertex_stride, const float* exist,
                             const float* edge, int Ep, const float* tverts, int Vt, const float* texist,
                             const float* tlabel, int Et, const int32_t* col4row, const int64_t* counts, int B, int V,
                             float w_vertex, float w_exist, float w_edge, float* dverts, float* dexist, float* dedge,
                             float* losses, void* ws, size_t ws_bytes, void* stream);
int wf3d_loss_terms(const float* verts, long sample_stride, long vertex_stride, const float* exist,
                    const float* edge, int Ep, const float* tverts, int Vt, const float* texist,
                    const float* tlabel, int Et, const int32_t* m_pred, const int32_t* m_tgt, const int32_t* m_off,
                    int n_match, int B, int V, float w_vertex, float w_exist, float w_edge, float* dverts,
                    float* dexist, float* dedge, float* losses, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Row f-2: the step tail of the reference's loop, train.py:141-142 —
 * torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm) followed by torch.optim.Adam.step()
 * (lr, betas, eps, L2 weight_decay; train.py:96) — over ALL tensors in two launches.
 * Tables are HOST arrays of `ntensors` device pointers / element counts.  params[i] == NULL marks a tensor that is only
 * part of the norm and only scaled (the lazily created point_pool_proj, which the reference's optimizer never sees,
 * SURVEY.md §9 Q1).  grads are scaled in place by min(1, max_norm / (norm + 1e-6)) exactly like clip_grad_norm_
 * (max_norm <= 0: no clipping); hyper-parameters are doubles so that 1 - beta, the bias corrections and lr / bc1 are
 * formed in double before their single rounding to fp32, as torch forms them; `step` is the 1-based Adam step of every tensor; total_norm (device scalar, optional)
 * receives the pre-clip global L2 norm.  ws: wf3d_clip_adam_ws_floats() floats.
 * ------------------------------------------------------------------------ */
size_t wf3d_clip_adam_ws_floats(const long* numel, int ntensors);
int wf3d_clip_adam_step(float* const* params, float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                        const long* numel, int ntensors, double max_norm, double lr, double beta1, double beta2, double eps,
                        double weight_decay, int step, float* ws, size_t ws_floats, float* total_norm, void* stream);

/* Training-step meter (reference train.py:145-157: total_loss.item() twice and a host copy of sample 0's vertices on EVERY
 * step, for the loss history, the best loss and a monitoring RMSE).  One small launch per step folds them into a
 * device-resident record; the loop copies it back only when it logs.
 *   state[0] = steps recorded, [1] = best total loss, [2] = best vertex RMSE, [3..7] = last total / vertex / existence /
 *   edge loss and RMSE, [8 .. 8 + capacity) = ring of the last `capacity` total losses (step t at 8 + t % capacity);
 *   the caller zeroes state[0] once.  RMSE = sqrt(mean((pred - target)^2)) over the first count0[0] (<= max_v) vertices
 *   x 3 coordinates of sample 0; pred_stride / target_stride = floats between consecutive vertices.
 *   vertex_loss / existence_loss / edge_loss / count0 may be NULL (recorded as 0 / count = max_v). */
int wf3d_meter_update(const float* total, const float* vertex_loss, const float* existence_loss, const float* edge_loss,
                      const float* pred_vertices, long pred_stride, const float* target_vertices, long target_stride,
                      const int64_t* count0, int max_v, float* state, int capacity, void* stream);

/* ------------------------------------------------------------------------
 * Row f-3: the input pipeline, datasets/building3d.py:95-158.  Clouds are parsed once, normalised once on
 * the device in float64 (the raw coordinates are UTM metres: colour / 256 for columns [color_lo, color_hi), centroid and
 * max-norm of xyz, :101-121) and stay resident; every batch is then ONE gather kernel: random rows (:127), flips and the
 * rotation about z (:130-145) applied in float64 and rounded to fp32 once (:158).  The random choices [B, P] and
 * aug[b] = {flip_x (+-1), flip_y (+-1), cos t, sin t} come from the host's numpy RNG in the reference's draw order.
 * Packed layout: cloud i owns rows first[i] .. first[i+1]-1 of raw / out ([rows, C] float64).
 * ------------------------------------------------------------------------ */
int wf3d_cloud_normalize(const double* raw, const long* first, int nclouds, int C, int color_lo, int color_hi, int normalize,
                         double* out, double* centroid, double* max_distance, void* stream);
int wf3d_cloud_sample(const double* norm, const long* first, const int* cloud, const int* choice, const double* aug, int B,
                      int P, int C, float* out, void* stream);
/* HOST function: the numbers of a whitespace-separated text file (.xyz: 8 per row; np.loadtxt, :98) into out[0..max_vals);
 * returns the count in the file (call again with a larger buffer if > max_vals), -1 unreadable, -2 not a number. */
long wf3d_parse_floats(const char* path, double* out, long max_vals);
/* Same (`#` comments skipped, C-locale numbers), and *ncols = values per row — np.loadtxt's second dimension;
 * -3 if two non-empty rows hold different numbers of values. */
long wf3d_parse_table(const char* path, double* out, long max_vals, long* ncols);

/* ------------------------------------------------------------------------
 * Row f-4: evaluation post-processing behind the path, evaluate.py:74-110 and
 * eval/ap_calculator.py:8-36.
 * wf3d_edge_endpoints: for every sample s and every candidate edge e = (i < j < counts[s]) in the model's lexicographic
 * order: keep[s, e] = probs[s, e] > threshold (:79) and edge_vertices[s, e] = the two predicted end points, higher z first
 * (:88-89; equal z: vertex j first) — all samples in one launch, so the host needs one copy instead of one per sample.
 * verts strides in floats (the `vertices` output is a view).  Entries e >= E_s are left untouched.
 * wf3d_hausdorff_lines: out[n, m] = max(h(P_n, T_m), h(T_m, P_n)) over `sample_points` (<= 32) equidistant points per
 * segment, point k = start + w_k * diff with np.linspace's weights; float64.
 * ------------------------------------------------------------------------ */
int wf3d_edge_endpoints(const float* verts, long sample_stride, long vertex_stride, const int32_t* counts, const float* probs,
                        int B, int V, int max_e, float threshold, float* edge_vertices, unsigned char* keep, void* stream);
int wf3d_hausdorff_lines(const double* p_start, const double* p_diff, const double* t_start, const double* t_diff, int N, int M,
                         int sample_points, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WF3D_H */
