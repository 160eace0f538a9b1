/*
 * gamer_hip.h -- C ABI of libgamer_hip.so: the MI355X (gfx950) kernels of the Qwen3Multi
 * SMB-decoder train step.
 *
 * The reference (wzf2000/GAMER) is pure Python/PyTorch and has no FFI of its own; the boundary
 * it exposes for this path is the nn.Module `Qwen3MultiWithTemperature`
 * (ref:SeqRec/models/generative/Qwen3Multi/model.py:883-1013).  `gamer_amd/` mirrors that module
 * in Python and drives the entry points below through ctypes (INTEGRATION.md shows the stub).
 * Each entry point names the reference code it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HIP), fp32 / bf16 / int32 / int64 as typed, 16-byte aligned
 *     (gamer_bf16 rows: 8-byte aligned, leading dims multiples of 4 elements unless an entry point says more);
 *   - matrices are row-major with an explicit leading dimension (elements) that is a multiple of 4;
 *   - token-major activations: row t = b*S + s;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - return value: 0 = ok, <0 = bad argument (see gamer_last_error()), >0 = hipError_t;
 *   - no entry point allocates, frees or synchronises (all are hipGraph-capturable).
 *   - dropout: keep-mask = hash(seed, element index) >= p*2^32, scale 1/(1-p); the same
 *     (seed, index) pair regenerates the mask in the backward kernels.  p = 0 disables it.
 */
#ifndef GAMER_HIP_H
#define GAMER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GAMER_ABI_VERSION 9
#define GAMER_AMAX_WORDS 256      /* words of one maximum slot (1 KB): see gamer_absmax_f32 */

/* bf16 activations of the AMP variant (the reference's --bf16 run, ref:SeqRec/tasks/train_SMB_decoder.py:114-118,
 * 407-408: HF Trainer autocast): raw bfloat16 bits.  Entry points with the suffix _bf16 are the same operation with
 * every ACTIVATION pointer (GEMM operands / results, what autocast holds in bf16) typed gamer_bf16; parameters, the
 * residual stream, normalisation statistics, losses, gradients of parameters and optimizer state stay fp32.        */
typedef uint16_t gamer_bf16;

int gamer_abi_version(void);
/* Thread-local message for the last <0 / >0 return of any entry point. */
const char* gamer_last_error(void);
/* The GAMER_* environment switches that select between kernels of one entry point (INTEGRATION.md lists them) are read once per
 * process and cached; after changing one inside a running process (tests, A/B tools) call this to have them read again.
 * Not on the reference's path (it has no such switches).  ABI version 9. */
int gamer_reload_env(void);

/* ------------------------------------------------------------------------------------------
 * Router + mask predicates + expert lists (integer work).
 * Replaces Qwen3MultiDecoderRouter.forward (ref:.../Qwen3Multi/router.py:74-201) and the mask
 * builders _update_session_wise_causal_mask / _update_session_multi_cross_mask
 * (ref:.../Qwen3Multi/model.py:691-741, 573-630): the [B,1,S,S] additive masks are never
 * materialised; the attention kernels evaluate  allowed(i,j) = j<=i && klevel[j] < qlevel[i].
 *   ids, attn_mask (may be NULL = all ones), actions (may be NULL = zeros): int64 [B,S]
 *   behavior_lut: int32 [vocab], token id -> behaviour index (0..NB-1) or -1
 * outputs, all int32 [B,S]:
 *   expert      position index 0..num_positions (0 = pad/eos)        (router.py:83-104)
 *   beh_idx     behaviour index for the FFN injection                 (router.py:110-148)
 *   act_idx     behaviour index for the cross-attention biases        (router.py:158-195)
 *   kl_self / kl_cross   key level   (INT32_MAX when attention_mask[j]==0)
 *   ql_cross             query level (= actions[i]); the self query level is the constant 1
 *   empty_self / empty_cross   1 when row i has no allowed key (softmax is then uniform over
 *                              all S keys, see model.py:586-602 + SDPA)
 *   tile_empty_self / tile_empty_cross  int32 [B, ceil(S/32)]: any empty row in that 32-row tile
 *   bad_token   int32 [1]: incremented for every behaviour-position token that is neither
 *               pad/eos nor in behavior_lut (the reference would raise IndexError later)
 * ---------------------------------------------------------------------------------------- */
int gamer_router_fwd(const int64_t* ids, const int64_t* attn_mask, const int64_t* actions,
                     const int32_t* behavior_lut, int vocab, int B, int S, int num_positions,
                     int pad_id, int eos_id,
                     int32_t* expert, int32_t* beh_idx, int32_t* act_idx,
                     int32_t* kl_self, int32_t* kl_cross, int32_t* ql_cross,
                     int32_t* empty_self, int32_t* empty_cross,
                     int32_t* tile_empty_self, int32_t* tile_empty_cross,
                     int32_t* bad_token, void* stream);

/* Session spans for the Qwen3SessionMulti variant (same weights, session-wise masks;
 * ref:.../Qwen3SessionMulti/model.py:545-551 in-item mask, :556-613 cross mask, :676-728 self mask, :983-984
 * RoPE positions = extended_session_ids).  The reference's masks
 *   self : allowed unless (not (same item and j <= i)) and sess[j] >= sess[i]
 *   cross: allowed unless sess[j] >= sess[i] or act[j] >= act[i]              (both: padded keys masked)
 * become per-query key spans when the session ids do not decrease along the kept tokens (the dataset's layout,
 * SMB_dataset.py:194-222): with lim_i = 1 + last kept j with sess[j] < sess[i],
 *   span_self [B,S,4] int32 = (i, lim_i, first token of i's item, 0)   keys j <= i except lim_i <= j < item start
 *   span_cross[B,S,4] int32 = (lim_i - 1, INT32_MAX, 0, 0)             keys j <= lim_i - 1 (and kl[j] < ql[i])
 * which gamer_attn_fwd / gamer_attn_bwd take as `q_span` (hi, hole_lo, hole_hi, unused).
 * Call after gamer_router_fwd (kl_cross / ql_cross are inputs): empty_* and tile_empty_* are OVERWRITTEN with the
 * session masks' empty rows.  pos_ids int32 [B,S] = extended_session_ids clamped to [0, n_rope_positions)
 * (extended_session_ids NULL: the index in the sequence).  violations int32 [1] is incremented for every row
 * whose ids are out of order (allowed keys not a causal prefix) or whose position is out of range; such rows
 * are clamped, the host decides whether to raise.                                                      */
int gamer_session_spans(const int64_t* session_ids, const int64_t* extended_session_ids,
                        const int64_t* attn_mask, const int32_t* kl_cross, const int32_t* ql_cross,
                        int B, int S, int num_positions, int n_rope_positions,
                        int32_t* span_self, int32_t* span_cross, int32_t* pos_ids,
                        int32_t* empty_self, int32_t* empty_cross,
                        int32_t* tile_empty_self, int32_t* tile_empty_cross, int32_t* violations,
                        void* stream);

/* Expert token lists for the position-routed FFN (replaces the boolean-mask gather/scatter loop of
 * MyQwen3SparseMLP.forward, ref:SeqRec/models/generative/Qwen3Moe/FFN.py:63-68, and its 6 host
 * syncs per layer).  Deterministic order: expert-major, then token order.
 *   expert int32 [T]; perm int32 [T] (sorted slot -> token); slot int32 [T] (token -> sorted slot);
 *   offsets int32 [num_experts+1]; work int32 [(B+1)*num_experts] scratch.                     */
int gamer_expert_lists(const int32_t* expert, int B, int S, int num_experts,
                       int32_t* perm, int32_t* slot, int32_t* offsets, int32_t* work, void* stream);

/* ------------------------------------------------------------------------------------------
 * Embedding (nn.Embedding(vocab,H,padding_idx), ref:.../Qwen3Multi/model.py:263,779).
 * bwd: dW[ids[t]] += dx[t] for ids[t] != pad_id (atomic adds into an already initialised dW:
 * the tied head's wgrad is written first).
 * ---------------------------------------------------------------------------------------- */
int gamer_embedding_fwd(const int64_t* ids, const float* W, int V, int T, int H, float* x, void* stream);
int gamer_embedding_bwd(const int64_t* ids, const float* dx, int V, int T, int H, int pad_id, float* dW,
                        void* stream);
/* The same gradient WITHOUT float atomics (same bits on every run): a stable counting sort of the tokens by id (integer
 * counters only), then every table row adds its tokens' rows of dx in token order - pieces of 256 tokens, the pieces in order.
 * dW += as above.  ws: scratch of gamer_embedding_bwd_ordered_ws_bytes(V, T, H) bytes, 16-byte aligned; V <= 8191, H % 4 == 0. */
int64_t gamer_embedding_bwd_ordered_ws_bytes(int V, int T, int H);
int gamer_embedding_bwd_ordered(const int64_t* ids, const float* dx, int V, int T, int H, int pad_id, float* dW,
                                void* ws, int64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * RMSNorm over the hidden dim (Qwen3RMSNorm, transformers/models/qwen3/modeling_qwen3.py;
 * call sites model.py:205,222,239,869).  y[dst(t)] = w * x[t] * rsqrt(mean(x[t]^2)+eps)
 *   dst_rows: optional int32 [T] row scatter (token -> sorted slot) used for the FFN input;
 *   ldy: leading dim of y (>= H), lets the FFN input carry the behaviour embedding after col H.
 * bwd: dx[t] (+)= rstd*(w*dy - xhat*mean(w*dy*xhat)) ; dw_partial[block][H] deterministic partials
 *   that gamer_colsum_reduce folds into dw.  mask_out (optional): the updated dx is the gradient of the residual
 *   stream and the next consumer is a residual branch with dropout; its input gradient mask(seed) * dx is written to
 *   mask_out[mask_rows ? mask_rows[t] : t] in the same pass (= gamer_residual_dropout_bwd(dx, ..., seed)).
 * ---------------------------------------------------------------------------------------- */
int gamer_rmsnorm_fwd(const float* x, const float* w, int T, int H, float eps,
                      const int32_t* dst_rows, float* y, int ldy, void* stream);
int gamer_rmsnorm_bwd(const float* x, const float* w, const float* dy, int lddy,
                      const int32_t* dy_rows, int T, int H, float eps, int accumulate_dx,
                      float* dx, float* dw_partial, int n_partial,
                      float* mask_out, const int32_t* mask_rows, float p_drop, uint64_t seed, void* stream);
/* bf16: y / dy / mask_out are activations (what the next GEMM reads, what a dgrad GEMM wrote); x, dx stay fp32. */
int gamer_rmsnorm_fwd_bf16(const float* x, const float* w, int T, int H, float eps,
                           const int32_t* dst_rows, gamer_bf16* y, int ldy, void* stream);
int gamer_rmsnorm_bwd_bf16(const float* x, const float* w, const gamer_bf16* dy, int lddy,
                           const int32_t* dy_rows, int T, int H, float eps, int accumulate_dx,
                           float* dx, float* dw_partial, int n_partial,
                           gamer_bf16* mask_out, const int32_t* mask_rows, float p_drop, uint64_t seed, void* stream);
/* out[c] (+)= sum_r partial[r][c] */
int gamer_colsum_reduce(const float* partial, int rows, int cols, int accumulate, float* out,
                        void* stream);
/* the same for n tables `stride` floats apart in one launch: outs[i][c] (+)= sum_r partial[i * stride + r * cols + c]; outs is a
 * DEVICE array of n device pointers (the RMSNorm weight gradients of a whole backward pass at its end)                        */
int gamer_colsum_reduce_batched(const float* partial, int64_t stride, int rows, int cols, int n, float* const* outs,
                                int accumulate, void* stream);

/* Behaviour-embedding concat for the injected FFN layers (FFN.py:60-62):
 * y[dst(t)][col0 .. col0+E) = table[idx[t]];   bwd: dtable[idx[t]] += dy[dst(t)][col0..]
 * bwd, (ABI 7) `partial` (scratch of partial_numel floats, at least n_rows_table * E; NULL = the round-1 form with float atomics):
 * every workgroup writes its table to a row of it and a second kernel folds the rows in a fixed order - the same bits on
 * every run.                                                                                                    */
int gamer_rowtable_fwd(const float* table, const int32_t* idx, const int32_t* dst_rows, int T, int E,
                       float* y, int ldy, int col0, void* stream);
int gamer_rowtable_bwd(const float* dy, int lddy, int col0, const int32_t* idx,
                       const int32_t* dy_rows, int T, int E, int n_rows_table, float* dtable,
                       float* partial, int64_t partial_numel, void* stream);
int gamer_rowtable_fwd_bf16(const float* table, const int32_t* idx, const int32_t* dst_rows, int T, int E,
                            gamer_bf16* y, int ldy, int col0, void* stream);
int gamer_rowtable_bwd_bf16(const gamer_bf16* dy, int lddy, int col0, const int32_t* idx,
                            const int32_t* dy_rows, int T, int E, int n_rows_table, float* dtable,
                            float* partial, int64_t partial_numel, void* stream);

/* ------------------------------------------------------------------------------------------
 * fp32 MFMA GEMM  (v_mfma_f32_32x32x2_f32; exact fp32, replaces every nn.Linear on the path:
 * model.py:93-99,145-149,1001 and FFN.py:25-27).
 *     C[m][n] (=|+=) alpha * sum_k A(m,k) * B(n,k)
 * Operand element (r,k) lives at ptr[r*rs + k*ks]; exactly one of rs/ks is 1.
 *   Linear fwd   Y = X W^T      : A = X (ks=1),  B = W (ks=1)
 *   dgrad        dX = dY W      : A = dY (ks=1), B(n,k) = W[k][n] (rs=1)
 *   wgrad        dW = dY^T X    : A(m,k) = dY[k][m] (rs=1), B(n,k) = X[k][n] (rs=1)
 * Grouping (position-routed experts): group_offsets int32 [groups+1] on the DEVICE, or NULL for
 * one group.
 *   group_mode 0: rows of A/C are segmented (C rows offsets[g]..offsets[g+1] use B + g*strideB)
 *   group_mode 1: the contraction index is segmented (wgrad): C + g*strideC = sum over k in
 *                 segment g; split along k in chunks of `kchunk`, combined with fp32 atomics
 *                 (C must be zero-initialised or hold the value to accumulate into).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const float* A; int64_t a_rs, a_ks;
    const float* B; int64_t b_rs, b_ks;
    float* C; int64_t ldc;
    int M, N, K;
    float alpha;
    int accumulate;              /* 0: C = , 1: C += (mode 0 only) */
    int groups;                  /* >= 1 */
    int group_mode;              /* 0 rows, 1 contraction */
    const int32_t* group_offsets;/* device, groups+1 entries, or NULL when groups == 1 */
    int64_t strideB, strideC;    /* elements between consecutive groups */
    int kchunk;                  /* mode 1: contraction chunk per workgroup (multiple of 32) */
    /* Optional fused epilogue of the decoder layer's residual sites (model.py:217,235,241), Linear-forward
     * layout only:  C[map(m)][n] = resid[map(m)][n] + dropout(alpha * acc[m][n])  with map = row_map
     * (sorted slot -> token, for the expert down projection) or identity; the dropout mask is the one
     * gamer_residual_dropout_bwd regenerates from (seed, element index of C).  resid == NULL: plain GEMM. */
    const float* resid;
    const int32_t* row_map;
    float p_drop;
    uint64_t seed;
    /* Optional row-dot epilogue (plain layouts, M % 128 == 0, N % 128 == 0, one group): with heads = N / 64,
     *   rowdot_out[(m / rowdot_S) * heads * rowdot_S + head * rowdot_S + m % rowdot_S] =
     *       sum over the 64 columns of `head` of C[m][.] * rowdot_other[m][.]      (rowdot_other has leading dim ldc)
     * i.e. flash attention's delta[b][head][i] = dO . O computed by the o_proj dgrad GEMM that produces dO, while the
     * tile is still in LDS.  rowdot_out == NULL: off. */
    const float* rowdot_other;
    float* rowdot_out;
    int rowdot_S;
    /* Optional q|k|v epilogue (Linear-forward layout, one group, M % 128 == 0, N == (qk_nq + 2 * qk_nkv) * 64, no other
     * epilogue): the projection of Qwen3MultiAttention (model.py:88-101) leaves the GEMM as
     *   C (= q|k|v, the values the backward needs: with the behaviour-level biases added when qk_bias_q != NULL),
     *   qk_q_rot [M, qk_nq*64] / qk_k_rot [M, qk_nkv*64] = RoPE(RMSNorm_head(q or k) * weight)
     * exactly as gamer_qknorm_rope_fwd computes them from C - while the tile is still in LDS (a wave's 64-column patch is
     * one head), saving that kernel's pass over q|k|v.  Arguments as gamer_qknorm_rope_fwd.  qk_q_rot == NULL: off. */
    const float* qk_wq; const float* qk_wk;
    float qk_eps;
    const float* qk_cos; const float* qk_sin;
    const float* qk_bias_q; const float* qk_bias_k; const float* qk_bias_v;
    const int32_t* qk_act_idx;
    const int32_t* qk_pos_ids;
    float* qk_q_rot; float* qk_k_rot;
    int qk_S, qk_nq, qk_nkv;
    /* optional (ABI 5), gamer_gemm_f32_split with group_mode 0 only: B pre-cut into its three bf16 pieces by
     * gamer_split3_planes - three bf16 arrays indexed exactly like B (piece s of B's element e at b_planes + s *
     * b_plane_stride + e; 8-byte aligned).  Full tiles then stage B without the cut (weights are re-cut by every row tile
     * otherwise: 4040 times per step at batch 1024); edge tiles and gamer_gemm_f32 ignore it.  Results are bit-identical. */
    const gamer_bf16* b_planes;
    int64_t b_plane_stride;
    /* (ABI 6) gamer_gemm_f32_split with terms = 3 only: device words holding the bits of max |A| and max |B| over the operand
     * tensors (gamer_absmax_f32; for grouped B: over all groups) - the kernel derives each tensor's power-of-two scale from them. */
    const uint32_t* amax_a;
    const uint32_t* amax_b;
    /* (ABI 7) optional, group_mode 0 with the plain or row-dot epilogue (no accumulate / resid / q|k|v epilogue): the kernel folds
     * the bits of max |C[m][n] stored| over the columns n >= amax_c_col0 (a multiple of 64) into the amax slot `amax_c` (as
     * gamer_amax_sink does for the element-wise producers): the maximum of an operand of the NEXT matrix product - the v columns
     * of the q|k|v projection, dO out of the o_proj input gradient - without a pass of gamer_absmax_f32 over it.  NULL = off. */
    uint32_t* amax_c;
    int amax_c_col0;
    /* (ABI 7) optional, group_mode 1 (weight gradient): a workspace of at least (number of k chunks, with one extra per group when
     * group_offsets is given) x ceil(M / 128) x ceil(N / 128) x 16384 floats.  The chunks then store their partial tiles there
     * with plain stores and a second kernel adds them to C in chunk order: a DETERMINISTIC weight gradient (same bits on every
     * run; the default combines the chunks with fp32 atomics, whose order varies).  NULL = atomics. */
    float* wgrad_ws;
    int64_t wgrad_ws_floats;
    /* SwiGLU-backward epilogue (gamer_gemm_f32_split, terms = 3; group_mode 0, alpha = 1, no other epilogue): this GEMM is the
     * input gradient of the experts' down projection, C[m][n] = d(hm)[m][n] with N = the intermediate size.  C is NOT stored;
     * instead sw_gu - the [M][sw_ld] buffer of the fused gate|up projection, gate in columns 0 .. N - 1, up in N .. 2 N - 1 - is
     * overwritten with d gate | d up exactly as gamer_swiglu_bwd_ld(sw_gu, sw_ld, M, N, C, p_drop, seed) would do (p_drop / seed of
     * this descriptor): d(hm) never goes to memory.  amax_c (amax_c_col0 = 0) then receives max |d gate|, |d up|.  NULL = off. */
    float* sw_gu;
    int64_t sw_ld;
    /* (ABI 9) group_mode 0 with group_offsets: several consecutive row groups share one B - group g multiplies with
     * B + (g / group_div) * strideB.  The injecting layers' experts (ref:SeqRec/models/generative/Qwen3Moe/FFN.py:60-68) sort their
     * rows by (expert, behaviour): group_div = num_behavior + 1 groups per expert.  0 / 1 = one B per group.  The 128 x 128 kernel
     * only (the kernels of include-note "kernel choice" take descriptors with group_div <= 1). */
    int group_div;
    /* (ABI 9) with sw_gu: a table [groups][2 N] added to the gate | up values of group g's rows BEFORE the SwiGLU backward is
     * evaluated on them - the share of the gate|up projection that depends on the row's group only (gamer_inject_table_fwd:
     * the behaviour-embedding columns of the injecting layers), which gamer_swiglu_fwd_ld_tbl added on the fly in the forward
     * without storing the sums.  NULL = off. */
    const float* sw_tbl;
    /* (ABI 9) gamer_gemm_f32_split(terms = 3), Linear-forward layout (both operands k-contiguous) with N = 256 and b_planes given:
     * the packed pieces of B^T ([K][256], dense; gamer_split2h_transpose_multi from b_planes) at B's offset in their buffer.  Lets
     * the output-stationary kernel (csrc/gemm_os.hip) take the forwards with 256 output features and a long contraction - o_proj
     * (K = 384), the experts' down projection (K = 512, grouped) - with the plain store or the residual epilogue.  NULL = off. */
    const gamer_bf16* b_planes_t;
    /* (ABI 9) SwiGLU FORWARD in the epilogue (gamer_gemm_f32_split, terms = 3; Linear-forward layout, N = 2 I: the experts' fused
     * gate|up projection, ref:SeqRec/models/generative/Qwen3Moe/FFN.py:25-27): besides C = gate | up the call leaves
     * sw_hm[m][c] = dropout(silu(gate[m][c] + tg) * (up[m][c] + tu)), [M][N / 2] contiguous, exactly what
     * gamer_swiglu_fwd_ld(C, ldc, M, N / 2, p_drop, seed, sw_hm) - with sw_tbl: gamer_swiglu_fwd_ld_tbl(..., sw_tbl, sw_row_group) -
     * computes from C (same arithmetic, same dropout masks), without the pass over C when the activation-stationary kernel takes the
     * call; otherwise the library runs that kernel itself after the GEMM.  amax_c then receives max |sw_hm|.  sw_tbl here is
     * [groups][N] with the row groups of group_offsets (group_div groups per B); sw_row_group: int32 [M], the group of every row.  */
    float* sw_hm;
    const int32_t* sw_row_group;
} gamer_gemm_desc;

int gamer_gemm_f32(const gamer_gemm_desc* d, void* stream);
/* (ABI 9) planes_t <- the packed pieces (see b_planes) of the TRANSPOSES of n row-major matrices whose pieces are in `planes`:
 * table = int64 [n][3] = (element offset, rows, cols), rows and cols multiples of 4; the transpose's pieces land at the same offset. */
int gamer_split2h_transpose_multi(const gamer_bf16* planes, const int64_t* table, int n, gamer_bf16* planes_t, void* stream);

/* The same GEMM (same descriptor, same layouts, epilogues and fp32 inputs / outputs) with the products formed on the
 * bf16 matrix pipe: every operand value is cut EXACTLY into three bf16 pieces (x = x0 + x1 + x2, 8 + 8 + 8 significant
 * bits) on its way into LDS and a.b is accumulated in fp32 from the piece products a_i.b_j, each of which is exact:
 * terms = 9 -> all nine (no bit of either operand is dropped; the only rounding is the fp32 accumulation of
 * v_mfma_f32_32x32x16_bf16), terms = 6 -> those with i + j <= 2 (the three omitted products are each below 2^-24 |a.b|).
 * Same F.linear sites as gamer_gemm_f32 (model.py:93-99,145-149,1001, FFN.py:25-27); selected per Engine
 * (`matmul="split6" | "split9"`), never implicitly.  Inf / NaN operands produce NaN. */
int gamer_gemm_f32_split(const gamer_gemm_desc* d, int terms, void* stream);
/* (ABI 7) Row-range guard of terms = 3, Linear-forward and input-gradient layouts (group_mode 0, A k-contiguous): one scale per
 * operand TENSOR means a row of A whose largest magnitude is 2^r below the tensor's is only computed to 2^(r - 38) of ITS OWN
 * result - an outlier element anywhere in A does that to every other row.  With the guard ON (the default) the kernel tracks
 * the row maxima of the A tile it stages and a tile holding a non-zero row more than 2^16 below the tensor's maximum is
 * recomputed from the fp32 operands with v_mfma_f32_32x32x2_f32 (exact fp32 products) before its epilogue - decided on the
 * device, no host synchronisation; other tiles are bit-identical to the unguarded kernel.  Not covered: the B operand and the
 * weight-gradient layout (error there stays relative to sum |a_k b_k|).  gamer_split3_guard(1 / 0) turns it on / off for the
 * process, a negative argument only queries; returns the PREVIOUS setting (not an error code). */
int gamer_split3_guard(int on);
/* A tensor's maximum ("amax slot") is GAMER_AMAX_WORDS consecutive 32-bit words of device memory, zero before the first kernel
 * that writes it: producers publish the bits of max |x| into words 16 i (i = workgroup index modulo 16; same-address atomics of
 * thousands of workgroups would serialise), consumers take the maximum over them.
 * terms = 3 (ABI 6): the two-way fp16 form - every operand value times its tensor's power-of-two scale s (largest magnitude
 * brought into [2^13, 2^14)) is cut into h0 = fp16(x s), h1 = fp16(x s - h0) and a.b = (a0.b0 + a0.b1 + a1.b0) / (s_a s_b) is
 * accumulated in fp32 by v_mfma_f32_32x32x16_f16: |x s - h0 - h1| <= 2^-22 |x s| for elements within 2^17 of the tensor's
 * largest (smaller ones: absolute 2^-25 of the scaled unit).  Against fp64 the result is at the six-product form's error
 * (rms 1.2e-8 of sum |a_k b_k|; fp32 MFMA 4.4e-8) at half its matrix instructions.  Needs d->amax_a / d->amax_b.
 * Behind this entry point, chosen per descriptor (same layouts, same results contract; environment switches for A/B runs in
 * brackets): the 128 x 128-tile kernel of csrc/gemm.hip; for a plain Linear forward with K <= 256, packed B pieces and >= 16 k rows
 * the activation-stationary kernel of csrc/gemm_as.hip, which scales A per ROW (amax_a is not read; no guard needed)
 * [GAMER_GEMM_AS=0]; for a plain input gradient with N = 256 the output-stationary kernel of csrc/gemm_os.hip, whose guard is a
 * second pass with a scale per row [GAMER_GEMM_OS=0]; for a weight gradient (group_mode 1) with wgrad_ws whose C is whole
 * 256 x 256 tiles the large-tile kernel of csrc/gemm_wg.hip, bit-identical to the 128 x 128 kernel [GAMER_GEMM_WG=0].
 *
 * gamer_absmax_f32: *out = max(*out, bits of max |x|) over `batch` matrices [rows, cols] with leading dimension ld
 * (ld % 4 == 0, 16-byte aligned), `stride` elements apart; *out must hold 0 (or an earlier maximum) on entry.  One pass over
 * x; a NaN anywhere leaves a NaN pattern (the GEMM then scales by 1 and the NaN propagates). */
int gamer_absmax_f32(const float* x, int batch, int64_t stride, int rows, int cols, int64_t ld, uint32_t* out, void* stream);
/* n dense tensors inside one buffer in one launch (the parameters a pass multiplies by): table[2 e] = offset, table[2 e + 1] =
 * numel of tensor e in elements (multiples of 4; int64, device memory), out + e * GAMER_AMAX_WORDS = its slot. */
int gamer_absmax_multi_f32(const float* base, const int64_t* table, int n, uint32_t* out, void* stream);
/* The `b_planes` operand of gamer_gemm_f32_split(terms = 3): PACKED pieces at the offsets of the values they replace - the 16
 * bytes of elements 4 i .. 4 i + 3 of entry e of `table` (as gamer_absmax_multi_f32; slots = that call's output, which fixes the
 * scale s(e) exactly as the GEMM derives it from amax_b) hold the four fp16 pieces h0 of base[.] * s(e), then the four pieces h1.
 * The GEMM then loads B as before (from `b_planes` + B's offset; b_plane_stride is not used) and stages it without the cut:
 * parameters are cut once per pass instead of once per row tile of the activations; results are bit-identical.  A B whose
 * contiguous extent (K when k-contiguous, N when row-contiguous) is not a multiple of 4 is cut in the kernel as if b_planes were
 * NULL (a packed quad cannot be masked element-wise at a ragged edge). */
int gamer_split2h_planes_multi(const float* base, const int64_t* table, int n, const uint32_t* slots, gamer_bf16* planes,
                               void* stream);
/* The same maximum as a by-product of the kernel that WRITES the tensor: gamer_amax_sink(out0, out1) arms the next launch - on
 * the calling host thread - of one of gamer_rmsnorm_fwd (y), gamer_rowtable_fwd (the columns it writes), gamer_swiglu_fwd (hm),
 * gamer_swiglu_bwd (out0: d gate, out1: d up), gamer_silu_gate_bwd (out0: da, out1: dgate), gamer_rmsnorm_bwd (mask_out),
 * gamer_ce_bwd (d logits), gamer_qknorm_rope_fwd (out0: q_rot, out1: k_rot; fp32), gamer_attn_fwd_split (o), gamer_attn_bwd_split
 * (dv), gamer_qknorm_rope_bwd (dq, dk; fp32): the kernel folds the bits of max |value it stores| into *out0 / *out1
 * (atomicMax; words hold 0 or an earlier maximum; NULL = none) and the launch disarms the sink.  Any other launch in between
 * leaves it armed.  Saves the separate pass of gamer_absmax_f32 over the tensor. */
int gamer_amax_sink(uint32_t* out0, uint32_t* out1);
/* the same with a third word: gamer_qknorm_rope_fwd with behaviour biases (the cross block) folds max |v + bias_v| - the v columns
 * of q|k|v as the attention reads them - into *out2 (out0 / out1: q_rot / k_rot as above) */
int gamer_amax_sink3(uint32_t* out0, uint32_t* out1, uint32_t* out2);
/* gamer_attn_split_amax(q, k, v, d_o) arms the next gamer_attn_fwd_split / gamer_attn_bwd_split on the calling host thread to form
 * its products in the three-product fp16 form of gamer_gemm_f32_split(terms = 3): q, k, v (d_o: backward; NULL in the forward)
 * are device words holding the bits of max |.| of those operand tensors; P is scaled by 2^13, dS by a bound derived from the
 * maxima.  Same results within the forms' tolerances (tests/test_ops_gpu.py runs the attention grid in this form at the same
 * bars); not armed = the six-product bf16 form.  The armed form needs p_drop < 0.75 (P / (1 - p_drop) must stay below 2^3 for its
 * fixed 2^13 scale; the entry points reject larger values - use the six-product form there). */
int gamer_attn_split_amax(const uint32_t* q, const uint32_t* k, const uint32_t* v, const uint32_t* d_o);
/* planes[s * plane_stride + i] = piece s (s = 0, 1, 2) of x[i] under the exact three-way cut above (n % 4 == 0,
 * plane_stride % 4 == 0, 16-byte aligned x, 8-byte aligned planes): the `b_planes` operand of gamer_gemm_f32_split.  The
 * engine cuts its fp32 master parameters once per step with it (98 MB read, 147 MB written). */
int gamer_split3_planes(const float* x, gamer_bf16* planes, int64_t n, int64_t plane_stride, void* stream);

/* ------------------------------------------------------------------------------------------
 * bf16 MFMA GEMM (v_mfma_f32_32x32x16_bf16, fp32 accumulation): the nn.Linear sites of the reference's --bf16 run
 * (autocast casts both operands of every F.linear to bf16, model.py:93-99,145-149,1001, FFN.py:25-27).
 *   group_mode 0  C[m][n] = sum_k A[m][k] * B[n][k]: BOTH operands k-contiguous (row-major [M,K] and [N,K], leading
 *       dims lda / ldb, K % 64 == 0).  Linear forward: A = X, B = W.  dgrad dX = dY W: A = dY, B = the TRANSPOSED
 *       bf16 copy of W (gamer_cast_params_bf16 keeps it), K = the layer's output width (pad columns must be finite
 *       in A and zero in B).  C is bf16 [M, ldc] (accumulate: C += ), or - with `resid` - the fp32 residual stream:
 *       C[map(m)][n] = resid[map(m)][n] + dropout(acc) exactly as gamer_gemm_f32's fused epilogue (same mask).
 *       Row segments (experts) as in gamer_gemm_f32.  rowdot_*: as gamer_gemm_f32, `other` bf16, the dot product uses
 *       the rounded C.
 *   group_mode 1  (wgrad) C[m][n] += sum_t A[t][m] * B[t][n]: A [K rows, lda] and B [K rows, ldb] token-major bf16
 *       (lda >= round_up(M, 8), ldb >= round_up(N, 8): whole 16-byte chunks are read), C fp32 [M, ldc] accumulated
 *       with atomics over token chunks of `kchunk` (multiple of 64); the token range may be segmented (experts,
 *       C + g * strideC).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const gamer_bf16* A; int64_t lda;
    const gamer_bf16* B; int64_t ldb;
    void* C; int64_t ldc;        /* gamer_bf16* (mode 0), float* (mode 0 with resid, mode 1) */
    int M, N, K;
    int accumulate;              /* mode 0, bf16 C only */
    int groups;
    int group_mode;
    const int32_t* group_offsets;
    int64_t strideB, strideC;
    int kchunk;
    const float* resid;
    const int32_t* row_map;
    float p_drop;
    uint64_t seed;
    const gamer_bf16* rowdot_other;
    float* rowdot_out;
    int rowdot_S;
    /* optional (ABI 5): the q|k|v projection of an attention with qknorm_rope_fwd's work in the epilogue, as qk_* of
     * gamer_gemm_desc but with the AMP arithmetic of gamer_qknorm_rope_fwd_bf16: C = raw q|k|v (bf16; v + bias_v in the cross
     * attention), q_rot / k_rot = RoPE(w * rmsnorm(q | k (+ bias))) as bf16 [M, nq*64] / [M, nkv*64].  qk_q_rot != NULL
     * selects it; needs N = (nq + 2 nkv) * 64, M %% 128 == 0, one group, no accumulate / resid / rowdot. */
    const float* qk_wq; const float* qk_wk; float qk_eps;
    const float* qk_cos; const float* qk_sin;
    const float* qk_bias_q; const float* qk_bias_k; const float* qk_bias_v;
    const int32_t* qk_act_idx; const int32_t* qk_pos_ids;
    gamer_bf16* qk_q_rot; gamer_bf16* qk_k_rot;
    int qk_S, qk_nq, qk_nkv;
    /* SwiGLU-backward epilogue (group_mode 0, no other epilogue, no accumulate): as gamer_gemm_desc.sw_gu - this GEMM is the input
     * gradient of the experts' down projection; C = d(hm), rounded to bf16, is consumed in the tile and NOT stored, sw_gu (the
     * [M][sw_ld] bf16 buffer of the fused gate|up projection, N = the intermediate size) becomes d gate | d up exactly as
     * gamer_swiglu_bwd_ld_bf16 would make it (p_drop / seed of this descriptor).  NULL = off. */
    gamer_bf16* sw_gu;
    int64_t sw_ld;
    /* group_mode 1: the ordered form of the weight gradient, as gamer_gemm_desc.wgrad_ws (chunk partial tiles to this scratch of
     * (chunks + groups if grouped) x tiles x 16384 floats, then added to C in chunk order: no fp32 atomics).  NULL = atomics. */
    float* wgrad_ws;
    int64_t wgrad_ws_floats;
} gamer_gemm_bf16_desc;

int gamer_gemm_bf16(const gamer_gemm_bf16_desc* d, void* stream);

/* bf16 operand copies of the fp32 master parameters, refreshed after every optimizer step (one launch):
 * table[5 * e + ...] describes matrix e of the flat fp32 buffer: { src offset, dst offset in `out`, dst offset in
 * `out_t` or -1, rows | cols << 32, ldt | first_tile << 32 } with first_tile = sum over earlier entries of
 * ceil(rows / 32) * ceil(cols / 32); n_tiles = that sum over all entries.  out[dst + r * cols + c] = bf16(W[r][c]);
 * out_t[dst_t + c * ldt + r] = bf16(W[r][c]) (columns >= rows of a row of out_t are never written: keep them zero). */
int gamer_cast_params_bf16(const float* flat, gamer_bf16* out, gamer_bf16* out_t, const int64_t* table,
                           int n_entries, int n_tiles, void* stream);

/* ------------------------------------------------------------------------------------------
 * q/k per-head RMSNorm + RoPE (+ per-behaviour q/k/v bias for the cross attention)
 * (model.py:88-101; Qwen3MoeRMSNorm; apply_rotary_pos_emb / rotate_half).
 *   qkv: [T, (nq+2*nkv)*64] output of the fused q|k|v projection; head_dim is 64.
 *   cos/sin: [S,64] tables (Qwen3RotaryEmbedding, positions 0..S-1)
 *   pos_ids: int32 [T] row of the tables for every token, NULL = t % S (Qwen3Multi); the session model passes
 *            gamer_session_spans' pos_ids
 *   bias_q [NB1, nq*64], bias_k/bias_v [NB1, nkv*64], act_idx int32 [T]  (all NULL for self)
 *   out: q_rot [T,nq*64], k_rot [T,nkv*64]; v (+bias) is updated in place inside qkv.
 * bwd: dq_rot, dk_rot -> dqkv[:, :q|k] (pre-norm grads); dv is already in dqkv[:, v];
 *   dwq/dwk [64] and dbias_* are ADDED to (zero-initialise them) in a fixed order: every wave writes its sums to a
 *   row of `partial` (scratch, partial_numel floats; 8192 * (1 + nb1) * 64 is always enough, a smaller buffer only
 *   lowers the number of waves; at least (nq + 2 nkv) * (1 + nb1) * 64) and a second kernel folds the rows.
 * bf16 (gamer_*_bf16): qkv / q_rot / k_rot / dq_rot / dk_rot / dqkv are activations.  The arithmetic follows the
 *   reference under autocast: self attention = Qwen3MoeRMSNorm on a bf16 tensor (normalised value rounded to bf16
 *   before the fp32 weight multiply); cross attention = fp32 bias added to the bf16 projection first, norm in fp32;
 *   RoPE in fp32; q_rot / k_rot / v rounded to bf16 (SDPA's autocast).  The biased pre-norm q/k are not written back
 *   (the backward re-adds bias_q / bias_k, so both are required there for the cross attention).
 * ---------------------------------------------------------------------------------------- */
int gamer_qknorm_rope_fwd(float* qkv, int T, int S, int nq, int nkv,
                          const float* wq, const float* wk, float eps,
                          const float* cos_t, const float* sin_t,
                          const float* bias_q, const float* bias_k, const float* bias_v,
                          const int32_t* act_idx,
                          float* q_rot, float* k_rot, const int32_t* pos_ids, void* stream);
int gamer_qknorm_rope_bwd(const float* qkv, const float* dq_rot, const float* dk_rot,
                          int T, int S, int nq, int nkv,
                          const float* wq, const float* wk, float eps,
                          const float* cos_t, const float* sin_t,
                          const float* bias_q, const float* bias_k,
                          const int32_t* act_idx, int nb1,
                          float* dqkv, float* dwq, float* dwk,
                          float* dbias_q, float* dbias_k, float* dbias_v, const int32_t* pos_ids,
                          float* partial, int64_t partial_numel, void* stream);
int gamer_qknorm_rope_fwd_bf16(gamer_bf16* qkv, int T, int S, int nq, int nkv,
                               const float* wq, const float* wk, float eps,
                               const float* cos_t, const float* sin_t,
                               const float* bias_q, const float* bias_k, const float* bias_v,
                               const int32_t* act_idx,
                               gamer_bf16* q_rot, gamer_bf16* k_rot, const int32_t* pos_ids, void* stream);
int gamer_qknorm_rope_bwd_bf16(const gamer_bf16* qkv, const gamer_bf16* dq_rot, const gamer_bf16* dk_rot,
                               int T, int S, int nq, int nkv,
                               const float* wq, const float* wk, float eps,
                               const float* cos_t, const float* sin_t,
                               const float* bias_q, const float* bias_k,
                               const int32_t* act_idx, int nb1,
                               gamer_bf16* dqkv, float* dwq, float* dwk,
                               float* dbias_q, float* dbias_k, float* dbias_v, const int32_t* pos_ids,
                               float* partial, int64_t partial_numel, void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-behaviour flash attention, fp32 MFMA, head_dim 64, GQA group nq/nkv in {1,2,4}.
 * Replaces sdpa_attention_forward + the additive masks (model.py:133-143): online softmax over
 * allowed(i,j) = j<=i && kl[j] < ql[i]; rows flagged empty use p = 1/S over all S keys.
 *   q [T,nq*64] (ldq), k [T,nkv*64] (ldk), v rows at v + t*ldv (ldv lets v live inside qkv)
 *   ql == NULL means query level 1 for every row (self attention).
 *   o [T,nq*64]; lse [B,nq,S] (log-sum-exp of the scaled scores; 0 for empty rows)
 *   dropout on the probabilities: p_drop, seed (attention_dropout, model.py:139)
 *   row_perm/tile_kind/tile_maxpos (all NULL = natural order): query-row order from
 *   gamer_attn_row_order; it only changes which rows share a wave, never the result.
 *   uniform_len (fwd only; 0 or S = training semantics): evaluation by re-running the whole sequence each
 *   generation step.  In the reference's cached decode an empty row is uniform over the keys that existed
 *   when the row was computed (model.py:603-617), so row i spans max(uniform_len, i+1) keys; pass the
 *   prompt length.
 *   q_span (NULL = Qwen3Multi's plain causal rule): int32 [B,S,4] = (hi, hole_lo, hole_hi, unused) per query
 *   row, hi <= i: key j is allowed iff j <= hi and not (hole_lo <= j < hole_hi) and kl[j] < ql[i]
 *   (gamer_session_spans); row_empty / tile_empty must then describe that predicate.
 *   delta_ready (bwd, with ds_work): delta already holds dO.O (the row-dot epilogue of gamer_gemm_f32 on the
 *   o_proj dgrad wrote it): the delta kernel is skipped.
 * bwd: delta[b,h,i] = dO.O, dq and dk/dv, no atomics.  ds_work == NULL: two recompute kernels (7 matmuls per
 *   tile pair).  ds_work = B*nq*ceil(S/32)^2*1024 floats of scratch: the dk/dv kernel spills its dS tiles there
 *   and dq is one matmul per tile that streams them back (5 matmuls; +2 x that many bytes of HBM traffic).
 * ---------------------------------------------------------------------------------------- */
/* Stable partition of every sequence's query rows: rows with an allowed key first (ascending position),
 * "empty" rows (uniform over all keys, no scores needed) behind them, so that a 32-row wave tile is all
 * normal (causal key range only) or all empty (no Q.K^T).  perm int32 [B,S] slot -> position;
 * tile_kind int32 [B,ceil(S/32)] bit0 has normal rows, bit1 has empty rows; tile_maxpos: largest
 * position of a normal row in the tile or -1.                                                   */
int gamer_attn_row_order(const int32_t* row_empty, int B, int S, int32_t* perm, int32_t* tile_kind,
                         int32_t* tile_maxpos, void* stream);
int gamer_attn_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                   const int32_t* kl, const int32_t* ql, const int32_t* row_empty,
                   const int32_t* tile_empty,
                   int B, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed,
                   float* o, float* lse, const int32_t* row_perm, const int32_t* tile_kind,
                   const int32_t* tile_maxpos, int uniform_len, const int32_t* q_span, void* stream);
int gamer_attn_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                   const float* o, const float* d_o, const float* lse,
                   const int32_t* kl, const int32_t* ql, const int32_t* row_empty,
                   const int32_t* tile_empty,
                   int B, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed,
                   float* delta, float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                   const int32_t* row_perm, const int32_t* tile_kind, const int32_t* tile_maxpos,
                   float* ds_work, const int32_t* q_span, int delta_ready, void* stream);

/* The same attention with fp32 inputs / outputs and every matrix product
 * formed on the bf16 matrix pipe from an exact three-way bf16 cut of both operands, six piece products per fp32 product,
 * fp32 accumulation - the attention counterpart of gamer_gemm_f32_split (csrc/attention_split.hip; same predicate,
 * empty-row rule, row order and dropout mask function as gamer_attn_fwd / _bwd, so the three kernels of either form
 * regenerate the same masks).  Replaces ref:SeqRec/models/generative/Qwen3Multi/model.py:133-143 in the
 * Engine(matmul="split6") step.  bwd: ds_work == NULL = recompute form (dQ kernel + dK/dV kernel, 7 products per tile
 * pair); ds_work = the scratch of gamer_attn_bwd (B*nq*ceil(S/32)^2*1024 floats) = the dK/dV kernel spills its fp32 dS
 * tiles and dQ = dS K is one product per tile pair (5 products).  delta_ready != 0: delta already holds dO.O (row-dot
 * epilogue of the o_proj dgrad GEMM), otherwise it is computed here.
 * (ABI 7) uniform_len / q_span as in gamer_attn_fwd / _bwd: the evaluation re-run's span of an "empty" row (forward, both
 * forms) and the per-query key spans of Qwen3SessionMulti (gamer_session_spans) - the latter in the three-product form only
 * (armed by gamer_attn_split_amax; backward: recompute form). */
int gamer_attn_fwd_split(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                         const int32_t* kl, const int32_t* ql, const int32_t* row_empty,
                         int B, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed,
                         float* o, float* lse,
                         const int32_t* row_perm, const int32_t* tile_kind, const int32_t* tile_maxpos,
                         int uniform_len, const int32_t* q_span, void* stream);
int gamer_attn_bwd_split(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                         const float* o, const float* d_o, const float* lse,
                         const int32_t* kl, const int32_t* ql, const int32_t* row_empty, const int32_t* tile_empty,
                         int B, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed,
                         float* delta, float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                         const int32_t* row_perm, const int32_t* tile_kind, const int32_t* tile_maxpos,
                         int delta_ready, float* ds_work, const int32_t* q_span, void* stream);

/* bf16 attention of the reference's --bf16 run (autocast casts q, k, v AND the additive mask of
 * sdpa_attention_forward to bf16).  Same predicate, layouts and dropout mask function as gamer_attn_fwd / _bwd, with
 * ONE semantic difference that the reference itself has under bf16 and that the fixtures tests/golden/{small,full}_bf16.npz pin:
 * finfo(float32).min is not representable in bf16, the mask becomes -inf, and SDPA returns 0 for a query row with
 * no allowed key - so an "empty" row gives o = 0 (lse = 0) and no gradient instead of the fp32 run's uniform average
 * over all S keys.  (No tile_empty / uniform_len arguments: nothing depends on them.)
 *   Optional row order (row_perm, tile_maxpos from gamer_attn_row_order, row_empty [B,S]; all three or none, not together
 *   with q_span): the query rows are visited through row_perm - rows with an allowed key first - so the tiles behind the
 *   last such row of a sequence cost nothing (in the cross attention 70 % of the rows have none); results are identical
 *   to the call without it, element for element.
 *   q [T,nq*64] (ldq), k [T,nkv*64] (ldk), v rows at v + t*ldv, o / d_o [T,nq*64]: bf16, leading dims multiples of 8;
 *   lse, delta [B,nq,S] fp32; dq / dk / dv bf16 (leading dims multiples of 4).
 *   Scores and softmax statistics fp32, un-normalised probabilities rounded to bf16 for the second product, fp32
 *   accumulation.  bwd = delta (unless delta_ready: gamer_gemm_bf16's row-dot epilogue wrote it; o may then be NULL),
 *   dK/dV kernel, dQ kernel (probabilities recomputed in both; no atomics, no workspace).                          */
int gamer_attn_fwd_bf16(const gamer_bf16* q, int ldq, const gamer_bf16* k, int ldk, const gamer_bf16* v, int ldv,
                        const int32_t* kl, const int32_t* ql, int B, int S, int nq, int nkv, float scale,
                        float p_drop, uint64_t seed, gamer_bf16* o, float* lse, const int32_t* q_span,
                        const int32_t* row_perm, const int32_t* tile_maxpos, const int32_t* row_empty, void* stream);
int gamer_attn_bwd_bf16(const gamer_bf16* q, int ldq, const gamer_bf16* k, int ldk, const gamer_bf16* v, int ldv,
                        const gamer_bf16* o, const gamer_bf16* d_o, const float* lse, const int32_t* kl,
                        const int32_t* ql, int B, int S, int nq, int nkv, float scale, float p_drop, uint64_t seed,
                        float* delta, gamer_bf16* dq, int lddq, gamer_bf16* dk, int lddk, gamer_bf16* dv, int lddv,
                        const int32_t* q_span, int delta_ready, const int32_t* row_perm, const int32_t* tile_maxpos,
                        const int32_t* row_empty, void* stream);

/* ------------------------------------------------------------------------------------------
 * Post-LN BERT-style encoder of the discriminative baselines (SURVEY section 8(f) row 4;
 * ref:SeqRec/modules/layers/transformer.py:12-183).  The projections and FFN matmuls use gamer_gemm_f32.
 *   activations (FeedForward.get_hidden_act, :100-109): 0 none, 1 relu, 2 gelu (erf form), 3 swish, 4 tanh,
 *   5 sigmoid, 6 elu.
 * bias_act_fwd: x[T,N] <- x + bias (kept for the backward); y = act(x + bias) (y NULL with act 0: bias only).
 * bias_act_bwd: dx = dy * act'(pre) (dx may alias dy); db_partial [n_partial, N] per-workgroup column sums
 *   of dx, to be summed with gamer_colsum_reduce (deterministic).
 * layernorm_fwd: v = x (+ res); y = (v - mean) / sqrt(var + eps) * w + b (biased variance, nn.LayerNorm);
 *   v_out (optional, may alias x) and mean / rstd [T] are what the backward takes.  H <= 1024.
 * layernorm_bwd: dx (= d v) and per-workgroup partial sums dw_partial / db_partial [n_partial, H].
 * attn_dense_fwd/bwd (MultiHeadAttention.forward, :47-78): o = dropout(softmax(q k^T * scale + mask)) v for
 *   head_dim <= 64 and S <= 128, one workgroup per (batch, head) with the probabilities in LDS; q/k/v/o are
 *   [B*S, ld] with head h at column h*head_dim; mask is ADDITIVE fp32, element (b,h,i,j) at
 *   mask[b*strides[0] + h*strides[1] + i*strides[2] + j*strides[3]] (0 strides broadcast; mask NULL: none);
 *   lse [B,H,S].  The backward recomputes the probabilities and regenerates the dropout mask from the seed.
 * ---------------------------------------------------------------------------------------- */
int gamer_bias_act_fwd(float* x, const float* bias, int T, int N, int act, float* y, void* stream);
int gamer_bias_act_bwd(const float* pre, const float* dy, int T, int N, int act, float* dx,
                       float* db_partial, int n_partial, void* stream);
int gamer_layernorm_fwd(const float* x, const float* res, const float* w, const float* b, int T, int H,
                        float eps, float* v_out, float* y, float* mean, float* rstd, void* stream);
int gamer_layernorm_bwd(const float* v, const float* w, const float* mean, const float* rstd,
                        const float* dy, int T, int H, float* dx, float* dw_partial, float* db_partial,
                        int n_partial, void* stream);
int gamer_attn_dense_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                         const float* mask, const int64_t* mask_strides, int B, int S, int H, int head_dim,
                         float scale, float p_drop, uint64_t seed, float* o, int ldo, float* lse, void* stream);
int gamer_attn_dense_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                         const float* mask, const int64_t* mask_strides, int B, int S, int H, int head_dim,
                         float scale, float p_drop, uint64_t seed, const float* o, const float* d_o, int ldo,
                         const float* lse, float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                         void* stream);

/* ------------------------------------------------------------------------------------------
 * Elementwise pieces of the decoder layer (model.py:147,217,235,241; FFN.py:25-27).
 * ---------------------------------------------------------------------------------------- */
/* x_out[t] = x_in[t] + drop(delta[src(t)])   src_rows optional (token -> sorted slot);
 * x_out may alias x_in (in place).                                                              */
int gamer_residual_dropout_fwd(const float* x_in, const float* delta, const int32_t* src_rows, int T,
                               int H, float p_drop, uint64_t seed, float* x_out, void* stream);
/* ddelta[src(t)] = drop_mask * dx[t]                                                            */
int gamer_residual_dropout_bwd(const float* dx, const int32_t* src_rows, int T, int H,
                               float p_drop, uint64_t seed, float* ddelta, void* stream);
/* hm = drop(silu(g) * u)   (n elements, g/u/hm contiguous)                                      */
int gamer_swiglu_fwd(const float* g, const float* u, int64_t n, float p_drop, uint64_t seed,
                     float* hm, void* stream);
/* in place: g <- dg, u <- du given dhm                                                          */
int gamer_swiglu_bwd(float* g, float* u, const float* dhm, int64_t n, float p_drop, uint64_t seed,
                     void* stream);
/* The same two on the output of a FUSED gate|up projection (ref:SeqRec/models/generative/Qwen3Moe/FFN.py:25-27: gate_proj and
 * up_proj read the same input - here one GEMM against the stacked [2 I, din] weight): gu is [T][ld], gate in columns 0 .. I - 1,
 * up in columns I .. 2 I - 1 (ld >= 2 I, a multiple of 4); hm / dhm are [T][I] contiguous.  Same dropout masks as the contiguous
 * form (element (t, c) <-> flat index t I + c).  gamer_amax_sink: out0 = max |hm|; backward out0 = max |d gate|, out1 = max |d up|
 * (pass the same word twice for the maximum of the whole [T][2 I] gradient).                                                    */
int gamer_swiglu_fwd_ld(const float* gu, int64_t ld, int T, int I, float p_drop, uint64_t seed, float* hm, void* stream);
int gamer_swiglu_bwd_ld(float* gu, int64_t ld, int T, int I, const float* dhm, float p_drop, uint64_t seed, void* stream);
/* (ABI 9) The same two with a row table: row t's gate | up values are gu[t] + tbl[row_group[t]] (tbl [groups][2 I] fp32, row_group
 * int32 [T]) - the sums are used, not stored.  The injecting layers of MyQwen3SparseMLP (ref:SeqRec/models/generative/Qwen3Moe/
 * FFN.py:60-68) concatenate a behaviour embedding to the expert input; its share of gate|up depends on (expert, behaviour) only
 * (gamer_inject_table_fwd), so the projection runs on the 256 hidden columns and the table row is added where the values are read. */
int gamer_swiglu_fwd_ld_tbl(const float* gu, int64_t ld, int T, int I, float p_drop, uint64_t seed, float* hm,
                            const float* tbl, const int32_t* row_group, void* stream);
int gamer_swiglu_bwd_ld_tbl(float* gu, int64_t ld, int T, int I, const float* dhm, float p_drop, uint64_t seed,
                            const float* tbl, const int32_t* row_group, void* stream);
/* (ABI 9) The behaviour-embedding share of the injecting layers' gate|up projection (FFN.py:60-68: h = cat(x, Eb[beh]); gate|up =
 * W_e h with W [E * 2I rows][ldw], the embedding columns at col0 .. col0 + EB - 1; plain fp32 FMA arithmetic, fixed summation orders):
 *   gamer_inject_table_fwd   tbl[(e * NB1 + b) * 2I + n] = sum_j Eb[b][j] W[e * 2I + n][col0 + j]          (Eb [NB1][EB])
 *   gamer_segment_colsum     out[s][c] = sum over the rows of segment s of x[row][c]; rows sorted by segment, offsets int32 [nseg + 1];
 *                            ws: gamer_segment_colsum_ws_floats(rows, cols, nseg) floats of scratch
 *   gamer_inject_table_bwd   with seg = the segment sums of d(gate|up) over the (expert, behaviour) row groups ([E * NB1][2I]):
 *                            dW[e * 2I + n][col0 + j] += sum_b seg[e * NB1 + b][n] Eb[b][j],
 *                            dEb[b][j] += sum_e sum_n seg[e * NB1 + b][n] W[e * 2I + n][col0 + j];  scratch: NB1 * E * EB floats
 * Together they replace the [T, EB] columns of the expert input, of its gradient and gamer_rowtable_fwd / _bwd on them.           */
int gamer_inject_table_fwd(const float* Eb, const float* W, int64_t ldw, int col0, int E, int twoI, int NB1, int EB, float* tbl,
                           void* stream);
int64_t gamer_segment_colsum_ws_floats(int rows, int cols, int nseg);
int gamer_segment_colsum(const float* x, int64_t ld, int rows, int cols, const int32_t* offsets, int nseg, float* ws,
                         int64_t ws_floats, float* out, void* stream);
int gamer_inject_table_bwd(const float* seg, const float* Eb, const float* W, int64_t ldw, int col0, int E, int twoI, int NB1, int EB,
                           float* dW, float* dEb, float* scratch, void* stream);
int gamer_swiglu_fwd_ld_bf16(const gamer_bf16* gu, int64_t ld, int T, int I, float p_drop, uint64_t seed, gamer_bf16* hm,
                             void* stream);
int gamer_swiglu_bwd_ld_bf16(gamer_bf16* gu, int64_t ld, int T, int I, const gamer_bf16* dhm, float p_drop, uint64_t seed,
                             void* stream);
/* out = a * silu(gate)  (cross-attention output gate, model.py:147); with resid != NULL the block's residual add
 * is fused in: out = resid + drop(a * silu(gate)) (model.py:235), same mask as gamer_residual_dropout_fwd(seed)  */
int gamer_silu_gate_fwd(const float* a, const float* gate, int64_t n, float* out, const float* resid,
                        float p_drop, uint64_t seed, void* stream);
/* d = drop_mask(seed) * dout (p_drop = 0: d = dout); da = d*silu(gate) ; dgate = d*a*silu'(gate)              */
int gamer_silu_gate_bwd(const float* a, const float* gate, const float* dout, int64_t n,
                        float* da, float* dgate, float p_drop, uint64_t seed, void* stream);
/* bf16: g / u / hm / dhm, a / gate / da / dgate are activations; the gate's `out` and `dout` are the fp32 residual
 * stream and its gradient (resid is required).  Computed in fp32, rounded once on store.                          */
int gamer_swiglu_fwd_bf16(const gamer_bf16* g, const gamer_bf16* u, int64_t n, float p_drop, uint64_t seed,
                          gamer_bf16* hm, void* stream);
int gamer_swiglu_bwd_bf16(gamer_bf16* g, gamer_bf16* u, const gamer_bf16* dhm, int64_t n, float p_drop,
                          uint64_t seed, void* stream);
int gamer_silu_gate_fwd_bf16(const gamer_bf16* a, const gamer_bf16* gate, int64_t n, float* out,
                             const float* resid, float p_drop, uint64_t seed, void* stream);
int gamer_silu_gate_bwd_bf16(const gamer_bf16* a, const gamer_bf16* gate, const float* dout, int64_t n,
                             gamer_bf16* da, gamer_bf16* dgate, float p_drop, uint64_t seed, void* stream);

/* ------------------------------------------------------------------------------------------
 * Tied LM head loss: temperature + shifted cross entropy (model.py:904-922;
 * transformers/loss/loss_utils.py ForCausalLMLoss / fixed_cross_entropy).
 *   logits [T, ldl] (first V columns valid) are divided by `temperature` IN PLACE (as upstream),
 *   target of row (b,s) is labels[b][s+1] (ignore_index for s = S-1),
 *   row_loss [T] per-row CE (0 where ignored), lse_out [T]; loss_sum[0] = sum CE and
 *   count[0] = #targets != ignore, both reduced in a fixed order (deterministic).
 * bwd: logits <- (softmax - onehot) * dloss / (denom * temperature), denom = count_dev[0] when
 *   count_dev != NULL (mean reduction) else denom_host (num_items_in_batch from the trainer); dloss_dev (optional,
 *   device scalar) multiplies dloss: the incoming gradient of the loss under autograd, read without a host sync.
 * ---------------------------------------------------------------------------------------- */
/* bad_label[0] += number of labels that are neither ignore_index nor inside [0, V): the reference's
 * nn.CrossEntropyLoss raises on those (transformers/loss/loss_utils.py fixed_cross_entropy); gamer_ce_* skip them, so
 * the host checks this counter and raises (no synchronisation here).                                          */
int gamer_check_labels(const int64_t* labels, int64_t n, int V, int ignore_index, int32_t* bad_label,
                       void* stream);
int gamer_ce_fwd(float* logits, int ldl, const int64_t* labels, int B, int S, int V,
                 float temperature, int ignore_index, float* lse_out, float* row_loss,
                 float* loss_sum, float* count, void* stream);
int gamer_ce_bwd(float* logits, int ldl, const int64_t* labels, int B, int S, int V,
                 float temperature, int ignore_index, const float* lse, const float* count_dev,
                 float denom_host, float dloss, const float* dloss_dev, void* stream);
/* bf16 logits (what the reference's lm_head returns under autocast; the in-place /temperature rounds to bf16 again,
 * as upstream); lse / loss / count are fp32 (loss_utils.py upcasts the logits).  bwd writes d(logits) as bf16.     */
int gamer_ce_fwd_bf16(gamer_bf16* logits, int ldl, const int64_t* labels, int B, int S, int V,
                      float temperature, int ignore_index, float* lse_out, float* row_loss,
                      float* loss_sum, float* count, void* stream);
int gamer_ce_bwd_bf16(gamer_bf16* logits, int ldl, const int64_t* labels, int B, int S, int V,
                      float temperature, int ignore_index, const float* lse, const float* count_dev,
                      float denom_host, float dloss, const float* dloss_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * HF Trainer update: clip_grad_norm_(max_norm) + AdamW (transformers/trainer.py;
 * ref:SeqRec/tasks/train_SMB_decoder.py:396-428).  Flat fp32 buffers of n elements; the first
 * n_decay elements get weight decay (norm weights live behind them).
 *   gamer_sumsq: partial[i] = sum of squares of chunk i (n_partial chunks, deterministic)
 *   gamer_adamw: norm = sqrt(sum partial) ; coef = min(1, max_norm/(norm+1e-6)) ;
 *                grad_scale multiplies the gradient first (1/world, 1/accum); writes norm_out[0].
 * ---------------------------------------------------------------------------------------- */
int gamer_sumsq(const float* g, int64_t n, float* partial, int n_partial, void* stream);
int gamer_adamw(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_decay,
                float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                float max_norm, float grad_scale, const float* partial, int n_partial,
                float* norm_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Evaluation path: trie-constrained beam-search scoring.
 * Replaces, per generation step, log_softmax(next_token_logits) + PrefixConstrainedLogitsProcessor with
 * the reference's item Trie (ref:SeqRec/generation/trie.py:5-104; call sites
 * ref:SeqRec/tasks/test_SMB_decoder.py:120,470-500; HF generation/logits_process.py
 * PrefixConstrainedLogitsProcessor).  The trie is a CSR array: node 0 is the root, the children of node x are
 * edges child_start[x] .. child_start[x+1]-1 with token child_tok[e] and target node child_node[e].
 *   gamer_trie_logprobs: scores[n][0..V) = -inf, then for every child token of node[n]:
 *       scores[n][tok] = logits[row_index[n]][tok] - logsumexp(logits[row_index[n]][0..V)) + beam_score[n]
 *       (node[n] < 0: nothing allowed, what Trie.get returns for an unknown prefix)
 *   gamer_trie_advance:  next[n] = child of node[n] along token[n], or -1
 * ---------------------------------------------------------------------------------------- */
int gamer_trie_logprobs(const float* logits, int64_t ld, const int32_t* row_index, const float* beam_score,
                        const int32_t* node, const int32_t* child_start, const int32_t* child_tok, int N, int V,
                        float* scores, void* stream);
int gamer_trie_advance(const int32_t* node, const int64_t* token, const int32_t* child_start,
                       const int32_t* child_tok, const int32_t* child_node, int N, int32_t* next, void* stream);

/* Attention of one new token per beam against the K/V cache of a generation run
 * (Qwen3MultiAttention.forward with past_key_value, ref:.../Qwen3Multi/model.py:118-121; the decode branches of
 * the mask builders, model.py:603-617 and 718-741).  N = B*nb beams; the prompt part of the cache is stored once
 * per sample (the reference keeps nb identical copies), the generated part per beam:
 *   q  [N, nq*64] (ldq)   queries of the new tokens (after q-norm + RoPE at their position)
 *   kp/vp  prompt keys / values, row b*L0 + j, leading dims ldkp / ldvp
 *   key_ok int32 [B, L0]  1 = prompt key j may be attended by the new rows of sample b
 *   kg/vg  generated keys / values, row n*tmax + g (g < t, the new token itself is g = t-1), leading dim ldg
 *   gen_ok 1 = generated keys are attended (self attention), 0 = masked (cross attention)
 *   uniform int32 [B] or NULL: 1 = the new rows of sample b have no allowed key -> mean of V over all L0+t keys
 *   o  [N, nq*64]
 * t = 0 (no generated position: kg / vg are not read): the rows of q are prompt rows attending the prompt keys key_ok allows - the
 * last prompt row of the last layer's cross attention in a generation's prompt pass (Engine.forward(last_row_logits=True)).       */
int gamer_attn_decode(const float* q, int ldq, const float* kp, int ldkp, const float* vp, int ldvp,
                      const int32_t* key_ok, const float* kg, const float* vg, int ldg, int tmax, int t, int gen_ok,
                      const int32_t* uniform, int B, int nb, int L0, int nq, int nkv, float scale, float* o,
                      void* stream);
/* (ABI 9, round 6) The new token's keys / values join the generated part of the cache in one launch: kg / vg [N][tmax][C] (row stride
 * ldg) <- k [N][C] (ldk), v [N][C] (ldv: the v columns of the q|k|v buffer) at position g; C, the leading dims % 4 == 0, 16-byte aligned.
 * Replaces two strided tensor copies per attention and token of the cached decode step (ref:.../Qwen3Multi/model.py:118-121). */
int gamer_kv_append(const float* k, int ldk, const float* v, int ldv, float* kg, float* vg, int ldg, int tmax, int g, int N, int C,
                    void* stream);
/* (ABI 9) The same with its two products in the three-piece fp16 form of the train step (gamer_attn_fwd_split with
 * gamer_attn_split_amax): amax_k / amax_v = gamer_absmax_f32 slots holding max |kp|, max |vp| (the prompt caches do not change during a
 * generation: measured once per session); the queries are scaled per workgroup inside the kernel, the generated positions stay in fp32.
 * gamer_amax_sink: out0 = max |o| (the operand of the o_proj GEMM that follows). */
int gamer_attn_decode_split(const float* q, int ldq, const float* kp, int ldkp, const float* vp, int ldvp,
                            const int32_t* key_ok, const float* kg, const float* vg, int ldg, int tmax, int t, int gen_ok,
                            const int32_t* uniform, int B, int nb, int L0, int nq, int nkv, float scale, float* o,
                            const uint32_t* amax_k, const uint32_t* amax_v, void* stream);

/* misc */
int gamer_fill_f32(float* p, int64_t n, float value, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GAMER_HIP_H */
