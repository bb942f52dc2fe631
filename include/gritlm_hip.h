/*
 * gritlm_hip.h -- C ABI of libgritlm_hip.so, the MI355X (gfx950 / CDNA4) native engine for the
 * GritLM embedding-encode / contrastive-training hot path.
 *
 * The reference (ContextualAI/gritlm) is 100 % Python and has no FFI of its own:
 * every "kernel" there is a PyTorch op.  This header therefore DEFINES the drop-in boundary; each
 * entry point cites the reference op(s) it replaces (paths relative to the upstream checkout).
 *
 * Conventions (all entry points):
 *   - plain pointers to DEVICE memory + explicit sizes/strides; no torch types;
 *   - `stream` is a hipStream_t passed as void*; NOTHING synchronises the device or allocates;
 *   - bf16 tensors are passed as `const void*` to 16-bit storage, row-major, innermost contiguous;
 *   - return 0 on success, negative GRIT_E_* on failure (then grit_last_error_string() explains);
 *   - the *_workspace_* size queries have no error channel: they return 0 for sizes their compute entry point rejects
 *     (which then reports the error); sizes are range-checked BEFORE any arithmetic on them (tools/fuzz_abi.py);
 *   - re-entrant and thread-safe for distinct streams (forward from the Python main thread,
 *     backward from autograd worker threads).
 */
#ifndef GRITLM_HIP_H
#define GRITLM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRIT_ABI_VERSION 5

enum {
  GRIT_OK = 0,
  GRIT_E_BADARG = -1,      /* null pointer / non-positive size / misaligned pointer            */
  GRIT_E_UNSUPPORTED = -2, /* shape outside what the gfx950 kernels are built for              */
  GRIT_E_LAUNCH = -3,      /* hipLaunchKernel / runtime error                                   */
  GRIT_E_RCCL = -4         /* RCCL unavailable (librccl.so not loadable) or an ncclResult_t error (grit_comm_*) */
};

/* GEMM epilogues (grit_gemm_bf16_nt) */
enum {
  GRIT_EPI_STORE = 0,    /* C = bf16(acc)                                                       */
  GRIT_EPI_RESIDUAL = 1, /* C = bf16(acc + residual)  (residual may alias C)                    */
  GRIT_EPI_SWIGLU = 2,   /* weight rows interleaved gate/up in blocks of grit_swiglu_block() rows;
                            C[:, N/2] = bf16(silu(bf16(gate)) * bf16(up))                        */
  GRIT_EPI_ROPE = 3,     /* (grit_gemm_bf16_nt_rope) STORE + rotary embedding on the leading columns */
  GRIT_EPI_SWIGLU_STACKED = 4, /* SWIGLU on STACKED weights W = [gate (N/2 rows); up (N/2 rows)] -- the layout of a module whose
                            gate_proj / up_proj parameters are row-slices of one buffer (training engine); the interleave happens in
                            the per-lane LDS-DMA source address, the arithmetic is GRIT_EPI_SWIGLU's                         */
  GRIT_EPI_SWIGLU_STACKED_SAVE = 5, /* SWIGLU_STACKED that ALSO writes the bf16 pre-activations [gate | up] ([M, N], leading dimension
                            ldr) through the `residual` pointer (an OUTPUT here): what the backward pass keeps (training forward)   */
  GRIT_EPI_SWIGLU_BWD = 6, /* backward of SwiGLU fused behind the dgrad GEMM d_act = d_h @ W_down^T (N = intermediate size):
                            `residual` = saved [gate | up] ([M, 2N], ldr); C = [d_gate | d_up] ([M, 2N], ldc >= 2N):
                            d_gate = d_act * up * s (1 + gate (1 - s)), d_up = d_act * gate * s, s = sigmoid(gate), d_act = bf16(acc)  */
  GRIT_EPI_RESIDUAL_F32 = 7 /* fp32 residual stream (the encoder's high-precision mode): C and residual are FP32 [M, N] (ldc / ldr in
                            floats, residual may alias C): C = residual + acc, no rounding of the Linear output or of the sum -- what
                            hidden_states = residual + hidden_states (:769,:775) computes when the reference runs in fp32           */
};

/* pooling modes: gritlm/gritlm.py:188-214 */
enum { GRIT_POOL_MEAN = 0, GRIT_POOL_WEIGHTEDMEAN = 1, GRIT_POOL_CLS = 2, GRIT_POOL_LASTTOKEN = 3 };

int grit_version(void);
/* thread-local message of the last failing call on this thread ("" if none) */
const char* grit_last_error_string(void);

/* ---- encoder forward: scripts/modeling_mistral_gritlm.py ------------------------------------ */

/* embed_tokens(input_ids), :918,994.  table [V,H] bf16, ids [T] int64 -> out [T,H] bf16.
 * ids outside [0,V) are clamped (the reference raises IndexError). */
int grit_embed_gather(const void* table, const int64_t* ids, void* out, int64_t T, int H, int64_t V,
                      void* stream);

/* MistralRMSNorm.forward, :84-89: y = w * bf16(x * rsqrt(mean(x^2) + eps)), fp32 math. x,y [T,H] bf16. */
int grit_rmsnorm_fwd(const void* x, const void* w, void* y, int64_t T, int H, float eps, void* stream);

/* The fp32 residual stream of the encoder's high-precision mode (GRIT_EPI_RESIDUAL_F32): embed_tokens widened to fp32 (exact), and
 * MistralRMSNorm (:84-89) of an fp32 row with ONE rounding, y = bf16(w * (x * rsqrt(mean(x^2) + eps))) -- the bf16 operand of the next
 * projection.  out / x: [T,H] fp32, 16-byte aligned; w, y bf16. */
int grit_embed_gather_f32(const void* table, const int64_t* ids, float* out, int64_t T, int H, int64_t V, void* stream);
int grit_rmsnorm_fwd_f32in(const float* x, const void* w, void* y, int64_t T, int H, float eps, void* stream);

/* apply_rotary_pos_emb, :138-163, in place on the q and k heads of a fused [T, row_stride] bf16 buffer
 * (columns [0,(nq+nkv)*d) hold q heads then k heads); positions are t % S (:984-989).
 * cos/sin: fp32 tables [S, d/2] (MistralRotaryEmbedding :93-126; cos[j] == cos[j+d/2]).
 * inverse != 0 applies the transposed rotation (backward of RoPE). */
int grit_rope_qk_inplace(void* qkv, const float* cos_tab, const float* sin_tab, int64_t T, int S, int nq,
                         int nkv, int d, int64_t row_stride, int inverse, void* stream);

/* same with explicit positions (packed / un-padded batches: positions[t] = index of token t inside its sequence);
 * tables have table_rows rows. */
int grit_rope_qk_inplace_pos(void* qkv, const float* cos_tab, const float* sin_tab, const int32_t* positions, int64_t T,
                             int table_rows, int nq, int nkv, int d, int64_t row_stride, int inverse, void* stream);

/* nn.Linear without bias (q/k/v/o_proj :225-228,655-657,703; MLP :177-178):
 *   C[M,N] = A[M,K] * W[N,K]^T, bf16 in, fp32 MFMA accumulate, bf16 out, with a fused epilogue.
 * Requirements: K % 64 == 0, N % 16 == 0, lda/ldw/ldc/ldr % 8 == 0, pointers 16-byte aligned.
 * SWIGLU: N counts the interleaved gate+up rows (2*I), N % 64 == 0; C has N/2 columns. */
int grit_gemm_bf16_nt(const void* A, const void* W, void* C, int64_t M, int N, int K, int64_t lda,
                      int64_t ldw, int64_t ldc, int epilogue, const void* residual, int64_t ldr,
                      void* stream);

/* The fused QKV projection with apply_rotary_pos_emb (:138-163) in the epilogue: C = A W^T (STORE), and the leading rope_cols
 * columns (q heads then k heads, head_dim 128; rope_cols = (nq+nkv)*128) are rotated in the lane that produced them -- bit-identical to
 * grit_gemm_bf16_nt followed by grit_rope_qk_inplace[_pos], without the extra read + write of the activation.
 * Row m sits at position positions[m] (int32, device) or, when positions is NULL, m % S.  cos/sin: fp32 [table_rows, 64].
 * N and rope_cols must be multiples of 128. */
int grit_gemm_bf16_nt_rope(const void* A, const void* W, void* C, int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldc,
                           const float* cos_tab, const float* sin_tab, const int32_t* positions, int S, int table_rows,
                           int rope_cols, void* stream);

/* Row-interleave granularity the SWIGLU epilogue expects: packed row r = 2*blk*(r/blk) + r%blk holds gate row r,
 * the next blk rows the matching up rows (blk = 32 for the 32x32x16 kernel generation). */
int grit_swiglu_block(void);

/* attention_mask [B,S] int64 (HF layout, 0 = padding) -> key bitmask [B, ceil(S/64)] uint64
 * (replaces _prepare_4d_attention_mask(_for_sdpa), :1017-1020,1033-1036: no [B,1,S,S] tensor). */
int grit_mask_pack(const int64_t* mask, uint64_t* bits, int B, int S, void* stream);

/* Bidirectional (is_causal=False) attention core with key-padding mask and GQA, replacing
 * repeat_kv + SDPA (:182-191, :690-698).  qkv: fused [B*S, qkv_stride] bf16 with q heads at column 0,
 * k heads at nq*d, v heads at (nq+nkv)*d (RoPE already applied).  out [B*S, out_stride] bf16
 * (= attn_output.transpose(1,2).reshape(B,S,nq*d)).  lse (nullable) [B, nq, S] fp32 = log-sum-exp of
 * the scaled scores (saved for backward).  head_dim d must be 128. */
int grit_attn_bidir_fwd(const void* qkv, const uint64_t* key_bits, void* out, float* lse, int B, int S,
                        int nq, int nkv, int d, int64_t qkv_stride, int64_t out_stride, float scale,
                        void* stream);

/* Packed (un-padded) variant: sequence b occupies rows [cu_seqlens[b], cu_seqlens[b+1]) of qkv/out ([T, stride]), every row
 * is a real token (the reference pads to the batch maximum and masks, gritlm/gritlm.py:120-127; its flash-attention path
 * un-pads the same way, modeling_mistral_gritlm.py:575-615).  cu_seqlens: int32 [B+1] on the device; max_len sizes the grid.
 * lse (nullable) is [T, nq]. */
int grit_attn_bidir_varlen_fwd(const void* qkv, const int32_t* cu_seqlens, void* out, float* lse, int B, int max_len, int nq,
                               int nkv, int d, int64_t qkv_stride, int64_t out_stride, float scale, void* stream);

/* ---- fp16-operand precision policy of the embedding path ("f16_operands", round 5) ----------------------------------------------
 * The reference states its outputs in fp32 terms (README cosine tables; north-star tolerance 1 - cos < 1e-4 against the fp32 run of
 * scripts/modeling_mistral_gritlm.py:936-1096).  With bf16 MFMA operands the error of a 32-layer forward is ~4e-4 even with an fp32
 * residual stream (profiles/r04_depth_parity.json): the 8-bit mantissa of the operands is the floor.  v_mfma_f32_*_f16 runs at the bf16
 * rate with 3 more mantissa bits, and bf16 weights convert to fp16 exactly for 6.1e-5 <= |w| < 65520 (smaller ones lose at most 3e-8
 * absolute).  In this policy the residual stream is fp32 (grit_embed_gather_f32, GRIT_EPI_RESIDUAL_F32) and every MFMA operand --
 * RMSNorm output, q | k | v, P, the attention output, the SwiGLU activation, the weights -- is fp16, each rounded ONCE from fp32.
 * Every kernel that rounds to fp16 flags a result beyond the fp16 range (inf / nan stored) in a per-device word that
 * grit_f16_overflow_flag reads: the host raises instead of returning a silently saturated embedding. */

/* MistralRMSNorm (:84-89) of an fp32 row, y = fp16(w * (x * rsqrt(mean(x^2) + eps))); w bf16, y fp16 [T,H]. */
int grit_rmsnorm_fwd_f32in_f16(const float* x, const void* w, void* y, int64_t T, int H, float eps, void* stream);

/* grit_gemm_bf16_nt on fp16 operands (A, W fp16; fp32 accumulate).  Epilogues: GRIT_EPI_STORE (C = fp16(acc)), GRIT_EPI_SWIGLU
 * (C = fp16(silu(gate) * up) evaluated in fp32, interleaved weight rows; GRIT_EPI_SWIGLU_STACKED: the same on [gate; up] stacked rows --
 * the training engine's weight layout, used by GradCache pass 1 under an fp16 policy), GRIT_EPI_RESIDUAL_F32 (C, residual fp32:
 * C = residual + acc), GRIT_EPI_RESIDUAL (C, residual fp16: the fp16 residual stream, below). */
int grit_gemm_f16_nt(const void* A, const void* W, void* C, int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldc, int epilogue,
                     const void* residual, int64_t ldr, void* stream);

/* grit_gemm_bf16_nt_rope on fp16 operands: the rotation (:138-163) runs on the fp32 accumulators with the fp32 tables as passed (build
 * them unrounded), q | k | v are rounded once, to fp16. */
int grit_gemm_f16_nt_rope(const void* A, const void* W, void* C, int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldc,
                          const float* cos_tab, const float* sin_tab, const int32_t* positions, int S, int table_rows, int rope_cols,
                          void* stream);

/* grit_attn_bidir_fwd / grit_attn_bidir_varlen_fwd with qkv and out in fp16 (P rounded to fp16, fp32 statistics and accumulators). */
int grit_attn_bidir_f16_fwd(const void* qkv, const uint64_t* key_bits, void* out, float* lse, int B, int S, int nq, int nkv, int d,
                            int64_t qkv_stride, int64_t out_stride, float scale, void* stream);
int grit_attn_bidir_varlen_f16_fwd(const void* qkv, const int32_t* cu_seqlens, void* out, float* lse, int B, int max_len, int nq, int nkv,
                                   int d, int64_t qkv_stride, int64_t out_stride, float scale, void* stream);
/* The same attention on fp16 operands with is_causal=True (ABI 5), optionally under a sliding window of `window` keys (0: none; meaning as
 * grit_attn_causal_window_fwd): the causal prompt pass of a unified / generative model and the 'cc' embedding attention under the fp16
 * policies (mask :1005-1031). */
int grit_attn_causal_f16_fwd(const void* qkv, const uint64_t* key_bits, void* out, float* lse, int B, int S, int nq, int nkv, int d,
                             int64_t qkv_stride, int64_t out_stride, float scale, int window, void* stream);
int grit_attn_causal_varlen_f16_fwd(const void* qkv, const int32_t* cu_seqlens, void* out, float* lse, int B, int max_len, int nq, int nkv,
                                    int d, int64_t qkv_stride, int64_t out_stride, float scale, int window, void* stream);

/* "f16_stream" (the same policy with the residual stream itself in fp16 instead of fp32: 16-bit epilogue and norm traffic, 0.975 of
 * the default's docs/s instead of 0.955; 1 - cos 6e-6 emulated): grit_embed_gather on an fp16 copy of the table (a 16-bit row copy),
 * grit_gemm_f16_nt with GRIT_EPI_RESIDUAL (C = fp16(fp16(acc) + residual), residual fp16, may alias C), and MistralRMSNorm (:84-89) of an
 * fp16 row with one rounding, to fp16 (out_is_f16 != 0) or to bf16 (last_hidden_state). */
int grit_rmsnorm_fwd_f16in(const void* x, const void* w, void* y, int out_is_f16, int64_t T, int H, float eps, void* stream);

/* *host_flag = 1 if a kernel of this policy stored an inf / nan on the current device since the last clear, else 0.  Enqueues a 4-byte
 * D2H copy (and, with clear != 0, the reset) on `stream` and WAITS for that stream: the one entry point of this library that
 * synchronises, meant to be called once per encode() next to the copy of the embeddings. */
int grit_f16_overflow_flag(int* host_flag, int clear, void* stream);

/* ---- sparse MoE MLP (Mixtral): scripts/modeling_mixtral_gritlm.py:797-882 ---------------------- */

/* Grouped GEMM: the M_total rows of A (and C) are the concatenation of num_groups row ranges, group g has
 * group_counts[g] rows (int32, DEVICE memory: the host never waits for the routing) and multiplies against
 * W + g * w_group_stride ([N,K] each):  C[r,:] = A[a_rows ? a_rows[r] : r, :] * W_g^T.   a_rows (nullable, int32
 * [M_total], device) gathers the A rows, i.e. the token permutation of :861-873 without materialising it.
 * Epilogues: GRIT_EPI_STORE, GRIT_EPI_SWIGLU (w1/w3 rows interleaved per expert, grit_swiglu_block()). */
int grit_gemm_bf16_nt_grouped(const void* A, const int32_t* a_rows, const void* W, void* C, const int32_t* group_counts,
                              int num_groups, int64_t M_total, int N, int K, int64_t lda, int64_t ldw,
                              int64_t w_group_stride, int64_t ldc, int epilogue, void* stream);

/* grit_gemm_bf16_nt_grouped on fp16 operands (round 6: Mixtral's expert MLP under the "f16_operands" policy; replaces the expert loop of
 * scripts/modeling_mixtral_gritlm.py:861-880 when the model runs at the north-star tolerance): A (gathered through a_rows), W [E,N,K] and C
 * fp16, ONE rounding of the fp32 accumulator.  Epilogues: GRIT_EPI_STORE, GRIT_EPI_SWIGLU, GRIT_EPI_SWIGLU_STACKED.  Overflow is flagged
 * like grit_gemm_f16_nt. */
int grit_gemm_f16_nt_grouped(const void* A, const int32_t* a_rows, const void* W, void* C, const int32_t* group_counts,
                             int num_groups, int64_t M_total, int N, int K, int64_t lda, int64_t ldw,
                             int64_t w_group_stride, int64_t ldc, int epilogue, void* stream);

/* Router of the "f16_operands" policy (:843-849 as the reference computes it in fp32: nothing rounded).  h [T,H] fp32 is the residual
 * stream ENTERING the block's post-attention RMSNorm (:939); the norm (ln_w [H] bf16, eps) is folded in: logits = rsqrt(mean h^2 + eps) *
 * ((h * ln_w) gate_w^T), softmax, top-2 (ties: lower index), renormalise.  experts [T,2] int32, weights [T,2] fp32 (not rounded). */
int grit_moe_router_top2_f32(const float* h, const void* ln_w, float eps, const void* gate_w, int32_t* experts, float* weights, int64_t T,
                             int H, int E, void* stream);

/* Combine of the "f16_operands" policy (:876-880 + decoder :945 in fp32): out[t] = residual[t] + weights[t,0]*y[rows[t,0]] +
 * weights[t,1]*y[rows[t,1]]; y [2T,H] fp16 (the grouped w2 GEMM's single rounding), residual (nullable) / out [T,H] fp32 (may alias). */
int grit_moe_combine_f32(const void* y, const int32_t* rows, const float* weights, const float* residual, float* out, int64_t T, int H,
                         void* stream);

/* Router (:843-849): logits = bf16(x gate_w^T), softmax in fp32, top-2 (ties: lower index), renormalise, round to bf16.
 * x [T,H] bf16, gate_w [E,H] bf16 (E in {4,8,16}) -> experts [T,2] int32, weights [T,2] fp32 (bf16-representable). */
int grit_moe_router_top2(const void* x, const void* gate_w, int32_t* experts, float* weights, int64_t T, int H, int E,
                         void* stream);

/* Stable counting sort of the 2T (token,k) pairs by expert (replaces torch.where/.tolist() per expert, :859-870):
 * counts [E] int32, row_token [2T] int32 (sorted row -> token), rows [T,2] int32 ((token,k) -> sorted row);
 * workspace: grit_moe_index_workspace_ints(T, E) int32 of device memory. */
int64_t grit_moe_index_workspace_ints(int64_t T, int E);
int grit_moe_index(const int32_t* experts, int64_t T, int E, int32_t* counts, int32_t* row_token, int32_t* rows,
                   int32_t* workspace, void* stream);

/* out[t] = residual[t] + (weights[t,0]*y[rows[t,0]] + weights[t,1]*y[rows[t,1]]) with the bf16 roundings of :876, :880 and
 * the decoder's residual add (:945).  y [2T,H] bf16, residual (nullable) / out [T,H] bf16 (out may alias residual). */
int grit_moe_combine(const void* y, const int32_t* rows, const float* weights, const void* residual, void* out, int64_t T,
                     int H, void* stream);

/* Backward of grit_moe_combine (autograd of `final_hidden_states += expert_out * routing_weight`, scripts/modeling_mixtral_gritlm.py
 * :861-880): for every routed row r (2T of them): dy[r] = bf16(w(r) * dout[token(r)]); dw[token(r), slot(r)] = <y[r], dout[token(r)]>
 * in fp32.  dout [T,H] bf16, y [2T,H] bf16 (the expert outputs of the forward, sorted row order), dy [2T,H] bf16, dw [T,2] fp32. */
int grit_moe_combine_bwd(const void* dout, const void* y, const int32_t* row_token, const int32_t* rows, const float* weights,
                         void* dy, float* dw, int64_t T, int H, void* stream);

/* Router backward (autograd of `routing_weights = softmax(gate(x), float); topk(2); /= sum` , scripts/modeling_mixtral_gritlm.py:843-849,
 * through the gate Linear): x [T,H] bf16 (the block's input), gate_w [E,H] bf16, experts [T,2] int32 (the forward's selection), dw [T,2] fp32
 * (d loss / d routing weights: grit_moe_combine_bwd), aux_dlogits [T,E] fp32 or NULL (added to the logits' gradient: the auxiliary
 * load-balancing loss).  The logits are recomputed in fp32.  Outputs: dlogits [T,E] fp32 and dx_out [T,H] bf16 = bf16(f32(dx_in) +
 * dlogits @ gate_w); dx_in [T,H] bf16 or NULL = the experts' share of the input gradient; dx_out may alias dx_in.  E in {4, 8, 16}. */
int grit_moe_router_bwd(const void* x, const void* gate_w, const int32_t* experts, const float* dw, const float* aux_dlogits,
                        const void* dx_in, void* dx_out, float* dlogits, int64_t T, int H, int E, void* stream);

/* gate.weight gradient: grad [E,H] bf16 (in/out) = bf16(f32(grad) + f32(bf16(dlogits^T [E,T] @ x [T,H]))) -- `grad.add_(dW.to(bf16))`;
 * fp32 sums in a fixed two-level order (bit-reproducible).  workspace: grit_moe_router_wgrad_workspace_floats(T, H, E) floats. */
int64_t grit_moe_router_wgrad_workspace_floats(int64_t T, int H, int E);
int grit_moe_router_wgrad(const void* x, const float* dlogits, void* grad, float* workspace, int64_t T, int H, int E, void* stream);

/* grit_gemm_bf16_nt_grouped with the training epilogues of the expert MLP (forward with saved pre-activations, backward):
 * STORE, SWIGLU, SWIGLU_STACKED (w = [gate rows; up rows] per expert, the layout of `experts.gate_up_proj`), SWIGLU_STACKED_SAVE
 * (additionally writes bf16 [gate | up] of every sorted row through `residual`, ldr >= N), SWIGLU_BWD (C = [d_gate | d_up] from the
 * saved [gate | up] in `residual`, ldr >= 2N, ldc >= 2N).  Per group the semantics of grit_gemm_bf16_nt. */
int grit_gemm_bf16_nt_grouped_epi(const void* A, const int32_t* a_rows, const void* W, void* C, const void* residual,
                                  const int32_t* group_counts, int num_groups, int64_t M_total, int N, int K, int64_t lda,
                                  int64_t ldw, int64_t w_group_stride, int64_t ldc, int64_t ldr, int epilogue, void* stream);

/* Causal variants (key <= query in addition to the key-padding mask): the generative branch of unified training
 * (MistralSdpaAttention with is_causal=True; causal mask :1005-1031).  Same layouts as the bidirectional entry points. */
int grit_attn_causal_fwd(const void* qkv, const uint64_t* key_bits, void* out, float* lse, int B, int S,
                         int nq, int nkv, int d, int64_t qkv_stride, int64_t out_stride, float scale, void* stream);
int grit_attn_causal_varlen_fwd(const void* qkv, const int32_t* cu_seqlens, void* out, float* lse, int B, int max_len, int nq,
                                int nkv, int d, int64_t qkv_stride, int64_t out_stride, float scale, void* stream);

/* Sliding-window causal attention (Mistral's `sliding_window`: modeling_mistral_gritlm.py:381-385 and the sliding-window causal mask of
 * :1005-1036): query q sees the `window` keys q - window + 1 .. q that the key mask allows (window >= 1; window >= S is plain causal
 * attention).  `window` counts keys INCLUDING the query's own: config.sliding_window for the mask the reference's eager / sdpa paths
 * build under its pinned transformers 4.37.2, config.sliding_window + 1 for its flash-attention path (window_size = (W, W)). */
int grit_attn_causal_window_fwd(const void* qkv, const uint64_t* key_bits, void* out, float* lse, int B, int S, int nq, int nkv, int d,
                                int64_t qkv_stride, int64_t out_stride, float scale, int window, void* stream);
int grit_attn_causal_window_varlen_fwd(const void* qkv, const int32_t* cu_seqlens, void* out, float* lse, int B, int max_len, int nq,
                                       int nkv, int d, int64_t qkv_stride, int64_t out_stride, float scale, int window, void* stream);

/* ---- pooling + normalise: gritlm/gritlm.py:178-218,156-158; training/model.py:151-165 --------- */

/* hidden [B,S,H] bf16; mask [B,S] int64 (attention mask); instr_len (nullable) [B] int32: the first
 * instr_len[b] positions are excluded from the pool but were attended to (gritlm.py:144-153).
 * out [B,H] fp32 = pooled (and L2-normalised, eps 1e-12, if normalize != 0).
 * inv_norm (nullable) [B] fp32 receives 1/max(||pooled||,eps) (saved for backward).
 * An all-masked row divides by zero exactly like the reference (:213-214). */
int grit_pool_norm_fwd(const void* hidden, const int64_t* mask, const int32_t* instr_len, float* out,
                       float* inv_norm, int B, int S, int H, int mode, int normalize, void* stream);

/* Packed variant: hidden [T,H] bf16 with document b at rows [cu_seqlens[b], cu_seqlens[b+1]) (all real tokens);
 * the first instr_len[b] rows of a document are excluded from mean / weightedmean pools. */
int grit_pool_norm_varlen_fwd(const void* hidden, const int32_t* cu_seqlens, const int32_t* instr_len, float* out,
                              float* inv_norm, int B, int H, int mode, int normalize, void* stream);

/* backward of the above w.r.t. hidden: y = forward output [B,H] fp32, dy [B,H] fp32 -> dhidden [B,S,H] bf16 */
int grit_pool_norm_bwd(const float* y, const float* dy, const float* inv_norm, const int64_t* mask,
                       const int32_t* instr_len, void* dhidden, int B, int S, int H, int mode,
                       int normalize, void* stream);

/* packed variant of the backward: dhidden [T,H] bf16 (rows of document b at [cu_seqlens[b], cu_seqlens[b+1])) */
int grit_pool_norm_varlen_bwd(const float* y, const float* dy, const float* inv_norm, const int32_t* cu_seqlens,
                              const int32_t* instr_len, void* dhidden, int B, int H, int mode, int normalize,
                              void* stream);

/* Two dense GEMMs with the same K and epilogue (STORE or RESIDUAL) in one launch -- autograd's weight-gradient GEMMs of nn.Linear
 * (q/k/v_proj + down_proj of a layer: 384 + 896 tiles of 256 x 256 fill 5 whole waves of 256 CUs instead of 2 + 4 partial ones).
 * Per problem exactly grit_gemm_bf16_nt (same bits). */
int grit_gemm_bf16_nt_pair(const void* A1, const void* W1, void* C1, const void* residual1, int64_t M1, int N1, int64_t lda1, int64_t ldw1,
                           int64_t ldc1, int64_t ldr1, const void* A2, const void* W2, void* C2, const void* residual2, int64_t M2, int N2,
                           int64_t lda2, int64_t ldw2, int64_t ldc2, int64_t ldr2, int K, int epilogue, void* stream);

/* ---- contrastive loss: gritlm/training/model.py:36-47,62-64 --------------------------------- */

/* scores = q p^T / tau (fp32 MFMA, exact f32), target[i] = i * (Np / Nq), CrossEntropyLoss(mean).
 * q [Nq,H] fp32, p [Np,H] fp32 (already gathered across ranks, rank order).
 * scores: workspace fp32 [Nq,Np] (holds d loss / d scores afterwards); loss: fp32 [1] = the mean loss; loss_rows: fp32 [Nq],
 * loss_rows[i] = the loss term of query row i (summed in a fixed order: the result is bit-reproducible, no float atomics).
 * (ABI 2 renamed this entry point from grit_infonce_fwd_bwd, whose single `loss` pointer had grown to 1 + Nq floats.)
 * Gradients are produced only for the caller's local rows, exactly the rows that carry grad in the
 * reference after `_dist_gather_tensor` (:49-60): dq [nq_loc,H] for q rows [q_off, q_off+nq_loc),
 * dp [np_loc,H] for p rows [p_off, p_off+np_loc).  dq/dp may be NULL (forward only). */
int grit_infonce_rows_fwd_bwd(const float* q, const float* p, float inv_temperature, float* scores, float* loss,
                              float* loss_rows, float* dq, float* dp, int Nq, int Np, int H, int q_off, int nq_loc,
                              int p_off, int np_loc, void* stream);

/* ---- cross-rank exchange on RCCL: DistributedContrastiveLoss._dist_gather_tensor, gritlm/training/model.py:49-60 ------------------ */

#define GRIT_COMM_ID_BYTES 128
/* Rank 0 draws a unique id (ncclGetUniqueId) into id_out[GRIT_COMM_ID_BYTES] (HOST memory); the host program hands the bytes to every
 * rank over its own channel (torch.distributed store / broadcast, MPI, a file). */
int grit_comm_unique_id(void* id_out);
/* ncclCommInitRank on the calling thread's current HIP device; collective over all `world` ranks.  *comm_out is an opaque handle. */
int grit_comm_init(const void* id, int world, int rank, void** comm_out);
/* ONE grouped all-gather of both towers (ncclGroupStart / 2 x ncclAllGather / ncclGroupEnd) on `stream`:
 * q_local [nq_rows, H], p_local [np_rows, H] fp32 -> q_all [world * nq_rows, H], p_all [world * np_rows, H], rank-major (rank r's rows
 * at [r * n, (r + 1) * n): the torch.cat order of :57-58 that the targets arange(B) * G rely on).  No staging copies; the caller owns
 * every buffer and orders `stream` against its compute stream (events).  Either tower may be empty (n rows = 0).  The backward of the
 * gather is not a collective (each rank differentiates its own rows only). */
int grit_comm_allgather_packed(void* comm, const float* q_local, int64_t nq_rows, const float* p_local, int64_t np_rows, int H,
                               float* q_all, float* p_all, void* stream);
int grit_comm_destroy(void* comm);
/* side stream restricted to the first n_cus compute units (hipExtStreamCreateWithCUMask) for the collective's kernels */
int grit_stream_create_cu_mask(int n_cus, void** stream_out);
int grit_stream_destroy(void* stream);

/* ---- helpers -------------------------------------------------------------------------------- */

/* bf16 [R,C] (row stride ld_in) -> [C,R] (row stride ld_out): operands of the dgrad / wgrad GEMMs
 * (autograd of nn.Linear: dX = dY W needs W^T K-contiguous, dW = dY^T X needs dY^T and X^T).  C % 8 == 0; R is
 * arbitrary, ld_out >= roundup(R, 8): columns R..roundup(R,8) of the output are zero-filled. */
int grit_transpose_bf16(const void* in, void* out, int64_t R, int64_t C, int64_t ld_in, int64_t ld_out, void* stream);

/* ---- backward of the encoder (contrastive step, GradCache pass 2: grad_cache.py:213-242) ------ */

/* RMSNorm backward (+ optional residual-gradient add: dx = dres + d/dx).  dy,x [T,H] bf16, w [H] bf16 ->
 * dx [T,H] bf16 (may alias dres); dw [H] fp32 is ACCUMULATED (+=).  dw_partial: fp32 workspace
 * [grit_rmsnorm_bwd_workspace_rows(T), H]. */
int64_t grit_rmsnorm_bwd_workspace_rows(int64_t T);
int grit_rmsnorm_bwd(const void* dy, const void* x, const void* w, const void* dres, void* dx, float* dw_partial, float* dw,
                     int64_t T, int H, float eps, void* stream);

/* MistralMLP activation (:177-178) on the training engine's layout gu = [gate | up] ([T,2I] bf16, output of ONE GEMM
 * against the concatenated weight): act[T,I] = silu(gate)*up, and its backward d(gu) from d(act). */
int grit_swiglu_fwd(const void* gu, void* act, int64_t T, int I, void* stream);
int grit_swiglu_bwd(const void* gu, const void* dact, void* dgu, int64_t T, int I, void* stream);

/* Attention backward (flash-style recompute, deterministic, no atomics).  qkv/out/lse as in grit_attn_bidir_fwd,
 * dout [B*S,out_stride] bf16 -> dqkv [B*S,qkv_stride] bf16 (gradients w.r.t. post-RoPE q, k and v; apply
 * grit_rope_qk_inplace(inverse=1) afterwards).  delta: fp32 workspace [B,nq,S]. */
int grit_attn_bidir_bwd(const void* qkv, const uint64_t* key_bits, const void* out, const void* dout, const float* lse,
                        float* delta, void* dqkv, int B, int S, int nq, int nkv, int d, int64_t qkv_stride,
                        int64_t out_stride, float scale, void* stream);

/* Packed variant (layouts of grit_attn_bidir_varlen_fwd: qkv/out/dout/dqkv [T, stride], lse and the delta workspace
 * [T, nq]); T = cu_seqlens[B]. */
int grit_attn_bidir_varlen_bwd(const void* qkv, const int32_t* cu_seqlens, const void* out, const void* dout, const float* lse,
                               float* delta, void* dqkv, int B, int max_len, int64_t T, int nq, int nkv, int d,
                               int64_t qkv_stride, int64_t out_stride, float scale, void* stream);

int grit_attn_causal_bwd(const void* qkv, const uint64_t* key_bits, const void* out, const void* dout, const float* lse,
                         float* delta, void* dqkv, int B, int S, int nq, int nkv, int d, int64_t qkv_stride,
                         int64_t out_stride, float scale, void* stream);
int grit_attn_causal_varlen_bwd(const void* qkv, const int32_t* cu_seqlens, const void* out, const void* dout, const float* lse,
                                float* delta, void* dqkv, int B, int max_len, int64_t T, int nq, int nkv, int d,
                                int64_t qkv_stride, int64_t out_stride, float scale, void* stream);
/* backward of the sliding-window variants (autograd through MistralSdpaAttention with the sliding-window mask) */
int grit_attn_causal_window_bwd(const void* qkv, const uint64_t* key_bits, const void* out, const void* dout, const float* lse,
                                float* delta, void* dqkv, int B, int S, int nq, int nkv, int d, int64_t qkv_stride,
                                int64_t out_stride, float scale, int window, void* stream);
int grit_attn_causal_window_varlen_bwd(const void* qkv, const int32_t* cu_seqlens, const void* out, const void* dout, const float* lse,
                                       float* delta, void* dqkv, int B, int max_len, int64_t T, int nq, int nkv, int d,
                                       int64_t qkv_stride, int64_t out_stride, float scale, int window, void* stream);

/* ---- generative branch: NextTokenLoss, gritlm/training/model.py:66-107 -------------------------- */

/* Cross entropy over the vocabulary on logits [T,V] bf16 (leading dimension ld; the lm_head GEMM output, upcast to fp32 on the
 * fly like logits.float(), modeling_mistral_gritlm.py:1177).  labels [T] int64, -100 = ignore_index (rows already shifted by the
 * caller: label[t] = token t+1).  lse [T], loss_row [T] fp32 (0 for ignored rows); the caller reduces (sum / count). */
int grit_ce_fwd(const void* logits, int64_t ld, const int64_t* labels, float* lse, float* loss_row, int64_t T, int V, void* stream);
/* In place: logits <- d loss / d logits = (softmax - onehot) * scale * (*dev_scale) as bf16 (ignored rows: 0).
 * dev_scale (nullable) is a device scalar, e.g. 1 / #valid tokens, so the host never waits for the count. */
int grit_ce_bwd(void* logits, int64_t ld, const int64_t* labels, const float* lse, const float* dev_scale, float scale, int64_t T,
                int V, void* stream);

/* ---- token-by-token decode (generation from cached document KV): rag/eval.py:237-302, gritlm.py:131-140 ---------- */

/* out[b,:] = x[b,:] W^T for 1..8 rows (HBM-bound GEMV; larger batches: grit_gemm_bf16_nt).  Epilogues as the GEMM:
 * STORE, RESIDUAL (out = bf16(xW^T) + residual), SWIGLU (W = interleaved gate/up rows, out [B, N/2]). */
int grit_gemv_bf16(const void* x, const void* W, void* out, int B, int N, int K, int64_t ldx, int64_t ldw, int64_t ldo,
                   int epilogue, const void* residual, int64_t ldr, void* stream);
/* The same GEMV with MistralRMSNorm (:84-89) applied to x on the fly (x is the raw residual stream, ln_weight [K] bf16): saves the
 * separate 1-row RMSNorm launch of a decode step.  Epilogues STORE and SWIGLU. */
int grit_rmsnorm_gemv_bf16(const void* x, const void* ln_weight, float eps, const void* W, void* out, int B, int N, int K,
                           int64_t ldx, int64_t ldw, int64_t ldo, int epilogue, void* stream);
/* The DEFERRED form: out = rsqrt(mean x^2 + eps) * (W (x * ln_weight)) -- the scale multiplies the finished dot products and the sum of
 * squares is reduced beside them from the x pieces the product loads anyway, so nothing waits in front of the weight stream (the exact
 * form above has to know the row's RMS before its first product).  x_n is never rounded to bf16 (the reference rounds it twice, :84-89):
 * one rounding FEWER than the reference's arithmetic, not the same bits as RMSNorm followed by the GEMV.  Same arguments. */
int grit_rmsnorm_gemv_bf16_deferred(const void* x, const void* ln_weight, float eps, const void* W, void* out, int B, int N, int K,
                                    int64_t ldx, int64_t ldw, int64_t ldo, int epilogue, void* stream);
/* RoPE (:138-163) of the new token's q and k at position lens[b] + append of its k, v to the cache, one launch: q is rotated in place in
 * qkv [B, qkv_stride]; cos/sin tables [Lmax, d/2] fp32 as grit_rope_qk_inplace. */
int grit_rope_kv_append(void* qkv, const float* cos_tab, const float* sin_tab, void* cache_k, void* cache_v, const int32_t* lens, int B,
                        int nq, int nkv, int d, int Lmax, int64_t qkv_stride, void* stream);
/* Append the (already rotated) k, v of one new token per sequence: qkv [B, qkv_stride] -> cache_{k,v}[b, h, lens[b], :],
 * caches [B, nkv, Lmax, d] bf16 (the layout encode(get_cache=True) returns per layer), lens int32 [B] on the device. */
int grit_kv_append(const void* qkv, void* cache_k, void* cache_v, const int32_t* lens, int B, int nq, int nkv, int d, int Lmax,
                   int64_t qkv_stride, void* stream);
/* One query row per sequence and head against keys 0..lens[b] of the cache (flash-decoding split + combine), head_dim 128.
 * q [B, q_stride] (heads at h*d), out [B, out_stride]; workspace: grit_attn_decode_workspace_floats(...) fp32. */
int64_t grit_attn_decode_workspace_floats(int B, int nq, int nkv, int Lmax);
int grit_attn_decode(const void* q, const void* cache_k, const void* cache_v, const int32_t* lens, void* out, float* workspace,
                     int B, int nq, int nkv, int d, int Lmax, int64_t q_stride, int64_t out_stride, float scale, void* stream);
/* The same with the step's RoPE and KV-append folded in: qkv [B, qkv_stride] is the raw fused projection of the new token; q is rotated
 * on the fly at position lens[b], k is rotated and k, v are appended to the caches by the workgroup whose key slice contains that
 * position (= grit_rope_kv_append + grit_attn_decode in one launch; the caches are written). */
int grit_attn_decode_rope(const void* qkv, const float* cos_tab, const float* sin_tab, void* cache_k, void* cache_v, const int32_t* lens,
                          void* out, float* workspace, int B, int nq, int nkv, int d, int Lmax, int64_t qkv_stride, int64_t out_stride,
                          float scale, void* stream);
/* Greedy step: next[b] = argmax_v logits[b,v] (bf16 logits, lowest index on ties); optionally history[b, *step] = next[b],
 * *step += 1 and lens[b] += 1 -- all on the device, so a whole decode step is one HIP graph. */
int grit_argmax_advance(const void* logits, int64_t ld, int V, int64_t* next, int32_t* lens, int64_t* history,
                        int64_t hist_stride, int32_t* step, int B, void* stream);

/* The decode step on fp16 operands (ABI 5): the continuation of encode(get_cache=True) under the encoder's "f16_operands" / "f16_stream"
 * policies (gritlm.py:131-140 hands the document K/V to model.generate, rag/eval.py:237-302).  Formats: the residual stream, the fused
 * q|k|v row of the new token and the logits are fp32 (residual adds, q's rotation and the argmax see unrounded values); the weights, the
 * K/V cache and every GEMV operand row (the stream's copy in front of a norm, ctx, act) are IEEE fp16, rounded once from fp32.  Values
 * beyond the fp16 range raise grit_f16_overflow_flag().
 *   grit_gemv_f16: x fp16 [B,K], W fp16 [N,K]; STORE: out fp32 [B,N]; RESIDUAL: out fp32 = residual fp32 + x W^T and, if out16 is given,
 *                  out16 = fp16(out) (the operand copy the next norm + GEMV reads); SWIGLU: out fp16 [B,N/2] = fp16(silu(g) * u), one
 *                  rounding.  ld* in elements of the respective format.
 *   grit_rmsnorm_gemv_f16_deferred: x = the fp16 copy of the stream, ln_weight bf16 (as stored), W fp16; out = rsqrt(mean x^2 + eps) *
 *                  (W (x * ln_weight)) accumulated in fp32; STORE -> fp32, SWIGLU -> fp16.
 *   grit_attn_decode_rope_f16: qkv fp32 [B, qkv_stride]; q is rotated and scaled without a rounding; k (rotated) and v are rounded to fp16
 *                  once, into caches [B, nkv, Lmax, d] fp16; out (ctx) fp16.  Workspace as grit_attn_decode.
 *   grit_argmax_advance_f32: grit_argmax_advance on fp32 logits. */
int grit_gemv_f16(const void* x, const void* W, void* out, int B, int N, int K, int64_t ldx, int64_t ldw, int64_t ldo,
                  int epilogue, const void* residual, int64_t ldr, void* out16, int64_t ldo16, void* stream);
int grit_rmsnorm_gemv_f16_deferred(const void* x, const void* ln_weight, float eps, const void* W, void* out, int B, int N, int K,
                                   int64_t ldx, int64_t ldw, int64_t ldo, int epilogue, void* stream);
int grit_attn_decode_rope_f16(const void* qkv, const float* cos_tab, const float* sin_tab, void* cache_k, void* cache_v,
                              const int32_t* lens, void* out, float* workspace, int B, int nq, int nkv, int d, int Lmax,
                              int64_t qkv_stride, int64_t out_stride, float scale, void* stream);
int grit_argmax_advance_f32(const void* logits, int64_t ld, int V, int64_t* next, int32_t* lens, int64_t* history,
                            int64_t hist_stride, int32_t* step, int B, void* stream);

/* Sparse-MoE decode (ABI 5; MixtralSparseMoeBlock.forward, scripts/modeling_mixtral_gritlm.py:839-882, at 1..8 rows): x [B,K] times ONE
 * matrix of the stack W [E,N,K] -- matrix expert[0], an int32 in DEVICE memory (the router's choice for the row, so that the decode step
 * stays one HIP graph); w_expert_stride = elements between consecutive matrices.  Epilogues STORE and SWIGLU; formats of grit_gemv_bf16 /
 * grit_gemv_f16 (f16 STORE writes fp32). */
int grit_gemv_bf16_expert(const void* x, const void* W, void* out, const int32_t* expert, int64_t w_expert_stride, int B, int N, int K,
                          int64_t ldx, int64_t ldw, int64_t ldo, int epilogue, void* stream);
int grit_gemv_f16_expert(const void* x, const void* W, void* out, const int32_t* expert, int64_t w_expert_stride, int B, int N, int K,
                         int64_t ldx, int64_t ldw, int64_t ldo, int epilogue, void* stream);
/* ... and its combine on fp16 operands (:876-880 in fp32): h[b] += weights[b,0] y[2b] + weights[b,1] y[2b+1] (h [B,H] the fp32 residual stream,
 * y [2B,H] the chosen experts' fp32 outputs), h16 = fp16(h) (the operand copy the next norm + GEMV reads; beyond the range: the overflow flag). */
int grit_moe_decode_combine_f32(float* h, void* h16, const float* y, const float* weights, int B, int H, void* stream);

/* A prompt chunk on top of a cached prefix without a token-by-token loop (ABI 5): what model.generate() does with the query tokens it is
 * handed next to past_key_values (rag/eval.py:277-302 -- the attention mask covers the cache, the new tokens attend to it and causally to
 * each other).  V rows are tokens of B <= V sequences: row v belongs to sequence cache_row[v] of the caches [B, nkv, Lmax, d] and sits at
 * position lens[v] of it (consecutive tokens of one sequence: prefix, prefix + 1, ...).  grit_rope_kv_append_rows rotates q (in place) and
 * k at position lens[v] and appends every row's k, v to its sequence's cache; grit_attn_decode_rows then lets every row attend to keys
 * 0 .. lens[v] of its sequence.  f16 = 0: bf16 rows and caches (grit_rope_kv_append / grit_attn_decode arithmetic, bit for bit);
 * f16 != 0: the fp16-operand formats (fp32 q|k|v rows, fp16 caches and ctx).  Workspace: grit_attn_decode_workspace_floats(V, ...). */
int grit_rope_kv_append_rows(void* qkv, const float* cos_tab, const float* sin_tab, void* cache_k, void* cache_v, const int32_t* lens,
                             const int32_t* cache_row, int V, int nq, int nkv, int d, int Lmax, int64_t qkv_stride, int f16, void* stream);
int grit_attn_decode_rows(const void* q, const void* cache_k, const void* cache_v, const int32_t* lens, const int32_t* cache_row, void* out,
                          float* workspace, int V, int nq, int nkv, int d, int Lmax, int64_t q_stride, int64_t out_stride, float scale,
                          int f16, void* stream);

/* ---- RAG index search: rag/index.py:97-104 (scores = queries @ embeddings; torch.topk) ------------------------ */

/* Brute-force kNN by inner product.  queries [Q,H] fp32 (contiguous); embeddings: element (n,h) at n*emb_stride_n + h*emb_stride_h, so
 * both the [N,H] layout and the reference's [H,N] index layout (index.py:141) are accepted; exact-f32 MFMA scores, then the k best per
 * query (descending, lower index first on ties): out_scores [Q,k] fp32, out_index [Q,k] int64.  k <= 1024.
 * workspace: grit_knn_workspace_bytes(Q, N, k) bytes of device memory. */
int64_t grit_knn_workspace_bytes(int Q, int64_t N, int k);
int grit_knn_topk(const float* queries, const float* embeddings, int Q, int64_t N, int H, int64_t emb_stride_n, int64_t emb_stride_h,
                  int k, void* workspace, float* out_scores, int64_t* out_index, void* stream);

/* embedding backward (nn.Embedding's weight gradient, scripts/modeling_mistral_gritlm.py:918,994), deterministic: no atomics.
 * order [T] int64 = a STABLE argsort of the token ids, sorted_ids[i] = ids[order[i]]; dh [T,H] bf16; grad [V,H] bf16 (in/out):
 * grad[id, :] = bf16(grad[id, :] + sum of dh[t, :] over the tokens t with ids[t] == id, added in token order in fp32).
 * Rows of ids that do not occur are neither read nor written.  Ids outside [0, V) are clamped to row 0 / row V - 1 BEFORE the runs of
 * equal ids are formed (a sorted sequence stays sorted under the clamp), so every gradient row has exactly one writer whatever the ids.
 * One workgroup sums one run: a run of thousands of rows (the pad token of a padded batch) is a serial tail of about 1 us per 8 rows --
 * the packed training path never feeds pad rows in. */
int grit_embed_scatter_add_sorted(const void* dh, const int64_t* sorted_ids, const int64_t* order, void* grad, int64_t T, int H,
                                  int64_t V, void* stream);

/* acc (bf16, n values) += x (fp32): folds an fp32 gradient into a bf16 .grad buffer */
int grit_accum_bf16_from_f32(void* acc, const float* x, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GRITLM_HIP_H */
