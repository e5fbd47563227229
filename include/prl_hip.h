/*
 * prl_hip.h: C ABI of libprl_hip.so, the MI355X (gfx950) implementation of
 * the policy-update hot path of an RL post-training trainer.
 *
 * The reference this library stands in for is pure Python on top of torch; it
 * has no FFI of its own for this path.  Each entry point below therefore names
 * the reference CALLABLE it replaces (file:line relative to the reference
 * root), and the Python package next to this header keeps those callables'
 * names and signatures on top of these functions.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - every buffer is allocated and owned by the caller; the library allocates
 *     nothing and keeps no state between calls except a thread-local error text;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     kernels are enqueued on it and the call returns without synchronising;
 *   - return value: PRL_OK (0) or a negative PRL_E* code, with a description
 *     available from prl_last_error() on the calling thread;
 *   - logits are row-major [T, V] with `row_stride` ELEMENTS between rows,
 *     dtype PRL_F32 or PRL_BF16; all arithmetic is fp32;
 *   - a packed row holds several samples back to back; token t belongs to
 *     sample seg_id[t], sample s covers [seg_start[s], seg_start[s+1]).
 */
#ifndef PRL_HIP_H
#define PRL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRL_OK 0
#define PRL_EINVAL (-1)   /* bad argument (null pointer, negative size, ...) */
#define PRL_ELAUNCH (-2)  /* the HIP runtime refused a launch or a query    */
#define PRL_EDTYPE (-3)   /* unsupported dtype code                          */

#define PRL_F32 0
#define PRL_BF16 1

#define PRL_RATIO_TOKEN 0
#define PRL_RATIO_SEQUENCE 1

#define PRL_ADV_DRGRPO 0
#define PRL_ADV_DRGRPO_NEGCLIPPED 1
#define PRL_ADV_RLOO 2
#define PRL_ADV_OPO 3

/* Text of the last error on this thread ("" if none). Never NULL. */
const char* prl_last_error(void);

/* ABI version of the library: bumped when a signature changes. */
int prl_abi_version(void);

/*
 * Bytes of device scratch prl_logprob_entropy_fwd wants for a [T, V] problem.
 * 0 means "none needed" (enough rows to fill the chip with one workgroup per
 * row).  Passing less (or NULL) is allowed and only costs speed at small T.
 */
size_t prl_logprob_workspace_bytes(int64_t T, int64_t V);

/*
 * Per-token log-probability and entropy over the vocabulary in ONE pass over
 * the logits.  Replaces the chain
 *     shift_logits            src/prime_rl/trainer/rl/loss.py:51-58
 *     / temperature           src/prime_rl/trainer/rl/train.py:287
 *     selective_log_softmax   src/prime_rl/trainer/rl/loss.py:13-39
 *     compute_entropy         src/prime_rl/trainer/rl/loss.py:43-47
 * without materialising the shifted or scaled copies.
 *
 *   shift = 1: token t is scored with logits row t-1; token 0 with an all-zero
 *              row (logprob = -ln V, entropy = lse = ln V); the last logits
 *              row is never read.
 *   shift = 0: token t is scored with row t (the logits are already shifted).
 *
 * Outputs (fp32 [T], any may be NULL to skip it):
 *   logprob[t] = x[ids[t]]/temperature - lse[t]
 *   entropy[t] = lse[t] - sum_v softmax(x/temperature)_v * x_v/temperature
 *   lse[t]     = logsumexp(x/temperature)
 * An id outside [0, V) yields NaN in logprob[t] (torch.gather would raise).
 *
 * Accuracy against the reference's fp32 torch results: logprob and lse within 2e-6 at V = 152 K.  The entropy is
 * within 1e-5 of a float64 evaluation of loss.py:43-47; at V = 152 K and temperature 0.6 that is 2.6e-4 .. 2.8e-4
 * away from what the reference's own fp32 softmax-multiply-sum returns on the CPU, because that fp32 evaluation is
 * itself that far from float64 there (reference-made vectors tests/golden/full_v*.npz hold both values;
 * tests/test_gpu_config_shapes.py asserts both distances).  At the small vocabularies of the other vectors the
 * entropy meets 1e-4 against the reference directly.
 */
int prl_logprob_entropy_fwd(const void* logits, int logits_dtype, int64_t T, int64_t V, int64_t row_stride,
                            const int64_t* ids, float temperature, int shift, float* logprob, float* entropy,
                            float* lse, void* workspace, size_t workspace_bytes, void* stream);

/*
 * prl_logprob_entropy_fwd for callers that cannot allocate device memory or hand over a workspace: code under
 * HIP graph capture, tight serving loops that score a few tokens per call, harnesses that forbid hipMalloc.
 * It has no counterpart in the reference.  Same arguments, same kernels and bit-identical results; the partial
 * soft-max states of rows that are split over several workgroups (T < 1024) live in a fixed array of
 * prl_logprob_static_scratch_bytes() = 49152 bytes of device memory that every GPU the library runs on holds
 * until the process ends.  The call makes no hipMalloc / hipFree; a problem whose scratch would not fit the
 * array is refused with PRL_EINVAL (no size the split rule produces today does).  The first call on a device
 * looks the array's address up (a host-side runtime query): make it outside stream capture.  All calls on one
 * GPU share the array, so calls that overlap on one GPU (two streams, two threads) race; calls on one stream
 * are ordered.  Nothing in the array needs clearing between calls.
 */
size_t prl_logprob_static_scratch_bytes(void);
int prl_logprob_entropy_fwd_static(const void* logits, int logits_dtype, int64_t T, int64_t V, int64_t row_stride,
                                   const int64_t* ids, float temperature, int shift, float* logprob, float* entropy,
                                   float* lse, void* stream);

/*
 * Gradient of sum_t g[t] * logprob[t] with respect to the logits, in one pass:
 *     dlogits[r, v] = g[t]/temperature * (1[v == ids[t]] - exp(x[r,v]/temperature - lse[t])),
 *     t = r + shift.
 * With shift = 1 the last row of dlogits is zero.  Rows whose g[t] is zero are
 * written as zeros without reading the logits.  Replaces what autograd runs for
 * loss.backward() through selective_log_softmax, the division and shift_logits
 * (src/prime_rl/trainer/rl/train.py:310 back through :288-286).
 *
 *   g       fp32 [T]   dL/dlogprob (from prl_grpo_loss)
 *   gscale  optional device scalar multiplied into every g[t] (autograd's
 *           incoming gradient of the loss); NULL = 1
 *   dlogits [T, V] with out_row_stride elements between rows, dtype out_dtype
 */
int prl_logprob_bwd(const void* logits, int logits_dtype, int64_t T, int64_t V, int64_t row_stride,
                    const int64_t* ids, const float* lse, const float* g, const float* gscale, float temperature,
                    int shift, void* dlogits, int out_dtype, int64_t out_row_stride, void* stream);

/*
 * prl_logprob_bwd over a selection of rows: row r is written only where row_select[r] != 0 (uint8 [T], device),
 * every other row of dlogits keeps what it holds.  The second pass behind prl_policy_rows_seq_fwd_bwd, for the
 * rows that call has marked as not written (one workgroup per row and piece, which leaves at once where the
 * mark is 0).
 */
int prl_logprob_bwd_selected(const void* logits, int logits_dtype, int64_t T, int64_t V, int64_t row_stride,
                             const int64_t* ids, const float* lse, const float* g, const float* gscale,
                             float temperature, int shift, void* dlogits, int out_dtype, int64_t out_row_stride,
                             const uint8_t* row_select, void* stream);

/*
 * Sample boundaries of a packed row from its position ids.  Replaces
 * get_response_lengths (src/prime_rl/trainer/utils.py:36-69) bit for bit:
 * index i >= 1 opens a new sample iff pos[i] == 0, pos[i-1] != 0, i+1 < T and
 * pos[i+1] == 1.
 *
 *   seg_id    int32 [T]               sample index of every token (may be NULL)
 *   seg_start int64 [max(T, 1) + 1]   start offsets, seg_start[n_seg] = T
 *   n_seg     int32 [1]               number of samples, at least 1: an empty row (T == 0) is ONE sample of
 *                                     length 0, as the reference's `boundaries = [0]` makes it (:52, :63-67)
 */
int prl_segments_from_position_ids(const int64_t* pos, int64_t T, int32_t* seg_id, int64_t* seg_start,
                                   int32_t* n_seg, void* stream);

/*
 * Ratio-clip policy loss over the samples of a packed row and its gradient
 * with respect to the log-probabilities.  Replaces compute_loss
 * (src/prime_rl/trainer/rl/loss.py:61-124) and autograd through it.
 *
 *   lp, old, adv  fp32 [T];  mask uint8 [T] (0/1, torch.bool layout)
 *   seg_start     int64 [n_seg + 1] on the device;  n_seg int32 [1] on the device
 *   max_seg       capacity of the per-sample arrays (>= the value in n_seg)
 *   ratio_type    PRL_RATIO_TOKEN | PRL_RATIO_SEQUENCE;  length_norm 0/1
 *   loss_scale    the normaliser; max(loss_scale, 1) is applied as in the reference
 * Outputs
 *   loss      fp32 [1]
 *   ratio, clipped, is_clipped
 *             token mode: fp32 [T], one value per token;
 *             sequence mode: fp32 [max_seg], one value per sample
 *   g         fp32 [T]  dloss/dlp (NULL to skip)
 *   seg_loss  fp32 [max_seg] per-sample loss terms before the final sum (scratch, required)
 */
int prl_grpo_loss(const float* lp, const float* old, const float* adv, const uint8_t* mask,
                  const int64_t* seg_start, const int32_t* n_seg, int64_t max_seg, int64_t T, int ratio_type,
                  int length_norm, float clip_ratio, float loss_scale, float* loss, float* ratio, float* clipped,
                  float* is_clipped, float* g, float* seg_loss, void* stream);

/*
 * The forward half of one micro-batch in a single call: prl_logprob_entropy_fwd
 * with shift = 1, prl_segments_from_position_ids and prl_grpo_loss, chained on
 * `stream`.  Replaces the trainer's sequence src/prime_rl/trainer/rl/train.py:285-303
 * (shift, divide, selective_log_softmax, get_response_lengths, compute_loss,
 * compute_entropy).  The per-sample arrays (ratio, clipped, is_clipped in
 * sequence mode, seg_loss) need room for T / 2 + 1 entries, seg_start for
 * T + 1.  entropy, ratio, clipped, is_clipped, g and seg_id may be NULL.
 * skip_masked = 1 leaves tokens whose mask is 0 unscored: their logits row is
 * not read and their logprob, entropy and lse come back as 0.  Loss, gradient
 * and every masked statistic are unchanged, because the reference uses those
 * values only under the mask (loss.py:93, :104; train.py:313-315); with prompts
 * masked out this removes their share of the forward's HBM traffic.
 */
int prl_policy_forward(const void* logits, int logits_dtype, int64_t T, int64_t V, int64_t row_stride,
                       const int64_t* ids, const int64_t* pos, const float* adv, const float* old,
                       const uint8_t* mask, int skip_masked, float temperature, int ratio_type, int length_norm,
                       float clip_ratio, float loss_scale, float* logprob, float* entropy, float* lse, float* loss, float* ratio,
                       float* clipped, float* is_clipped, float* g, float* seg_loss, int32_t* seg_id,
                       int64_t* seg_start, int32_t* n_seg, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Everything of prl_policy_forward that comes after the log-probabilities:
 * sample boundaries from the position ids (src/prime_rl/trainer/utils.py:36-69),
 * the ratio-clip loss and its three logged tensors
 * (src/prime_rl/trainer/rl/loss.py:61-124) and g = dL/dlogprob.  One launch of
 * one workgroup for rows of up to 16384 tokens.  Buffer sizes as for
 * prl_policy_forward.
 */
int prl_policy_tail(const int64_t* pos, const float* logprob, const float* old, const float* adv,
                    const uint8_t* mask, int64_t T, int ratio_type, int length_norm, float clip_ratio,
                    float loss_scale, float* loss, float* ratio, float* clipped, float* is_clipped, float* g,
                    float* seg_loss, int32_t* seg_id, int64_t* seg_start, int32_t* n_seg, void* stream);

/*
 * Token-mode forward AND backward of one packed row in a single pass over the
 * logits.  Each workgroup keeps one bf16 row of logits in registers, so the
 * row is read from memory once and the gradient is written once: per token V
 * logits in, V gradient elements out, and no second read for the backward.
 * Replaces, for ratio_type "token", src/prime_rl/trainer/rl/train.py:285-310
 * from the shift to `loss.backward()` reaching the logits: logprob, entropy
 * and lse as prl_logprob_entropy_fwd with shift = 1 gives them, and
 *   dlogits[t-1, :] = g[t] / temperature * (onehot(ids[t]) - softmax(logits[t-1] / temperature)),
 *   g[t] = mask[t] * [ratio <= clip] * (-adv[t] * ratio / max(loss_scale, 1)),  ratio = exp(logprob[t] - old[t])
 * (src/prime_rl/trainer/rl/loss.py:90, 98-104 differentiated), i.e. the
 * gradient of the scaled loss for an incoming gradient of 1; dlogits[T-1, :] = 0.
 * The scalar loss and the logged ratio tensors come from prl_policy_tail run
 * on the logprob this call wrote.  skip_masked as in prl_policy_forward.
 *
 * Limits (prl_policy_rows_supported returns 1 when they hold): bf16 logits and
 * gradient, V a multiple of 8 and at most 155648, row strides multiples of 8,
 * both base pointers 16-byte aligned.  Sequence-mode losses and fp32 logits
 * use prl_policy_forward + prl_logprob_bwd.
 */
int prl_policy_rows_supported(int logits_dtype, int64_t T, int64_t V, int64_t row_stride, int out_dtype,
                              int64_t out_row_stride);
int prl_policy_rows_fwd_bwd(const void* logits, int logits_dtype, int64_t T, int64_t V, int64_t row_stride,
                            const int64_t* ids, const float* adv, const float* old, const uint8_t* mask,
                            int skip_masked, float temperature, float clip_ratio, float loss_scale, float* logprob,
                            float* entropy, float* lse, void* dlogits, int out_dtype, int64_t out_row_stride,
                            void* stream);

/*
 * Measurement aid, no counterpart in the reference: the NEXT prl_policy_rows_fwd_bwd call made from this thread
 * stamps the two HIP events (hipEvent_t, created by the caller with timing enabled) with the start and the end of its
 * row kernel on the GPU (the launch goes through hipExtLaunchKernelGGL), so that hipEventElapsedTime(start, stop) is
 * the kernel's own duration: what a profiler's kernel trace reports, without an event record on either side of the
 * launch holding the queue up.  One launch per request; two NULLs take a request back.
 */
int prl_policy_rows_time_next(void* start_event, void* stop_event);

/*
 * prl_policy_rows_fwd_bwd over GATHERED rows: the model's LM head was run only at the positions whose next token
 * is scored, so `logits` holds R rows, row r being the logits at position positions[r] of a packed row of T tokens
 * (it predicts token positions[r] + 1).  With prompts masked out of the loss (src/prime_rl/orchestrator/batch.py:21-62
 * gives prompt tokens mask False) the reference still computes and differentiates the logits of every position
 * (src/prime_rl/trainer/model.py:251-254, src/prime_rl/trainer/rl/train.py:285-310) although loss, gradient and
 * every logged statistic use them only under the mask (loss.py:93, :104; train.py:313-315); leaving those rows out
 * of the LM head's three products and out of this call changes none of them.
 *
 *   logits, dlogits  bf16 [R, V] (limits as prl_policy_rows_supported(.., R, ..) says)
 *   positions        int64 [R], device; 0 <= positions[r] < T - 1, in any order, no position twice.  A row whose
 *                    position is outside that range is not read and gets a zero gradient row.
 *   ids, adv, old, mask, logprob, entropy, lse: [T], indexed by token, as for prl_policy_rows_fwd_bwd
 * Every token that no row predicts comes back unscored (logprob = entropy = lse = 0), as skip_masked = 1 leaves a
 * masked token; token 0 is scored against an all-zero row when mask[0] is set; a listed row whose token is masked
 * out is not read (zero gradient row, token unscored).  R = 0 is allowed.  Follow with prl_policy_tail as usual.
 *
 * dlogits = NULL: the rows are only scored (logprob, entropy, lse as above), nothing is differentiated and
 * out_dtype / out_row_stride are ignored.  For passes that need the logged values but no gradient: the reference's
 * logprob recompute (src/prime_rl/trainer/rl/train.py:227-245) and rows whose advantages are all zero.
 */
int prl_policy_rows_gathered_fwd_bwd(const void* logits, int logits_dtype, int64_t R, int64_t V, int64_t row_stride,
                                     const int64_t* positions, int64_t T, const int64_t* ids, const float* adv,
                                     const float* old, const uint8_t* mask, float temperature, float clip_ratio,
                                     float loss_scale, float* logprob, float* entropy, float* lse, void* dlogits,
                                     int out_dtype, int64_t out_row_stride, void* stream);

/*
 * Sequence-mode forward AND backward of one packed row in a single pass over the logits (ratio_type "sequence":
 * one importance ratio per sample, src/prime_rl/trainer/rl/loss.py:92-96, 107-108; the chain replaced is the same
 * src/prime_rl/trainer/rl/train.py:285-310 as for prl_policy_rows_fwd_bwd).  A token's gradient needs the summed
 * log-ratio of its whole sample.  The workgroups of a sample run at the same time (they start in row order, one
 * per compute unit); each publishes its token's masked log-ratio to device memory and waits, for a bounded time,
 * until its sample's other tokens are there; then
 *   dlogits[t-1, :] = g_s / temperature * (onehot(ids[t]) - softmax(logits[t-1] / temperature))   for mask[t],
 * with g_s the sample's dloss/dlogprob as prl_grpo_loss defines it (incoming gradient 1), zero rows elsewhere.
 * logprob, entropy, lse as prl_logprob_entropy_fwd with shift = 1 (skip_masked as in prl_policy_forward).
 *
 * pending (uint8 [T], device, written for every row): 1 where the row's gradient has NOT been written, because its
 * sample is longer than max_wait_len tokens (at most about 250 can be resident together) or because its
 * workgroup gave up waiting after spin_limit sweeps.  The caller finishes with
 *   prl_policy_tail (loss, logged tensors, g)  and  prl_logprob_bwd_selected(..., g, ..., row_select = pending).
 * Results do not depend on which rows ended up pending beyond rounding (the second pass computes the same
 * gradient from the same g).  workspace: prl_policy_rows_seq_workspace_bytes(T) bytes, 16-byte aligned; the call
 * clears what it needs of it.  Same limits as prl_policy_rows_fwd_bwd (prl_policy_rows_supported).
 */
size_t prl_policy_rows_seq_workspace_bytes(int64_t T);
int prl_policy_rows_seq_fwd_bwd(const void* logits, int logits_dtype, int64_t T, int64_t V, int64_t row_stride,
                                const int64_t* ids, const int64_t* pos, const float* adv, const float* old,
                                const uint8_t* mask, int skip_masked, float temperature, int length_norm,
                                float clip_ratio, float loss_scale, float* logprob, float* entropy, float* lse,
                                void* dlogits, int out_dtype, int64_t out_row_stride, uint8_t* pending,
                                int max_wait_len, int spin_limit, void* workspace, size_t workspace_bytes,
                                void* stream);

/*
 * x[T, V] *= scale[0] in place, where scale is a device scalar; when it is
 * exactly 1 the kernel returns without touching x.  Lets the backward of
 * prl_policy_rows_fwd_bwd honour an incoming loss gradient other than 1
 * (autograd's grad_output at src/prime_rl/trainer/rl/train.py:310) without a
 * host synchronisation to look at its value.
 */
int prl_scale_rows_if_not_one(void* x, int dtype, int64_t T, int64_t V, int64_t row_stride, const float* scale,
                              void* stream);

/*
 * RMSNorm over the rows of a [rows, H] activation, forward and backward:
 *   y = w * cast(x32 * rsqrt(mean(x32^2) + eps)),  x32 = float(x), cast to the dtype of x,
 * the arithmetic of the normalisation module in the transformer blocks the
 * trainer runs.  Counterpart of the fused kernels the reference switches on
 * with `model.impl = "liger_kernel"` (src/prime_rl/trainer/model.py:74-75,
 * src/prime_rl/trainer/config.py:179-184).  x, y, gy, gx and w share `dtype`
 * (bf16 or fp32); rstd (fp32 [rows]) is written by the forward for the
 * backward; gw receives the weight gradient summed over the rows in a fixed
 * order.  Row strides in elements; pointers and strides 16-byte aligned, H a
 * multiple of 8 (bf16) or 4 (fp32); prl_rmsnorm_supported says whether a row
 * length is handled.  The backward needs prl_rmsnorm_bwd_workspace_bytes of
 * caller scratch.
 */
int prl_rmsnorm_supported(int dtype, int64_t H);
size_t prl_rmsnorm_bwd_workspace_bytes(int dtype, int64_t rows, int64_t H);
/* x_group > 0: the rows of x come in groups of x_group (x_row_stride inside a group, x_group_stride from group to
 * group, in elements), e.g. the heads of a token inside the wider row of a stacked q/k/v projection; 0: plain rows. */
int prl_rmsnorm_fwd(const void* x, int64_t x_row_stride, const void* w, void* y, int64_t y_row_stride, float* rstd,
                    int dtype, int64_t rows, int64_t H, float eps, int64_t x_group, int64_t x_group_stride,
                    void* stream);
int prl_rmsnorm_bwd(const void* x, int64_t x_row_stride, const void* w, const float* rstd, const void* gy,
                    int64_t gy_row_stride, void* gx, int64_t gx_row_stride, void* gw, int dtype, int64_t rows,
                    int64_t H, void* workspace, size_t workspace_bytes, int64_t x_group, int64_t x_group_stride,
                    void* stream);
/* The residual connection folded into the normalisation that follows it, as a transformer block has it twice
 * (`hidden = residual + branch; normed = norm(hidden)`, the decoder layer of the blocks loaded at
 * src/prime_rl/trainer/model.py:71-88): sum = cast(x + res) is written out (the next residual) and normalised in
 * the same pass; the backward adds `gadd`, the gradient that reaches `sum` past the normalisation, to the rounded
 * gx.  Results are those of a separate addition kernel before / after prl_rmsnorm_fwd / _bwd, bit for bit; what is
 * saved is one read of the sum forward and one write and one read of gx backward.  Plain rows only (no groups);
 * the workspace is prl_rmsnorm_bwd_workspace_bytes as above; x here is the stored sum. */
int prl_add_rmsnorm_fwd(const void* x, int64_t x_row_stride, const void* res, int64_t res_row_stride, void* sum,
                        int64_t sum_row_stride, const void* w, void* y, int64_t y_row_stride, float* rstd, int dtype,
                        int64_t rows, int64_t H, float eps, void* stream);
int prl_add_rmsnorm_bwd(const void* x, int64_t x_row_stride, const void* w, const float* rstd, const void* gy,
                        int64_t gy_row_stride, const void* gadd, int64_t gadd_row_stride, void* gx,
                        int64_t gx_row_stride, void* gw, int dtype, int64_t rows, int64_t H, void* workspace,
                        size_t workspace_bytes, void* stream);

/*
 * Rotary position embedding of query and key in one launch:
 *   out = x * cos + rotate_half(x) * sin   over the last dimension D,
 * or, with inverse = 1, the transpose of that map (the backward applied to the
 * incoming gradients).  Same switch in the reference as prl_rmsnorm_*
 * (`model.impl = "liger_kernel"`, src/prime_rl/trainer/model.py:74-75); the
 * arithmetic is transformers' apply_rotary_pos_emb.  q is [B, q_heads, T, D],
 * k is [B, k_heads, T, D], any strides with a contiguous last dimension:
 * q_strides / k_strides are HOST arrays of 6 element strides (batch, head,
 * token of the input, then of the output).  cos, sin are [B, T, D] with the
 * given batch and token strides.  Everything 16-byte aligned; D a multiple of
 * 16 (bf16) or 8 (fp32) (prl_rope_supported).
 */
int prl_rope_supported(int dtype, int64_t D);
int prl_rope(const void* q, void* q_out, const int64_t* q_strides, int64_t q_heads, const void* k, void* k_out,
             const int64_t* k_strides, int64_t k_heads, const void* cos_, const void* sin_, int64_t cs_batch_stride,
             int64_t cs_token_stride, int dtype, int64_t B, int64_t T, int64_t D, int inverse, void* stream);

/*
 * out = silu(gate) * up over n contiguous elements, and the backward
 *   gup = gout * silu(gate),  ggate = gout * up * silu'(gate).
 * The gated MLP activation of the transformer blocks; third kernel of the
 * reference's `model.impl = "liger_kernel"` switch
 * (src/prime_rl/trainer/model.py:74-75).  All tensors share `dtype` (bf16 or
 * fp32), are 16-byte aligned, and n is a multiple of 8 (bf16) or 4 (fp32).
 */
int prl_swiglu_fwd(const void* gate, const void* up, void* out, int dtype, int64_t n, void* stream);
int prl_swiglu_bwd(const void* gate, const void* up, const void* gout, void* ggate, void* gup, int dtype, int64_t n,
                   void* stream);
/* The same over `rows` rows of `cols` elements with row pitches (in elements): gate and up as column blocks of ONE
 * wider matrix (the output of one product over the concatenated gate / up weights), and their gradients likewise. */
int prl_swiglu_rows_fwd(const void* gate, const void* up, int64_t in_pitch, void* out, int64_t out_pitch, int dtype,
                        int64_t rows, int64_t cols, void* stream);
int prl_swiglu_rows_bwd(const void* gate, const void* up, int64_t in_pitch, const void* gout, int64_t gout_pitch,
                        void* ggate, void* gup, int64_t grad_pitch, int dtype, int64_t rows, int64_t cols, void* stream);

/*
 * Gradient accumulation over micro-batches into ONE flat fp32 buffer, one launch for all parameters:
 *   acc[dst_offset[k] + i] += scale[0] * src[k][i]      for every tensor k, i < numel[k].
 * Stands where FSDP2 sums every micro-batch's gradients in the fp32 reduce dtype
 * (src/prime_rl/trainer/model.py:102-104; the micro-batch loop is src/prime_rl/trainer/rl/train.py:271-310).
 * All arrays are DEVICE memory.  `src` holds one pointer per tensor (NULL: no gradient this time);
 * chunk c covers elements [chunk_start[c], chunk_start[c] + prl_accumulate_chunk_elems()) of tensor
 * chunk_tensor[c].  `scale` may be NULL (= 1); it is read on the device, so a captured launch can be
 * replayed under another scale.  Chunks must not overlap in `acc`.
 */
int64_t prl_accumulate_chunk_elems(void);
int prl_accumulate(const void* const* src, int src_dtype, const int64_t* dst_offset, const int64_t* numel,
                   const int32_t* chunk_tensor, const int64_t* chunk_start, int64_t n_chunks, const float* scale,
                   float* acc, void* stream);
/* The first pass of a step: acc = scale * src for every tensor (zeros where src[i] is NULL) instead of
 * acc += ...; the sums then need no clearing between steps and their old contents are not read. */
int prl_accumulate_first(const void* const* src, int src_dtype, const int64_t* dst_offset, const int64_t* numel,
                   const int32_t* chunk_tensor, const int64_t* chunk_start, int64_t n_chunks, const float* scale,
                   float* acc, void* stream);
/* Either of the two (`first` != 0: prl_accumulate_first), and chunk_squares[c] = the sum of the squares of chunk c's
 * new sums, for every chunk whose tensor has a gradient in this call (and 0 for one without, when `first`); other
 * entries stay as they are.  When the first call of a step is a `first` one over all chunks and every later call
 * of the step goes through here with the same chunk table, the entries add up to the squared norm of `acc`: the
 * gradient norm that clipping needs (torch.nn.utils.clip_grad_norm_, src/prime_rl/trainer/rl/train.py:336-338)
 * without another pass over the sums.  `chunk_squares`: n_chunks floats of device memory. */
int prl_accumulate_squares(int first, const void* const* src, int src_dtype, const int64_t* dst_offset,
                           const int64_t* numel, const int32_t* chunk_tensor, const int64_t* chunk_start,
                           int64_t n_chunks, const float* scale, float* acc, float* chunk_squares, void* stream);

/*
 * out[i] = part[0][i] + part[1][i] + ... + part[nparts - 1][i] for i < n, added in that order in fp32 and rounded
 * once to bf16; the parts (fp32 or bf16) lie `part_stride` elements apart.  Last step of a weight gradient that
 * was computed as a few batched products over ranges of the tokens of a long packed row (the reference leaves
 * dW = gy^T x to one cuBLAS call inside autograd, src/prime_rl/trainer/rl/train.py:310).  n and part_stride
 * multiples of 8, pointers 16-byte aligned.
 */
int prl_sum_parts(const void* parts, int parts_dtype, int64_t part_stride, int32_t nparts, int64_t n, void* out,
                  void* stream);

/*
 * Causal self-attention over a packed row of SHORT samples (each at most 128 tokens, head dimension 128,
 * bf16), hand-written for MFMA.  Stands where the reference's attention runs through flash-attn's
 * variable-length kernel (`attn_implementation = "flash_attention_2"` with position ids that restart per
 * sample, src/prime_rl/trainer/model.py:58-60, :251-254).
 * Every tensor is [token][head][128] with the last dimension contiguous; `*_strides` = {token stride, head
 * stride} in elements (HOST arrays).  `cu_seqlens` (DEVICE, int32 [n_samples + 1]) holds the first token of
 * every sample and T at the end.  `lse2` (fp32 [q_heads][T]) receives the base-2 log-sum-exp of the scaled
 * scores; the backward reads it, needs `delta` (fp32 [q_heads][T]) as scratch (its contents afterwards are
 * unspecified), and writes dq, dk, dv (tokens
 * outside every sample are left untouched).  Samples longer than 128 tokens are the caller's to exclude
 * (prl_attn_short_supported); a table that breaks the rule is clamped, never followed out of bounds.
 */
int prl_attn_short_supported(int dtype, int64_t head_dim, int64_t max_len, int64_t q_heads, int64_t kv_heads);
int prl_attn_short_fwd(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides,
                       const void* v, const int64_t* v_strides, void* out, const int64_t* out_strides, float* lse2,
                       const int32_t* cu_seqlens, int32_t n_samples, int64_t T, int32_t q_heads, int32_t kv_heads,
                       float scale, void* stream);
int prl_attn_short_bwd(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides,
                       const void* v, const int64_t* v_strides, const void* out, const int64_t* out_strides,
                       const void* dout, const int64_t* dout_strides, const float* lse2, float* delta, void* dq,
                       const int64_t* dq_strides, void* dk, const int64_t* dk_strides, void* dv,
                       const int64_t* dv_strides, const int32_t* cu_seqlens, int32_t n_samples, int64_t T,
                       int32_t q_heads, int32_t kv_heads, float scale, void* stream);

/*
 * The same attention for samples of ANY length (head dimension 128, bf16), tiled flash-style over 128 x 128
 * blocks of the score matrix with a running soft-max; stands where the reference's recipes with long samples
 * (wordle: 4096 tokens, hendrycks_math: 2048, deepscaler: 8192) run flash-attn's variable-length kernel
 * (src/prime_rl/trainer/model.py:58-60, :251-254).  Arguments as prl_attn_short_*; `max_len` is the longest
 * sample of the table (HOST value; it sizes the grid: a longer sample than stated would be cut off there, never
 * followed out of bounds).
 */
int prl_attn_long_supported(int dtype, int64_t head_dim, int64_t max_len, int64_t q_heads, int64_t kv_heads);
int prl_attn_long_fwd(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides,
                      const void* v, const int64_t* v_strides, void* out, const int64_t* out_strides, float* lse2,
                      const int32_t* cu_seqlens, int32_t n_samples, int64_t max_len, int64_t T, int32_t q_heads,
                      int32_t kv_heads, float scale, void* stream);
int prl_attn_long_bwd(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides,
                      const void* v, const int64_t* v_strides, const void* out, const int64_t* out_strides,
                      const void* dout, const int64_t* dout_strides, const float* lse2, float* delta, void* dq,
                      const int64_t* dq_strides, void* dk, const int64_t* dk_strides, void* dv,
                      const int64_t* dv_strides, const int32_t* cu_seqlens, int32_t n_samples, int64_t max_len,
                      int64_t T, int32_t q_heads, int32_t kv_heads, float scale, void* stream);

/*
 * AdamW over one flat fp32 shard, fused with what surrounds it at the end of a step: param, exp_avg, exp_avg_sq
 * and grad are fp32 [n] (n a multiple of 4); every gradient is first multiplied by grad_scale[0] (DEVICE
 * memory, NULL = 1: the clipping factor of clip_grad_norm_, never copied to the host); `compute_copy`
 * (bf16 or fp32 [n], may be NULL) receives the new weights for the next forward; zero_grad != 0 clears grad.
 * The update rule is torch.optim.AdamW's, which the reference runs after clip_grad_norm_
 * (src/prime_rl/trainer/optim.py:19-25, src/prime_rl/trainer/rl/train.py:336-342); `step` counts from 1.
 */
int prl_adamw_flat(float* param, float* grad, float* exp_avg, float* exp_avg_sq, void* compute_copy, int copy_dtype,
                   int64_t n, const float* grad_scale, int zero_grad, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int64_t step, void* stream);

/*
 * Query / key RMSNorm over every head followed by the rotary embedding, reading the query and key blocks of a
 * stacked projection output y [T][(q_heads + k_heads [+ v_heads]) * 128] (row pitch y_pitch, in elements) and
 * writing q_out [T][q_heads][128] and k_out [T][k_heads][128] (contiguous) plus rstd [T][q_heads + k_heads].
 * The backward takes the gradients of q_out and k_out ({token stride, head stride} in elements, HOST arrays)
 * and writes the gradient with respect to y's query and key blocks INTO gy (same layout as y, pitch gy_pitch),
 * and gw [256] = the two norm weights' gradients (query's, then key's), in y's dtype; gq and gk may be the query
 * and key blocks of gy itself (the kernel then works in place).  cos / sin are
 * [T][128] with a token stride.  Head dimension 128 only.  These are the steps of the transformers attention
 * block between its projections and its attention call; the reference's switch for fused block kernels is
 * `model.impl = "liger_kernel"` (src/prime_rl/trainer/model.py:74-75).
 */
int prl_qk_norm_rope_supported(int dtype, int64_t head_dim);
size_t prl_qk_norm_rope_bwd_workspace_bytes(int dtype, int64_t T, int64_t q_heads, int64_t k_heads);
int prl_qk_norm_rope_fwd(const void* y, int64_t y_pitch, const void* wq, const void* wk, const void* cos_,
                         const void* sin_, int64_t cs_token_stride, void* q_out, void* k_out, float* rstd, int dtype,
                         int64_t T, int64_t q_heads, int64_t k_heads, float eps, void* stream);
int prl_qk_norm_rope_bwd(const void* gq, const int64_t* gq_strides, const void* gk, const int64_t* gk_strides,
                         const void* y, int64_t y_pitch, const void* wq, const void* wk, const float* rstd,
                         const void* cos_, const void* sin_, int64_t cs_token_stride, void* gy, int64_t gy_pitch,
                         void* gw, void* workspace, size_t workspace_bytes, int dtype, int64_t T, int64_t q_heads,
                         int64_t k_heads, void* stream);

/*
 * The three matrix products of the LM head, hand-written for the MFMA units (bf16 in, fp32 accumulation, bf16
 * out).  They replace the final projection of the model's forward (src/prime_rl/trainer/model.py:251-254, whose
 * logits src/prime_rl/trainer/rl/train.py:285 picks up) and the two products autograd runs for it behind
 * loss.backward() (src/prime_rl/trainer/rl/train.py:310), which the reference leaves to its BLAS library:
 *     prl_lm_head_fwd         logits[T, V]  = hidden[T, H] . weight[V, H]^T
 *     prl_lm_head_bwd_input   dhidden[T, H] = dlogits[T, V] . weight[V, H]
 *     prl_lm_head_bwd_weight  dweight[V, H] = dlogits[T, V]^T . hidden[T, H]
 * Every matrix is row-major bf16 with `*_stride` ELEMENTS between rows (multiples of 8; outputs multiples of 4),
 * 16-byte aligned.  T, H and V must be multiples of 64 (prl_lm_head_supported); rows or columns beyond a multiple
 * of 256 are handled inside the kernels.  Sums run in fp32 in the order of the K index within a 64-wide step and
 * over the steps front to back; a product whose K is split over workgroups (dhidden with few rows) adds its fp32
 * partial results in split order, so results are reproducible run to run.
 * workspace: prl_lm_head_workspace_bytes(T, H, V) bytes of device scratch, 16-byte aligned, needed by the two
 * backward products (fp32 partial results of a K split; for the weight gradient also the narrower of hidden and
 * dlogits transposed); the forward needs none.  A product that is split over K (few output tiles, a long sum)
 * needs its output contiguous (stride = row length).
 * The same entry points serve any bias-free linear layer y = x W^T (T rows of x, H = in, V = out): the decoder
 * blocks' projections run on them where they were measured ahead of the BLAS library.
 */
int prl_lm_head_supported(int dtype, int64_t T, int64_t H, int64_t V);
size_t prl_lm_head_workspace_bytes(int64_t T, int64_t H, int64_t V);
int prl_lm_head_fwd(const void* hidden, int64_t hidden_stride, const void* weight, int64_t weight_stride,
                    void* logits, int64_t logits_stride, int64_t T, int64_t H, int64_t V, void* stream);
int prl_lm_head_bwd_input(const void* dlogits, int64_t dlogits_stride, const void* weight, int64_t weight_stride,
                          void* dhidden, int64_t dhidden_stride, int64_t T, int64_t H, int64_t V, void* workspace,
                          size_t workspace_bytes, void* stream);
int prl_lm_head_bwd_weight(const void* dlogits, int64_t dlogits_stride, const void* hidden, int64_t hidden_stride,
                           void* dweight, int64_t dweight_stride, int64_t T, int64_t H, int64_t V, void* workspace,
                           size_t workspace_bytes, void* stream);

/*
 * Number of nodes of a captured HIP graph (`graph` is a hipGraph_t, child graphs included) and how many of them
 * are memset nodes.  No counterpart in the reference, which never captures graphs
 * (src/prime_rl/trainer/model.py:219-230 hands the blocks to a tracing compiler instead): the whole-model
 * trainer calls this between capture and instantiation and refuses to replay a pass that holds memset nodes,
 * which library calls record under capture.  Host-only: launches nothing.
 */
int prl_graph_node_counts(void* graph, int64_t* n_nodes, int64_t* n_memset_nodes);

/*
 * Group-relative advantages: B rewards in consecutive groups of G.  Replaces
 * compute_advantages (src/prime_rl/orchestrator/advantage.py:71-103) for the
 * four kinds of its REGISTRY (:63-68).  `lengths` (int64 [B], completion
 * lengths) is read by PRL_ADV_OPO only and may be NULL otherwise.
 */
int prl_group_advantage(const float* rewards, const int64_t* lengths, int64_t B, int64_t G, int kind,
                        float* advantages, void* stream);

/*
 * Statistics of logged vectors: mean, median, std (unbiased), min, max, as the
 * trainer logs them per key (src/prime_rl/trainer/utils.py:235-256: cat, gather,
 * mean / median / std / min / max), computed where the data lives.
 *
 * A statistic ("key") is the union of any number of device pieces, typically
 * one per micro-batch.  A piece may carry a 0/1 selection mask and may ask for
 * exp() of its values (probabilities from log-probabilities,
 * src/prime_rl/trainer/rl/train.py:313-314).  The median is the lower middle
 * value, as torch.median returns it, found by an exact radix select over four
 * passes.  Call order:
 *     prl_stats_begin; prl_stats_pass level 0, 1, 2, 3; prl_stats_finish
 * Between passes a data-parallel job may all-reduce the scratch across ranks to
 * obtain the statistics of all ranks' values without gathering them: after
 * level 0 sum the regions cnt, mom and the level-0 histogram and take the
 * maximum of ext; after levels 1..3 sum that level's histogram (region offsets
 * from prl_stats_layout; the histogram of level L is the L-th quarter of `hist`).
 */
#define PRL_STAT_IDENTITY 0
#define PRL_STAT_EXP 1

typedef struct prl_stat_piece {
  const float* x;      /* device, fp32 [n] */
  const uint8_t* mask; /* device, 0/1 per entry, or NULL = all selected */
  int64_t n;
  int32_t key;         /* which statistic the piece belongs to: 0 <= key < n_keys */
  int32_t op;          /* PRL_STAT_IDENTITY or PRL_STAT_EXP */
} prl_stat_piece;

size_t prl_stats_workspace_bytes(int n_keys);
/* offsets5 (host): byte offsets of the regions cnt (int64 [K][2]), mom (double [K][2]),
 * ext (int32 [K][2]), hist (int32 [4][K][256]) and the total size */
int prl_stats_layout(int n_keys, int64_t* offsets5);
int prl_stats_begin(void* workspace, int n_keys, void* stream);
/* pieces: HOST array of n_pieces descriptors (copied into the launch) */
int prl_stats_pass(int level, const prl_stat_piece* pieces, int n_pieces, int n_keys, void* workspace, void* stream);
/* out: fp32 [n_keys][5] = {mean, median, std, min, max}, NaN-filled for a key with nothing selected or with
 * a NaN among the selected; counts: int64 [n_keys] selected entries (may be NULL) */
int prl_stats_finish(void* workspace, int n_keys, float* out, int64_t* counts, void* stream);

/* One vector, one call: the five statistics of the selected entries of x
 * (src/prime_rl/trainer/utils.py:247-251). */
size_t prl_masked_stats_workspace_bytes(void);
int prl_masked_stats(const float* x, const uint8_t* mask, int64_t n, float* out5, int64_t* count, void* workspace,
                     size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PRL_HIP_H */
