// One call for the forward half of a micro-batch: sample boundaries, per-token
// log-probability and entropy, ratio-clip loss and dL/dlogprob.  It only chains
// the three entry points it is made of; the point is one host call (and one
// argument check) instead of three on a path where the host, not the GPU, can
// become the limit.
#include "prl_common.h"

namespace prl {
int logprob_fwd_impl(const void* logits, int logits_dtype, int64_t T, int64_t V, int64_t row_stride,
                     const int64_t* ids, float temperature, int shift, const uint8_t* skip, float* logprob,
                     float* entropy, float* lse, void* workspace, size_t workspace_bytes, void* stream);
}

extern "C" int prl_policy_forward(const void* logits, int logits_dtype, int64_t T, int64_t V, int64_t row_stride,
                                  const int64_t* ids, const int64_t* pos, const float* adv, const float* old,
                                  const uint8_t* mask, int skip_masked, float temperature, int ratio_type,
                                  int length_norm, float clip_ratio, float loss_scale, float* logprob, float* entropy, float* lse,
                                  float* loss, float* ratio, float* clipped, float* is_clipped, float* g,
                                  float* seg_loss, int32_t* seg_id, int64_t* seg_start, int32_t* n_seg,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  PRL_REQUIRE(logprob, "policy_forward: logprob must not be NULL");
  int rc = prl::logprob_fwd_impl(logits, logits_dtype, T, V, row_stride, ids, temperature, 1,
                                 skip_masked ? mask : nullptr, logprob, entropy, lse, workspace, workspace_bytes, stream);
  if (rc != PRL_OK) return rc;
  rc = prl_segments_from_position_ids(pos, T, seg_id, seg_start, n_seg, stream);
  if (rc != PRL_OK) return rc;
  const int64_t max_seg = T > 0 ? T / 2 + 1 : 0;  // a sample needs two tokens to open
  return prl_grpo_loss(logprob, old, adv, mask, seg_start, n_seg, max_seg, T, ratio_type, length_norm, clip_ratio,
                       loss_scale, loss, ratio, clipped, is_clipped, g, seg_loss, stream);
}
