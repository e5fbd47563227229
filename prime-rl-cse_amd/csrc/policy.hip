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


namespace prl {

// ---------------------------------------------------------------------------
// Rows of up to kSmallMaxTokens tokens: sample boundaries, loss, its gradient and the final
// sum in ONE launch of one workgroup.  The three separate kernels cost about 4 us each plus
// three launch boundaries, which is several percent of a micro-batch whose two big kernels
// take ~0.35 ms; fused, the whole bookkeeping is one ~6 us launch.
constexpr int kSmallThreads = 1024;
constexpr int kSmallWaves = kSmallThreads / kWave;
constexpr int64_t kSmallMaxTokens = 16384;
constexpr int64_t kSmallMaxSeg = kSmallMaxTokens / 2 + 1;

__device__ __forceinline__ float clamp_max_(float x, float c) { return x > c ? c : x; }

__global__ __launch_bounds__(kSmallThreads) void k_policy_small(
    const int64_t* __restrict__ pos, const float* __restrict__ lp, const float* __restrict__ old,
    const float* __restrict__ adv, const uint8_t* __restrict__ mask, int64_t T, int ratio_type, int length_norm,
    float clip, float scale, float* __restrict__ loss, float* __restrict__ ratio, float* __restrict__ clipped,
    float* __restrict__ is_clipped, float* __restrict__ g, float* __restrict__ seg_loss_out,
    int32_t* __restrict__ seg_id, int64_t* __restrict__ seg_start, int32_t* __restrict__ n_seg) {
  __shared__ int wave_total[kSmallWaves];
  __shared__ float red[kSmallWaves];
  __shared__ int starts[kSmallMaxSeg + 1];  // sample start offsets (T < 2^31 here)
  __shared__ float seg_term[kSmallMaxSeg];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
  const float inv_scale = 1.0f / scale;

  // 1. sample boundaries (the rule of src/prime_rl/trainer/utils.py:55-60), 1024 tokens per step
  int carry = 0;
  if (tid == 0) {
    starts[0] = 0;
    seg_start[0] = 0;
  }
  for (int64_t base = 0; base < T; base += kSmallThreads) {
    const int64_t i = base + tid;
    const bool b = i >= 1 && i + 1 < T && pos[i] == 0 && pos[i - 1] != 0 && pos[i + 1] == 1;
    const unsigned long long bal = __ballot(b);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_total[wid] = __popcll(bal);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kSmallWaves; ++w) {
      const int c = wave_total[w];
      off += (w < wid) ? c : 0;
      total += c;
    }
    const int sid = carry + off + before + (b ? 1 : 0);
    if (i < T) {
      if (seg_id) seg_id[i] = sid;
      if (b) {
        starts[sid] = (int)i;
        seg_start[sid] = i;
      }
    }
    carry += total;
    __syncthreads();
  }
  const int n = T > 0 ? carry + 1 : 0;
  if (tid == 0) {
    n_seg[0] = n;
    seg_start[n] = T;
    starts[n] = (int)T;
  }
  __syncthreads();

  if (ratio_type == PRL_RATIO_TOKEN) {
    // 2a. one ratio per token (src/prime_rl/trainer/rl/loss.py:90, 98-104); the loss is the sum over all
    //     tokens in a fixed order, which does not depend on where the samples begin
    float acc = 0.0f;
    for (int64_t i = tid; i < T; i += kSmallThreads) {
      const float r = expf(lp[i] - old[i]);
      const float c = clamp_max_(r, clip);
      const float ad = adv[i];
      const bool m = mask[i] != 0;
      if (ratio) ratio[i] = r;
      if (clipped) clipped[i] = c;
      if (is_clipped) is_clipped[i] = r > clip ? 1.0f : 0.0f;
      if (m) acc += -c * ad;
      if (g) {
        const float through = (r <= clip) ? -(inv_scale * ad) : 0.0f;
        g[i] = m ? through * r : 0.0f;
      }
    }
    const float tot = block_sum<kSmallWaves>(acc, red);
    if (tid == 0) loss[0] = tot / scale;
    return;
  }

  // 2b. one ratio per sample (src/prime_rl/trainer/rl/loss.py:92-96, 107-108): a wave per sample
  for (int s = wid; s < n; s += kSmallWaves) {
    const int a = starts[s], b = starts[s + 1];
    float s_lr = 0.0f, s_adv = 0.0f, s_cnt = 0.0f;
    for (int i = a + lane; i < b; i += kWave) {
      if (mask[i] != 0) {
        s_lr += lp[i] - old[i];
        s_adv += adv[i];
        s_cnt += 1.0f;
      }
    }
    s_lr = wave_sum(s_lr);
    s_adv = wave_sum(s_adv);
    s_cnt = wave_sum(s_cnt);
    const float cnt = fmaxf(s_cnt, 1.0f);
    const float av = length_norm ? s_lr / cnt : s_lr;
    const float r = expf(clamp_max_(av, 10.0f));
    const float c = clamp_max_(r, clip);
    if (lane == 0) {
      const float term = (-c * s_adv) / cnt;
      seg_term[s] = term;
      if (seg_loss_out) seg_loss_out[s] = term;
      if (ratio) ratio[s] = r;
      if (clipped) clipped[s] = c;
      if (is_clipped) is_clipped[s] = r > clip ? 1.0f : 0.0f;
    }
    if (g) {
      const float g_c = -((inv_scale / cnt) * s_adv);
      const float g_r = (r <= clip) ? g_c : 0.0f;
      const float g_b = g_r * r;
      const float g_a = (av <= 10.0f) ? g_b : 0.0f;
      const float g_s = length_norm ? g_a / cnt : g_a;
      for (int i = a + lane; i < b; i += kWave) g[i] = mask[i] != 0 ? g_s : 0.0f;
    }
  }
  __syncthreads();
  float acc = 0.0f;
  for (int s = tid; s < n; s += kSmallThreads) acc += seg_term[s];
  const float tot = block_sum<kSmallWaves>(acc, red);
  if (tid == 0) loss[0] = tot / scale;
}

}  // namespace prl

extern "C" int prl_policy_tail(const int64_t* pos, const float* logprob, const float* old, const float* adv,
                               const uint8_t* mask, int64_t T, int ratio_type, int length_norm, float clip_ratio,
                               float loss_scale, float* loss, float* ratio, float* clipped, float* is_clipped, float* g,
                               float* seg_loss, int32_t* seg_id, int64_t* seg_start, int32_t* n_seg, void* stream) {
  prl::clear_error();
  PRL_REQUIRE(T >= 0, "policy_tail: T must be >= 0");
  PRL_REQUIRE(ratio_type == PRL_RATIO_TOKEN || ratio_type == PRL_RATIO_SEQUENCE, "policy_tail: bad ratio_type %d",
              ratio_type);
  if (T > 0 && T <= prl::kSmallMaxTokens) {
    PRL_REQUIRE(pos && logprob && adv && old && mask && loss && seg_start && n_seg,
                "policy_tail: pos, logprob, adv, old, mask, loss, seg_start and n_seg must not be NULL");
    const float scale = loss_scale > 1.0f ? loss_scale : 1.0f;  // max(loss_scale, 1)
    hipLaunchKernelGGL(prl::k_policy_small, dim3(1), dim3(prl::kSmallThreads), 0, (hipStream_t)stream, pos, logprob,
                       old, adv, mask, T, ratio_type, length_norm ? 1 : 0, clip_ratio, scale, loss, ratio, clipped,
                       is_clipped, g, seg_loss, seg_id, seg_start, n_seg);
    PRL_CHECK_LAUNCH("policy_tail(small)");
    return PRL_OK;
  }
  int rc = prl_segments_from_position_ids(pos, T, seg_id, seg_start, n_seg, stream);
  if (rc != PRL_OK) return rc;
  const int64_t max_seg = T > 0 ? T / 2 + 1 : 0;  // a sample needs two tokens to open
  return prl_grpo_loss(logprob, old, adv, mask, seg_start, n_seg, max_seg, T, ratio_type, length_norm, clip_ratio,
                       loss_scale, loss, ratio, clipped, is_clipped, g, seg_loss, stream);
}

extern "C" int prl_policy_forward(const void* logits, int logits_dtype, int64_t T, int64_t V, int64_t row_stride,
                                  const int64_t* ids, const int64_t* pos, const float* adv, const float* old,
                                  const uint8_t* mask, int skip_masked, float temperature, int ratio_type,
                                  int length_norm, float clip_ratio, float loss_scale, float* logprob, float* entropy,
                                  float* lse, float* loss, float* ratio, float* clipped, float* is_clipped, float* g,
                                  float* seg_loss, int32_t* seg_id, int64_t* seg_start, int32_t* n_seg,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  PRL_REQUIRE(logprob, "policy_forward: logprob must not be NULL");
  int rc = prl::logprob_fwd_impl(logits, logits_dtype, T, V, row_stride, ids, temperature, 1,
                                 skip_masked ? mask : nullptr, logprob, entropy, lse, workspace, workspace_bytes, stream);
  if (rc != PRL_OK) return rc;
  return prl_policy_tail(pos, logprob, old, adv, mask, T, ratio_type, length_norm, clip_ratio, loss_scale, loss, ratio,
                         clipped, is_clipped, g, seg_loss, seg_id, seg_start, n_seg, stream);
}
