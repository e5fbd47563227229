// Ratio-clip policy loss over the samples of a packed row, with its gradient
// with respect to the per-token log-probabilities in closed form.
//
// The reference loops over the samples in Python and issues about ten small
// torch kernels per sample (src/prime_rl/trainer/rl/loss.py:89-115); autograd
// later replays them backwards.  Here one workgroup owns one sample, all samples
// run in one launch, and the gradient is written in the same pass.  The loss is
// summed per sample first and over the samples afterwards, in a fixed order, so
// the result does not change from run to run.
#include "prl_common.h"

namespace prl {

constexpr int kLossThreads = 256;
constexpr int kLossWaves = kLossThreads / kWave;
constexpr int kLossMaxBlocks = 2048;

// torch.clamp(x, max=c): NaN stays NaN
__device__ __forceinline__ float clamp_max(float x, float c) { return x > c ? c : x; }

__global__ __launch_bounds__(kLossThreads) void k_grpo_loss(
    const float* __restrict__ lp, const float* __restrict__ old, const float* __restrict__ adv,
    const uint8_t* __restrict__ mask, const int64_t* __restrict__ seg_start, const int32_t* __restrict__ n_seg_p,
    int64_t max_seg, int ratio_type, int length_norm, float clip, float inv_scale, float* __restrict__ ratio,
    float* __restrict__ clipped, float* __restrict__ is_clipped, float* __restrict__ g,
    float* __restrict__ seg_loss) {
  __shared__ float red[kLossWaves];
  const int tid = threadIdx.x;
  int64_t n_seg = n_seg_p[0];
  if (n_seg > max_seg) n_seg = max_seg;
  for (int64_t s = blockIdx.x; s < n_seg; s += gridDim.x) {
    const int64_t a = seg_start[s], b = seg_start[s + 1];
    if (ratio_type == PRL_RATIO_TOKEN) {
      // one ratio per token (src/prime_rl/trainer/rl/loss.py:90, 98-104)
      float acc = 0.0f;
      for (int64_t i = a + tid; i < b; i += kLossThreads) {
        const float r = expf(lp[i] - old[i]);
        const float c = clamp_max(r, clip);
        const float ad = adv[i];
        const bool m = mask[i] != 0;
        if (ratio) ratio[i] = r;
        if (clipped) clipped[i] = c;
        if (is_clipped) is_clipped[i] = r > clip ? 1.0f : 0.0f;
        if (m) acc += -c * ad;
        if (g) {
          // d/dlp of -min(r, clip) * adv / scale; the clamp passes the gradient
          // at equality, and a ratio of +inf gives 0 * inf = NaN as autograd does
          const float through = (r <= clip) ? -(inv_scale * ad) : 0.0f;
          g[i] = m ? through * r : 0.0f;
        }
      }
      const float tot = block_sum<kLossWaves>(acc, red);
      if (tid == 0) seg_loss[s] = tot;
    } else {
      // one ratio per sample (src/prime_rl/trainer/rl/loss.py:92-96, 107-108)
      float s_lr = 0.0f, s_adv = 0.0f, s_cnt = 0.0f;
      for (int64_t i = a + tid; i < b; i += kLossThreads) {
        if (mask[i] != 0) {
          s_lr += lp[i] - old[i];
          s_adv += adv[i];
          s_cnt += 1.0f;
        }
      }
      s_lr = block_sum<kLossWaves>(s_lr, red);
      s_adv = block_sum<kLossWaves>(s_adv, red);
      s_cnt = block_sum<kLossWaves>(s_cnt, red);
      const float n = fmaxf(s_cnt, 1.0f);
      const float av = length_norm ? s_lr / n : s_lr;
      const float bv = clamp_max(av, 10.0f);
      const float r = expf(bv);
      const float c = clamp_max(r, clip);
      if (tid == 0) {
        seg_loss[s] = (-c * s_adv) / n;
        if (ratio) ratio[s] = r;
        if (clipped) clipped[s] = c;
        if (is_clipped) is_clipped[s] = r > clip ? 1.0f : 0.0f;
      }
      if (g) {
        const float g_c = -((inv_scale / n) * s_adv);
        const float g_r = (r <= clip) ? g_c : 0.0f;
        const float g_b = g_r * r;
        const float g_a = (av <= 10.0f) ? g_b : 0.0f;
        const float g_s = length_norm ? g_a / n : g_a;
        for (int64_t i = a + tid; i < b; i += kLossThreads) g[i] = mask[i] != 0 ? g_s : 0.0f;
      }
    }
  }
}

// total = sum of the per-sample terms in index order; loss = total / max(scale, 1)
__global__ __launch_bounds__(kLossThreads) void k_grpo_loss_finish(const float* __restrict__ seg_loss,
                                                                    const int32_t* __restrict__ n_seg_p,
                                                                    int64_t max_seg, float scale,
                                                                    float* __restrict__ loss) {
  __shared__ float red[kLossWaves];
  int64_t n_seg = n_seg_p[0];
  if (n_seg > max_seg) n_seg = max_seg;
  float acc = 0.0f;
  for (int64_t s = threadIdx.x; s < n_seg; s += kLossThreads) acc += seg_loss[s];
  const float tot = block_sum<kLossWaves>(acc, red);
  if (threadIdx.x == 0) loss[0] = tot / scale;
}

}  // namespace prl

using namespace prl;

extern "C" int prl_grpo_loss(const float* lp, const float* old, const float* adv, const uint8_t* mask,
                             const int64_t* seg_start, const int32_t* n_seg, int64_t max_seg, int64_t T,
                             int ratio_type, int length_norm, float clip_ratio, float loss_scale, float* loss,
                             float* ratio, float* clipped, float* is_clipped, float* g, float* seg_loss,
                             void* stream) {
  clear_error();
  PRL_REQUIRE(T >= 0 && max_seg >= 0, "grpo_loss: T and max_seg must be >= 0");
  PRL_REQUIRE(ratio_type == PRL_RATIO_TOKEN || ratio_type == PRL_RATIO_SEQUENCE, "grpo_loss: bad ratio_type %d",
              ratio_type);
  PRL_REQUIRE(loss && seg_start && n_seg, "grpo_loss: loss, seg_start and n_seg must not be NULL");
  PRL_REQUIRE(max_seg == 0 || seg_loss, "grpo_loss: seg_loss scratch must not be NULL");
  PRL_REQUIRE(T == 0 || (lp && old && adv && mask), "grpo_loss: lp, old, adv, mask must not be NULL");
  const float scale = loss_scale > 1.0f ? loss_scale : 1.0f;  // max(loss_scale, 1)
  hipStream_t st = (hipStream_t)stream;
  if (T > 0 && max_seg > 0) {
    const unsigned nblk = (unsigned)(max_seg < kLossMaxBlocks ? max_seg : kLossMaxBlocks);
    hipLaunchKernelGGL(k_grpo_loss, dim3(nblk), dim3(kLossThreads), 0, st, lp, old, adv, mask, seg_start, n_seg,
                       max_seg, ratio_type, length_norm ? 1 : 0, clip_ratio, 1.0f / scale, ratio, clipped, is_clipped,
                       g, seg_loss);
    PRL_CHECK_LAUNCH("grpo_loss");
  }
  hipLaunchKernelGGL(k_grpo_loss_finish, dim3(1), dim3(kLossThreads), 0, st, seg_loss, n_seg,
                     (T > 0 ? max_seg : (int64_t)0), scale, loss);
  PRL_CHECK_LAUNCH("grpo_loss(finish)");
  return PRL_OK;
}
