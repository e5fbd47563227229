// Group-relative advantages: B rewards in consecutive groups of G, one wave per
// group.  The four kinds are those of the reference's registry
// (src/prime_rl/orchestrator/advantage.py:9-68).
#include "prl_common.h"

namespace prl {

constexpr int kAdvThreads = 256;

__global__ __launch_bounds__(kAdvThreads) void k_group_advantage(const float* __restrict__ rewards,
                                                                 const int64_t* __restrict__ lengths, int64_t B,
                                                                 int64_t G, int kind, float* __restrict__ out) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t waves_per_block = kAdvThreads / kWave;
  const int64_t n_groups = B / G;
  for (int64_t grp = blockIdx.x * waves_per_block + threadIdx.x / kWave; grp < n_groups;
       grp += (int64_t)gridDim.x * waves_per_block) {
    const float* r = rewards + grp * G;
    float* o = out + grp * G;
    float num = 0.0f, den = 0.0f;
    if (kind == PRL_ADV_OPO) {
      // baseline = sum(r * len) / sum(len), lengths cast to fp32 (advantage.py:55-57)
      const int64_t* len = lengths + grp * G;
      for (int64_t i = lane; i < G; i += kWave) {
        const float w = (float)len[i];
        num += r[i] * w;
        den += w;
      }
      num = wave_sum(num);
      den = wave_sum(den);
    } else {
      for (int64_t i = lane; i < G; i += kWave) num += r[i];
      num = wave_sum(num);
      den = (float)G;
    }
    const float base = num / den;
    for (int64_t i = lane; i < G; i += kWave) {
      float a = r[i] - base;
      if (kind == PRL_ADV_DRGRPO_NEGCLIPPED) a = fmaxf(a, 0.0f);                    // advantage.py:32
      if (kind == PRL_ADV_RLOO) a = (a * (float)G) / (float)(G - 1);                // advantage.py:41-43; G = 1 gives NaN there too
      o[i] = a;
    }
  }
}

}  // namespace prl

using namespace prl;

extern "C" int prl_group_advantage(const float* rewards, const int64_t* lengths, int64_t B, int64_t G, int kind,
                                   float* advantages, void* stream) {
  clear_error();
  PRL_REQUIRE(B >= 0 && G >= 1, "group_advantage: need B >= 0 and G >= 1");
  PRL_REQUIRE(B % G == 0, "group_advantage: B=%lld is not a multiple of G=%lld", (long long)B, (long long)G);
  PRL_REQUIRE(kind >= PRL_ADV_DRGRPO && kind <= PRL_ADV_OPO, "group_advantage: unknown kind %d", kind);
  if (B == 0) return PRL_OK;
  PRL_REQUIRE(rewards && advantages, "group_advantage: rewards and advantages must not be NULL");
  PRL_REQUIRE(kind != PRL_ADV_OPO || lengths, "group_advantage: OPO needs the completion lengths");
  const int64_t n_groups = B / G;
  int64_t nblk = (n_groups + 3) / 4;
  if (nblk > 1024) nblk = 1024;
  hipLaunchKernelGGL(k_group_advantage, dim3((unsigned)nblk), dim3(kAdvThreads), 0, (hipStream_t)stream, rewards,
                     lengths, B, G, kind, advantages);
  PRL_CHECK_LAUNCH("group_advantage");
  return PRL_OK;
}
