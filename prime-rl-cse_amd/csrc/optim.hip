// AdamW over ONE flat fp32 shard, with everything the end of a step does to that shard in the same pass:
// the gradient-clipping factor (read from device memory: it comes from a norm that was never copied to the
// host), the update itself, the bf16 copy of the new weights for the next forward, and the clearing of the
// gradient accumulator.  The reference runs torch.optim.AdamW over FSDP2's fp32 shards after
// clip_grad_norm_ (src/prime_rl/trainer/optim.py:19-25, src/prime_rl/trainer/rl/train.py:336-342); as separate
// passes that is the norm, the scaling, the optimizer, the cast and the zero fill: 30 GB of traffic for a 0.6B
// model against 20 GB here.  Same update rule as torch's (decoupled weight decay, bias corrections, eps added
// to sqrt(v) / sqrt(1 - beta2^t)).
#include "prl_common.h"

namespace prl {

struct AdamArgs {
  float lr, beta1, beta2, eps, weight_decay, step_size, inv_bc2_sqrt;
};

template <int COPY>  // 0: no compute copy, 1: bf16, 2: fp32
__global__ __launch_bounds__(256) void k_adamw_flat(float* __restrict__ p, float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    char* __restrict__ copy, int64_t nvec,
                                                    const float* __restrict__ grad_scale, int zero_grad, AdamArgs a) {
  const float gs = grad_scale ? grad_scale[0] : 1.0f;
  const float decay = 1.0f - a.lr * a.weight_decay;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    float4 pv = reinterpret_cast<float4*>(p)[i], gv = reinterpret_cast<float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float* pe = reinterpret_cast<float*>(&pv);
    float* ge = reinterpret_cast<float*>(&gv);
    float* me = reinterpret_cast<float*>(&mv);
    float* ve = reinterpret_cast<float*>(&vv);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = ge[k] * gs;
      float pk = pe[k] * decay;
      const float mk = me[k] + (gk - me[k]) * (1.0f - a.beta1);
      const float vk = ve[k] * a.beta2 + gk * gk * (1.0f - a.beta2);
      const float denom = sqrtf(vk) * a.inv_bc2_sqrt + a.eps;
      pk -= a.step_size * (mk / denom);
      pe[k] = pk, me[k] = mk, ve[k] = vk;
    }
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
    if (zero_grad) reinterpret_cast<float4*>(g)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if constexpr (COPY == 1) {
      uint2 o;
      o.x = pack_bf16(pe[0], pe[1]);
      o.y = pack_bf16(pe[2], pe[3]);
      reinterpret_cast<uint2*>(copy)[i] = o;
    } else if constexpr (COPY == 2) {
      reinterpret_cast<float4*>(copy)[i] = pv;
    }
  }
}

}  // namespace prl

using namespace prl;

extern "C" int prl_adamw_flat(float* param, float* grad, float* exp_avg, float* exp_avg_sq, void* compute_copy,
                              int copy_dtype, int64_t n, const float* grad_scale, int zero_grad, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int64_t step, void* stream) {
  clear_error();
  PRL_REQUIRE(n >= 0 && n % 4 == 0, "adamw_flat: the element count must be a multiple of 4 (got %lld)", (long long)n);
  PRL_REQUIRE(step >= 1, "adamw_flat: steps count from 1");
  PRL_REQUIRE(compute_copy == nullptr || copy_dtype == PRL_F32 || copy_dtype == PRL_BF16,
              "adamw_flat: unsupported dtype %d for the compute copy", copy_dtype);
  if (n == 0) return PRL_OK;
  PRL_REQUIRE(param && grad && exp_avg && exp_avg_sq, "adamw_flat: param, grad and both moments must not be NULL");
  PRL_REQUIRE(((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) |
                reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(compute_copy) & 7) == 0,
              "adamw_flat: buffers must be 16-byte aligned (the bf16 copy 8-byte)");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  AdamArgs a{lr, beta1, beta2, eps, weight_decay, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2))};
  const int64_t nvec = n / 4;
  const int64_t want = (nvec + 255) / 256;
  const dim3 grid((unsigned)(want < 16384 ? want : 16384)), block(256);
  hipStream_t st = (hipStream_t)stream;
  char* copy = static_cast<char*>(compute_copy);
  if (!copy)
    hipLaunchKernelGGL(k_adamw_flat<0>, grid, block, 0, st, param, grad, exp_avg, exp_avg_sq, copy, nvec, grad_scale,
                       zero_grad, a);
  else if (copy_dtype == PRL_BF16)
    hipLaunchKernelGGL(k_adamw_flat<1>, grid, block, 0, st, param, grad, exp_avg, exp_avg_sq, copy, nvec, grad_scale,
                       zero_grad, a);
  else
    hipLaunchKernelGGL(k_adamw_flat<2>, grid, block, 0, st, param, grad, exp_avg, exp_avg_sq, copy, nvec, grad_scale,
                       zero_grad, a);
  PRL_CHECK_LAUNCH("adamw_flat");
  return PRL_OK;
}
