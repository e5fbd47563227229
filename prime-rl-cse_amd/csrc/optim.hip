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

// Build-time tunables (tools/tune_adamw.py): vectors of four elements a thread has in flight per trip, whether
// the streams bypass the caches' retention (nothing here is read again before 18 GB more have gone by), and
// the cap on the grid.  Measured on the 0.6B model's 596 M elements (30 bytes each): a capped grid that strides
// over the buffers 3.69 ms (4.84 TB/s); one workgroup per 256 vectors and no stride 3.28 ms; two vectors per
// thread, non-temporal streams and no stride 3.12 ms (5.73 TB/s).  Every variant gives the same bits.
#ifndef PRL_ADAM_UNROLL
#define PRL_ADAM_UNROLL 2
#endif
#ifndef PRL_ADAM_NT
#define PRL_ADAM_NT 1
#endif
#ifndef PRL_ADAM_MAX_BLOCKS
#define PRL_ADAM_MAX_BLOCKS (1 << 22)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));

template <typename V>
__device__ __forceinline__ V stream_load(const V* p) {
#if PRL_ADAM_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
template <typename V>
__device__ __forceinline__ void stream_store(V* p, V v) {
#if PRL_ADAM_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

template <int COPY>  // 0: no compute copy, 1: bf16, 2: fp32
__global__ __launch_bounds__(256) void k_adamw_flat(float* __restrict__ p, float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    char* __restrict__ copy, int64_t nvec,
                                                    const float* __restrict__ grad_scale, int zero_grad, AdamArgs a) {
  constexpr int U = PRL_ADAM_UNROLL;
  const float gs = grad_scale ? grad_scale[0] : 1.0f;
  const float decay = 1.0f - a.lr * a.weight_decay;
  f32x4* p4 = reinterpret_cast<f32x4*>(p);
  f32x4* g4 = reinterpret_cast<f32x4*>(g);
  f32x4* m4 = reinterpret_cast<f32x4*>(m);
  f32x4* v4 = reinterpret_cast<f32x4*>(v);
  // a trip of the grid covers U consecutive stretches of gridDim.x * 256 vectors; a thread takes the same place in each
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t base = (int64_t)blockIdx.x * 256 + threadIdx.x; base < nvec; base += stride * U) {
    f32x4 pv[U], gv[U], mv[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * stride;
      if (i < nvec) {
        pv[u] = stream_load(p4 + i), gv[u] = stream_load(g4 + i);
        mv[u] = stream_load(m4 + i), vv[u] = stream_load(v4 + i);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * stride;
      if (i >= nvec) break;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gk = gv[u][k] * gs;
        float pk = pv[u][k] * decay;
        const float mk = mv[u][k] + (gk - mv[u][k]) * (1.0f - a.beta1);
        const float vk = vv[u][k] * a.beta2 + gk * gk * (1.0f - a.beta2);
        const float denom = sqrtf(vk) * a.inv_bc2_sqrt + a.eps;
        pk -= a.step_size * (mk / denom);
        pv[u][k] = pk, mv[u][k] = mk, vv[u][k] = vk;
      }
      stream_store(p4 + i, pv[u]);
      stream_store(m4 + i, mv[u]);
      stream_store(v4 + i, vv[u]);
      if (zero_grad) stream_store(g4 + i, f32x4{0.0f, 0.0f, 0.0f, 0.0f});
      if constexpr (COPY == 1) {
        u32x2v o;
        o.x = pack_bf16(pv[u][0], pv[u][1]);
        o.y = pack_bf16(pv[u][2], pv[u][3]);
        stream_store(reinterpret_cast<u32x2v*>(copy) + i, o);
      } else if constexpr (COPY == 2) {
        stream_store(reinterpret_cast<f32x4*>(copy) + i, pv[u]);
      }
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
  const int64_t want = (nvec + 256 * PRL_ADAM_UNROLL - 1) / (256 * PRL_ADAM_UNROLL);
  const dim3 grid((unsigned)(want < PRL_ADAM_MAX_BLOCKS ? want : PRL_ADAM_MAX_BLOCKS)), block(256);
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
