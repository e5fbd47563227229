// Shared device helpers and host-side error plumbing for libprl_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "prl_hip.h"

namespace prl {

// ---------------------------------------------------------------- host: errors
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void clear_error();

#define PRL_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      ::prl::set_error(__VA_ARGS__);  \
      return PRL_EINVAL;              \
    }                                 \
  } while (0)

// Launch errors are read with hipGetLastError right after the launch; this does
// not synchronise, so the call stays legal under stream capture.
#define PRL_CHECK_LAUNCH(name)                                                        \
  do {                                                                                \
    hipError_t e__ = hipGetLastError();                                               \
    if (e__ != hipSuccess) {                                                          \
      ::prl::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));        \
      return PRL_ELAUNCH;                                                             \
    }                                                                                 \
  } while (0)

// 16 bytes as a native vector (the non-temporal load/store builtins take these, not HIP's uint4)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- device: constants
constexpr int kWave = 64;  // CDNA wavefront
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kNegBig = -3.0e38f;  // finite stand-in for "no value yet" in running maxima

// ---------------------------------------------------------------- device: wave and block reductions
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, kWave));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}

// Sum over a workgroup of NW waves; every thread gets the result.  `slots` is
// LDS for NW values; the trailing barrier makes it reusable straight away.
template <int NW, typename T>
__device__ __forceinline__ T block_sum(T v, T* slots) {
  v = wave_sum(v);
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  if (lane == 0) slots[wid] = v;
  __syncthreads();
  T r = slots[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) r += slots[i];
  __syncthreads();
  return r;
}

// ---------------------------------------------------------------- device: bf16 <-> fp32
__device__ __forceinline__ float bf16_lo(uint32_t pair) { return __uint_as_float(pair << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t pair) { return __uint_as_float(pair & 0xffff0000u); }
__device__ __forceinline__ float bf16_one(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// Two floats to a packed bf16 pair, round to nearest even, NaN stays NaN
// (a plain cast compiles to v_cvt_pk_bf16_f32 on gfx950).
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  bf16x2_t v;
  v.x = (__bf16)lo;
  v.y = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint16_t one_bf16(float x) {
  __bf16 h = (__bf16)x;
  return __builtin_bit_cast(uint16_t, h);
}

}  // namespace prl
