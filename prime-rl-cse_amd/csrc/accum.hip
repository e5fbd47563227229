// Gradient accumulation over micro-batches for a model that lives whole on the GPU: every parameter's
// fresh gradient (bf16 or fp32, wherever autograd put it) is added, scaled, into ONE flat fp32 buffer by
// one launch.  The reference gets the same sums from FSDP2's fp32 reduction of every micro-batch's
// gradients (src/prime_rl/trainer/model.py:102-104, src/prime_rl/trainer/rl/train.py:271-310).
//
// The tensors are cut into chunks of kAccumChunk elements and a workgroup takes one chunk; which tensor a
// chunk belongs to and where it starts is a table made once on the host (parameter sizes never change).
// Only the array of source pointers differs from launch to launch.  The scale is read from device memory
// so that a captured launch can be replayed with another value.
#include "prl_common.h"

namespace prl {

constexpr int kAccumThreads = 256;
constexpr int kAccumChunk = 8192;  // elements per workgroup: 16 KB of bf16 in, 32 KB of fp32 read and written

// FIRST: the sums start here (acc = scale * src, zeros where a tensor has no gradient) instead of continuing
// (acc += scale * src): the first pass of a step then needs neither a cleared buffer nor a read of it.
template <int DT, bool FIRST>
__global__ __launch_bounds__(kAccumThreads) void k_accumulate(const void* const* __restrict__ src,
                                                             const int64_t* __restrict__ dst_offset,
                                                             const int64_t* __restrict__ numel,
                                                             const int32_t* __restrict__ chunk_tensor,
                                                             const int64_t* __restrict__ chunk_start,
                                                             const float* __restrict__ scale,
                                                             float* __restrict__ acc) {
  const int tensor = chunk_tensor[blockIdx.x];
  const char* s = static_cast<const char*>(src[tensor]);
  if (!FIRST && s == nullptr) return;  // a parameter that got no gradient in this backward
  const int64_t start = chunk_start[blockIdx.x];
  int64_t n = numel[tensor] - start;
  if (n > kAccumChunk) n = kAccumChunk;
  const float sc = scale ? scale[0] : 1.0f;
  float* d = acc + dst_offset[tensor] + start;
  if (FIRST && s == nullptr) {  // the first pass of a step starts every sum: this one at zero
    for (int64_t i = threadIdx.x; i < n; i += kAccumThreads) d[i] = 0.0f;
    return;
  }
  constexpr int ES = DT == PRL_F32 ? 4 : 2;
  s += start * ES;
  // 8 elements per lane and turn where both sides are 16-byte aligned (the host aligns the flat buffer's
  // offsets; autograd's gradients are fresh allocations), single elements otherwise and for the tail
  const bool aligned = ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0;
  const int64_t nvec = aligned ? n >> 3 : 0;
  for (int64_t u = threadIdx.x; u < nvec; u += kAccumThreads) {
    float x[8];
    if constexpr (DT == PRL_F32) {
      const float4 a = reinterpret_cast<const float4*>(s)[2 * u], b = reinterpret_cast<const float4*>(s)[2 * u + 1];
      x[0] = a.x, x[1] = a.y, x[2] = a.z, x[3] = a.w, x[4] = b.x, x[5] = b.y, x[6] = b.z, x[7] = b.w;
    } else {
      const u32x4 v = reinterpret_cast<const u32x4*>(s)[u];
      x[0] = bf16_lo(v.x), x[1] = bf16_hi(v.x), x[2] = bf16_lo(v.y), x[3] = bf16_hi(v.y);
      x[4] = bf16_lo(v.z), x[5] = bf16_hi(v.z), x[6] = bf16_lo(v.w), x[7] = bf16_hi(v.w);
    }
    float4* dp = reinterpret_cast<float4*>(d) + 2 * u;
    float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = a;
    if (!FIRST) a = dp[0], b = dp[1];
    a.x = fmaf(x[0], sc, a.x), a.y = fmaf(x[1], sc, a.y), a.z = fmaf(x[2], sc, a.z), a.w = fmaf(x[3], sc, a.w);
    b.x = fmaf(x[4], sc, b.x), b.y = fmaf(x[5], sc, b.y), b.z = fmaf(x[6], sc, b.z), b.w = fmaf(x[7], sc, b.w);
    dp[0] = a, dp[1] = b;
  }
  for (int64_t i = (nvec << 3) + threadIdx.x; i < n; i += kAccumThreads) {
    float x;
    if constexpr (DT == PRL_F32) {
      x = reinterpret_cast<const float*>(s)[i];
    } else {
      x = __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(s)[i] << 16);
    }
    d[i] = FIRST ? x * sc : fmaf(x, sc, d[i]);
  }
}

}  // namespace prl

using namespace prl;

extern "C" int64_t prl_accumulate_chunk_elems(void) { return kAccumChunk; }

static int accumulate(bool first, const void* const* src, int src_dtype, const int64_t* dst_offset,
                      const int64_t* numel, const int32_t* chunk_tensor, const int64_t* chunk_start, int64_t n_chunks,
                      const float* scale, float* acc, void* stream) {
  clear_error();
  PRL_REQUIRE(n_chunks >= 0 && n_chunks <= 0x7fffffffLL, "accumulate: n_chunks=%lld out of range", (long long)n_chunks);
  PRL_REQUIRE(src_dtype == PRL_F32 || src_dtype == PRL_BF16, "accumulate: unsupported gradient dtype %d", src_dtype);
  if (n_chunks == 0) return PRL_OK;
  PRL_REQUIRE(src && dst_offset && numel && chunk_tensor && chunk_start && acc,
              "accumulate: the tables and the accumulator must not be NULL");
  const dim3 grid((unsigned)n_chunks), block(kAccumThreads);
  hipStream_t st = (hipStream_t)stream;
#define PRL_ACC(DT, FIRST)                                                                                         \
  hipLaunchKernelGGL((k_accumulate<DT, FIRST>), grid, block, 0, st, src, dst_offset, numel, chunk_tensor, chunk_start, \
                     scale, acc)
  if (src_dtype == PRL_F32) {
    if (first) PRL_ACC(PRL_F32, true); else PRL_ACC(PRL_F32, false);
  } else {
    if (first) PRL_ACC(PRL_BF16, true); else PRL_ACC(PRL_BF16, false);
  }
#undef PRL_ACC
  PRL_CHECK_LAUNCH("accumulate");
  return PRL_OK;
}

extern "C" int prl_accumulate(const void* const* src, int src_dtype, const int64_t* dst_offset, const int64_t* numel,
                              const int32_t* chunk_tensor, const int64_t* chunk_start, int64_t n_chunks,
                              const float* scale, float* acc, void* stream) {
  return accumulate(false, src, src_dtype, dst_offset, numel, chunk_tensor, chunk_start, n_chunks, scale, acc, stream);
}

extern "C" int prl_accumulate_first(const void* const* src, int src_dtype, const int64_t* dst_offset,
                                    const int64_t* numel, const int32_t* chunk_tensor, const int64_t* chunk_start,
                                    int64_t n_chunks, const float* scale, float* acc, void* stream) {
  return accumulate(true, src, src_dtype, dst_offset, numel, chunk_tensor, chunk_start, n_chunks, scale, acc, stream);
}
