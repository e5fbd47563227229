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
//
// SQ: the workgroup also leaves the sum of the squares of its chunk's NEW sums in chunk_squares[blockIdx.x].  A
// chunk that a launch does not touch (no gradient, not FIRST) keeps its entry, which is still right; so once a
// FIRST launch has written every entry and every later launch of the step is an SQ one, the entries add up to the
// squared norm of the whole buffer, and the step's gradient norm costs one read of 4 bytes per 8192 elements
// instead of one more pass over the sums.
template <int DT, bool FIRST, bool SQ>
__global__ __launch_bounds__(kAccumThreads) void k_accumulate(const void* const* __restrict__ src,
                                                             const int64_t* __restrict__ dst_offset,
                                                             const int64_t* __restrict__ numel,
                                                             const int32_t* __restrict__ chunk_tensor,
                                                             const int64_t* __restrict__ chunk_start,
                                                             const float* __restrict__ scale,
                                                             float* __restrict__ acc,
                                                             float* __restrict__ chunk_squares) {
  __shared__ float slots[kAccumThreads / kWave];
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
    if (SQ && threadIdx.x == 0) chunk_squares[blockIdx.x] = 0.0f;
    return;
  }
  float ss = 0.0f;
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
    if (SQ) {
      ss += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
      ss += (b.x * b.x + b.y * b.y) + (b.z * b.z + b.w * b.w);
    }
  }
  for (int64_t i = (nvec << 3) + threadIdx.x; i < n; i += kAccumThreads) {
    float x;
    if constexpr (DT == PRL_F32) {
      x = reinterpret_cast<const float*>(s)[i];
    } else {
      x = __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(s)[i] << 16);
    }
    const float y = FIRST ? x * sc : fmaf(x, sc, d[i]);
    d[i] = y;
    if (SQ) ss += y * y;
  }
  if (SQ) {  // (every thread of the workgroup gets here: the early returns above are taken by all or by none)
    ss = block_sum<kAccumThreads / kWave>(ss, slots);
    if (threadIdx.x == 0) chunk_squares[blockIdx.x] = ss;
  }
}

}  // namespace prl

using namespace prl;

extern "C" int64_t prl_accumulate_chunk_elems(void) { return kAccumChunk; }

static int accumulate(bool first, const void* const* src, int src_dtype, const int64_t* dst_offset,
                      const int64_t* numel, const int32_t* chunk_tensor, const int64_t* chunk_start, int64_t n_chunks,
                      const float* scale, float* acc, float* chunk_squares, void* stream) {
  clear_error();
  PRL_REQUIRE(n_chunks >= 0 && n_chunks <= 0x7fffffffLL, "accumulate: n_chunks=%lld out of range", (long long)n_chunks);
  PRL_REQUIRE(src_dtype == PRL_F32 || src_dtype == PRL_BF16, "accumulate: unsupported gradient dtype %d", src_dtype);
  if (n_chunks == 0) return PRL_OK;
  PRL_REQUIRE(src && dst_offset && numel && chunk_tensor && chunk_start && acc,
              "accumulate: the tables and the accumulator must not be NULL");
  const dim3 grid((unsigned)n_chunks), block(kAccumThreads);
  hipStream_t st = (hipStream_t)stream;
#define PRL_ACC(DT, FIRST, SQ)                                                                                   \
  hipLaunchKernelGGL((k_accumulate<DT, FIRST, SQ>), grid, block, 0, st, src, dst_offset, numel, chunk_tensor,      \
                     chunk_start, scale, acc, chunk_squares)
#define PRL_ACC_FORM(DT)                                                      \
  if (chunk_squares) {                                                        \
    if (first) PRL_ACC(DT, true, true); else PRL_ACC(DT, false, true);        \
  } else {                                                                    \
    if (first) PRL_ACC(DT, true, false); else PRL_ACC(DT, false, false);      \
  }
  if (src_dtype == PRL_F32) {
    PRL_ACC_FORM(PRL_F32)
  } else {
    PRL_ACC_FORM(PRL_BF16)
  }
#undef PRL_ACC_FORM
#undef PRL_ACC
  PRL_CHECK_LAUNCH("accumulate");
  return PRL_OK;
}

extern "C" int prl_accumulate(const void* const* src, int src_dtype, const int64_t* dst_offset, const int64_t* numel,
                              const int32_t* chunk_tensor, const int64_t* chunk_start, int64_t n_chunks,
                              const float* scale, float* acc, void* stream) {
  return accumulate(false, src, src_dtype, dst_offset, numel, chunk_tensor, chunk_start, n_chunks, scale, acc, nullptr,
                    stream);
}

extern "C" int prl_accumulate_first(const void* const* src, int src_dtype, const int64_t* dst_offset,
                                    const int64_t* numel, const int32_t* chunk_tensor, const int64_t* chunk_start,
                                    int64_t n_chunks, const float* scale, float* acc, void* stream) {
  return accumulate(true, src, src_dtype, dst_offset, numel, chunk_tensor, chunk_start, n_chunks, scale, acc, nullptr,
                    stream);
}

extern "C" int prl_accumulate_squares(int first, const void* const* src, int src_dtype, const int64_t* dst_offset,
                                      const int64_t* numel, const int32_t* chunk_tensor, const int64_t* chunk_start,
                                      int64_t n_chunks, const float* scale, float* acc, float* chunk_squares,
                                      void* stream) {
  clear_error();
  PRL_REQUIRE(chunk_squares != nullptr || n_chunks == 0, "accumulate_squares: chunk_squares must not be NULL");
  return accumulate(first != 0, src, src_dtype, dst_offset, numel, chunk_tensor, chunk_start, n_chunks, scale, acc,
                    chunk_squares, stream);
}

// ---------------------------------------------------------------- sum of partial products
//
// out[i] = part[0][i] + part[1][i] + ... (in that order, fp32 arithmetic), rounded to bf16: the last step of a weight
// gradient computed as a few batched products over ranges of the tokens (a long sum with a small output fills the
// chip only when it is cut up).  One pass and one rounding, where a reduction followed by a type conversion reads
// the partial results at half the memory rate and launches twice.
namespace prl {

template <int DTI>
__global__ __launch_bounds__(256) void k_sum_parts(const char* __restrict__ parts, int64_t part_stride_bytes,
                                                   int nparts, int64_t nvec, char* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // 8 consecutive outputs
  if (i >= nvec) return;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
#pragma unroll 4
  for (int p = 0; p < nparts; ++p) {
    const char* src = parts + p * part_stride_bytes;
    if constexpr (DTI == PRL_F32) {
      const u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src) + 2 * i);
      const u32x4 b = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src) + 2 * i + 1);
      acc[0] += __uint_as_float(a.x), acc[1] += __uint_as_float(a.y);
      acc[2] += __uint_as_float(a.z), acc[3] += __uint_as_float(a.w);
      acc[4] += __uint_as_float(b.x), acc[5] += __uint_as_float(b.y);
      acc[6] += __uint_as_float(b.z), acc[7] += __uint_as_float(b.w);
    } else {
      const u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src) + i);
      const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] += __uint_as_float(w[j] << 16);
        acc[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);
      }
    }
  }
  u32x4 v;
  v.x = pack_bf16(acc[0], acc[1]);
  v.y = pack_bf16(acc[2], acc[3]);
  v.z = pack_bf16(acc[4], acc[5]);
  v.w = pack_bf16(acc[6], acc[7]);
  reinterpret_cast<u32x4*>(out)[i] = v;
}

}  // namespace prl

extern "C" int prl_sum_parts(const void* parts, int parts_dtype, int64_t part_stride, int32_t nparts, int64_t n,
                             void* out, void* stream) {
  using namespace prl;
  clear_error();
  PRL_REQUIRE(parts_dtype == PRL_F32 || parts_dtype == PRL_BF16, "sum_parts: partial results must be fp32 or bf16");
  PRL_REQUIRE(nparts >= 1 && n >= 0 && n % 8 == 0 && part_stride >= n && part_stride % 8 == 0,
              "sum_parts: bad sizes (%d parts of %lld, stride %lld; multiples of 8)", nparts, (long long)n,
              (long long)part_stride);
  if (n == 0) return PRL_OK;
  PRL_REQUIRE(parts && out && ((reinterpret_cast<uintptr_t>(parts) | reinterpret_cast<uintptr_t>(out)) & 15) == 0,
              "sum_parts: pointers must be 16-byte aligned and not NULL");
  const int64_t nvec = n / 8;
  PRL_REQUIRE((nvec + 255) / 256 <= 0x7fffffffLL, "sum_parts: too many elements");
  const dim3 grid((unsigned)((nvec + 255) / 256)), block(256);
  const int es = parts_dtype == PRL_F32 ? 4 : 2;
  if (parts_dtype == PRL_F32)
    hipLaunchKernelGGL((k_sum_parts<PRL_F32>), grid, block, 0, (hipStream_t)stream, (const char*)parts,
                       part_stride * es, nparts, nvec, (char*)out);
  else
    hipLaunchKernelGGL((k_sum_parts<PRL_BF16>), grid, block, 0, (hipStream_t)stream, (const char*)parts,
                       part_stride * es, nparts, nvec, (char*)out);
  PRL_CHECK_LAUNCH("sum_parts");
  return PRL_OK;
}
