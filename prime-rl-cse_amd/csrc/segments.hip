// Sample boundaries of a packed row from its position ids, on the device.
//
// The reference walks the row in a Python loop that reads a device tensor
// element by element (src/prime_rl/trainer/utils.py:36-69): three host syncs
// per token.  Here one workgroup flags the boundaries and numbers the samples
// with a ballot-based prefix sum, 1024 tokens per step, with no host round trip.
#include "prl_common.h"

namespace prl {

constexpr int kSegThreads = 1024;

// Index i >= 1 opens a new sample iff pos[i] == 0, pos[i-1] != 0, i+1 < T and pos[i+1] == 1.
__device__ __forceinline__ bool opens_sample(const int64_t* __restrict__ pos, int64_t i, int64_t T) {
  if (i < 1 || i + 1 >= T) return false;
  return pos[i] == 0 && pos[i - 1] != 0 && pos[i + 1] == 1;
}

__global__ __launch_bounds__(kSegThreads) void k_segments(const int64_t* __restrict__ pos, int64_t T,
                                                          int32_t* __restrict__ seg_id,
                                                          int64_t* __restrict__ seg_start,
                                                          int32_t* __restrict__ n_seg,
                                                          unsigned long long* __restrict__ clear, int64_t n_clear) {
  constexpr int NW = kSegThreads / kWave;
  // words a later launch on this stream wants zeroed (the hand-off words of the one-pass sequence kernel)
  for (int64_t i = threadIdx.x; i < n_clear; i += kSegThreads) clear[i] = 0ull;
  __shared__ int wave_total[NW];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
  int carry = 0;  // boundaries in front of the current step
  if (tid == 0) seg_start[0] = 0;
  for (int64_t base = 0; base < T; base += kSegThreads) {
    const int64_t i = base + tid;
    const bool b = opens_sample(pos, i, T);
    const unsigned long long bal = __ballot(b);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_total[wid] = __popcll(bal);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const int c = wave_total[w];
      off += (w < wid) ? c : 0;
      total += c;
    }
    const int sid = carry + off + before + (b ? 1 : 0);
    if (i < T) {
      if (seg_id) seg_id[i] = sid;
      if (b) seg_start[sid] = i;
    }
    carry += total;
    __syncthreads();
  }
  if (tid == 0) {
    // an empty row is ONE sample of length 0, as the reference's `boundaries = [0]` makes it
    // (src/prime_rl/trainer/utils.py:52, :63-67)
    const int n = carry + 1;
    n_seg[0] = n;
    seg_start[n] = T;
  }
}

}  // namespace prl

using namespace prl;

namespace prl {
int segments_impl(const int64_t* pos, int64_t T, int32_t* seg_id, int64_t* seg_start, int32_t* n_seg,
                  unsigned long long* clear, int64_t n_clear, void* stream);
}

extern "C" int prl_segments_from_position_ids(const int64_t* pos, int64_t T, int32_t* seg_id, int64_t* seg_start,
                                              int32_t* n_seg, void* stream) {
  return segments_impl(pos, T, seg_id, seg_start, n_seg, nullptr, 0, stream);
}

int prl::segments_impl(const int64_t* pos, int64_t T, int32_t* seg_id, int64_t* seg_start, int32_t* n_seg,
                       unsigned long long* clear, int64_t n_clear, void* stream) {
  clear_error();
  PRL_REQUIRE(T >= 0, "segments_from_position_ids: T must be >= 0");
  PRL_REQUIRE(seg_start && n_seg, "segments_from_position_ids: seg_start and n_seg must not be NULL");
  PRL_REQUIRE(T == 0 || pos, "segments_from_position_ids: pos must not be NULL");
  PRL_REQUIRE(T < ((int64_t)1 << 31) - 2, "segments_from_position_ids: T too large");
  hipLaunchKernelGGL(k_segments, dim3(1), dim3(kSegThreads), 0, (hipStream_t)stream, pos, T, seg_id, seg_start,
                     n_seg, clear, clear ? n_clear : (int64_t)0);
  PRL_CHECK_LAUNCH("segments_from_position_ids");
  return PRL_OK;
}
