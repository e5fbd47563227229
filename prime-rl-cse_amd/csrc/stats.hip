// mean / median / std / min / max of the selected entries of a device vector.
//
// The reference copies every logged vector to the host, gathers it across
// ranks, calls five torch reductions with an .item() each and finally converts
// the whole vector to a Python list (src/prime_rl/trainer/utils.py:235-256).
// Here the five numbers are produced where the data lives: moments and extrema
// in one pass, the median by an exact 4 x 8-bit radix select on order-preserving
// integer keys (no sort, no host round trip).
#include "prl_common.h"

namespace prl {

constexpr int kStatThreads = 256;  // also the number of bins of one radix digit
constexpr int kStatWaves = kStatThreads / kWave;
constexpr int kStatMaxBlocks = 1024;

struct StatsWs {
  unsigned long long count;      // selected, non-NaN
  unsigned long long nan_count;  // selected NaNs
  double sum, sumsq;
  unsigned min_key, max_key;
  unsigned hist[4][256];  // digit histograms, most significant digit first
};

// fp32 -> uint32 whose unsigned order is the float order (-inf lowest, +inf highest)
__device__ __forceinline__ unsigned order_key(float x) {
  const unsigned u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_value(unsigned k) {
  const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

__global__ void k_stats_reset(StatsWs* ws) {
  unsigned* w = reinterpret_cast<unsigned*>(ws);
  for (int i = threadIdx.x; i < (int)(sizeof(StatsWs) / 4); i += blockDim.x) w[i] = 0u;
  __syncthreads();
  if (threadIdx.x == 0) {
    ws->min_key = 0xffffffffu;
    ws->max_key = 0u;
  }
}

// Walk the digit histograms of levels 0 .. level-1 to find the key prefix that
// holds the element of rank k = (count - 1) / 2 (the lower median, as
// torch.median returns it) and the rank left inside that prefix.  Every thread
// of the workgroup gets the same answer.  Needs count > 0.
__device__ void select_prefix(const StatsWs* ws, int level, unsigned* prefix_out, unsigned long long* k_out,
                              unsigned long long* scratch /* LDS: kStatWaves + 2 */) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
  unsigned prefix = 0;
  unsigned long long k = (ws->count - 1) / 2;
  for (int lv = 0; lv < level; ++lv) {
    const unsigned long long h = ws->hist[lv][tid];
    unsigned long long inc = h;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const unsigned long long up = __shfl_up(inc, o, kWave);
      if (lane >= o) inc += up;
    }
    if (lane == kWave - 1) scratch[wid] = inc;
    __syncthreads();
    unsigned long long off = 0;
#pragma unroll
    for (int w = 0; w < kStatWaves; ++w) off += (w < wid) ? scratch[w] : 0ull;
    inc += off;
    const unsigned long long exc = inc - h;
    if (h > 0 && exc <= k && k < inc) {  // exactly one bin qualifies
      scratch[kStatWaves] = (unsigned long long)tid;
      scratch[kStatWaves + 1] = k - exc;
    }
    __syncthreads();
    prefix = (prefix << 8) | (unsigned)scratch[kStatWaves];
    k = scratch[kStatWaves + 1];
    __syncthreads();
  }
  *prefix_out = prefix;
  *k_out = k;
}

// One pass over the data per radix digit.  LEVEL 0 also gathers the count, the
// sum, the sum of squares, the extrema and the NaN count.
template <int LEVEL>
__global__ __launch_bounds__(kStatThreads) void k_stats_pass(const float* __restrict__ x,
                                                             const uint8_t* __restrict__ mask, int64_t n,
                                                             StatsWs* __restrict__ ws) {
  __shared__ unsigned hist[256];
  __shared__ unsigned ext[2];
  __shared__ unsigned long long scratch[kStatWaves + 2];
  __shared__ double dred[kStatWaves];
  const int tid = threadIdx.x;
  hist[tid] = 0u;
  if (tid == 0) {
    ext[0] = 0xffffffffu;
    ext[1] = 0u;
  }
  unsigned prefix = 0;
  unsigned long long krem = 0;
  if (LEVEL > 0) {
    if (ws->count == 0) return;  // the same for every thread
    select_prefix(ws, LEVEL, &prefix, &krem, scratch);
  }
  __syncthreads();
  constexpr int shift = 24 - 8 * LEVEL;
  double sum = 0.0, sumsq = 0.0, cnt = 0.0, nans = 0.0;
  unsigned kmin = 0xffffffffu, kmax = 0u;
  for (int64_t i = (int64_t)blockIdx.x * kStatThreads + tid; i < n; i += (int64_t)gridDim.x * kStatThreads) {
    if (mask && mask[i] == 0) continue;
    const float v = x[i];
    if (v != v) {
      nans += 1.0;
      continue;
    }
    const unsigned key = order_key(v);
    if (LEVEL == 0) {
      cnt += 1.0;
      sum += (double)v;
      sumsq += (double)v * (double)v;
      kmin = key < kmin ? key : kmin;
      kmax = key > kmax ? key : kmax;
      atomicAdd(&hist[key >> 24], 1u);
    } else if ((key >> (shift + 8)) == prefix) {
      atomicAdd(&hist[(key >> shift) & 0xffu], 1u);
    }
  }
  if (LEVEL == 0) {
    atomicMin(&ext[0], kmin);
    atomicMax(&ext[1], kmax);
  }
  __syncthreads();
  if (hist[tid]) atomicAdd(&ws->hist[LEVEL][tid], hist[tid]);
  if (LEVEL == 0) {
    sum = block_sum<kStatWaves>(sum, dred);
    sumsq = block_sum<kStatWaves>(sumsq, dred);
    cnt = block_sum<kStatWaves>(cnt, dred);
    nans = block_sum<kStatWaves>(nans, dred);
    if (tid == 0) {
      if (cnt > 0.0) {
        atomicAdd(&ws->count, (unsigned long long)cnt);
        atomicAdd(&ws->sum, sum);
        atomicAdd(&ws->sumsq, sumsq);
        atomicMin(&ws->min_key, ext[0]);
        atomicMax(&ws->max_key, ext[1]);
      }
      if (nans > 0.0) atomicAdd(&ws->nan_count, (unsigned long long)nans);
    }
  }
}

__global__ __launch_bounds__(kStatThreads) void k_stats_finish(const StatsWs* __restrict__ ws,
                                                               float* __restrict__ out5,
                                                               int64_t* __restrict__ count_out) {
  __shared__ unsigned long long scratch[kStatWaves + 2];
  const unsigned long long n = ws->count;
  if (threadIdx.x == 0 && count_out) count_out[0] = (int64_t)(n + ws->nan_count);
  const float nanv = __builtin_nanf("");
  if (n == 0 || ws->nan_count > 0) {
    // nothing selected, or a NaN among the selected: every statistic is NaN, as torch's are
    if (threadIdx.x < 5) out5[threadIdx.x] = nanv;
    return;
  }
  unsigned key;
  unsigned long long krem;
  select_prefix(ws, 4, &key, &krem, scratch);
  if (threadIdx.x == 0) {
    const double mean = ws->sum / (double)n;
    double var = (ws->sumsq - ws->sum * mean) / (double)(n - 1);  // n == 1: 0/0 = NaN, as torch's unbiased std
    if (var < 0.0) var = 0.0;
    out5[0] = (float)mean;
    out5[1] = key_value(key);
    out5[2] = (float)sqrt(var);
    out5[3] = key_value(ws->min_key);
    out5[4] = key_value(ws->max_key);
  }
}

}  // namespace prl

using namespace prl;

extern "C" size_t prl_masked_stats_workspace_bytes(void) { return sizeof(StatsWs); }

extern "C" int prl_masked_stats(const float* x, const uint8_t* mask, int64_t n, float* out5, int64_t* count,
                                void* workspace, size_t workspace_bytes, void* stream) {
  clear_error();
  PRL_REQUIRE(n >= 0, "masked_stats: n must be >= 0");
  PRL_REQUIRE(out5, "masked_stats: out5 must not be NULL");
  PRL_REQUIRE(n == 0 || x, "masked_stats: x must not be NULL");
  PRL_REQUIRE(workspace && workspace_bytes >= sizeof(StatsWs) && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0,
              "masked_stats: workspace of %zu bytes (8-byte aligned) required", sizeof(StatsWs));
  PRL_REQUIRE(n < ((int64_t)1 << 32), "masked_stats: n too large");
  hipStream_t st = (hipStream_t)stream;
  StatsWs* ws = (StatsWs*)workspace;
  hipLaunchKernelGGL(k_stats_reset, dim3(1), dim3(256), 0, st, ws);
  PRL_CHECK_LAUNCH("masked_stats(reset)");
  if (n > 0) {
    int64_t nblk = (n + kStatThreads * 4 - 1) / (kStatThreads * 4);
    if (nblk > kStatMaxBlocks) nblk = kStatMaxBlocks;
    const dim3 grid((unsigned)nblk), block(kStatThreads);
    hipLaunchKernelGGL(k_stats_pass<0>, grid, block, 0, st, x, mask, n, ws);
    hipLaunchKernelGGL(k_stats_pass<1>, grid, block, 0, st, x, mask, n, ws);
    hipLaunchKernelGGL(k_stats_pass<2>, grid, block, 0, st, x, mask, n, ws);
    hipLaunchKernelGGL(k_stats_pass<3>, grid, block, 0, st, x, mask, n, ws);
    PRL_CHECK_LAUNCH("masked_stats(pass)");
  }
  hipLaunchKernelGGL(k_stats_finish, dim3(1), dim3(kStatThreads), 0, st, ws, out5, count);
  PRL_CHECK_LAUNCH("masked_stats(finish)");
  return PRL_OK;
}
