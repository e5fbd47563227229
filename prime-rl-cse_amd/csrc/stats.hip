// mean / median / std / min / max of logged vectors, for several statistics at
// once and without moving a single value to the host.
//
// The reference copies every logged vector to the host, gathers it across
// ranks, calls five torch reductions with an .item() each and finally converts
// the whole vector to a Python list (src/prime_rl/trainer/utils.py:235-256).
// Here every statistic ("key") is a list of device pieces (one per micro-batch);
// one launch per pass covers all pieces of all keys:
//   pass 0: count, sum, sum of squares, extrema, NaN count, and the histogram
//           of the most significant byte of an order-preserving integer key
//   pass 1..3: histogram of the next byte among the values that share the
//           prefix holding the median's rank
//   finish: the 4-byte prefix IS the lower median; moments give mean and std
// The scratch is laid out so that a data-parallel job can all-reduce it between
// passes (sums for counts, moments and histograms; max for the extrema) and get
// the exact statistics of the union of all ranks' values without gathering them.
#include "prl_common.h"

namespace prl {

constexpr int kStatThreads = 256;  // also the number of bins of one radix digit
constexpr int kStatWaves = kStatThreads / kWave;
constexpr int kStatMaxBlocks = 256;
constexpr int kStatTablePieces = 64;

struct PieceTable {
  prl_stat_piece p[kStatTablePieces];
};

// scratch layout for K keys, every region 8-byte aligned:
//   int64  cnt [K][2]        selected non-NaN values, selected NaNs
//   double mom [K][2]        sum, sum of squares
//   int32  ext [K][2]        max of ~a and max of a, a = signed order-preserving key (so both reduce with MAX)
//   int32  hist[4][K][256]   digit histograms, most significant digit first
struct Layout {
  int64_t cnt, mom, ext, hist, total;
};
__host__ __device__ inline Layout layout_for(int K) {
  Layout l;
  l.cnt = 0;
  l.mom = l.cnt + (int64_t)K * 16;
  l.ext = l.mom + (int64_t)K * 16;
  l.hist = l.ext + (int64_t)K * 8;
  l.total = l.hist + (int64_t)4 * K * 256 * 4;
  return l;
}

struct Ws {
  long long* cnt;
  double* mom;
  int* ext;
  int* hist;
  int K;
  __device__ int* hist_of(int level, int key) const { return hist + ((int64_t)level * K + key) * 256; }
};
__device__ __forceinline__ Ws ws_view(void* base, int K) {
  const Layout l = layout_for(K);
  char* b = (char*)base;
  Ws w;
  w.cnt = (long long*)(b + l.cnt);
  w.mom = (double*)(b + l.mom);
  w.ext = (int*)(b + l.ext);
  w.hist = (int*)(b + l.hist);
  w.K = K;
  return w;
}

// fp32 -> int32 whose signed order is the float order (-inf lowest, +inf highest)
__device__ __forceinline__ int order_key(float x) {
  const unsigned u = __float_as_uint(x);
  const unsigned k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // unsigned order
  return (int)(k ^ 0x80000000u);                                  // signed order
}
__device__ __forceinline__ float key_value(int a) {
  const unsigned k = (unsigned)a ^ 0x80000000u;
  const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}
__device__ __forceinline__ unsigned key_bits(int a) { return (unsigned)a ^ 0x80000000u; }  // unsigned-ordered bits

__global__ void k_stats_begin(void* base, int K) {
  const Layout l = layout_for(K);
  int* w = (int*)base;
  const int64_t words = l.total / 4;
  // every word is written exactly once, by the thread that owns it, with the value it starts from: the extrema
  // (kept as "largest so far" of an order key and of its complement) start at the lowest key, the rest at zero
  const int64_t ext_lo = l.ext / 4, ext_hi = ext_lo + 2 * (int64_t)K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (int64_t)gridDim.x * blockDim.x)
    w[i] = (i >= ext_lo && i < ext_hi) ? (int)0x80000000 : 0;
}

// Walk the digit histograms of levels 0 .. level-1 of `key` to find the prefix
// that holds the value of rank (count - 1) / 2 (the lower median, as
// torch.median returns it) and the rank left inside that prefix.  Every thread
// of the workgroup gets the same answer.  Needs count > 0.
__device__ void select_prefix(const Ws& ws, int key, int level, unsigned* prefix_out,
                              unsigned long long* scratch /* LDS: kStatWaves + 2 */) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
  unsigned prefix = 0;
  unsigned long long k = (unsigned long long)(ws.cnt[2 * key] - 1) / 2;
  for (int lv = 0; lv < level; ++lv) {
    const unsigned long long h = (unsigned long long)(unsigned)ws.hist_of(lv, key)[tid];
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
}

__device__ __forceinline__ float piece_value(const prl_stat_piece& pc, int64_t i) {
  const float v = pc.x[i];
  return pc.op == PRL_STAT_EXP ? expf(v) : v;
}

// grid = (blocks per piece, pieces).  LEVEL 0 also gathers count, sum, sum of
// squares, extrema and the NaN count.
template <int LEVEL>
__global__ __launch_bounds__(kStatThreads) void k_stats_pass(PieceTable table, void* base, int K) {
  __shared__ int hist[256];
  __shared__ int ext[2];
  __shared__ unsigned long long scratch[kStatWaves + 2];
  __shared__ double dred[kStatWaves];
  const prl_stat_piece pc = table.p[blockIdx.y];
  const int tid = threadIdx.x;
  if ((int64_t)blockIdx.x * kStatThreads >= pc.n) return;  // nothing for this workgroup
  const Ws ws = ws_view(base, K);
  hist[tid] = 0;
  if (tid == 0) {
    ext[0] = (int)0x80000000;
    ext[1] = (int)0x80000000;
  }
  unsigned prefix = 0;
  if (LEVEL > 0) {
    if (ws.cnt[2 * pc.key] <= 0) return;  // the same for every thread
    select_prefix(ws, pc.key, LEVEL, &prefix, scratch);
  }
  __syncthreads();
  constexpr int shift = 24 - 8 * LEVEL;
  double sum = 0.0, sumsq = 0.0, cnt = 0.0, nans = 0.0;
  int amin_neg = (int)0x80000000, amax = (int)0x80000000;
  for (int64_t i = (int64_t)blockIdx.x * kStatThreads + tid; i < pc.n; i += (int64_t)gridDim.x * kStatThreads) {
    if (pc.mask && pc.mask[i] == 0) continue;
    const float v = piece_value(pc, i);
    if (v != v) {
      nans += 1.0;
      continue;
    }
    const int a = order_key(v);
    const unsigned kb = key_bits(a);
    if (LEVEL == 0) {
      cnt += 1.0;
      sum += (double)v;
      sumsq += (double)v * (double)v;
      amin_neg = max(amin_neg, ~a);
      amax = max(amax, a);
      atomicAdd(&hist[kb >> 24], 1);
    } else if ((kb >> (shift + 8)) == prefix) {
      atomicAdd(&hist[(kb >> shift) & 0xffu], 1);
    }
  }
  if (LEVEL == 0) {
    atomicMax(&ext[0], amin_neg);
    atomicMax(&ext[1], amax);
  }
  __syncthreads();
  if (hist[tid]) atomicAdd(&ws.hist_of(LEVEL, pc.key)[tid], hist[tid]);
  if (LEVEL == 0) {
    sum = block_sum<kStatWaves>(sum, dred);
    sumsq = block_sum<kStatWaves>(sumsq, dred);
    cnt = block_sum<kStatWaves>(cnt, dred);
    nans = block_sum<kStatWaves>(nans, dred);
    if (tid == 0) {
      if (cnt > 0.0) {
        atomicAdd((unsigned long long*)&ws.cnt[2 * pc.key], (unsigned long long)cnt);
        atomicAdd(&ws.mom[2 * pc.key], sum);
        atomicAdd(&ws.mom[2 * pc.key + 1], sumsq);
        atomicMax(&ws.ext[2 * pc.key], ext[0]);
        atomicMax(&ws.ext[2 * pc.key + 1], ext[1]);
      }
      if (nans > 0.0) atomicAdd((unsigned long long*)&ws.cnt[2 * pc.key + 1], (unsigned long long)nans);
    }
  }
}

// one workgroup per key
__global__ __launch_bounds__(kStatThreads) void k_stats_finish(void* base, int K, float* __restrict__ out,
                                                               int64_t* __restrict__ counts) {
  __shared__ unsigned long long scratch[kStatWaves + 2];
  const int key = blockIdx.x;
  const Ws ws = ws_view(base, K);
  const long long n = ws.cnt[2 * key], nn = ws.cnt[2 * key + 1];
  float* o = out + 5 * key;
  if (threadIdx.x == 0 && counts) counts[key] = n + nn;
  if (n <= 0 || nn > 0) {
    // nothing selected, or a NaN among the selected: every statistic is NaN, as torch's are
    if (threadIdx.x < 5) o[threadIdx.x] = __builtin_nanf("");
    return;
  }
  unsigned bits;
  select_prefix(ws, key, 4, &bits, scratch);
  if (threadIdx.x == 0) {
    const double s = ws.mom[2 * key], ss = ws.mom[2 * key + 1];
    const double mean = s / (double)n;
    double var = (ss - s * mean) / (double)(n - 1);  // n == 1: 0/0 = NaN, as torch's unbiased std
    if (var < 0.0) var = 0.0;
    o[0] = (float)mean;
    o[1] = key_value((int)(bits ^ 0x80000000u));
    o[2] = (float)sqrt(var);
    o[3] = key_value(~ws.ext[2 * key]);
    o[4] = key_value(ws.ext[2 * key + 1]);
  }
}

static int check_ws(const void* ws, int n_keys, const char* who) {
  PRL_REQUIRE(n_keys >= 1 && n_keys <= 4096, "%s: n_keys must be in [1, 4096]", who);
  PRL_REQUIRE(ws && (reinterpret_cast<uintptr_t>(ws) & 7) == 0, "%s: workspace must be non-NULL and 8-byte aligned",
              who);
  return PRL_OK;
}

}  // namespace prl

using namespace prl;

extern "C" size_t prl_stats_workspace_bytes(int n_keys) { return n_keys > 0 ? (size_t)layout_for(n_keys).total : 0; }

extern "C" int prl_stats_layout(int n_keys, int64_t* offsets5) {
  clear_error();
  PRL_REQUIRE(n_keys >= 1 && offsets5, "stats_layout: n_keys >= 1 and a host array of 5 required");
  const Layout l = layout_for(n_keys);
  offsets5[0] = l.cnt;
  offsets5[1] = l.mom;
  offsets5[2] = l.ext;
  offsets5[3] = l.hist;
  offsets5[4] = l.total;
  return PRL_OK;
}

extern "C" int prl_stats_begin(void* workspace, int n_keys, void* stream) {
  clear_error();
  if (int rc = check_ws(workspace, n_keys, "stats_begin")) return rc;
  const int64_t words = layout_for(n_keys).total / 4;
  int64_t nblk = (words + 1023) / 1024;
  if (nblk > 256) nblk = 256;
  hipLaunchKernelGGL(k_stats_begin, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, workspace, n_keys);
  PRL_CHECK_LAUNCH("stats_begin");
  return PRL_OK;
}

extern "C" int prl_stats_pass(int level, const prl_stat_piece* pieces, int n_pieces, int n_keys, void* workspace,
                              void* stream) {
  clear_error();
  if (int rc = check_ws(workspace, n_keys, "stats_pass")) return rc;
  PRL_REQUIRE(level >= 0 && level <= 3, "stats_pass: level must be 0..3");
  PRL_REQUIRE(n_pieces >= 0 && (n_pieces == 0 || pieces), "stats_pass: bad piece list");
  for (int i = 0; i < n_pieces; ++i) {
    PRL_REQUIRE(pieces[i].n >= 0 && pieces[i].n < ((int64_t)1 << 31), "stats_pass: piece %d has a bad size", i);
    PRL_REQUIRE(pieces[i].n == 0 || pieces[i].x, "stats_pass: piece %d has no data pointer", i);
    PRL_REQUIRE(pieces[i].key >= 0 && pieces[i].key < n_keys, "stats_pass: piece %d names key %d of %d", i,
                pieces[i].key, n_keys);
    PRL_REQUIRE(pieces[i].op == PRL_STAT_IDENTITY || pieces[i].op == PRL_STAT_EXP, "stats_pass: piece %d: bad op", i);
  }
  hipStream_t st = (hipStream_t)stream;
  for (int lo = 0; lo < n_pieces; lo += kStatTablePieces) {
    const int cnt = n_pieces - lo < kStatTablePieces ? n_pieces - lo : kStatTablePieces;
    PieceTable table;
    int64_t nmax = 0;
    for (int i = 0; i < cnt; ++i) {
      table.p[i] = pieces[lo + i];
      if (table.p[i].n > nmax) nmax = table.p[i].n;
    }
    if (nmax == 0) continue;
    int64_t nblk = (nmax + kStatThreads * 4 - 1) / (kStatThreads * 4);
    if (nblk > kStatMaxBlocks) nblk = kStatMaxBlocks;
    const dim3 grid((unsigned)nblk, (unsigned)cnt), block(kStatThreads);
    switch (level) {
      case 0: hipLaunchKernelGGL(k_stats_pass<0>, grid, block, 0, st, table, workspace, n_keys); break;
      case 1: hipLaunchKernelGGL(k_stats_pass<1>, grid, block, 0, st, table, workspace, n_keys); break;
      case 2: hipLaunchKernelGGL(k_stats_pass<2>, grid, block, 0, st, table, workspace, n_keys); break;
      default: hipLaunchKernelGGL(k_stats_pass<3>, grid, block, 0, st, table, workspace, n_keys); break;
    }
    PRL_CHECK_LAUNCH("stats_pass");
  }
  return PRL_OK;
}

extern "C" int prl_stats_finish(void* workspace, int n_keys, float* out, int64_t* counts, void* stream) {
  clear_error();
  if (int rc = check_ws(workspace, n_keys, "stats_finish")) return rc;
  PRL_REQUIRE(out, "stats_finish: out must not be NULL");
  hipLaunchKernelGGL(k_stats_finish, dim3((unsigned)n_keys), dim3(kStatThreads), 0, (hipStream_t)stream, workspace,
                     n_keys, out, counts);
  PRL_CHECK_LAUNCH("stats_finish");
  return PRL_OK;
}

extern "C" size_t prl_masked_stats_workspace_bytes(void) { return prl_stats_workspace_bytes(1); }

extern "C" int prl_masked_stats(const float* x, const uint8_t* mask, int64_t n, float* out5, int64_t* count,
                                void* workspace, size_t workspace_bytes, void* stream) {
  clear_error();
  PRL_REQUIRE(n >= 0, "masked_stats: n must be >= 0");
  PRL_REQUIRE(out5, "masked_stats: out5 must not be NULL");
  PRL_REQUIRE(n == 0 || x, "masked_stats: x must not be NULL");
  PRL_REQUIRE(workspace && workspace_bytes >= prl_stats_workspace_bytes(1),
              "masked_stats: workspace of %zu bytes required", prl_stats_workspace_bytes(1));
  prl_stat_piece pc;
  pc.x = x;
  pc.mask = mask;
  pc.n = n;
  pc.key = 0;
  pc.op = PRL_STAT_IDENTITY;
  int rc = prl_stats_begin(workspace, 1, stream);
  for (int level = 0; level < 4 && rc == PRL_OK; ++level) rc = prl_stats_pass(level, &pc, 1, 1, workspace, stream);
  if (rc == PRL_OK) rc = prl_stats_finish(workspace, 1, out5, count, stream);
  return rc;
}
