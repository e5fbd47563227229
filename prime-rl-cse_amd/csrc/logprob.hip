// Per-token log-probability / entropy over the vocabulary, and the gradient with
// respect to the logits.  Both kernels stream every logit exactly once with
// 16-byte loads per lane and are bound by HBM bandwidth:
//   forward : reads V logits per token, writes 12 bytes
//   backward: reads V logits, writes V gradient elements per token
// The shift by one position and the division by the temperature of the
// reference chain (src/prime_rl/trainer/rl/loss.py:13-58,
// src/prime_rl/trainer/rl/train.py:285-288) are folded into indexing and into
// one multiply, so no shifted or scaled copy of the logits is ever written.
#include <mutex>
#include "logprob_dev.h"

namespace prl {

// One workgroup per token.
template <int DT, int NT>
__global__ __launch_bounds__(NT) void k_logprob_fwd(const char* __restrict__ logits, int64_t V, int64_t stride_bytes,
                                                    const int64_t* __restrict__ ids, float inv_temp, int shift,
                                                    const uint8_t* __restrict__ skip, float* __restrict__ lp,
                                                    float* __restrict__ ent, float* __restrict__ lse_out) {
  constexpr int NW = NT / kWave;
  __shared__ float lds[3 * NW];
#if PRL_FWD_REVERSE
  const int64_t t = (int64_t)gridDim.x - 1 - blockIdx.x;
#else
  const int64_t t = blockIdx.x;
#endif
  if (skip && skip[t] == 0) {  // a token nobody will look at: its row is not read
    if (threadIdx.x == 0) emit_skipped(t, lp, ent, lse_out);
    return;
  }
  if (shift && t == 0) {
    if (threadIdx.x == 0) emit_zero_row(V, t, lp, ent, lse_out);
    return;
  }
  const char* rp = logits + (t - shift) * stride_bytes;
  float x_label = __builtin_nanf("");
  if (threadIdx.x == 0) {
    const int64_t label = ids[t];
    if (label >= 0 && label < V) x_label = load_elem<DT>(rp, label);
  }
  Running r = {kNegBig, 0.0f, 0.0f};
  reduce_range<DT, NT>(rp, V, inv_temp * kLog2e, r);
  r = block_merge<NW>(r, lds);
  if (threadIdx.x == 0) emit_token(r, x_label, inv_temp, t, lp, ent, lse_out);
}

// Few tokens: `nsplit` workgroups share one row; partial statistics go to scratch.
template <int DT, int NT>
__global__ __launch_bounds__(NT) void k_logprob_fwd_part(const char* __restrict__ logits, int64_t V,
                                                         int64_t stride_bytes, int shift, float inv_temp, int nsplit,
                                                         int64_t chunk, const uint8_t* __restrict__ skip,
                                                         float4* __restrict__ part) {
  constexpr int NW = NT / kWave;
  __shared__ float lds[3 * NW];
  const int64_t t = blockIdx.x / nsplit;
  const int k = (int)(blockIdx.x % nsplit);
  if (shift && t == 0) return;
  if (skip && skip[t] == 0) return;
  constexpr int ES = DT == PRL_F32 ? 4 : 2;
  const int64_t e0 = k * chunk;
  int64_t n = V - e0;
  if (n > chunk) n = chunk;
  Running r = {kNegBig, 0.0f, 0.0f};
  if (n > 0) reduce_range<DT, NT>(logits + (t - shift) * stride_bytes + e0 * ES, n, inv_temp * kLog2e, r);
  r = block_merge<NW>(r, lds);
  if (threadIdx.x == 0) part[blockIdx.x] = make_float4(r.m, r.s, r.w, 0.0f);
}

template <int DT>
__global__ void k_logprob_fwd_merge(const char* __restrict__ logits, int64_t T, int64_t V, int64_t stride_bytes,
                                    const int64_t* __restrict__ ids, float inv_temp, int shift, int nsplit,
                                    const uint8_t* __restrict__ skip, const float4* __restrict__ part,
                                    float* __restrict__ lp, float* __restrict__ ent, float* __restrict__ lse_out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  if (skip && skip[t] == 0) {
    emit_skipped(t, lp, ent, lse_out);
    return;
  }
  if (shift && t == 0) {
    emit_zero_row(V, t, lp, ent, lse_out);
    return;
  }
  Running r = {kNegBig, 0.0f, 0.0f};
  for (int k = 0; k < nsplit; ++k) {
    const float4 q = part[t * nsplit + k];
    const Running b = {q.x, q.y, q.z};
    r = merge(r, b);
  }
  const int64_t label = ids[t];
  float x_label = __builtin_nanf("");
  if (label >= 0 && label < V) x_label = load_elem<DT>(logits + (t - shift) * stride_bytes, label);
  emit_token(r, x_label, inv_temp, t, lp, ent, lse_out);
}


// ---------------------------------------------------------------- backward

// Flat decomposition: the [T, V] matrix is cut into pieces of NT * UPT work units that never
// cross a row, and the workgroups walk the pieces with a grid stride, so that at any moment the
// whole grid reads one contiguous window of the logits and writes one contiguous window of the
// gradient (measured: a copy laid out this way moves 7 % more bytes per second than one
// workgroup per row).  Needs every row 16-byte aligned and V a multiple of the unit size.
template <int DTI, int DTO, int NT, int UPT>
__global__ __launch_bounds__(NT) void k_logprob_bwd_flat(const char* __restrict__ logits, int64_t T, int64_t V,
                                                         int64_t in_stride_bytes, const int64_t* __restrict__ ids,
                                                         const float* __restrict__ lse, const float* __restrict__ g,
                                                         const float* __restrict__ gscale, float inv_temp, int shift,
                                                         char* __restrict__ out, int64_t out_stride_bytes,
                                                         int64_t pieces_per_row, int64_t total_pieces) {
  constexpr int U = (DTI == PRL_BF16 || DTO == PRL_BF16) ? 8 : 4;
  constexpr int RAW = DTI == PRL_F32 ? U / 4 : 1;
  const int tid = threadIdx.x;
  const int64_t nunit = V / U;
  const float c = inv_temp * kLog2e;
  const float gs = (gscale ? gscale[0] : 1.0f) * inv_temp;
  int64_t row = blockIdx.x / pieces_per_row, piece = blockIdx.x % pieces_per_row;
  const int64_t drow = gridDim.x / pieces_per_row, dpiece = gridDim.x % pieces_per_row;
  for (int64_t p = blockIdx.x; p < total_pieces; p += gridDim.x) {
    const int64_t t = row + shift;
    const float gi = t < T ? g[t] * gs : 0.0f;
    const char* inp = logits + row * in_stride_bytes;
    char* outp = out + row * out_stride_bytes;
    const int64_t u0 = piece * (int64_t)(NT * UPT) + tid;
    if (gi == 0.0f) {
      float z[U];
#pragma unroll
      for (int j = 0; j < U; ++j) z[j] = 0.0f;
#pragma unroll
      for (int q = 0; q < UPT; ++q)
        if (u0 + q * NT < nunit) store_unit<DTO, U>(outp, u0 + q * NT, z);
    } else {
      const float lse2 = lse[t] * kLog2e;
      const int64_t label = ids[t];
      u32x4 raw[UPT][RAW];
#pragma unroll
      for (int q = 0; q < UPT; ++q)
        if (u0 + q * NT < nunit) load_unit<DTI, U>(inp, u0 + q * NT, raw[q]);
#pragma unroll
      for (int q = 0; q < UPT; ++q)
        if (u0 + q * NT < nunit) bwd_unit<DTI, DTO, U>(raw[q], outp, u0 + q * NT, c, lse2, gi, label);
    }
    piece += dpiece;
    row += drow;
    if (piece >= pieces_per_row) {
      piece -= pieces_per_row;
      ++row;
    }
  }
}

// The same pieces, one per workgroup on a (pieces per row, rows) grid: a workgroup handles NT * UPT units and ends.
// Workgroups start in the order of their index (x fastest), so the grid as a whole still sweeps the logits and the
// gradient front to back, and no wave lives longer than one load and one store per unit.
template <int DTI, int DTO, int NT, int UPT>
__global__ __launch_bounds__(NT) void k_logprob_bwd_once(const char* __restrict__ logits, int64_t T, int64_t V,
                                                         int64_t in_stride_bytes, const int64_t* __restrict__ ids,
                                                         const float* __restrict__ lse, const float* __restrict__ g,
                                                         const float* __restrict__ gscale, float inv_temp, int shift,
                                                         char* __restrict__ out, int64_t out_stride_bytes) {
  constexpr int U = (DTI == PRL_BF16 || DTO == PRL_BF16) ? 8 : 4;
  constexpr int RAW = DTI == PRL_F32 ? U / 4 : 1;
  const int64_t nunit = V / U;
  const int64_t row = blockIdx.y;
  const int64_t t = row + shift;
  const float gi = t < T ? g[t] * ((gscale ? gscale[0] : 1.0f) * inv_temp) : 0.0f;
  const char* inp = logits + row * in_stride_bytes;
  char* outp = out + row * out_stride_bytes;
  const int64_t u0 = (int64_t)blockIdx.x * (NT * UPT) + threadIdx.x;
  if (gi == 0.0f) {
    float z[U];
#pragma unroll
    for (int j = 0; j < U; ++j) z[j] = 0.0f;
#pragma unroll
    for (int q = 0; q < UPT; ++q)
      if (u0 + q * NT < nunit) store_unit<DTO, U, PRL_BWD_ONCE_NT_STORE != 0>(outp, u0 + q * NT, z);
    return;
  }
  const float c = inv_temp * kLog2e;
  const float lse2 = lse[t] * kLog2e;
  const int64_t label = ids[t];
  u32x4 raw[UPT][RAW];
#pragma unroll
  for (int q = 0; q < UPT; ++q)
    if (u0 + q * NT < nunit) load_unit<DTI, U>(inp, u0 + q * NT, raw[q]);
#pragma unroll
  for (int q = 0; q < UPT; ++q)
    if (u0 + q * NT < nunit) bwd_unit<DTI, DTO, U, PRL_BWD_ONCE_NT_STORE != 0>(raw[q], outp, u0 + q * NT, c, lse2, gi, label);
}

// grid.x = rows * nsplit; workgroup (row, k) covers elements [k*chunk, (k+1)*chunk) of the row.
template <int DTI, int DTO, int NT>
__global__ __launch_bounds__(NT) void k_logprob_bwd(const char* __restrict__ logits, int64_t T, int64_t V,
                                                    int64_t in_stride_bytes, const int64_t* __restrict__ ids,
                                                    const float* __restrict__ lse, const float* __restrict__ g,
                                                    const float* __restrict__ gscale, float inv_temp, int shift,
                                                    char* __restrict__ out, int64_t out_stride_bytes, int nsplit,
                                                    int64_t chunk, const uint8_t* __restrict__ row_select) {
  constexpr int ESI = DTI == PRL_F32 ? 4 : 2;
  constexpr int ESO = DTO == PRL_F32 ? 4 : 2;
  constexpr int U = (DTI == PRL_BF16 || DTO == PRL_BF16) ? 8 : 4;  // elements per work unit
  constexpr int RAW = DTI == PRL_F32 ? U / 4 : 1;                  // 16-byte loads per unit
  const int tid = threadIdx.x;
#if PRL_BWD_REVERSE
  const int64_t row = T - 1 - blockIdx.x / nsplit;
#else
  const int64_t row = blockIdx.x / nsplit;
#endif
  if (row_select && row_select[row] == 0) return;  // a row somebody else has written already: left as it is
  const int k = (int)(blockIdx.x % nsplit);
  const int64_t e0 = k * chunk;
  int64_t n = V - e0;
  if (n > chunk) n = chunk;
  if (n <= 0) return;
  const int64_t t = row + shift;

  float gi = 0.0f;  // g[t] / temperature; 0 for the row that shifting drops
  if (t < T) {
    gi = g[t];
    if (gscale) gi *= gscale[0];
    gi *= inv_temp;
  }
  const char* inp = logits + row * in_stride_bytes + e0 * ESI;
  char* outp = out + row * out_stride_bytes + e0 * ESO;
  const bool aligned = (((reinterpret_cast<uintptr_t>(inp) | reinterpret_cast<uintptr_t>(outp)) & 15) == 0);
  const int64_t nunit = aligned ? n / U : 0;

  if (gi == 0.0f) {
    // nothing to learn from this token: zeros, and the logits are not read
    float z[U];
#pragma unroll
    for (int j = 0; j < U; ++j) z[j] = 0.0f;
    for (int64_t u = tid; u < nunit; u += NT) store_unit<DTO, U>(outp, u, z);
    for (int64_t i = nunit * U + tid; i < n; i += NT) store_elem<DTO>(outp, i, 0.0f);
    return;
  }

  const float c = inv_temp * kLog2e;
  const float lse2 = lse[t] * kLog2e;
  const int64_t label = ids[t] - e0;  // relative to this workgroup's range; may be out of it
  const float ngi = -gi;

  auto finish_unit = [&](int64_t u, const u32x4* raw) {
    float y[U], d[U];
    decode_unit<DTI, U>(raw, c, y);
#pragma unroll
    for (int j = 0; j < U; ++j) d[j] = ngi * fast_exp2(y[j] - lse2);
    const uint64_t off = (uint64_t)(label - u * U);
    if (off < (uint64_t)U) {
#pragma unroll
      for (int j = 0; j < U; ++j) d[j] += (off == (uint64_t)j) ? gi : 0.0f;
    }
    store_unit<DTO, U>(outp, u, d);
  };

  constexpr int UNR = PRL_BWD_LOADS / RAW;  // units per iteration
  constexpr int64_t STEP = (int64_t)UNR * NT;
  int64_t u = tid;
  for (; u + (UNR - 1) * NT < nunit; u += STEP) {
    u32x4 raw[UNR][RAW];
#pragma unroll
    for (int q = 0; q < UNR; ++q) load_unit<DTI, U>(inp, u + q * NT, raw[q]);
    __builtin_amdgcn_sched_barrier(0);  // all loads issue before the first result is touched
#pragma unroll
    for (int q = 0; q < UNR; ++q) finish_unit(u + q * NT, raw[q]);
  }
  for (; u < nunit; u += NT) {
    u32x4 raw[RAW];
    load_unit<DTI, U>(inp, u, raw);
    finish_unit(u, raw);
  }
  // ragged end, or the whole range when a row pointer is not 16-byte aligned
  for (int64_t i = nunit * U + tid; i < n; i += NT) {
    const float y = load_elem<DTI>(inp, i) * c;
    float d = ngi * fast_exp2(y - lse2);
    if (i == label) d += gi;
    store_elem<DTO>(outp, i, d);
  }
}

// ---------------------------------------------------------------- host side

constexpr int kFwdThreads = PRL_FWD_THREADS;
constexpr int kBwdThreads = PRL_BWD_THREADS;
constexpr int64_t kFillBlocks = 2048;   // workgroups that keep 256 CUs x 8 busy
constexpr int64_t kMinChunk = 8192;     // do not split a row into pieces smaller than this
constexpr int kMaxSplit = 64;

static int pick_split(int64_t T, int64_t V) {
  if (T <= 0) return 1;
  if (T >= kFillBlocks / 2) {
#if PRL_FWD_SPLIT_PIECE > 0
    // many rows: the chip is full either way; pieces of about one trip per workgroup keep every wave short-lived
    int64_t want = (V + PRL_FWD_SPLIT_PIECE - 1) / PRL_FWD_SPLIT_PIECE;
    if (want > kMaxSplit) want = kMaxSplit;
    return want < 1 ? 1 : (int)want;
#else
    return 1;
#endif
  }
  int64_t want = (kFillBlocks + T - 1) / T;
  int64_t cap = V / kMinChunk;
  if (cap < 1) cap = 1;
  if (want > cap) want = cap;
  if (want > kMaxSplit) want = kMaxSplit;
  return (int)want;
}

static int64_t chunk_for(int64_t V, int nsplit) {
  int64_t c = (V + nsplit - 1) / nsplit;
  return (c + 63) / 64 * 64;  // keeps every piece 16-byte aligned relative to the row
}

}  // namespace prl

using namespace prl;

extern "C" size_t prl_logprob_workspace_bytes(int64_t T, int64_t V) {
  const int ns = pick_split(T, V);
  return ns > 1 ? (size_t)T * ns * sizeof(float4) : 0;
}

namespace prl {
int logprob_fwd_impl(const void* logits, int logits_dtype, int64_t T, int64_t V, int64_t row_stride,
                     const int64_t* ids, float temperature, int shift, const uint8_t* skip, float* logprob,
                     float* entropy, float* lse, void* workspace, size_t workspace_bytes, void* stream);
}

extern "C" int prl_logprob_entropy_fwd(const void* logits, int logits_dtype, int64_t T, int64_t V,
                                       int64_t row_stride, const int64_t* ids, float temperature, int shift,
                                       float* logprob, float* entropy, float* lse, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  return logprob_fwd_impl(logits, logits_dtype, T, V, row_stride, ids, temperature, shift, nullptr, logprob, entropy,
                          lse, workspace, workspace_bytes, stream);
}

// ---------------------------------------------------------------- the entry point that allocates nothing
//
// Same kernels as prl_logprob_entropy_fwd; the partial soft-max states of rows split over several workgroups go
// to a fixed array in device memory instead of a caller workspace.  Every GPU the library runs on holds its own
// copy of the array, at its own address, until the process ends.
constexpr size_t kStaticScratchBytes = 49152;  // 3072 partial states: T * nsplit stays below 2048 + T, T < 1024
constexpr int kMaxDevices = 64;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
static std::mutex g_scratch_mutex;
static void* g_scratch_ptr[kMaxDevices] = {};  // the array's device address, per device; `scratch` itself is only
                                               // a host-side handle and must never be passed to a kernel

extern "C" size_t prl_logprob_static_scratch_bytes(void) { return kStaticScratchBytes; }

extern "C" int prl_logprob_entropy_fwd_static(const void* logits, int logits_dtype, int64_t T, int64_t V,
                                              int64_t row_stride, const int64_t* ids, float temperature, int shift,
                                              float* logprob, float* entropy, float* lse, void* stream) {
  clear_error();
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    (void)hipGetLastError();
    set_error("logprob_entropy_fwd_static: no current device, or its index %d is beyond %d", dev, kMaxDevices - 1);
    return PRL_ELAUNCH;
  }
  void* ptr = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    ptr = g_scratch_ptr[dev];
    if (!ptr) {  // first call on this device: ask the runtime where this device's copy of the array lives
      const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
      if (e != hipSuccess || !ptr) {
        (void)hipGetLastError();
        set_error("logprob_entropy_fwd_static: hipGetSymbolAddress failed on device %d: %s", dev,
                  hipGetErrorString(e));
        return PRL_ELAUNCH;
      }
      g_scratch_ptr[dev] = ptr;
    }
  }
  if (T >= kFillBlocks / 2)  // many rows: one workgroup per row, which needs no scratch at all
    return logprob_fwd_impl(logits, logits_dtype, T, V, row_stride, ids, temperature, shift, nullptr, logprob, entropy,
                            lse, nullptr, 0, stream);
  const size_t need = prl_logprob_workspace_bytes(T, V);
  if (need > kStaticScratchBytes) {
    set_error("logprob_entropy_fwd_static: T=%lld V=%lld needs %zu bytes of scratch, the static array holds %zu; "
              "use prl_logprob_entropy_fwd with a workspace of the caller's",
              (long long)T, (long long)V, need, kStaticScratchBytes);
    return PRL_EINVAL;
  }
  // nothing in the array has to start from zero: every partial state the merge kernel reads was written by the
  // launch before it
  return logprob_fwd_impl(logits, logits_dtype, T, V, row_stride, ids, temperature, shift, nullptr, logprob, entropy,
                          lse, ptr, kStaticScratchBytes, stream);
}

// `skip`: optional 0/1 per token; tokens with 0 are not scored (their row is not read, their outputs are 0)
int prl::logprob_fwd_impl(const void* logits, int logits_dtype, int64_t T, int64_t V, int64_t row_stride,
                          const int64_t* ids, float temperature, int shift, const uint8_t* skip, float* logprob,
                          float* entropy, float* lse, void* workspace, size_t workspace_bytes, void* stream) {
  clear_error();
  PRL_REQUIRE(T >= 0 && V >= 1, "logprob_entropy_fwd: need T >= 0 and V >= 1 (got T=%lld V=%lld)", (long long)T,
              (long long)V);
  if (logits_dtype != PRL_F32 && logits_dtype != PRL_BF16) {
    set_error("logprob_entropy_fwd: unsupported logits dtype code %d", logits_dtype);
    return PRL_EDTYPE;
  }
  if (T == 0) return PRL_OK;
  PRL_REQUIRE(logits && ids, "logprob_entropy_fwd: logits and ids must not be NULL");
  PRL_REQUIRE(row_stride >= V, "logprob_entropy_fwd: row_stride %lld < V %lld", (long long)row_stride, (long long)V);
  PRL_REQUIRE(temperature > 0.0f && temperature - temperature == 0.0f,
              "logprob_entropy_fwd: temperature must be finite and positive");
  PRL_REQUIRE(shift == 0 || shift == 1, "logprob_entropy_fwd: shift must be 0 or 1");
  PRL_REQUIRE(T < (int64_t)1 << 31, "logprob_entropy_fwd: T too large");
  const int es = logits_dtype == PRL_F32 ? 4 : 2;
  PRL_REQUIRE((reinterpret_cast<uintptr_t>(logits) % es) == 0, "logprob_entropy_fwd: logits not element-aligned");
  hipStream_t st = (hipStream_t)stream;
  const float inv_temp = 1.0f / temperature;
  const int64_t sb = row_stride * es;
  const char* lg = (const char*)logits;

  int ns = pick_split(T, V);
  if (ns > 1 && (!workspace || workspace_bytes < (size_t)T * ns * sizeof(float4) ||
                 (reinterpret_cast<uintptr_t>(workspace) & 15)))
    ns = 1;  // no (usable) scratch: one workgroup per row is still correct
  if (ns == 1) {
    if (logits_dtype == PRL_F32)
      hipLaunchKernelGGL((k_logprob_fwd<PRL_F32, kFwdThreads>), dim3((unsigned)T), dim3(kFwdThreads), 0, st, lg, V, sb,
                         ids, inv_temp, shift, skip, logprob, entropy, lse);
    else
      hipLaunchKernelGGL((k_logprob_fwd<PRL_BF16, kFwdThreads>), dim3((unsigned)T), dim3(kFwdThreads), 0, st, lg, V,
                         sb, ids, inv_temp, shift, skip, logprob, entropy, lse);
    PRL_CHECK_LAUNCH("logprob_entropy_fwd");
    return PRL_OK;
  }
  const int64_t chunk = chunk_for(V, ns);
  float4* part = (float4*)workspace;
  const unsigned nblk = (unsigned)(T * ns);
  const unsigned mblk = (unsigned)((T + 255) / 256);
  if (logits_dtype == PRL_F32) {
    hipLaunchKernelGGL((k_logprob_fwd_part<PRL_F32, kFwdThreads>), dim3(nblk), dim3(kFwdThreads), 0, st, lg, V, sb,
                       shift, inv_temp, ns, chunk, skip, part);
    PRL_CHECK_LAUNCH("logprob_entropy_fwd(part)");
    hipLaunchKernelGGL((k_logprob_fwd_merge<PRL_F32>), dim3(mblk), dim3(256), 0, st, lg, T, V, sb, ids, inv_temp,
                       shift, ns, skip, part, logprob, entropy, lse);
  } else {
    hipLaunchKernelGGL((k_logprob_fwd_part<PRL_BF16, kFwdThreads>), dim3(nblk), dim3(kFwdThreads), 0, st, lg, V, sb,
                       shift, inv_temp, ns, chunk, skip, part);
    PRL_CHECK_LAUNCH("logprob_entropy_fwd(part)");
    hipLaunchKernelGGL((k_logprob_fwd_merge<PRL_BF16>), dim3(mblk), dim3(256), 0, st, lg, T, V, sb, ids, inv_temp,
                       shift, ns, skip, part, logprob, entropy, lse);
  }
  PRL_CHECK_LAUNCH("logprob_entropy_fwd(merge)");
  return PRL_OK;
}

static int logprob_bwd_impl(const void* logits, int logits_dtype, int64_t T, int64_t V, int64_t row_stride,
                            const int64_t* ids, const float* lse, const float* g, const float* gscale,
                            float temperature, int shift, void* dlogits, int out_dtype, int64_t out_row_stride,
                            const uint8_t* row_select, void* stream);

extern "C" int prl_logprob_bwd(const void* logits, int logits_dtype, int64_t T, int64_t V, int64_t row_stride,
                               const int64_t* ids, const float* lse, const float* g, const float* gscale,
                               float temperature, int shift, void* dlogits, int out_dtype, int64_t out_row_stride,
                               void* stream) {
  return logprob_bwd_impl(logits, logits_dtype, T, V, row_stride, ids, lse, g, gscale, temperature, shift, dlogits,
                          out_dtype, out_row_stride, nullptr, stream);
}

// Only the rows r with row_select[r] != 0 are written; the others keep what they hold.
extern "C" int prl_logprob_bwd_selected(const void* logits, int logits_dtype, int64_t T, int64_t V,
                                        int64_t row_stride, const int64_t* ids, const float* lse, const float* g,
                                        const float* gscale, float temperature, int shift, void* dlogits,
                                        int out_dtype, int64_t out_row_stride, const uint8_t* row_select,
                                        void* stream) {
  if (!row_select) {
    clear_error();
    set_error("logprob_bwd_selected: row_select must not be NULL");
    return PRL_EINVAL;
  }
  return logprob_bwd_impl(logits, logits_dtype, T, V, row_stride, ids, lse, g, gscale, temperature, shift, dlogits,
                          out_dtype, out_row_stride, row_select, stream);
}

static int logprob_bwd_impl(const void* logits, int logits_dtype, int64_t T, int64_t V, int64_t row_stride,
                            const int64_t* ids, const float* lse, const float* g, const float* gscale,
                            float temperature, int shift, void* dlogits, int out_dtype, int64_t out_row_stride,
                            const uint8_t* row_select, void* stream) {
  clear_error();
  PRL_REQUIRE(T >= 0 && V >= 1, "logprob_bwd: need T >= 0 and V >= 1 (got T=%lld V=%lld)", (long long)T,
              (long long)V);
  if ((logits_dtype != PRL_F32 && logits_dtype != PRL_BF16) || (out_dtype != PRL_F32 && out_dtype != PRL_BF16)) {
    set_error("logprob_bwd: unsupported dtype codes in=%d out=%d", logits_dtype, out_dtype);
    return PRL_EDTYPE;
  }
  if (T == 0) return PRL_OK;
  PRL_REQUIRE(logits && ids && lse && g && dlogits, "logprob_bwd: logits, ids, lse, g, dlogits must not be NULL");
  PRL_REQUIRE(row_stride >= V && out_row_stride >= V, "logprob_bwd: row strides must be >= V");
  PRL_REQUIRE(temperature > 0.0f && temperature - temperature == 0.0f,
              "logprob_bwd: temperature must be finite and positive");
  PRL_REQUIRE(shift == 0 || shift == 1, "logprob_bwd: shift must be 0 or 1");
  const int esi = logits_dtype == PRL_F32 ? 4 : 2, eso = out_dtype == PRL_F32 ? 4 : 2;
  PRL_REQUIRE((reinterpret_cast<uintptr_t>(logits) % esi) == 0 && (reinterpret_cast<uintptr_t>(dlogits) % eso) == 0,
              "logprob_bwd: buffers not element-aligned");
  int ns = pick_split(T, V);
  PRL_REQUIRE(T * (int64_t)ns < (int64_t)1 << 31, "logprob_bwd: T too large");
  hipStream_t st = (hipStream_t)stream;
  const float inv_temp = 1.0f / temperature;
  const char* lg = (const char*)logits;
  char* out = (char*)dlogits;
  const int64_t sbi = row_stride * esi, sbo = out_row_stride * eso;
#if PRL_BWD_FLAT
  {
    const int unit = (esi == 2 || eso == 2) ? 8 : 4;
    const bool flat_ok = V % unit == 0 && ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(dlogits) |
                                            (uintptr_t)sbi | (uintptr_t)sbo) & 15) == 0;
    // (with a selection of rows: one workgroup per row, which looks at its row's mark and leaves at once.  The
    // selection is what the one-pass sequence-mode kernel left over, normally nothing: 2048 workgroups that leave
    // at once cost ~3 us, the 150 K of the piece grid ~30 us.)
    if (flat_ok && !row_select) {
#if PRL_BWD_ONCE
      if (T <= 65535) {  // (the grid's second dimension holds the rows)
        constexpr int OUPT = PRL_BWD_ONCE_UPT;
        const int64_t oppr = (V / unit + (int64_t)kBwdThreads * OUPT - 1) / ((int64_t)kBwdThreads * OUPT);
        const dim3 ogrid((unsigned)oppr, (unsigned)T), oblock(kBwdThreads);
#define PRL_BWDO(DI, DO)                                                                                          \
  hipLaunchKernelGGL((k_logprob_bwd_once<DI, DO, kBwdThreads, OUPT>), ogrid, oblock, 0, st, lg, T, V, sbi, ids, lse, \
                     g, gscale, inv_temp, shift, out, sbo)
        if (logits_dtype == PRL_F32 && out_dtype == PRL_F32)
          PRL_BWDO(PRL_F32, PRL_F32);
        else if (logits_dtype == PRL_F32)
          PRL_BWDO(PRL_F32, PRL_BF16);
        else if (out_dtype == PRL_F32)
          PRL_BWDO(PRL_BF16, PRL_F32);
        else
          PRL_BWDO(PRL_BF16, PRL_BF16);
#undef PRL_BWDO
        PRL_CHECK_LAUNCH("logprob_bwd(once)");
        return PRL_OK;
      }
#endif
      constexpr int UPT = PRL_BWD_FLAT_UPT;
      const int64_t per_piece = (int64_t)kBwdThreads * UPT;
      const int64_t ppr = (V / unit + per_piece - 1) / per_piece;
      const int64_t total = T * ppr;
      const dim3 fgrid((unsigned)(total < PRL_BWD_FLAT_GRID ? total : PRL_BWD_FLAT_GRID)), fblock(kBwdThreads);
#define PRL_BWDF(DI, DO)                                                                                          \
  hipLaunchKernelGGL((k_logprob_bwd_flat<DI, DO, kBwdThreads, UPT>), fgrid, fblock, 0, st, lg, T, V, sbi, ids, lse, \
                     g, gscale, inv_temp, shift, out, sbo, ppr, total)
      if (logits_dtype == PRL_F32 && out_dtype == PRL_F32)
        PRL_BWDF(PRL_F32, PRL_F32);
      else if (logits_dtype == PRL_F32)
        PRL_BWDF(PRL_F32, PRL_BF16);
      else if (out_dtype == PRL_F32)
        PRL_BWDF(PRL_BF16, PRL_F32);
      else
        PRL_BWDF(PRL_BF16, PRL_BF16);
#undef PRL_BWDF
      PRL_CHECK_LAUNCH("logprob_bwd(flat)");
      return PRL_OK;
    }
  }
#endif
  const int64_t chunk = chunk_for(V, ns);
  const dim3 grid((unsigned)(T * ns)), block(kBwdThreads);
#define PRL_BWD(DI, DO)                                                                                             \
  hipLaunchKernelGGL((k_logprob_bwd<DI, DO, kBwdThreads>), grid, block, 0, st, lg, T, V, sbi, ids, lse, g, gscale, \
                     inv_temp, shift, out, sbo, ns, chunk, row_select)
  if (logits_dtype == PRL_F32 && out_dtype == PRL_F32)
    PRL_BWD(PRL_F32, PRL_F32);
  else if (logits_dtype == PRL_F32)
    PRL_BWD(PRL_F32, PRL_BF16);
  else if (out_dtype == PRL_F32)
    PRL_BWD(PRL_BF16, PRL_F32);
  else
    PRL_BWD(PRL_BF16, PRL_BF16);
#undef PRL_BWD
  PRL_CHECK_LAUNCH("logprob_bwd");
  return PRL_OK;
}
