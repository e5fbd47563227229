// Per-token log-probability / entropy over the vocabulary, and the gradient with
// respect to the logits.  Both kernels stream every logit exactly once with
// 16-byte loads per lane and are bound by HBM bandwidth:
//   forward : reads V logits per token, writes 12 bytes
//   backward: reads V logits, writes V gradient elements per token
// The shift by one position and the division by the temperature of the
// reference chain (src/prime_rl/trainer/rl/loss.py:13-58,
// src/prime_rl/trainer/rl/train.py:285-288) are folded into indexing and into
// one multiply, so no shifted or scaled copy of the logits is ever written.
#include "prl_common.h"

// Build-time tunables (the defaults are the measured best on MI355X; the
// tuning script under tools/ builds variants by overriding them).
#ifndef PRL_FWD_NT_LOAD
#define PRL_FWD_NT_LOAD 1  // non-temporal hint on the forward's streaming loads of the logits
#endif
#ifndef PRL_BWD_NT_LOAD
#define PRL_BWD_NT_LOAD 1  // the same for the backward's loads
#endif
#ifndef PRL_NT_STORE
#define PRL_NT_STORE 0  // non-temporal hint on the streaming stores of the gradient
#endif
#ifndef PRL_FWD_REVERSE
#define PRL_FWD_REVERSE 0  // 1: workgroups take the rows last to first
#endif
#ifndef PRL_BWD_REVERSE
#define PRL_BWD_REVERSE 0  // 1: the backward takes the rows last to first (the rows the forward read last are the
                           // ones most likely still in the Infinity Cache)
#endif
#ifndef PRL_FWD_THREADS
#define PRL_FWD_THREADS 256
#endif
#ifndef PRL_BWD_THREADS
#define PRL_BWD_THREADS 256
#endif
#ifndef PRL_FWD_UNR
#define PRL_FWD_UNR 4  // 16-byte loads in flight per lane in the forward
#endif
#ifndef PRL_BWD_LOADS
#define PRL_BWD_LOADS 4  // 16-byte loads per lane and iteration in the backward
#endif
#ifndef PRL_BWD_FLAT
#define PRL_BWD_FLAT 1  // 1: flat grid-stride backward when the layout allows it, else one workgroup per row
#endif
#ifndef PRL_BWD_FLAT_UPT
#define PRL_BWD_FLAT_UPT 2  // work units per thread and piece in the flat backward
#endif
#ifndef PRL_BWD_FLAT_GRID
#define PRL_BWD_FLAT_GRID 4096  // workgroups of the flat backward
#endif

namespace prl {

template <bool NT>
__device__ __forceinline__ u32x4 stream_load(const u32x4* p) {
  if constexpr (NT) {
    return __builtin_nontemporal_load(p);
  } else {
    return *p;
  }
}
__device__ __forceinline__ void stream_store(u32x4 v, u32x4* p) {
#if PRL_NT_STORE
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// Running soft-max statistics of the raw logits x seen so far.  With c = log2(e) / temperature
// (positive) and y = c * x:  m = max y,  s = sum 2^(y-m),  w = sum x 2^(y-m)  (x unscaled, so that
// the inner loop is one fma, one exp, one add and one fma per element).
struct Running {
  float m, s, w;
};

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

template <int N>
__device__ __forceinline__ void absorb(Running& r, const float (&x)[N], float c) {
  float mx = x[0];
#pragma unroll
  for (int i = 1; i < N; ++i) mx = fmaxf(mx, x[i]);
  const float mn = fmaxf(r.m, mx * c);   // c > 0: the largest x gives the largest y
  const float sc = fast_exp2(r.m - mn);  // one rescale per group, not per element
  constexpr int LANES = N < 4 ? N : 4;   // independent accumulation chains
  float es[LANES], ws[LANES];
#pragma unroll
  for (int i = 0; i < LANES; ++i) {
    const float e = fast_exp2(fmaf(x[i], c, -mn));
    es[i] = e;
    ws[i] = x[i] * e;
  }
#pragma unroll
  for (int i = LANES; i < N; ++i) {
    const float e = fast_exp2(fmaf(x[i], c, -mn));
    es[i % LANES] += e;
    ws[i % LANES] = fmaf(x[i], e, ws[i % LANES]);
  }
#pragma unroll
  for (int h = LANES / 2; h > 0; h >>= 1) {
#pragma unroll
    for (int i = 0; i < h; ++i) {
      es[i] += es[i + h];
      ws[i] += ws[i + h];
    }
  }
  r.s = fmaf(r.s, sc, es[0]);
  r.w = fmaf(r.w, sc, ws[0]);
  r.m = mn;
}

__device__ __forceinline__ void absorb_one(Running& r, float x, float c) {
  const float v[1] = {x};
  absorb<1>(r, v, c);
}

__device__ __forceinline__ Running merge(const Running& a, const Running& b) {
  Running o;
  o.m = fmaxf(a.m, b.m);
  const float sa = fast_exp2(a.m - o.m), sb = fast_exp2(b.m - o.m);
  o.s = a.s * sa + b.s * sb;
  o.w = a.w * sa + b.w * sb;
  return o;
}

__device__ __forceinline__ Running wave_merge(Running r) {
  const float M = wave_max(r.m);
  const float sc = fast_exp2(r.m - M);
  r.s = wave_sum(r.s * sc);
  r.w = wave_sum(r.w * sc);
  r.m = M;
  return r;
}

template <int NW>
__device__ __forceinline__ Running block_merge(Running r, float* lds /* 3*NW floats */) {
  r = wave_merge(r);
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  if (lane == 0) {
    lds[wid] = r.m;
    lds[NW + wid] = r.s;
    lds[2 * NW + wid] = r.w;
  }
  __syncthreads();
  Running o = {lds[0], lds[NW], lds[2 * NW]};
#pragma unroll
  for (int i = 1; i < NW; ++i) {
    const Running b = {lds[i], lds[NW + i], lds[2 * NW + i]};
    o = merge(o, b);
  }
  return o;
}

template <int DT>
__device__ __forceinline__ float load_elem(const char* p, int64_t i) {
  if constexpr (DT == PRL_F32) {
    return reinterpret_cast<const float*>(p)[i];
  } else {
    return bf16_one(reinterpret_cast<const uint16_t*>(p)[i]);
  }
}

// 16 bytes of logits -> VE raw values
template <int DT>
__device__ __forceinline__ void decode16_raw(const u32x4& v, float* y) {
  if constexpr (DT == PRL_F32) {
    y[0] = __uint_as_float(v.x);
    y[1] = __uint_as_float(v.y);
    y[2] = __uint_as_float(v.z);
    y[3] = __uint_as_float(v.w);
  } else {
    y[0] = bf16_lo(v.x);
    y[1] = bf16_hi(v.x);
    y[2] = bf16_lo(v.y);
    y[3] = bf16_hi(v.y);
    y[4] = bf16_lo(v.z);
    y[5] = bf16_hi(v.z);
    y[6] = bf16_lo(v.w);
    y[7] = bf16_hi(v.w);
  }
}

// 16 bytes of logits -> VE scaled values y = x * c
template <int DT>
__device__ __forceinline__ void decode16(const u32x4& v, float c, float* y) {
  if constexpr (DT == PRL_F32) {
    y[0] = __uint_as_float(v.x) * c;
    y[1] = __uint_as_float(v.y) * c;
    y[2] = __uint_as_float(v.z) * c;
    y[3] = __uint_as_float(v.w) * c;
  } else {
    y[0] = bf16_lo(v.x) * c;
    y[1] = bf16_hi(v.x) * c;
    y[2] = bf16_lo(v.y) * c;
    y[3] = bf16_hi(v.y) * c;
    y[4] = bf16_lo(v.z) * c;
    y[5] = bf16_hi(v.z) * c;
    y[6] = bf16_lo(v.w) * c;
    y[7] = bf16_hi(v.w) * c;
  }
}

// Fold n elements starting at p (any element alignment) into r, striding the
// workgroup's NT threads over 16-byte vectors; the few elements in front of the
// first and behind the last whole aligned vector go one per thread.
template <int DT, int NT>
__device__ __forceinline__ void reduce_range(const char* p, int64_t n, float c, Running& r) {
  constexpr int ES = DT == PRL_F32 ? 4 : 2;
  constexpr int VE = 16 / ES;
  const int tid = threadIdx.x;
  int64_t head = (int64_t)(((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) / ES);
  if (head > n) head = n;
  const int64_t nvec = (n - head) / VE;
  const int64_t tail0 = head + nvec * VE;
  if (tid < head) absorb_one(r, load_elem<DT>(p, tid), c);
  if (tail0 + tid < n) absorb_one(r, load_elem<DT>(p, tail0 + tid), c);

  const u32x4* vp = reinterpret_cast<const u32x4*>(p + head * ES);
  int64_t vi = tid;
  constexpr int UNR = PRL_FWD_UNR;
  for (; vi + (UNR - 1) * NT < nvec; vi += UNR * NT) {
    u32x4 a[UNR];
#pragma unroll
    for (int q = 0; q < UNR; ++q) a[q] = stream_load<PRL_FWD_NT_LOAD>(vp + vi + q * NT);
    // fold four vectors at a time: one rescale per group keeps the exp count at one per element
#pragma unroll
    for (int q0 = 0; q0 < UNR; q0 += 4) {
      constexpr int GRP = UNR < 4 ? UNR : 4;
      float y[GRP * VE];
#pragma unroll
      for (int q = 0; q < GRP; ++q) decode16_raw<DT>(a[q0 + q], y + q * VE);
      absorb<GRP * VE>(r, y, c);
    }
  }
  for (; vi < nvec; vi += NT) {
    const u32x4 a0 = stream_load<PRL_FWD_NT_LOAD>(vp + vi);
    float y[VE];
    decode16_raw<DT>(a0, y);
    absorb<VE>(r, y, c);
  }
}

// Final values of one token from the merged statistics.
__device__ __forceinline__ void emit_token(const Running& r, float x_label, float inv_temp, int64_t t, float* lp,
                                           float* ent, float* lse_out) {
  const float lse = (r.m + fast_log2(r.s)) * kLn2;
  if (lse_out) lse_out[t] = lse;
  if (lp) lp[t] = x_label * inv_temp - lse;
  if (ent) ent[t] = lse - (r.w / r.s) * inv_temp;  // E[x] / temperature
}

__device__ __forceinline__ void emit_skipped(int64_t t, float* lp, float* ent, float* lse_out) {
  if (lse_out) lse_out[t] = 0.0f;
  if (lp) lp[t] = 0.0f;
  if (ent) ent[t] = 0.0f;
}

__device__ __forceinline__ void emit_zero_row(int64_t V, int64_t t, float* lp, float* ent, float* lse_out) {
  // an all-zero row: every probability is 1/V
  const float l = logf((float)V);
  if (lse_out) lse_out[t] = l;
  if (lp) lp[t] = 0.0f - l;
  if (ent) ent[t] = l;
}

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

template <int DT>
__device__ __forceinline__ void store_elem(char* p, int64_t i, float v) {
  if constexpr (DT == PRL_F32) {
    reinterpret_cast<float*>(p)[i] = v;
  } else {
    reinterpret_cast<uint16_t*>(p)[i] = one_bf16(v);
  }
}

// U values -> 16-byte stores
template <int DTO, int U>
__device__ __forceinline__ void store_unit(char* outp, int64_t unit, const float* d) {
  if constexpr (DTO == PRL_F32) {
    u32x4* o = reinterpret_cast<u32x4*>(outp) + unit * (U / 4);
#pragma unroll
    for (int q = 0; q < U / 4; ++q) {
      u32x4 v;
      v.x = __float_as_uint(d[4 * q + 0]);
      v.y = __float_as_uint(d[4 * q + 1]);
      v.z = __float_as_uint(d[4 * q + 2]);
      v.w = __float_as_uint(d[4 * q + 3]);
      stream_store(v, o + q);
    }
  } else {
    static_assert(U == 8 || DTO == PRL_F32, "bf16 output works on units of 8");
    u32x4 v;
    v.x = pack_bf16(d[0], d[1]);
    v.y = pack_bf16(d[2], d[3]);
    v.z = pack_bf16(d[4], d[5]);
    v.w = pack_bf16(d[6], d[7]);
    stream_store(v, reinterpret_cast<u32x4*>(outp) + unit);
  }
}

template <int DTI, int U>
__device__ __forceinline__ void load_unit(const char* inp, int64_t unit, u32x4* raw) {
  if constexpr (DTI == PRL_F32) {
    const u32x4* p = reinterpret_cast<const u32x4*>(inp) + unit * (U / 4);
#pragma unroll
    for (int q = 0; q < U / 4; ++q) raw[q] = stream_load<PRL_BWD_NT_LOAD>(p + q);
  } else {
    raw[0] = stream_load<PRL_BWD_NT_LOAD>(reinterpret_cast<const u32x4*>(inp) + unit);
  }
}

template <int DTI, int U>
__device__ __forceinline__ void decode_unit(const u32x4* raw, float c, float* y) {
  if constexpr (DTI == PRL_F32) {
#pragma unroll
    for (int q = 0; q < U / 4; ++q) decode16<PRL_F32>(raw[q], c, y + 4 * q);
  } else {
    decode16<PRL_BF16>(raw[0], c, y);
  }
}

// The math of one work unit: d = -gi * 2^(y - lse2), plus gi at the label, then the store.
template <int DTI, int DTO, int U>
__device__ __forceinline__ void bwd_unit(const u32x4* raw, char* outp, int64_t unit, float c, float lse2, float gi,
                                         int64_t label) {
  float y[U], d[U];
  decode_unit<DTI, U>(raw, c, y);
#pragma unroll
  for (int j = 0; j < U; ++j) d[j] = -gi * fast_exp2(y[j] - lse2);
  const uint64_t off = (uint64_t)(label - unit * U);
  if (off < (uint64_t)U) {
#pragma unroll
    for (int j = 0; j < U; ++j) d[j] += (off == (uint64_t)j) ? gi : 0.0f;
  }
  store_unit<DTO, U>(outp, unit, d);
}

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

// grid.x = rows * nsplit; workgroup (row, k) covers elements [k*chunk, (k+1)*chunk) of the row.
template <int DTI, int DTO, int NT>
__global__ __launch_bounds__(NT) void k_logprob_bwd(const char* __restrict__ logits, int64_t T, int64_t V,
                                                    int64_t in_stride_bytes, const int64_t* __restrict__ ids,
                                                    const float* __restrict__ lse, const float* __restrict__ g,
                                                    const float* __restrict__ gscale, float inv_temp, int shift,
                                                    char* __restrict__ out, int64_t out_stride_bytes, int nsplit,
                                                    int64_t chunk) {
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
  if (T <= 0 || T >= kFillBlocks / 2) return 1;
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

extern "C" int prl_logprob_bwd(const void* logits, int logits_dtype, int64_t T, int64_t V, int64_t row_stride,
                               const int64_t* ids, const float* lse, const float* g, const float* gscale,
                               float temperature, int shift, void* dlogits, int out_dtype, int64_t out_row_stride,
                               void* stream) {
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
    if (flat_ok) {
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
                     inv_temp, shift, out, sbo, ns, chunk)
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
