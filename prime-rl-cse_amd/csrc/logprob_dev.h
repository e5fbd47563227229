// Device helpers shared by the kernels that stream rows of logits: build-time tunables,
// 16-byte streaming loads and stores, the running soft-max statistics, and the per-unit
// gradient arithmetic.
#pragma once
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
#ifndef PRL_FWD_SPLIT_PIECE
#define PRL_FWD_SPLIT_PIECE 0  // > 0: also with many rows the forward splits every row into pieces of about this many elements
#endif
#ifndef PRL_BWD_LOADS
#define PRL_BWD_LOADS 4  // 16-byte loads per lane and iteration in the backward
#endif
#ifndef PRL_BWD_FLAT
#define PRL_BWD_FLAT 1  // 1: flat grid-stride backward when the layout allows it, else one workgroup per row
#endif
#ifndef PRL_BWD_ONCE
#define PRL_BWD_ONCE 1  // 1: the flat backward as one piece per workgroup on a (pieces, rows) grid, no grid stride
#endif
#ifndef PRL_BWD_ONCE_UPT
#define PRL_BWD_ONCE_UPT 1  // work units per thread in that form
#endif
#ifndef PRL_BWD_ONCE_NT_STORE
#define PRL_BWD_ONCE_NT_STORE 1  // non-temporal hint on that form's stores
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
template <bool NT = (PRL_NT_STORE != 0)>
__device__ __forceinline__ void stream_store(u32x4 v, u32x4* p) {
  if constexpr (NT) {
    __builtin_nontemporal_store(v, p);
  } else {
    *p = v;
  }
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

// ---------------------------------------------------------------- gradient units

template <int DT>
__device__ __forceinline__ void store_elem(char* p, int64_t i, float v) {
  if constexpr (DT == PRL_F32) {
    reinterpret_cast<float*>(p)[i] = v;
  } else {
    reinterpret_cast<uint16_t*>(p)[i] = one_bf16(v);
  }
}

// U values -> 16-byte stores
template <int DTO, int U, bool NTS = (PRL_NT_STORE != 0)>
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
      stream_store<NTS>(v, o + q);
    }
  } else {
    static_assert(U == 8 || DTO == PRL_F32, "bf16 output works on units of 8");
    u32x4 v;
    v.x = pack_bf16(d[0], d[1]);
    v.y = pack_bf16(d[2], d[3]);
    v.z = pack_bf16(d[4], d[5]);
    v.w = pack_bf16(d[6], d[7]);
    stream_store<NTS>(v, reinterpret_cast<u32x4*>(outp) + unit);
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
template <int DTI, int DTO, int U, bool NTS = (PRL_NT_STORE != 0)>
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
  store_unit<DTO, U, NTS>(outp, unit, d);
}

}  // namespace prl
