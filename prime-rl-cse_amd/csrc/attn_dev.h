// Device helpers shared by the attention kernels (attn_short.hip: samples of at most 128 tokens; attn_long.hip:
// samples of any length): MFMA 32x32x16 bf16 fragments out of registers, global memory and row-major LDS images
// (the transposing read ds_read_b64_tr_b16 for operands consumed along the token index), and the staging of
// [128][128] tiles of one head into LDS.
#pragma once
#include "prl_common.h"

namespace prl {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kAL = 128;     // longest sample
constexpr int kAD = 128;     // head dimension
#ifndef PRL_ATTN_PITCH
#define PRL_ATTN_PITCH 136
#endif
constexpr int kAP = PRL_ATTN_PITCH;  // LDS row pitch in elements: 272 B, a multiple of 16
constexpr int kAImg = kAL * kAP;  // elements of one [128][128] image
constexpr int kAThreads = 256;

struct AttnView {  // one [token][head][d] tensor of 16-bit elements, d contiguous
  const uint16_t* p;
  int64_t ts, hs;  // strides in elements
};

__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 as_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// accumulator registers 8s .. 8s+7 of a tile as the operand fragment of k-step s of the next product
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& x, int s) {
  u32x4 v;
  v.x = pack_bf16(x[8 * s + 0], x[8 * s + 1]);
  v.y = pack_bf16(x[8 * s + 2], x[8 * s + 3]);
  v.z = pack_bf16(x[8 * s + 4], x[8 * s + 5]);
  v.w = pack_bf16(x[8 * s + 6], x[8 * s + 7]);
  return as_frag(v);
}
// row of an accumulator tile that register i of lane half h holds
__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// rows [0, len) of one head each, tokens start .. start+len-1, into N natural [token][d] images; other rows zero.
// Rows past the end are read from the sample's last row and zeroed by a select, not skipped by a branch: a
// branch per load makes the compiler wait for every load before it issues the next (one memory round trip
// per 16 bytes); this way all of a thread's loads are in flight together.  Needs len >= 1.
template <int N>
__device__ __forceinline__ void stage_images(uint16_t* const (&img)[N], const AttnView* const (&x)[N], int64_t start,
                                             int len, const int (&head)[N]) {
  constexpr int kIt = kAL * kAD / 8 / kAThreads;
  u32x4 v[N][kIt];
#pragma unroll
  for (int n = 0; n < N; ++n) {
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
      const int id = it * kAThreads + threadIdx.x, tok = id >> 4, ch = id & 15;
      const int src = tok < len ? tok : len - 1;
      v[n][it] = *reinterpret_cast<const u32x4*>(x[n]->p + (start + src) * x[n]->ts + head[n] * x[n]->hs + ch * 8);
    }
  }
#pragma unroll
  for (int n = 0; n < N; ++n) {
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
      const int id = it * kAThreads + threadIdx.x, tok = id >> 4, ch = id & 15;
      const u32x4 zero = {0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(img[n] + tok * kAP + ch * 8) = tok < len ? v[n][it] : zero;
    }
  }
}
// fragment of a natural image for a product that sums over d: row `row`, d = 16 kk + 8 h .. + 7
__device__ __forceinline__ bf16x8 frag_rows(const uint16_t* img, int row, int kk, int h) {
  return as_frag(*reinterpret_cast<const u32x4*>(img + row * kAP + 16 * kk + 8 * h));
}
// Fragment of a natural [token][d] image for a product that sums over TOKENS and whose other operand is an
// accumulator tile: lane (r, h) needs column d = d0 + r at tokens tok0 + 16 s + 8 (j >> 2) + 4 h + (j & 3) in
// element j.  ds_read_b64_tr_b16 hands a 16-lane group a block of 4 rows x 16 columns transposed: lane 4q + p of
// the group supplies the address of row q, columns 4p .. 4p+3, and lane i receives column i of the 4 rows.  The
// four groups of a wave are (r < 16, h = 0), (r >= 16, h = 0), (r < 16, h = 1), (r >= 16, h = 1): columns
// d0 + 16 (r >> 4) .. + 15, rows tok0 + 16 s + 4 h .. + 3, and 8 rows further for elements 4 .. 7.
// Needs every lane of the wave active.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 frag_cols(const uint16_t* img, int d0, int tok0, int s, int lane) {
  const int i = lane & 15, grp = lane >> 4;
  const uint16_t* p = img + (tok0 + 16 * s + 4 * (grp >> 1) + (i >> 2)) * kAP + d0 + 16 * (grp & 1) + 4 * (i & 3);
  typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 8 * kAP));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return __builtin_bit_cast(bf16x8, v);
}
// the same fragment of a row read from global memory (token `tok` of the sample, or zeros past its end; read
// without a branch, as in stage_images; needs len >= 1)
__device__ __forceinline__ bf16x8 frag_global(const AttnView& x, int64_t start, int len, int head, int tok, int kk,
                                              int h) {
  const int src = tok < len ? tok : len - 1;
  const u32x4 v = *reinterpret_cast<const u32x4*>(x.p + (start + src) * x.ts + head * x.hs + 16 * kk + 8 * h);
  const u32x4 zero = {0u, 0u, 0u, 0u};
  return as_frag(tok < len ? v : zero);
}

__device__ __forceinline__ void sample_range(const int32_t* cu, int s, int64_t T, int64_t& start, int& len) {
  int64_t a = cu[s], b = cu[s + 1];
  if (a < 0) a = 0;
  if (b > T) b = T;
  int64_t n = b - a;
  if (n < 0) n = 0;
  if (n > kAL) n = kAL;  // the host checks the lengths; this only keeps a bad table inside the buffers
  start = a;
  len = (int)n;
}

// Which piece of work a workgroup of a 1-D grid takes.  Workgroups that read the same tiles (the query heads of a
// group read the same K and V; the halves of a dK / dV tile read the same Q, dO, K and V; in the long-sample kernels
// all query tiles of a group walk over the same K / V tiles) are dealt to ONE XCD, next to each other: the hardware
// hands workgroup i to XCD i % 8, and every XCD has an L2 of its own, so what such workgroups share is read from
// memory once and from that L2 afterwards.  XCD x takes the groups x, x + 8, ... one after the other, `per_group`
// consecutive workgroups each.  False: padding (the grid is 8 * ceil(groups / 8) * per_group workgroups).
__device__ __forceinline__ bool xcd_item(int per_group, int groups, int& group, int& within) {
  const int id = (int)blockIdx.x, slot = id >> 3;
  group = (id & 7) + 8 * (slot / per_group);
  within = slot % per_group;
  return group < groups;
}
static inline unsigned xcd_grid(int per_group, int groups) {
  return 8u * (unsigned)((groups + 7) / 8) * (unsigned)per_group;
}

static inline bool attn_view_ok(const void* p, int64_t ts, int64_t hs) {
  return p && (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ts % 8 == 0 && hs % 8 == 0 && ts > 0 && hs > 0;
}

}  // namespace prl
