// Causal self-attention over a packed row whose samples have ANY length (head dimension 128, bf16): the rows of
// the reference's wordle, hendrycks_math and deepscaler recipes (samples of 2048 .. 8192 tokens).  The reference
// gets this from flash-attn's variable-length kernel through transformers
// (`attn_implementation = "flash_attention_2"`, src/prime_rl/trainer/model.py:58-60, :251-254); PyTorch-ROCm's own
// variable-length flash kernels reach 0.22 PFLOP/s forward + backward on four samples of 4096 tokens.
//
// Same building blocks as attn_short.hip (scores computed transposed, S^T = K Q^T with MFMA 32x32x16 bf16, so the
// query sits on the lane; operands consumed along the token index come out of row-major LDS images through
// ds_read_b64_tr_b16), tiled flash-style over 128 x 128 blocks of the score matrix:
//   forward  one workgroup (4 waves x 32 queries) per (128-query tile, sample, query head) walks over the key
//            tiles 0 .. its own, K and V tile in LDS, running maximum and sum per query ON THE LANE (rescaling the
//            output accumulators is a multiplication by a per-lane scalar), blocks above the diagonal never touched;
//   dQ       the forward's walk with the stored log-sum-exp instead of the running one (dP^T = V dO^T,
//            dS^T = P^T (dP^T - delta), dQ^T += K^T dS^T); also leaves delta = rowsum(dO * O);
//   dK, dV   two workgroups (halves of the head dimension of the results) per (128-key tile, sample, key/value
//            head): wave w owns keys 32w .. 32w+31 of the tile, key on the lane, and walks over the query tiles from
//            its own to the sample's last, for every query head of the group, sums in registers: no atomics.
// Workgroups with the most blocks to walk start first.
#include "attn_dev.h"

// build-time tunables (tools/tune_attn_long.py)

namespace prl {

// tokens [start, start + len) of sample s, clamped into the row
__device__ __forceinline__ void sample_span(const int32_t* cu, int s, int64_t T, int64_t& start, int& len) {
  int64_t a = cu[s], b = cu[s + 1];
  if (a < 0) a = 0;
  if (b > T) b = T;
  int64_t n = b - a;
  if (n < 0) n = 0;
  if (n > 0x40000000) n = 0x40000000;
  start = a;
  len = (int)n;
}

// Two LDS layouts of a [128 tokens][128 d] tile, both row-major in 16-byte slots (8 elements):
//   kRows   pitch 136 elements (17 slots): for an image that is only read row-wise (ds_read_b128: its four 16-lane
//           groups then hit 16 different slots);
//   kBoth   pitch 160 elements (20 slots) with slot s of row t stored at slot s ^ ((t >> 2) & 3): for an image that
//           is (also) read transposed.  A transposing read's 32-lane group takes 64 bytes from each of four
//           consecutive rows; with 17 slots between rows those four pieces overlap on the banks (up to 4-way
//           conflicts, measured as twice the conflict-free LDS time in these kernels), with 20 they lie side by side
//           (4 x 16 banks), and the swizzle keeps the row-wise reads, whose 16-lane groups hold rows 4 apart,
//           conflict-free as well.  The 64 bytes behind every row's data are free for other things.
enum Layout { kRows = 0, kBoth = 1 };
template <int L> struct Lay;
template <> struct Lay<kRows> { static constexpr int pitch = 136; };
template <> struct Lay<kBoth> { static constexpr int pitch = 160; };
// Per-lane constants of the two kinds of read, worked out once per kernel so that every read is one base register
// plus an immediate: the swizzle of a row depends on (row >> 2) & 3 only, which for the rows a lane touches is the
// same in every 32-row block (row-wise reads: row = 32 mt + r) and in every 16-token step (transposing reads).
struct RowLane {  // row-wise fragment of row 32 mt + r: element offset = 32 mt * pitch + base + 32 (kk >> 1) + (kk & 1 ? odd : even)
  int base, even, odd;
};
template <int L>
__device__ __forceinline__ RowLane row_lane(int r, int h) {
  const int x = L == kBoth ? (r >> 2) & 3 : 0;
  return {r * Lay<L>::pitch, 8 * (h ^ x), 8 * ((2 + h) ^ x)};
}
template <int L>
__device__ __forceinline__ bf16x8 frag_rows_l(const uint16_t* img, const RowLane& t, int mt, int kk) {
  return as_frag(*reinterpret_cast<const u32x4*>(img + 32 * mt * Lay<L>::pitch + t.base + 32 * (kk >> 1) +
                                                 ((kk & 1) ? t.odd : t.even)));
}
struct TrLane {  // transposing fragment at tokens tok0 + 16 s, columns d0: offsets of the lane's two 8-byte pieces
  int lo, hi;
};
__device__ __forceinline__ TrLane tr_lane(int lane) {
  const int i = lane & 15, grp = lane >> 4, g1 = grp >> 1, q = i >> 2;
  const int b = 2 * (grp & 1) + ((i & 3) >> 1), within = 4 * (i & 1);
  constexpr int P = Lay<kBoth>::pitch;
  return {(4 * g1 + q) * P + 8 * (b ^ g1) + within, (4 * g1 + q + 8) * P + 8 * (b ^ ((g1 + 2) & 3)) + within};
}
// (see frag_cols: the same fragment out of a kBoth image; tok0 a multiple of 32, d0 of 32)
__device__ __forceinline__ bf16x8 frag_cols_l(const uint16_t* img, int d0, int tok0, int s, const TrLane& t) {
  const uint16_t* p = img + (tok0 + 16 * s) * Lay<kBoth>::pitch + d0;
  typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + t.lo));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + t.hi));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return __builtin_bit_cast(bf16x8, v);
}
// where thread `tid` of the staging loops puts its 16 bytes of round `it`: token 16 it + tid / 16, slot tid % 16;
// (token >> 2) & 3 is the thread's wave index in every round
template <int L>
__device__ __forceinline__ int stage_at(int it) {
  const int t4 = threadIdx.x >> 4, ch = threadIdx.x & 15, x = L == kBoth ? (threadIdx.x >> 6) & 3 : 0;
  return (16 * it + t4) * Lay<L>::pitch + 8 * (ch ^ x);
}

// rows row0 .. row0+127 of the sample, one head each, into two images of layouts L0 and L1; rows at or past `len`
// are zero.  Reads without a branch (see stage_images); needs len >= 1.
template <int L0, int L1>
__device__ __forceinline__ void stage_pair(uint16_t* img0, uint16_t* img1, const AttnView& x0, const AttnView& x1,
                                           int64_t start, int len, int row0, int head0, int head1) {
  constexpr int kIt = kAL * kAD / 8 / kAThreads;
  u32x4 v[2][kIt];
#pragma unroll
  for (int it = 0; it < kIt; ++it) {
    const int id = it * kAThreads + threadIdx.x, tok = row0 + (id >> 4), ch = id & 15;
    const int src = tok < len ? tok : len - 1;
    v[0][it] = *reinterpret_cast<const u32x4*>(x0.p + (start + src) * x0.ts + head0 * x0.hs + ch * 8);
    v[1][it] = *reinterpret_cast<const u32x4*>(x1.p + (start + src) * x1.ts + head1 * x1.hs + ch * 8);
  }
#pragma unroll
  for (int it = 0; it < kIt; ++it) {
    const int tok = 16 * it + (threadIdx.x >> 4);
    const u32x4 zero = {0u, 0u, 0u, 0u};
    *reinterpret_cast<u32x4*>(img0 + stage_at<L0>(it)) = row0 + tok < len ? v[0][it] : zero;
    *reinterpret_cast<u32x4*>(img1 + stage_at<L1>(it)) = row0 + tok < len ? v[1][it] : zero;
  }
}
// one image
template <int L0>
__device__ __forceinline__ void stage_one(uint16_t* img0, const AttnView& x0, int64_t start, int len, int row0,
                                          int head0) {
  constexpr int kIt = kAL * kAD / 8 / kAThreads;
  u32x4 v[kIt];
#pragma unroll
  for (int it = 0; it < kIt; ++it) {
    const int id = it * kAThreads + threadIdx.x, tok = row0 + (id >> 4), ch = id & 15;
    const int src = tok < len ? tok : len - 1;
    v[it] = *reinterpret_cast<const u32x4*>(x0.p + (start + src) * x0.ts + head0 * x0.hs + ch * 8);
  }
#pragma unroll
  for (int it = 0; it < kIt; ++it) {
    const int tok = 16 * it + (threadIdx.x >> 4);
    const u32x4 zero = {0u, 0u, 0u, 0u};
    *reinterpret_cast<u32x4*>(img0 + stage_at<L0>(it)) = row0 + tok < len ? v[it] : zero;
  }
}

// No LDS read or MFMA may be moved across this point; everything else may.  The product loops below are
// software-pipelined by hand with it: the fragments of step i + 1 are requested before the products of step i are
// issued.  Left alone, the compiler reuses ONE fragment register under this register budget and so puts a whole
// LDS round trip in front of every product.
#define PRL_PIPE_FENCE() __builtin_amdgcn_sched_barrier(0x476)

// One block of the forward walk: NMT 32-key blocks of the staged tile, from block mt0 on, against the wave's 32
// queries: scores, running maximum and sum (on the lane), probabilities, and their product with V added into `ot`.
// MASKED: the block lies on the diagonal or holds the sample's end.
template <int NMT, bool MASKED>
__device__ __forceinline__ void fwd_block(const uint16_t* Ks, const uint16_t* Vs, const bf16x8 (&qf)[8],
                                          f32x16 (&ot)[4], float& m, float& l, int mt0, int k0, int query, int len,
                                          float c, int h, const RowLane& kl, const TrLane& vl) {
  f32x16 st[NMT];
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) st[mt] = f32x16{};
  {
    bf16x8 kf[2][NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) kf[0][mt] = frag_rows_l<kRows>(Ks, kl, mt0 + mt, 0);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      if (kk < 7) {
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) kf[(kk + 1) & 1][mt] = frag_rows_l<kRows>(Ks, kl, mt0 + mt, kk + 1);
      }
      PRL_PIPE_FENCE();
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) st[mt] = mfma(kf[kk & 1][mt], qf[kk], st[mt]);
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MASKED) {
        const int key = k0 + 32 * (mt0 + mt) + acc_row(i, h);
        st[mt][i] = (key <= query && key < len) ? st[mt][i] : -INFINITY;
      }
      mx = fmaxf(mx, st[mt][i]);
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, kWave));
  // finite: every block walked holds a key that is at or before the lane's query and inside the sample
  const float m_new = fmaxf(m, mx);
  if (__any(m_new > m)) {  // (wave-uniform) some query's maximum moved: bring sums and accumulators to the new one
    const float alpha = __builtin_amdgcn_exp2f((m - m_new) * c);  // 0 at the first block (m = -inf), 1 where nothing moved
    l *= alpha;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) ot[dt][i] *= alpha;
    }
  }
  const float mc = m_new * c;
  m = m_new;
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float p = __builtin_amdgcn_exp2f(fmaf(st[mt][i], c, -mc));  // c > 0: a masked score stays -inf, p = 0
      st[mt][i] = p;
      l += p;
    }
  }
  bf16x8 vf[2][4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) vf[0][dt] = frag_cols_l(Vs, 32 * dt, 32 * mt0, 0, vl);
#pragma unroll
  for (int j = 0; j < 2 * NMT; ++j) {  // step j: 16 keys of block mt0 + j / 2
    if (j + 1 < 2 * NMT) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) vf[(j + 1) & 1][dt] = frag_cols_l(Vs, 32 * dt, 32 * (mt0 + (j + 1) / 2), (j + 1) & 1, vl);
    }
    PRL_PIPE_FENCE();
    const bf16x8 pf = acc_frag(st[j >> 1], j & 1);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) ot[dt] = mfma(vf[j & 1][dt], pf, ot[dt]);
  }
}

// lse2 is the base-2 log-sum-exp of the scaled scores, [head][token]
__global__ __launch_bounds__(kAThreads, 2) void k_attn_long_fwd(AttnView q, AttnView k, AttnView v, uint16_t* o,
                                                             int64_t o_ts, int64_t o_hs, float* lse2,
                                                             const int32_t* cu, int Hq, int G, int64_t T, float c,
                                                             int tiles, int groups) {
  extern __shared__ __align__(16) uint16_t lds[];
  uint16_t* Ks = lds;                             // read row-wise only
  uint16_t* Vs = lds + kAL * Lay<kRows>::pitch;   // read transposed only
  int group, within;
  if (!xcd_item(tiles * G, groups, group, within)) return;
  const int Hkv = Hq / G, s = group / Hkv, hkv = group % Hkv, hq = hkv * G + within % G;
  const int qt = tiles - 1 - within / G;  // the tiles with the longest walk first
  int64_t start;
  int len;
  sample_span(cu, s, T, start, len);
  const int q0 = qt * kAL;
  if (q0 >= len) return;  // the whole workgroup, before any barrier
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int query = q0 + 32 * w + r;
  const bool active = q0 + 32 * w < len;  // wave-uniform: this wave has queries (it stages and meets barriers anyway)
  bf16x8 qf[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) qf[kk] = frag_global(q, start, len, hq, query, kk, h);
  float m = -INFINITY, l = 0.0f;  // running maximum (raw scores) and this lane half's share of the running sum
  f32x16 ot[4] = {};
  const RowLane kl = row_lane<kRows>(r, h);
  const TrLane vl = tr_lane(lane);
#pragma unroll 1
  for (int kt = 0; kt <= qt; ++kt) {
    const int k0 = kt * kAL;
    __syncthreads();  // the previous tile's images are no longer read
    stage_pair<kRows, kBoth>(Ks, Vs, k, v, start, len, k0, hkv, hkv);
    __syncthreads();
    if (!active) continue;
    if (kt < qt) {  // below the diagonal: every key is before every query of the tile and inside the sample
      fwd_block<2, false>(Ks, Vs, qf, ot, m, l, 0, k0, query, len, c, h, kl, vl);
      fwd_block<2, false>(Ks, Vs, qf, ot, m, l, 2, k0, query, len, c, h, kl, vl);
    } else {
      int ntile = (len - k0 + 31) >> 5;  // blocks that hold keys at all ...
      if (ntile > w + 1) ntile = w + 1;  // ... and, causal, none beyond the wave's own queries
#pragma unroll 1
      for (int mt = 0; mt < ntile; ++mt) fwd_block<1, true>(Ks, Vs, qf, ot, m, l, mt, k0, query, len, c, h, kl, vl);
    }
  }
  if (!active) return;
  l += __shfl_xor(l, 32, kWave);
  if (query < len) {
    const float inv = 1.0f / l;
    uint16_t* orow = o + (start + query) * o_ts + hq * o_hs;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 pk;
        pk.x = pack_bf16(ot[dt][4 * g + 0] * inv, ot[dt][4 * g + 1] * inv);
        pk.y = pack_bf16(ot[dt][4 * g + 2] * inv, ot[dt][4 * g + 3] * inv);
        *reinterpret_cast<u32x2*>(orow + 32 * dt + 8 * g + 4 * h) = pk;
      }
    }
    if (h == 0) lse2[(int64_t)hq * T + start + query] = m * c + __builtin_amdgcn_logf(l);
  }
}

// dQ, and delta = rowsum(dO * O) [head][token] for the dK / dV kernel
__global__ __launch_bounds__(kAThreads, 2) void k_attn_long_bwd_dq(AttnView q, AttnView k, AttnView v, AttnView o,
                                                                AttnView dout, const float* lse2, float* delta,
                                                                uint16_t* dq, int64_t dq_ts, int64_t dq_hs,
                                                                const int32_t* cu, int Hq, int G, int64_t T, float c,
                                                                float scale, int tiles, int groups) {
  extern __shared__ __align__(16) uint16_t lds[];
  uint16_t* Ks = lds;                             // read row-wise (S) and transposed (dQ)
  uint16_t* Vs = lds + kAL * Lay<kBoth>::pitch;   // read row-wise only
  int group, within;
  if (!xcd_item(tiles * G, groups, group, within)) return;
  const int Hkv = Hq / G, s = group / Hkv, hkv = group % Hkv, hq = hkv * G + within % G;
  const int qt = tiles - 1 - within / G;
  int64_t start;
  int len;
  sample_span(cu, s, T, start, len);
  const int q0 = qt * kAL;
  if (q0 >= len) return;  // the whole workgroup, before any barrier
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int query = q0 + 32 * w + r;
  const bool active = q0 + 32 * w < len;
  bf16x8 qf[8], dof[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    qf[kk] = frag_global(q, start, len, hq, query, kk, h);
    dof[kk] = frag_global(dout, start, len, hq, query, kk, h);
  }
  float dl = 0.0f;
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    const bf16x8 of = frag_global(o, start, len, hq, query, kk, h);
#pragma unroll
    for (int j = 0; j < 8; ++j) dl = fmaf((float)dof[kk][j], (float)of[j], dl);
  }
  dl += __shfl_xor(dl, 32, kWave);
  float ls = 0.0f;
  if (query < len) {
    ls = lse2[(int64_t)hq * T + start + query];
    if (h == 0) delta[(int64_t)hq * T + start + query] = dl;
  }
  const bool real = query < len;
  f32x16 dqt[4] = {};
  const RowLane kl = row_lane<kBoth>(r, h), vl = row_lane<kRows>(r, h);
  const TrLane tl = tr_lane(lane);
#pragma unroll 1
  for (int kt = 0; kt <= qt; ++kt) {
    const int k0 = kt * kAL;
    __syncthreads();
    stage_pair<kBoth, kRows>(Ks, Vs, k, v, start, len, k0, hkv, hkv);
    __syncthreads();
    if (!active) continue;
    const bool diag = kt == qt;
    int ntile = (len - k0 + 31) >> 5;
    if (ntile > 4) ntile = 4;
    if (diag && ntile > w + 1) ntile = w + 1;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (mt < ntile) {
        f32x16 sa = {}, dp = {};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          sa = mfma(frag_rows_l<kBoth>(Ks, kl, mt, kk), qf[kk], sa);
          dp = mfma(frag_rows_l<kRows>(Vs, vl, mt, kk), dof[kk], dp);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = k0 + 32 * mt + acc_row(i, h);
          const bool valid = real && (!diag || (key <= query && key < len));
          const float p = valid ? __builtin_amdgcn_exp2f(fmaf(sa[i], c, -ls)) : 0.0f;
          sa[i] = p * (dp[i] - dl) * scale;  // dS^T
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 df = acc_frag(sa, s2);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) dqt[dt] = mfma(frag_cols_l(Ks, 32 * dt, 32 * mt, s2, tl), df, dqt[dt]);
        }
      }
    }
  }
  if (real) {
    uint16_t* row = dq + (start + query) * dq_ts + hq * dq_hs;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 pk;
        pk.x = pack_bf16(dqt[dt][4 * g + 0], dqt[dt][4 * g + 1]);
        pk.y = pack_bf16(dqt[dt][4 * g + 2], dqt[dt][4 * g + 3]);
        *reinterpret_cast<u32x2*>(row + 32 * dt + 8 * g + 4 * h) = pk;
      }
    }
  }
}

// dK and dV of one 128-key tile of one (sample, key/value head), summed over the G query heads that share it and
// over the query tiles from the key tile's own to the sample's last.  DSPLIT = 2: two workgroups per tile, each
// owning half of the head dimension of both results (64 accumulator registers, as in the short-sample kernel; both
// recompute S and dP).  DSPLIT = 1: one workgroup owns the whole head dimension (128 accumulator registers), S and
// dP are computed once and Q / dO staged once.
#ifndef PRL_ATTN_DKV_SPLIT
#define PRL_ATTN_DKV_SPLIT 1
#endif
#ifndef PRL_ATTN_DKV_ROUNDS
#define PRL_ATTN_DKV_ROUNDS 2
#endif
template <int DSPLIT>
__global__ __launch_bounds__(kAThreads, 2) void k_attn_long_bwd_dkv(AttnView q, AttnView k, AttnView v,
                                                                    AttnView dout, const float* lse2,
                                                                    const float* delta, uint16_t* dk, int64_t dk_ts,
                                                                    int64_t dk_hs, uint16_t* dv, int64_t dv_ts,
                                                                    int64_t dv_hs, const int32_t* cu, int Hkv, int G,
                                                                    int64_t T, float c, float scale, int tiles,
                                                                    int groups) {
  constexpr int ND = 4 / DSPLIT;  // 32-wide blocks of the head dimension this workgroup owns
  extern __shared__ __align__(16) uint16_t lds[];
  uint16_t* Qs = lds;                             // both read row-wise (S, dP) and transposed (dK, dV)
  uint16_t* Os = lds + kAL * Lay<kBoth>::pitch;   // dO
  // log-sum-exp and delta of the tile's 128 query rows live in the 64 bytes behind the data of the images' first
  // eight rows (16 floats per row): value n at row n / 16, float n % 16
  auto lse_at = [&](int n) { return reinterpret_cast<float*>(Qs + (n >> 4) * Lay<kBoth>::pitch + kAD) + (n & 15); };
  auto delta_at = [&](int n) { return reinterpret_cast<float*>(Os + (n >> 4) * Lay<kBoth>::pitch + kAD) + (n & 15); };
  int group, within;
  if (!xcd_item(tiles * DSPLIT, groups, group, within)) return;
  const int dpart = within % DSPLIT, kt = within / DSPLIT;  // the first key tiles have the longest walk
  const int s = group / Hkv, hkv = group % Hkv;
  int64_t start;
  int len;
  sample_span(cu, s, T, start, len);
  const int k0 = kt * kAL;
  if (k0 >= len) return;  // the whole workgroup, before any barrier
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int key = k0 + 32 * w + r;
  const bool active = k0 + 32 * w < len;  // wave-uniform: this wave has keys
  bf16x8 kf[8], vf[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    kf[kk] = frag_global(k, start, len, hkv, key, kk, h);
    vf[kk] = frag_global(v, start, len, hkv, key, kk, h);
  }
  const int nqt = (len + kAL - 1) / kAL;
  f32x16 dkt[ND] = {}, dvt[ND] = {};
  const RowLane rl = row_lane<kBoth>(r, h);
  const TrLane tl = tr_lane(lane);
#pragma unroll 1
  for (int g = 0; g < G; ++g) {
    const int hq = hkv * G + g;
#pragma unroll 1
    for (int qt = kt; qt < nqt; ++qt) {
      const int q0 = qt * kAL;
      __syncthreads();  // the previous images are no longer read
#if PRL_ATTN_DKV_ROUNDS == 1
      stage_pair<kBoth, kBoth>(Qs, Os, q, dout, start, len, q0, hq, hq);  // one round of loads, 64 registers in flight
#else
      stage_one<kBoth>(Qs, q, start, len, q0, hq);  // one image per round of loads: registers are scarce here
      stage_one<kBoth>(Os, dout, start, len, q0, hq);
#endif
      if (threadIdx.x < kAL) {
        const bool in = q0 + (int)threadIdx.x < len;
        *lse_at(threadIdx.x) = in ? lse2[(int64_t)hq * T + start + q0 + threadIdx.x] : 0.0f;
        *delta_at(threadIdx.x) = in ? delta[(int64_t)hq * T + start + q0 + threadIdx.x] : 0.0f;
      }
      __syncthreads();
      if (!active) continue;
      const bool diag = qt == kt;
      int mt_end = (len - q0 + 31) >> 5;  // 32-query blocks of this tile that hold queries at all
      if (mt_end > 4) mt_end = 4;
      const int mt_begin = diag ? w : 0;  // causal: on the diagonal only queries at or after the wave's keys
#pragma unroll 1
      for (int mt = 0; mt < 4; ++mt) {
        if (mt >= mt_begin && mt < mt_end) {
          f32x16 sa = {}, dp = {};
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) {
            sa = mfma(frag_rows_l<kBoth>(Qs, rl, mt, kk), kf[kk], sa);  // S: rows = queries, lane = key
            dp = mfma(frag_rows_l<kBoth>(Os, rl, mt, kk), vf[kk], dp);  // dP = dO V^T
          }
          // log-sum-exp and delta of the 16 query rows this lane half holds (rows 8j + 4h .. + 3, j = 0 .. 3): four
          // 16-byte reads each, and no branch around them.  Rows past the sample's end need no mask: their Q and dO
          // rows are zero in the images, so P there multiplies zeros in dV and dS comes out as zero.
          f32x4 lse_r[4], del_r[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            lse_r[j] = *reinterpret_cast<const f32x4*>(lse_at(32 * mt + 8 * j + 4 * h));
            del_r[j] = *reinterpret_cast<const f32x4*>(delta_at(32 * mt + 8 * j + 4 * h));
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            float p = __builtin_amdgcn_exp2f(fmaf(sa[i], c, -lse_r[i >> 2][i & 3]));
            if (diag) p = key <= q0 + 32 * mt + acc_row(i, h) ? p : 0.0f;  // causal, on the diagonal tile only
            sa[i] = p;
            dp[i] = p * (dp[i] - del_r[i >> 2][i & 3]) * scale;  // dS
          }
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf = acc_frag(sa, s2), df = acc_frag(dp, s2);
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
              const int d0 = 32 * (ND * dpart + dt);
              dvt[dt] = mfma(frag_cols_l(Os, d0, 32 * mt, s2, tl), pf, dvt[dt]);  // dV^T += dO^T P
              dkt[dt] = mfma(frag_cols_l(Qs, d0, 32 * mt, s2, tl), df, dkt[dt]);  // dK^T += Q^T dS
            }
          }
        }
      }
    }
  }
  if (key < len) {
    uint16_t* krow = dk + (start + key) * dk_ts + hkv * dk_hs + 32 * ND * dpart;
    uint16_t* vrow = dv + (start + key) * dv_ts + hkv * dv_hs + 32 * ND * dpart;
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 pk;
        pk.x = pack_bf16(dkt[dt][4 * g + 0], dkt[dt][4 * g + 1]);
        pk.y = pack_bf16(dkt[dt][4 * g + 2], dkt[dt][4 * g + 3]);
        *reinterpret_cast<u32x2*>(krow + 32 * dt + 8 * g + 4 * h) = pk;
        pk.x = pack_bf16(dvt[dt][4 * g + 0], dvt[dt][4 * g + 1]);
        pk.y = pack_bf16(dvt[dt][4 * g + 2], dvt[dt][4 * g + 3]);
        *reinterpret_cast<u32x2*>(vrow + 32 * dt + 8 * g + 4 * h) = pk;
      }
    }
  }
}

constexpr size_t kLongFwdLds = kAL * (Lay<kRows>::pitch + Lay<kBoth>::pitch) * sizeof(uint16_t);
constexpr size_t kLongDqLds = kAL * (Lay<kRows>::pitch + Lay<kBoth>::pitch) * sizeof(uint16_t);
constexpr size_t kLongDkvLds = 2 * kAL * Lay<kBoth>::pitch * sizeof(uint16_t);  // 80 KB: two workgroups fill the LDS exactly

// The kernels use more than the 64 KB of LDS a launch gets by default; raised once per thread and device.
static int raise_long_lds_limits() {
  static thread_local int done_for = -1;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return PRL_ELAUNCH;
  if (done_for == dev) return PRL_OK;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_long_fwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)kLongFwdLds) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_long_bwd_dq),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLongDqLds) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_long_bwd_dkv<PRL_ATTN_DKV_SPLIT>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLongDkvLds) != hipSuccess) {
    set_error("attn_long: cannot raise the dynamic LDS limit: %s", hipGetErrorString(hipGetLastError()));
    return PRL_ELAUNCH;
  }
  done_for = dev;
  return PRL_OK;
}

}  // namespace prl

using namespace prl;

extern "C" int prl_attn_long_supported(int dtype, int64_t head_dim, int64_t max_len, int64_t q_heads,
                                       int64_t kv_heads) {
  return dtype == PRL_BF16 && head_dim == kAD && max_len >= 1 && max_len <= ((int64_t)1 << 24) && kv_heads >= 1 &&
         q_heads >= kv_heads && q_heads % kv_heads == 0;
}

#define PRL_ATTN_LONG_CHECKS(name)                                                                             \
  PRL_REQUIRE(n_samples >= 0 && T >= 0 && q_heads >= 1 && kv_heads >= 1 && q_heads % kv_heads == 0,            \
              name ": bad sizes (samples %d, T %lld, heads %d / %d)", n_samples, (long long)T, q_heads,       \
              kv_heads);                                                                                       \
  PRL_REQUIRE(scale > 0.0f, name ": the score scale must be positive");                                       \
  PRL_REQUIRE(max_len >= 0 && max_len <= T, name ": the longest sample cannot be longer than the row");        \
  if (n_samples == 0 || T == 0 || max_len == 0) return PRL_OK;                                                 \
  PRL_REQUIRE(cu_seqlens, name ": the sample table must not be NULL")

extern "C" int prl_attn_long_fwd(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides,
                                 const void* v, const int64_t* v_strides, void* out, const int64_t* out_strides,
                                 float* lse2, const int32_t* cu_seqlens, int32_t n_samples, int64_t max_len,
                                 int64_t T, int32_t q_heads, int32_t kv_heads, float scale, void* stream) {
  clear_error();
  PRL_ATTN_LONG_CHECKS("attn_long_fwd");
  PRL_REQUIRE(q_strides && k_strides && v_strides && out_strides && lse2, "attn_long_fwd: NULL argument");
  PRL_REQUIRE(attn_view_ok(q, q_strides[0], q_strides[1]) && attn_view_ok(k, k_strides[0], k_strides[1]) &&
                  attn_view_ok(v, v_strides[0], v_strides[1]) && attn_view_ok(out, out_strides[0], out_strides[1]),
              "attn_long_fwd: tensors must be 16-byte aligned with token and head strides that are multiples of 8");
  if (int rc = raise_long_lds_limits()) return rc;
  const AttnView qv{static_cast<const uint16_t*>(q), q_strides[0], q_strides[1]};
  const AttnView kv{static_cast<const uint16_t*>(k), k_strides[0], k_strides[1]};
  const AttnView vv{static_cast<const uint16_t*>(v), v_strides[0], v_strides[1]};
  const int tiles = (int)((max_len + kAL - 1) / kAL), groups = n_samples * kv_heads, G = q_heads / kv_heads;
  PRL_REQUIRE((int64_t)((groups + 7) / 8) * tiles * q_heads < ((int64_t)1 << 28), "attn_long_fwd: too much work for one launch");
  hipLaunchKernelGGL(k_attn_long_fwd, dim3(xcd_grid(tiles * G, groups)), dim3(kAThreads), kLongFwdLds, (hipStream_t)stream, qv, kv, vv,
                     static_cast<uint16_t*>(out), out_strides[0], out_strides[1], lse2, cu_seqlens, q_heads, G, T,
                     scale * kLog2e, tiles, groups);
  PRL_CHECK_LAUNCH("attn_long_fwd");
  return PRL_OK;
}

extern "C" int prl_attn_long_bwd(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides,
                                 const void* v, const int64_t* v_strides, const void* out,
                                 const int64_t* out_strides, const void* dout, const int64_t* dout_strides,
                                 const float* lse2, float* delta, void* dq, const int64_t* dq_strides, void* dk,
                                 const int64_t* dk_strides, void* dv, const int64_t* dv_strides,
                                 const int32_t* cu_seqlens, int32_t n_samples, int64_t max_len, int64_t T,
                                 int32_t q_heads, int32_t kv_heads, float scale, void* stream) {
  clear_error();
  PRL_ATTN_LONG_CHECKS("attn_long_bwd");
  PRL_REQUIRE(q_strides && k_strides && v_strides && out_strides && dout_strides && dq_strides && dk_strides &&
                  dv_strides && lse2 && delta,
              "attn_long_bwd: NULL argument");
  PRL_REQUIRE(attn_view_ok(q, q_strides[0], q_strides[1]) && attn_view_ok(k, k_strides[0], k_strides[1]) &&
                  attn_view_ok(v, v_strides[0], v_strides[1]) && attn_view_ok(out, out_strides[0], out_strides[1]) &&
                  attn_view_ok(dout, dout_strides[0], dout_strides[1]) &&
                  attn_view_ok(dq, dq_strides[0], dq_strides[1]) && attn_view_ok(dk, dk_strides[0], dk_strides[1]) &&
                  attn_view_ok(dv, dv_strides[0], dv_strides[1]),
              "attn_long_bwd: tensors must be 16-byte aligned with token and head strides that are multiples of 8");
  if (int rc = raise_long_lds_limits()) return rc;
  const AttnView qv{static_cast<const uint16_t*>(q), q_strides[0], q_strides[1]};
  const AttnView kv{static_cast<const uint16_t*>(k), k_strides[0], k_strides[1]};
  const AttnView vv{static_cast<const uint16_t*>(v), v_strides[0], v_strides[1]};
  const AttnView ov{static_cast<const uint16_t*>(out), out_strides[0], out_strides[1]};
  const AttnView dov{static_cast<const uint16_t*>(dout), dout_strides[0], dout_strides[1]};
  const int G = q_heads / kv_heads;
  const float c = scale * kLog2e;
  const int tiles = (int)((max_len + kAL - 1) / kAL), groups = n_samples * kv_heads;
  PRL_REQUIRE((int64_t)((groups + 7) / 8) * tiles * q_heads < ((int64_t)1 << 28), "attn_long_bwd: too much work for one launch");
  hipLaunchKernelGGL(k_attn_long_bwd_dq, dim3(xcd_grid(tiles * G, groups)), dim3(kAThreads), kLongDqLds,
                     (hipStream_t)stream, qv, kv, vv, ov, dov, lse2, delta, static_cast<uint16_t*>(dq), dq_strides[0],
                     dq_strides[1], cu_seqlens, q_heads, G, T, c, scale, tiles, groups);
  PRL_CHECK_LAUNCH("attn_long_bwd (dq)");
  hipLaunchKernelGGL(k_attn_long_bwd_dkv<PRL_ATTN_DKV_SPLIT>, dim3(xcd_grid(tiles * PRL_ATTN_DKV_SPLIT, groups)),
                     dim3(kAThreads), kLongDkvLds, (hipStream_t)stream, qv, kv, vv, dov, lse2, delta,
                     static_cast<uint16_t*>(dk), dk_strides[0], dk_strides[1], static_cast<uint16_t*>(dv), dv_strides[0],
                     dv_strides[1], cu_seqlens, kv_heads, G, T, c, scale, tiles, groups);
  PRL_CHECK_LAUNCH("attn_long_bwd (dk, dv)");
  return PRL_OK;
}
