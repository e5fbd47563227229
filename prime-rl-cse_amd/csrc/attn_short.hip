// Causal self-attention over a packed row whose samples are SHORT (at most 128 tokens, head dimension 128):
// the rows of the reference's reverse_text recipe (seq_len 128) and of any recipe with short completions.
// The reference gets its attention from flash-attn's variable-length kernel through transformers
// (`attn_implementation = "flash_attention_2"`, src/prime_rl/trainer/model.py:58-60, :251-254); flash-style
// kernels tile long sequences, and on 256 sequences of 128 tokens they are bound by their own latency
// (49 us forward, 77 us backward per layer measured with PyTorch-ROCm's on this row shape).  A sample this
// short needs no tiling at all: its whole K and V fit the LDS of one compute unit, the whole score matrix
// of one head fits the registers of four waves, and the soft-max is a plain two-pass one.
//
// Forward: one workgroup (4 waves) per (sample, query head); wave w owns queries 32w .. 32w+31 and, the
// mask being causal, key tiles 0 .. w.  Scores are computed TRANSPOSED (S^T = K Q^T, MFMA 32x32x16 bf16):
// the query then sits on the lane and the keys in the accumulator registers, so row maxima and sums are
// in-register reductions plus one exchange between the two lane halves, and the accumulators are already
// the B operand of O^T = V^T P^T (rows of an accumulator tile = the k index of the next product).
// Backward: one kernel per (sample, key/value head) for rows with many samples (k_attn_short_bwd_fused, below),
// else two kernels.  `dq` has the forward's shape (S^T, dP^T = V dO^T, dS^T, dQ^T = K^T dS^T); `dkv`
// runs two workgroups per (sample, key/value head), one per half of the head dimension of the results, wave w
// owning keys 32w .. 32w+31, with the key on the
// lane (S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS) and sums over the query heads of the group
// in registers, so nothing is added across workgroups and nothing needs an atomic.
// Operands that a product consumes along the token index (V in the forward, K, Q and dO in the backward)
// come out of the same natural [token][d] LDS images through the transposing read ds_read_b64_tr_b16.
#include "attn_dev.h"

namespace prl {

// lse2 is the base-2 log-sum-exp of the scaled scores, [head][token]
__global__ __launch_bounds__(kAThreads) void k_attn_short_fwd(AttnView q, AttnView k, AttnView v, uint16_t* o,
                                                              int64_t o_ts, int64_t o_hs, float* lse2,
                                                              const int32_t* cu, int Hq, int G, int64_t T, float c,
                                                              int groups) {
  extern __shared__ __align__(16) uint16_t lds[];
  uint16_t* Ks = lds;
  uint16_t* Vs = lds + kAImg;
  int group, within;  // group = (sample, key/value head); its G query heads share K and V
  if (!xcd_item(G, groups, group, within)) return;
  const int Hkv = Hq / G, s = group / Hkv, hkv = group % Hkv, hq = hkv * G + within;
  int64_t start;
  int len;
  sample_range(cu, s, T, start, len);
  if (len == 0) return;  // the whole workgroup, before any barrier
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int query = 32 * w + r;
  bf16x8 qf[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) qf[kk] = frag_global(q, start, len, hq, query, kk, h);
  {
    uint16_t* const imgs[2] = {Ks, Vs};
    const AttnView* const views[2] = {&k, &v};
    const int heads[2] = {hkv, hkv};
    stage_images<2>(imgs, views, start, len, heads);
  }
  __syncthreads();
  int ntile = (len + 31) >> 5;
  if (ntile > w + 1) ntile = w + 1;  // causal: no key tile beyond the wave's own queries
  if (32 * w >= len) return;         // no query of this wave exists (after the only barrier)

  f32x16 st[4];
  float m = -INFINITY;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    if (mt < ntile) {
      f32x16 acc = {};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) acc = mfma(frag_rows(Ks, 32 * mt + r, kk, h), qf[kk], acc);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = 32 * mt + acc_row(i, h);
        const float sv = (key <= query && key < len) ? acc[i] : -INFINITY;
        acc[i] = sv;
        m = fmaxf(m, sv);
      }
      st[mt] = acc;
    }
  }
  m = fmaxf(m, __shfl_xor(m, 32, kWave));  // key 0 is never masked, so m is finite
  float l = 0.0f;
  const float mc = m * c;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    if (mt < ntile) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float p = __builtin_amdgcn_exp2f(fmaf(st[mt][i], c, -mc));  // c > 0: a masked score stays -inf, p = 0
        st[mt][i] = p;
        l += p;
      }
    }
  }
  l += __shfl_xor(l, 32, kWave);

  f32x16 ot[4] = {};
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    if (mt < ntile) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = acc_frag(st[mt], s2);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) ot[dt] = mfma(frag_cols(Vs, 32 * dt, 32 * mt, s2, lane), pf, ot[dt]);
      }
    }
  }
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
    if (h == 0) lse2[(int64_t)hq * T + start + query] = mc + __builtin_amdgcn_logf(l);
  }
}

// dQ, and delta = rowsum(dO * O) [head][token] for the dK / dV kernel
__global__ __launch_bounds__(kAThreads) void k_attn_short_bwd_dq(AttnView q, AttnView k, AttnView v, AttnView o,
                                                                 AttnView dout, const float* lse2, float* delta,
                                                                 uint16_t* dq, int64_t dq_ts, int64_t dq_hs,
                                                                 const int32_t* cu, int Hq, int G, int64_t T,
                                                                 float c, float scale, int groups) {
  extern __shared__ __align__(16) uint16_t lds[];
  uint16_t* Ks = lds;
  uint16_t* Vs = lds + kAImg;
  int group, within;
  if (!xcd_item(G, groups, group, within)) return;
  const int Hkv = Hq / G, s = group / Hkv, hkv = group % Hkv, hq = hkv * G + within;
  int64_t start;
  int len;
  sample_range(cu, s, T, start, len);
  if (len == 0) return;  // the whole workgroup, before any barrier
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int query = 32 * w + r;
  bf16x8 qf[8], dof[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    qf[kk] = frag_global(q, start, len, hq, query, kk, h);
    dof[kk] = frag_global(dout, start, len, hq, query, kk, h);
  }
  // delta of the lane's query: this lane's 8 of every 16 d, then the other half's
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
  {
    uint16_t* const imgs[2] = {Ks, Vs};
    const AttnView* const views[2] = {&k, &v};
    const int heads[2] = {hkv, hkv};
    stage_images<2>(imgs, views, start, len, heads);
  }
  __syncthreads();
  int ntile = (len + 31) >> 5;
  if (ntile > w + 1) ntile = w + 1;
  if (32 * w >= len) return;

  f32x16 dqt[4] = {};
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    if (mt < ntile) {
      f32x16 sa = {}, dp = {};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        sa = mfma(frag_rows(Ks, 32 * mt + r, kk, h), qf[kk], sa);
        dp = mfma(frag_rows(Vs, 32 * mt + r, kk, h), dof[kk], dp);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = 32 * mt + acc_row(i, h);
        const bool valid = key <= query && key < len && query < len;
        const float p = valid ? __builtin_amdgcn_exp2f(fmaf(sa[i], c, -ls)) : 0.0f;
        sa[i] = p * (dp[i] - dl) * scale;  // dS^T
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 df = acc_frag(sa, s2);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dqt[dt] = mfma(frag_cols(Ks, 32 * dt, 32 * mt, s2, lane), df, dqt[dt]);
      }
    }
  }
  if (query < len) {
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

// dK and dV of one (sample, key/value head), summed over the G query heads that share it.  Two workgroups
// per pair, each owning HALF of the head dimension of both results (64 accumulator registers instead of 128)
// and staging one query head at a time (70 KB of LDS), so that two workgroups fit a compute unit (at most 256
// registers per lane).  Both halves recompute the scores; the matrix cores are idle nine tenths of the time in
// this kernel either way.  Measured against one workgroup per pair with both heads staged at once (360
// registers, one workgroup per compute unit): 120 against 125 us per layer at 16 K tokens.
__global__ __launch_bounds__(kAThreads, 2) void k_attn_short_bwd_dkv(AttnView q, AttnView k, AttnView v,
                                                                     AttnView dout, const float* lse2,
                                                                     const float* delta, uint16_t* dk, int64_t dk_ts,
                                                                     int64_t dk_hs, uint16_t* dv, int64_t dv_ts,
                                                                     int64_t dv_hs, const int32_t* cu, int Hkv, int G,
                                                                     int64_t T, float c, float scale, int groups) {
  extern __shared__ __align__(16) uint16_t lds[];
  uint16_t* Qs = lds;
  uint16_t* Os = lds + kAImg;  // dO
  float* row_lse = reinterpret_cast<float*>(lds + 2 * kAImg);
  float* row_delta = row_lse + kAL;
  int pair, dhalf;  // the two halves read the same Q, dO, K and V
  if (!xcd_item(2, groups, pair, dhalf)) return;
  const int s = pair / Hkv, hkv = pair % Hkv;
  int64_t start;
  int len;
  sample_range(cu, s, T, start, len);
  if (len == 0) return;  // the whole workgroup, before any barrier
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int key = 32 * w + r;
  bf16x8 kf[8], vf[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    kf[kk] = frag_global(k, start, len, hkv, key, kk, h);
    vf[kk] = frag_global(v, start, len, hkv, key, kk, h);
  }
  const int qtiles = (len + 31) >> 5;
  f32x16 dkt[2] = {}, dvt[2] = {};
  for (int g = 0; g < G; ++g) {
    const int hq = hkv * G + g;
    __syncthreads();  // the previous head's images are no longer read
    for (int j = 0; j < 2; ++j) {  // one image per round of loads: the registers a round needs are scarce here
      uint16_t* const imgs[1] = {j ? Os : Qs};
      const AttnView* const views[1] = {j ? &dout : &q};
      const int hs[1] = {hq};
      stage_images<1>(imgs, views, start, len, hs);
    }
    if (threadIdx.x < kAL) {
      const bool in = (int)threadIdx.x < len;
      row_lse[threadIdx.x] = in ? lse2[(int64_t)hq * T + start + threadIdx.x] : 0.0f;
      row_delta[threadIdx.x] = in ? delta[(int64_t)hq * T + start + threadIdx.x] : 0.0f;
    }
    __syncthreads();
    if (32 * w < len) {  // wave-uniform: this wave has keys
#pragma unroll 1
      for (int mt = 0; mt < 4; ++mt) {  // query tiles; causal: only those at or after the wave's keys
        if (mt >= w && mt < qtiles) {
          f32x16 sa = {}, dp = {};
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) {
            sa = mfma(frag_rows(Qs, 32 * mt + r, kk, h), kf[kk], sa);  // S: rows = queries, lane = key
            dp = mfma(frag_rows(Os, 32 * mt + r, kk, h), vf[kk], dp);  // dP = dO V^T
          }
          // log-sum-exp and delta of the 16 query rows this lane half holds (rows 8j + 4h .. + 3, j = 0 .. 3): four
          // 16-byte reads each and no branch around them (a load inside the conditional made the compiler emit one
          // branch and one LDS round trip per element).  Rows past the sample's end need no mask: their Q and dO
          // rows are zero in the images, so P there multiplies zeros in dV and dS comes out as zero.
          f32x4 lse_r[4], del_r[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            lse_r[j] = *reinterpret_cast<const f32x4*>(row_lse + 32 * mt + 8 * j + 4 * h);
            del_r[j] = *reinterpret_cast<const f32x4*>(row_delta + 32 * mt + 8 * j + 4 * h);
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            float p = __builtin_amdgcn_exp2f(fmaf(sa[i], c, -lse_r[i >> 2][i & 3]));
            p = key <= 32 * mt + acc_row(i, h) ? p : 0.0f;  // causal
            sa[i] = p;
            dp[i] = p * (dp[i] - del_r[i >> 2][i & 3]) * scale;  // dS
          }
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf = acc_frag(sa, s2), df = acc_frag(dp, s2);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
              const int d0 = 32 * (2 * dhalf + dt);
              dvt[dt] = mfma(frag_cols(Os, d0, 32 * mt, s2, lane), pf, dvt[dt]);  // dV^T += dO^T P
              dkt[dt] = mfma(frag_cols(Qs, d0, 32 * mt, s2, lane), df, dkt[dt]);  // dK^T += Q^T dS
            }
          }
        }
      }
    }
  }
  if (key < len) {
    uint16_t* krow = dk + (start + key) * dk_ts + hkv * dk_hs + 64 * dhalf;
    uint16_t* vrow = dv + (start + key) * dv_ts + hkv * dv_hs + 64 * dhalf;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
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

// The rows of Q, dO and O of one head that a thread stages: chunk `ch` (8 elements) of tokens 16 it + tid / 16.
struct HeadRows {
  u32x4 q[8], dout[8], o[8];
};
__device__ __forceinline__ void fetch_head(HeadRows& x, const AttnView& q, const AttnView& dout, const AttnView& o,
                                           int64_t start, int len, int hq) {
  const int ch = threadIdx.x & 15, t0 = threadIdx.x >> 4;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int tok = 16 * it + t0, src = tok < len ? tok : len - 1;  // no branch around a load (see stage_images)
    x.q[it] = *reinterpret_cast<const u32x4*>(q.p + (start + src) * q.ts + hq * q.hs + ch * 8);
    x.dout[it] = *reinterpret_cast<const u32x4*>(dout.p + (start + src) * dout.ts + hq * dout.hs + ch * 8);
    x.o[it] = *reinterpret_cast<const u32x4*>(o.p + (start + src) * o.ts + hq * o.hs + ch * 8);
  }
}
__device__ __forceinline__ float dot_bf16x8(u32x4 a, u32x4 b, float acc) {
  const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    acc = fmaf(__uint_as_float(aw[i] << 16), __uint_as_float(bw[i] << 16), acc);
    acc = fmaf(__uint_as_float(aw[i] & 0xffff0000u), __uint_as_float(bw[i] & 0xffff0000u), acc);
  }
  return acc;
}

// The whole backward of one (sample, key/value head) in ONE workgroup: dK and dV summed over the G query heads of
// the group and every head's dQ.  The two-kernel form above computes the scores twice (once with the query on the
// lane for dQ, once with the key on the lane for dK / dV: an accumulator tile can only be summed over its
// register index by the next product) and reads q, k, v and dO twice; here the scores are computed once, with
// the key on the lane, dS goes through LDS as a [key][query] image, and the waves, now owning 32 queries each,
// read it back transposed for dQ^T = K^T dS^T: five products per head instead of seven, every input read once,
// delta = rowsum(dO * O) formed while Q and dO are staged and never written to memory.  Four images (K, Q, dO,
// dS) are 137 KB of LDS, so one workgroup per compute unit, with the whole register file: the next head's rows
// are requested before this head's products, so only the first head's loads are waited for.
__global__ __launch_bounds__(kAThreads, 1) void k_attn_short_bwd_fused(
    AttnView q, AttnView k, AttnView v, AttnView o, AttnView dout, const float* lse2, uint16_t* dq, int64_t dq_ts,
    int64_t dq_hs, uint16_t* dk, int64_t dk_ts, int64_t dk_hs, uint16_t* dv, int64_t dv_ts, int64_t dv_hs,
    const int32_t* cu, int Hkv, int G, int64_t T, float c, float scale) {
  extern __shared__ __align__(16) uint16_t lds[];
  uint16_t* Ks = lds;
  uint16_t* Qs = lds + kAImg;
  uint16_t* Os = lds + 2 * kAImg;  // dO
  uint16_t* Ds = lds + 3 * kAImg;  // dS as [key][query]
  float* row_lse = reinterpret_cast<float*>(lds + 4 * kAImg);
  float* row_delta = row_lse + kAL;
  const int s = (int)blockIdx.x / Hkv, hkv = (int)blockIdx.x % Hkv;
  int64_t start;
  int len;
  sample_range(cu, s, T, start, len);
  if (len == 0) return;  // the whole workgroup, before any barrier
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int ch = threadIdx.x & 15, t0 = threadIdx.x >> 4;
  const int own = 32 * w + r;  // the lane's key in the first half of a head's work, its query in the second
  HeadRows rows;
  fetch_head(rows, q, dout, o, start, len, hkv * G);
  bf16x8 kf[8], vf[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    kf[kk] = frag_global(k, start, len, hkv, own, kk, h);
    vf[kk] = frag_global(v, start, len, hkv, own, kk, h);
  }
#pragma unroll
  for (int kk = 0; kk < 8; ++kk)  // the K image for dQ is what the waves hold as fragments: no second read
    *reinterpret_cast<u32x4*>(Ks + own * kAP + 16 * kk + 8 * h) = __builtin_bit_cast(u32x4, kf[kk]);
  const int tiles = (len + 31) >> 5;
  const int key_or_never = own < len ? own : (1 << 20);  // keys past the sample's end take part in no score
  f32x16 dkt[4] = {}, dvt[4] = {};
#pragma unroll 1
  for (int g = 0; g < G; ++g) {
    const int hq = hkv * G + g;
    __syncthreads();  // the previous head's images are no longer read
    {
      const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int tok = 16 * it + t0;
        const bool in = tok < len;
        *reinterpret_cast<u32x4*>(Qs + tok * kAP + ch * 8) = in ? rows.q[it] : zero;
        *reinterpret_cast<u32x4*>(Os + tok * kAP + ch * 8) = in ? rows.dout[it] : zero;
        float d = dot_bf16x8(rows.dout[it], rows.o[it], 0.0f);
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) d += __shfl_xor(d, off, 16);
        if (ch == 0) row_delta[tok] = in ? d : 0.0f;
      }
    }
    if (threadIdx.x < kAL)
      row_lse[threadIdx.x] = (int)threadIdx.x < len ? lse2[(int64_t)hq * T + start + threadIdx.x] : 0.0f;
    if (g + 1 < G) fetch_head(rows, q, dout, o, start, len, hq + 1);
    __syncthreads();
    if (32 * w < len) {  // wave-uniform: this wave has keys
#pragma unroll 1
      for (int mt = w; mt < tiles; ++mt) {  // query tiles; causal: only those at or after the wave's keys
        f32x16 sa = {}, dp = {};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          sa = mfma(frag_rows(Qs, 32 * mt + r, kk, h), kf[kk], sa);  // S: rows = queries, lane = key
          dp = mfma(frag_rows(Os, 32 * mt + r, kk, h), vf[kk], dp);  // dP = dO V^T
        }
        // Query rows past the sample's end need no mask: their Q and dO rows are zero in the images and their
        // log-sum-exp and delta are stored as zero, so P multiplies zeros in dV and dS comes out as zero.
        f32x4 lse_r[4], del_r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          lse_r[j] = *reinterpret_cast<const f32x4*>(row_lse + 32 * mt + 8 * j + 4 * h);
          del_r[j] = *reinterpret_cast<const f32x4*>(row_delta + 32 * mt + 8 * j + 4 * h);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float p = __builtin_amdgcn_exp2f(fmaf(sa[i], c, -lse_r[i >> 2][i & 3]));
          p = key_or_never <= 32 * mt + acc_row(i, h) ? p : 0.0f;  // causal, and no key past the end
          sa[i] = p;
          dp[i] = p * (dp[i] - del_r[i >> 2][i & 3]) * scale;  // dS
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pf = acc_frag(sa, s2), df = acc_frag(dp, s2);
          const u32x4 dw = __builtin_bit_cast(u32x4, df);  // queries 16 s2 + 4 h .. + 3 and 8 further, of this tile
          uint16_t* drow = Ds + own * kAP + 32 * mt + 16 * s2 + 4 * h;
          *reinterpret_cast<u32x2*>(drow) = u32x2{dw.x, dw.y};
          *reinterpret_cast<u32x2*>(drow + 8) = u32x2{dw.z, dw.w};
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            dvt[dt] = mfma(frag_cols(Os, 32 * dt, 32 * mt, s2, lane), pf, dvt[dt]);  // dV^T += dO^T P
            dkt[dt] = mfma(frag_cols(Qs, 32 * dt, 32 * mt, s2, lane), df, dkt[dt]);  // dK^T += Q^T dS
          }
        }
      }
    }
    __syncthreads();  // dS of every key tile is in LDS
    if (32 * w < len) {  // wave-uniform: this wave has queries
      f32x16 dqt[4] = {};
#pragma unroll 1
      for (int mt = 0; mt <= w; ++mt) {  // key tiles; causal: none beyond the wave's queries (all exist: 32 w < len)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 df = frag_cols(Ds, 32 * w, 32 * mt, s2, lane);  // dS^T: rows = keys, lane = query
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) dqt[dt] = mfma(frag_cols(Ks, 32 * dt, 32 * mt, s2, lane), df, dqt[dt]);
        }
      }
      if (own < len) {
        uint16_t* row = dq + (start + own) * dq_ts + hq * dq_hs;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            u32x2 pk;
            pk.x = pack_bf16(dqt[dt][4 * j + 0], dqt[dt][4 * j + 1]);
            pk.y = pack_bf16(dqt[dt][4 * j + 2], dqt[dt][4 * j + 3]);
            *reinterpret_cast<u32x2*>(row + 32 * dt + 8 * j + 4 * h) = pk;
          }
        }
      }
    }
  }
  if (own < len) {
    uint16_t* krow = dk + (start + own) * dk_ts + hkv * dk_hs;
    uint16_t* vrow = dv + (start + own) * dv_ts + hkv * dv_hs;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u32x2 pk;
        pk.x = pack_bf16(dkt[dt][4 * j + 0], dkt[dt][4 * j + 1]);
        pk.y = pack_bf16(dkt[dt][4 * j + 2], dkt[dt][4 * j + 3]);
        *reinterpret_cast<u32x2*>(krow + 32 * dt + 8 * j + 4 * h) = pk;
        pk.x = pack_bf16(dvt[dt][4 * j + 0], dvt[dt][4 * j + 1]);
        pk.y = pack_bf16(dvt[dt][4 * j + 2], dvt[dt][4 * j + 3]);
        *reinterpret_cast<u32x2*>(vrow + 32 * dt + 8 * j + 4 * h) = pk;
      }
    }
  }
}

static bool view_ok(const void* p, int64_t ts, int64_t hs) { return attn_view_ok(p, ts, hs); }

// From this many (sample, key/value head) groups on, the backward is the one-workgroup-per-group kernel (135
// against 169 us on 128 samples of 128 tokens, 63 against 81 us on 64, the same 29 us on 16, 16 / 8 heads);
// below, the two kernels, whose grids are 2 and 4 times as large, spread a small row over more compute units.
#ifndef PRL_ATTN_SHORT_FUSED_MIN
#define PRL_ATTN_SHORT_FUSED_MIN 128
#endif

constexpr size_t kFusedLds = 4 * kAImg * sizeof(uint16_t) + 2 * kAL * sizeof(float);
constexpr size_t kFwdLds = 2 * kAImg * sizeof(uint16_t);
constexpr size_t kDqLds = 2 * kAImg * sizeof(uint16_t);
constexpr size_t kDkvLds = 2 * kAImg * sizeof(uint16_t) + 2 * kAL * sizeof(float);

// The kernels use more than the 64 KB of LDS a launch gets by default; raised once per process and device.
static int raise_lds_limits() {
  static thread_local int done_for = -1;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return PRL_ELAUNCH;
  if (done_for == dev) return PRL_OK;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_short_fwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)kFwdLds) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_short_bwd_dq),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDqLds) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_short_bwd_dkv),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDkvLds) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_short_bwd_fused),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFusedLds) != hipSuccess) {
    set_error("attn_short: cannot raise the dynamic LDS limit: %s", hipGetErrorString(hipGetLastError()));
    return PRL_ELAUNCH;
  }
  done_for = dev;
  return PRL_OK;
}

}  // namespace prl

using namespace prl;

extern "C" int prl_attn_short_supported(int dtype, int64_t head_dim, int64_t max_len, int64_t q_heads,
                                        int64_t kv_heads) {
  return dtype == PRL_BF16 && head_dim == kAD && max_len >= 1 && max_len <= kAL && kv_heads >= 1 &&
         q_heads >= kv_heads && q_heads % kv_heads == 0;
}

#define PRL_ATTN_COMMON_CHECKS(name)                                                                          \
  PRL_REQUIRE(n_samples >= 0 && T >= 0 && q_heads >= 1 && kv_heads >= 1 && q_heads % kv_heads == 0,           \
              name ": bad sizes (samples %d, T %lld, heads %d / %d)", n_samples, (long long)T, q_heads,      \
              kv_heads);                                                                                      \
  PRL_REQUIRE(scale > 0.0f, name ": the score scale must be positive");                                      \
  if (n_samples == 0 || T == 0) return PRL_OK;                                                                \
  PRL_REQUIRE(cu_seqlens, name ": the sample table must not be NULL");                                       \
  PRL_REQUIRE((int64_t)(n_samples + 8) * q_heads <= 0x3fffffffLL, name ": too many (sample, head) pairs")

extern "C" int prl_attn_short_fwd(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides,
                                  const void* v, const int64_t* v_strides, void* out, const int64_t* out_strides,
                                  float* lse2, const int32_t* cu_seqlens, int32_t n_samples, int64_t T,
                                  int32_t q_heads, int32_t kv_heads, float scale, void* stream) {
  clear_error();
  PRL_ATTN_COMMON_CHECKS("attn_short_fwd");
  PRL_REQUIRE(q_strides && k_strides && v_strides && out_strides && lse2, "attn_short_fwd: NULL argument");
  PRL_REQUIRE(view_ok(q, q_strides[0], q_strides[1]) && view_ok(k, k_strides[0], k_strides[1]) &&
                  view_ok(v, v_strides[0], v_strides[1]) && view_ok(out, out_strides[0], out_strides[1]),
              "attn_short_fwd: tensors must be 16-byte aligned with token and head strides that are multiples of 8");
  if (int rc = raise_lds_limits()) return rc;
  const AttnView qv{static_cast<const uint16_t*>(q), q_strides[0], q_strides[1]};
  const AttnView kv{static_cast<const uint16_t*>(k), k_strides[0], k_strides[1]};
  const AttnView vv{static_cast<const uint16_t*>(v), v_strides[0], v_strides[1]};
  const int groups = n_samples * kv_heads;
  hipLaunchKernelGGL(k_attn_short_fwd, dim3(xcd_grid(q_heads / kv_heads, groups)), dim3(kAThreads), kFwdLds,
                     (hipStream_t)stream, qv, kv, vv, static_cast<uint16_t*>(out), out_strides[0], out_strides[1],
                     lse2, cu_seqlens, q_heads, q_heads / kv_heads, T, scale * kLog2e, groups);
  PRL_CHECK_LAUNCH("attn_short_fwd");
  return PRL_OK;
}

extern "C" int prl_attn_short_bwd(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides,
                                  const void* v, const int64_t* v_strides, const void* out,
                                  const int64_t* out_strides, const void* dout, const int64_t* dout_strides,
                                  const float* lse2, float* delta, void* dq, const int64_t* dq_strides, void* dk,
                                  const int64_t* dk_strides, void* dv, const int64_t* dv_strides,
                                  const int32_t* cu_seqlens, int32_t n_samples, int64_t T, int32_t q_heads,
                                  int32_t kv_heads, float scale, void* stream) {
  clear_error();
  PRL_ATTN_COMMON_CHECKS("attn_short_bwd");
  PRL_REQUIRE(q_strides && k_strides && v_strides && out_strides && dout_strides && dq_strides && dk_strides &&
                  dv_strides && lse2 && delta,
              "attn_short_bwd: NULL argument");
  PRL_REQUIRE(view_ok(q, q_strides[0], q_strides[1]) && view_ok(k, k_strides[0], k_strides[1]) &&
                  view_ok(v, v_strides[0], v_strides[1]) && view_ok(out, out_strides[0], out_strides[1]) &&
                  view_ok(dout, dout_strides[0], dout_strides[1]) && view_ok(dq, dq_strides[0], dq_strides[1]) &&
                  view_ok(dk, dk_strides[0], dk_strides[1]) && view_ok(dv, dv_strides[0], dv_strides[1]),
              "attn_short_bwd: tensors must be 16-byte aligned with token and head strides that are multiples of 8");
  if (int rc = raise_lds_limits()) return rc;
  const AttnView qv{static_cast<const uint16_t*>(q), q_strides[0], q_strides[1]};
  const AttnView kv{static_cast<const uint16_t*>(k), k_strides[0], k_strides[1]};
  const AttnView vv{static_cast<const uint16_t*>(v), v_strides[0], v_strides[1]};
  const AttnView ov{static_cast<const uint16_t*>(out), out_strides[0], out_strides[1]};
  const AttnView dov{static_cast<const uint16_t*>(dout), dout_strides[0], dout_strides[1]};
  const int G = q_heads / kv_heads;
  const float c = scale * kLog2e;
  const int groups = n_samples * kv_heads;
  if (groups >= PRL_ATTN_SHORT_FUSED_MIN) {
    hipLaunchKernelGGL(k_attn_short_bwd_fused, dim3(groups), dim3(kAThreads), kFusedLds, (hipStream_t)stream, qv, kv,
                       vv, ov, dov, lse2, static_cast<uint16_t*>(dq), dq_strides[0], dq_strides[1],
                       static_cast<uint16_t*>(dk), dk_strides[0], dk_strides[1], static_cast<uint16_t*>(dv),
                       dv_strides[0], dv_strides[1], cu_seqlens, kv_heads, G, T, c, scale);
    PRL_CHECK_LAUNCH("attn_short_bwd (fused)");
    return PRL_OK;
  }
  hipLaunchKernelGGL(k_attn_short_bwd_dq, dim3(xcd_grid(G, groups)), dim3(kAThreads), kDqLds,
                     (hipStream_t)stream, qv, kv, vv, ov, dov, lse2, delta, static_cast<uint16_t*>(dq), dq_strides[0],
                     dq_strides[1], cu_seqlens, q_heads, G, T, c, scale, groups);
  PRL_CHECK_LAUNCH("attn_short_bwd (dq)");
  hipLaunchKernelGGL(k_attn_short_bwd_dkv, dim3(xcd_grid(2, groups)), dim3(kAThreads), kDkvLds,
                     (hipStream_t)stream, qv, kv, vv, dov, lse2, delta, static_cast<uint16_t*>(dk), dk_strides[0],
                     dk_strides[1], static_cast<uint16_t*>(dv), dv_strides[0], dv_strides[1], cu_seqlens, kv_heads, G,
                     T, c, scale, groups);
  PRL_CHECK_LAUNCH("attn_short_bwd (dk, dv)");
  return PRL_OK;
}
