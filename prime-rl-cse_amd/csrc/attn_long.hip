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

// rows row0 .. row0+127 of the sample, one head each, into N natural [token][d] images; rows at or past `len`
// are zero.  Reads without a branch (see stage_images); needs len >= 1.
template <int N>
__device__ __forceinline__ void stage_tile(uint16_t* const (&img)[N], const AttnView* const (&x)[N], int64_t start,
                                           int len, int row0, const int (&head)[N]) {
  constexpr int kIt = kAL * kAD / 8 / kAThreads;
  u32x4 v[N][kIt];
#pragma unroll
  for (int n = 0; n < N; ++n) {
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
      const int id = it * kAThreads + threadIdx.x, tok = row0 + (id >> 4), ch = id & 15;
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
      *reinterpret_cast<u32x4*>(img[n] + tok * kAP + ch * 8) = row0 + tok < len ? v[n][it] : zero;
    }
  }
}

// lse2 is the base-2 log-sum-exp of the scaled scores, [head][token]
__global__ __launch_bounds__(kAThreads, 2) void k_attn_long_fwd(AttnView q, AttnView k, AttnView v, uint16_t* o,
                                                             int64_t o_ts, int64_t o_hs, float* lse2,
                                                             const int32_t* cu, int Hq, int G, int64_t T, float c,
                                                             int tiles, int groups) {
  extern __shared__ __align__(16) uint16_t lds[];
  uint16_t* Ks = lds;
  uint16_t* Vs = lds + kAImg;
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
  uint16_t* const imgs[2] = {Ks, Vs};
  const AttnView* const views[2] = {&k, &v};
  const int heads[2] = {hkv, hkv};
#pragma unroll 1
  for (int kt = 0; kt <= qt; ++kt) {
    const int k0 = kt * kAL;
    __syncthreads();  // the previous tile's images are no longer read
    stage_tile<2>(imgs, views, start, len, k0, heads);
    __syncthreads();
    if (!active) continue;
    const bool diag = kt == qt;
    int ntile = 4;  // 32-key blocks of this tile the wave needs: all of them below the diagonal (they end before q0 < len)
    f32x16 st[4];
    float mx = -INFINITY;
    if (!diag) {
      // four independent accumulation chains: the LDS reads of one step are in flight while the products of the
      // step before run
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) st[mt] = f32x16{};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) st[mt] = mfma(frag_rows(Ks, 32 * mt + r, kk, h), qf[kk], st[mt]);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, st[mt][i]);
      }
    } else {
      ntile = (len - k0 + 31) >> 5;  // blocks that hold keys at all ...
      if (ntile > w + 1) ntile = w + 1;  // ... and, causal, none beyond the wave's own queries
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        if (mt < ntile) {
          f32x16 acc = {};
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) acc = mfma(frag_rows(Ks, 32 * mt + r, kk, h), qf[kk], acc);
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int key = k0 + 32 * mt + acc_row(i, h);
            acc[i] = (key <= query && key < len) ? acc[i] : -INFINITY;
            mx = fmaxf(mx, acc[i]);
          }
          st[mt] = acc;
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, kWave));
    const float m_new = fmaxf(m, mx);  // finite: key k0 of every tile walked is at or before every query of the tile
    if (__any(m_new > m)) {  // (wave-uniform) some query's maximum moved: bring sums and accumulators to the new one
      const float alpha = __builtin_amdgcn_exp2f((m - m_new) * c);  // 0 at the first tile (m = -inf), 1 where nothing moved
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
    for (int mt = 0; mt < 4; ++mt) {
      if (mt < ntile) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float p = __builtin_amdgcn_exp2f(fmaf(st[mt][i], c, -mc));  // c > 0: a masked score stays -inf, p = 0
          st[mt][i] = p;
          l += p;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pf = acc_frag(st[mt], s2);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) ot[dt] = mfma(frag_cols(Vs, 32 * dt, 32 * mt, s2, lane), pf, ot[dt]);
        }
      }
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
  uint16_t* Ks = lds;
  uint16_t* Vs = lds + kAImg;
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
  uint16_t* const imgs[2] = {Ks, Vs};
  const AttnView* const views[2] = {&k, &v};
  const int heads[2] = {hkv, hkv};
#pragma unroll 1
  for (int kt = 0; kt <= qt; ++kt) {
    const int k0 = kt * kAL;
    __syncthreads();
    stage_tile<2>(imgs, views, start, len, k0, heads);
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
          sa = mfma(frag_rows(Ks, 32 * mt + r, kk, h), qf[kk], sa);
          dp = mfma(frag_rows(Vs, 32 * mt + r, kk, h), dof[kk], dp);
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
          for (int dt = 0; dt < 4; ++dt) dqt[dt] = mfma(frag_cols(Ks, 32 * dt, 32 * mt, s2, lane), df, dqt[dt]);
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
  uint16_t* Qs = lds;
  uint16_t* Os = lds + kAImg;  // dO
  float* row_lse = reinterpret_cast<float*>(lds + 2 * kAImg);
  float* row_delta = row_lse + kAL;
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
#pragma unroll 1
  for (int g = 0; g < G; ++g) {
    const int hq = hkv * G + g;
#pragma unroll 1
    for (int qt = kt; qt < nqt; ++qt) {
      const int q0 = qt * kAL;
      __syncthreads();  // the previous images are no longer read
#if PRL_ATTN_DKV_ROUNDS == 1
      {  // both images in one round of loads (64 registers in flight)
        uint16_t* const imgs[2] = {Qs, Os};
        const AttnView* const views[2] = {&q, &dout};
        const int hs[2] = {hq, hq};
        stage_tile<2>(imgs, views, start, len, q0, hs);
      }
#else
      for (int j = 0; j < 2; ++j) {  // one image per round of loads: registers are scarce here
        uint16_t* const imgs[1] = {j ? Os : Qs};
        const AttnView* const views[1] = {j ? &dout : &q};
        const int hs[1] = {hq};
        stage_tile<1>(imgs, views, start, len, q0, hs);
      }
#endif
      if (threadIdx.x < kAL) {
        const bool in = q0 + (int)threadIdx.x < len;
        row_lse[threadIdx.x] = in ? lse2[(int64_t)hq * T + start + q0 + threadIdx.x] : 0.0f;
        row_delta[threadIdx.x] = in ? delta[(int64_t)hq * T + start + q0 + threadIdx.x] : 0.0f;
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
            sa = mfma(frag_rows(Qs, 32 * mt + r, kk, h), kf[kk], sa);  // S: rows = queries, lane = key
            dp = mfma(frag_rows(Os, 32 * mt + r, kk, h), vf[kk], dp);  // dP = dO V^T
          }
          // log-sum-exp and delta of the 16 query rows this lane half holds (rows 8j + 4h .. + 3, j = 0 .. 3): four
          // 16-byte reads each, and no branch around them.  Rows past the sample's end need no mask: their Q and dO
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
              dvt[dt] = mfma(frag_cols(Os, d0, 32 * mt, s2, lane), pf, dvt[dt]);  // dV^T += dO^T P
              dkt[dt] = mfma(frag_cols(Qs, d0, 32 * mt, s2, lane), df, dkt[dt]);  // dK^T += Q^T dS
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

constexpr size_t kLongFwdLds = 2 * kAImg * sizeof(uint16_t);
constexpr size_t kLongDqLds = 2 * kAImg * sizeof(uint16_t);
constexpr size_t kLongDkvLds = 2 * kAImg * sizeof(uint16_t) + 2 * kAL * sizeof(float);

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
