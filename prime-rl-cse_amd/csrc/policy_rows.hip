// Token-mode policy update of a packed row in ONE pass over the logits: the
// forward statistics (log-probability, entropy, log-sum-exp) and the gradient
// with respect to the logits come out of the same read.
//
// Why this is possible on MI355X: a compute unit has 512 KB of vector registers,
// and one bf16 row of a 152 K vocabulary is 304 KB.  A workgroup of 1024 threads
// (all 16 wave slots the 128-register budget allows) loads its row ONCE into
// registers, 19 x 16 bytes per lane, reduces max and sums over the registers
// (an exact two-pass soft-max, no rescaling), derives the token's loss gradient
// g = dL/dlogprob on the spot (in token mode it depends on nothing but this token:
// src/prime_rl/trainer/rl/loss.py:90, 98-104), and writes the gradient row
//   dlogits[row] = g / temperature * (onehot(label) - softmax(row / temperature))
// straight from the registers.  HBM traffic per token: V logits in, V gradient
// elements out, which is what the separate backward kernel alone moves; the
// forward kernel's read of the logits disappears.
//
// It replaces, for `ratio_type = "token"` and bf16 logits, the chain
//   shift_logits -> / temperature -> selective_log_softmax -> compute_entropy
//   -> compute_loss -> loss.backward() down to the logits
// of src/prime_rl/trainer/rl/train.py:285-310.  The gradient is computed for an
// incoming loss gradient of 1; `prl_scale_rows_if_not_one` rescales it in the
// (unusual) case that the caller's backward arrives with another value.
// Sequence mode needs the whole sample's sum before any token's gradient is
// known, and fp32 rows do not fit the registers: both take the two-kernel path.
#include "logprob_dev.h"

namespace prl {

constexpr int kRowThreads = 1024;
constexpr int kRowWaves = kRowThreads / kWave;
constexpr int kRowMaxChunks = 19;  // 16-byte chunks per lane: 19 * 1024 * 8 = 155,648 bf16 logits

// The row lives in registers as packed bf16 pairs and is unpacked three times (maximum, sums,
// gradient).  Left alone, the compiler keeps the unpacked floats of one pass for the next, which
// doubles the register need and spills; this empty statement makes the packed values opaque so
// that every pass unpacks again (two ALU operations per pair, no memory traffic).
template <int NCH>
__device__ __forceinline__ void forget_unpacked(u32x4 (&buf)[NCH]) {
#pragma unroll
  for (int j = 0; j < NCH; ++j) asm volatile("" : "+v"(buf[j]));
}

template <int NCH>
__global__ __launch_bounds__(kRowThreads) void k_policy_rows(
    const char* __restrict__ logits, int64_t T, int64_t V, int64_t in_stride_bytes, const int64_t* __restrict__ ids,
    const float* __restrict__ old, const float* __restrict__ adv, const uint8_t* __restrict__ mask, int skip_masked,
    float inv_temp, float clip, float inv_scale, float* __restrict__ lp_out, float* __restrict__ ent_out,
    float* __restrict__ lse_out, char* __restrict__ dlogits, int64_t out_stride_bytes) {
  __shared__ float red_m[kRowWaves], red_s[kRowWaves], red_w[kRowWaves];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
  const int64_t row = blockIdx.x;  // row of the logits and of the gradient
  const int64_t t = row + 1;       // the token that row predicts
  const int64_t nchunk = V >> 3;
  char* outp = dlogits + row * out_stride_bytes;

  // Addressing: a uniform base per 16 KB slab of the row plus one 32-bit lane offset, so that the
  // NCH loads (and stores) of a lane share a single address register.
  const uint32_t lane_off = (uint32_t)tid * 16u;
  const int nchunk32 = (int)nchunk;

  auto zero_row = [&]() {
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      if (tid < nchunk32 - j * kRowThreads)
        stream_store(z, reinterpret_cast<u32x4*>(outp + (int64_t)j * (kRowThreads * 16) + lane_off));
    }
  };

  if (t >= T) {
    // the last row predicts nothing; its workgroup also emits token 0, which no row predicts
    if (tid == 0) {
      if (skip_masked && mask[0] == 0)
        emit_skipped(0, lp_out, ent_out, lse_out);
      else
        emit_zero_row(V, 0, lp_out, ent_out, lse_out);
    }
    zero_row();
    return;
  }
  const bool m = mask[t] != 0;
  if (skip_masked && !m) {  // a token nobody looks at: no read, outputs 0, gradient 0
    if (tid == 0) emit_skipped(t, lp_out, ent_out, lse_out);
    zero_row();
    return;
  }

  // 1. the whole row into registers; lanes past the end hold the most negative finite bf16
  const char* rp = logits + row * in_stride_bytes;
  const u32x4 fill = {0xff7fff7fu, 0xff7fff7fu, 0xff7fff7fu, 0xff7fff7fu};
  u32x4 buf[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    buf[j] = tid < nchunk32 - j * kRowThreads
                 ? stream_load<true>(reinterpret_cast<const u32x4*>(rp + (int64_t)j * (kRowThreads * 16) + lane_off))
                 : fill;
  }
  const int64_t label = ids[t];
  float x_label = __builtin_nanf("");
  if (label >= 0 && label < V) x_label = load_elem<PRL_BF16>(rp, label);
  const float old_t = old[t], adv_t = adv[t];

  // 2. maximum
  forget_unpacked(buf);
  float mx = kNegBig;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    float x[8];
    decode16_raw<PRL_BF16>(buf[j], x);
#pragma unroll
    for (int i = 0; i < 8; ++i) mx = fmaxf(mx, x[i]);
  }
  mx = wave_max(mx);
  if (lane == 0) red_m[wid] = mx;
  __syncthreads();
  mx = red_m[0];
#pragma unroll
  for (int i = 1; i < kRowWaves; ++i) mx = fmaxf(mx, red_m[i]);
  const float c2 = inv_temp * kLog2e;
  const float M = mx * c2;  // c2 > 0: the largest logit gives the largest scaled logit

  // 3. s = sum 2^(y - M), w = sum x 2^(y - M) with y = x * c2
  forget_unpacked(buf);
  float es[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ws[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    float x[8];
    decode16_raw<PRL_BF16>(buf[j], x);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float e = fast_exp2(fmaf(x[i], c2, -M));
      es[i & 3] += e;
      ws[i & 3] = fmaf(x[i], e, ws[i & 3]);
    }
  }
  float s = wave_sum((es[0] + es[2]) + (es[1] + es[3]));
  float w = wave_sum((ws[0] + ws[2]) + (ws[1] + ws[3]));
  if (lane == 0) {
    red_s[wid] = s;
    red_w[wid] = w;
  }
  __syncthreads();
  s = red_s[0];
  w = red_w[0];
#pragma unroll
  for (int i = 1; i < kRowWaves; ++i) {
    s += red_s[i];
    w += red_w[i];
  }

  // 4. the token's values and its loss gradient (same expressions as the loss kernels, so that the
  //    g they report is the g used here)
  const float lse2 = M + fast_log2(s);
  const float lse = lse2 * kLn2;
  const float lp = x_label * inv_temp - lse;
  if (tid == 0) {
    if (lse_out) lse_out[t] = lse;
    if (lp_out) lp_out[t] = lp;
    if (ent_out) ent_out[t] = lse - (w / s) * inv_temp;
  }
  const float r = expf(lp - old_t);
  const float through = (r <= clip) ? -(inv_scale * adv_t) : 0.0f;
  const float g = m ? through * r : 0.0f;
  const float gi = g * inv_temp;
  if (gi == 0.0f) {  // masked, clipped or zero advantage: nothing to learn from this token
    zero_row();
    return;
  }

  // 5. the gradient row, from the registers
  forget_unpacked(buf);
  const int label_chunk = (int)(label >> 3), label_elem = (int)(label & 7);
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    if (tid < nchunk32 - j * kRowThreads) {
      float y[8], d[8];
      decode16<PRL_BF16>(buf[j], c2, y);
#pragma unroll
      for (int i = 0; i < 8; ++i) d[i] = -gi * fast_exp2(y[i] - lse2);
      if (label_chunk == j * kRowThreads + tid) {
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] += (label_elem == i) ? gi : 0.0f;
      }
      u32x4 v;
      v.x = pack_bf16(d[0], d[1]);
      v.y = pack_bf16(d[2], d[3]);
      v.z = pack_bf16(d[4], d[5]);
      v.w = pack_bf16(d[6], d[7]);
      stream_store(v, reinterpret_cast<u32x4*>(outp + (int64_t)j * (kRowThreads * 16) + lane_off));
    }
  }
}

// x *= scale[0] over T rows of V elements, unless scale[0] is exactly 1 (then nothing is read or written)
template <int DT>
__global__ __launch_bounds__(256) void k_scale_rows_if_not_one(char* __restrict__ x, int64_t T, int64_t V,
                                                               int64_t stride_bytes,
                                                               const float* __restrict__ scale) {
  const float sc = scale[0];
  if (sc == 1.0f) return;
  constexpr int ES = DT == PRL_F32 ? 4 : 2;
  for (int64_t row = blockIdx.x; row < T; row += gridDim.x) {
    char* p = x + row * stride_bytes;
    const bool aligned = ((reinterpret_cast<uintptr_t>(p)) & 15) == 0;
    constexpr int VE = 16 / ES;
    const int64_t nvec = aligned ? V / VE : 0;
    for (int64_t u = threadIdx.x; u < nvec; u += 256) {
      u32x4 v = reinterpret_cast<u32x4*>(p)[u];
      float y[VE];
      decode16<DT>(v, sc, y);
      store_unit<DT, VE>(p, u, y);
    }
    for (int64_t i = nvec * VE + threadIdx.x; i < V; i += 256) store_elem<DT>(p, i, load_elem<DT>(p, i) * sc);
  }
}

static int chunks_needed(int64_t V) { return (int)(((V >> 3) + kRowThreads - 1) / kRowThreads); }

}  // namespace prl

using namespace prl;

extern "C" int prl_policy_rows_supported(int logits_dtype, int64_t T, int64_t V, int64_t row_stride, int out_dtype,
                                         int64_t out_row_stride) {
  if (logits_dtype != PRL_BF16 || out_dtype != PRL_BF16) return 0;
  if (T < 1 || T >= ((int64_t)1 << 31) || V < 8 || (V & 7)) return 0;
  if (row_stride < V || out_row_stride < V || (row_stride & 7) || (out_row_stride & 7)) return 0;
  return chunks_needed(V) <= kRowMaxChunks ? 1 : 0;
}

extern "C" int prl_policy_rows_fwd_bwd(const void* logits, int logits_dtype, int64_t T, int64_t V, int64_t row_stride,
                                       const int64_t* ids, const float* adv, const float* old, const uint8_t* mask,
                                       int skip_masked, float temperature, float clip_ratio, float loss_scale,
                                       float* logprob, float* entropy, float* lse, void* dlogits, int out_dtype,
                                       int64_t out_row_stride, void* stream) {
  clear_error();
  PRL_REQUIRE(prl_policy_rows_supported(logits_dtype, T, V, row_stride, out_dtype, out_row_stride),
              "policy_rows_fwd_bwd: shape or dtype not supported (needs bf16 in and out, V a multiple of 8 and at "
              "most %d, row strides multiples of 8); ask prl_policy_rows_supported first",
              kRowMaxChunks * kRowThreads * 8);
  PRL_REQUIRE(logits && ids && adv && old && mask && logprob && dlogits,
              "policy_rows_fwd_bwd: logits, ids, adv, old, mask, logprob and dlogits must not be NULL");
  PRL_REQUIRE(((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(dlogits)) & 15) == 0,
              "policy_rows_fwd_bwd: logits and dlogits must be 16-byte aligned");
  PRL_REQUIRE(temperature > 0.0f && temperature - temperature == 0.0f,
              "policy_rows_fwd_bwd: temperature must be finite and positive");
  const float scale = loss_scale > 1.0f ? loss_scale : 1.0f;  // max(loss_scale, 1)
  const float inv_temp = 1.0f / temperature;
  const int need = chunks_needed(V);
  const dim3 grid((unsigned)T), block(kRowThreads);
  hipStream_t st = (hipStream_t)stream;
#define PRL_ROWS(N)                                                                                                \
  hipLaunchKernelGGL((k_policy_rows<N>), grid, block, 0, st, (const char*)logits, T, V, row_stride * 2, ids, old, \
                     adv, mask, skip_masked ? 1 : 0, inv_temp, clip_ratio, 1.0f / scale, logprob, entropy, lse,   \
                     (char*)dlogits, out_row_stride * 2)
  if (need <= 2)
    PRL_ROWS(2);
  else if (need <= 4)
    PRL_ROWS(4);
  else if (need <= 8)
    PRL_ROWS(8);
  else if (need <= 12)
    PRL_ROWS(12);
  else if (need <= 16)
    PRL_ROWS(16);
  else
    PRL_ROWS(19);
#undef PRL_ROWS
  PRL_CHECK_LAUNCH("policy_rows_fwd_bwd");
  return PRL_OK;
}

extern "C" int prl_scale_rows_if_not_one(void* x, int dtype, int64_t T, int64_t V, int64_t row_stride,
                                         const float* scale, void* stream) {
  clear_error();
  PRL_REQUIRE(T >= 0 && V >= 1 && row_stride >= V, "scale_rows_if_not_one: need T >= 0, V >= 1, row_stride >= V");
  if (dtype != PRL_F32 && dtype != PRL_BF16) {
    set_error("scale_rows_if_not_one: unsupported dtype code %d", dtype);
    return PRL_EDTYPE;
  }
  if (T == 0) return PRL_OK;
  PRL_REQUIRE(x && scale, "scale_rows_if_not_one: x and scale must not be NULL");
  const int es = dtype == PRL_F32 ? 4 : 2;
  const dim3 grid((unsigned)(T < 4096 ? T : 4096)), block(256);
  if (dtype == PRL_F32)
    hipLaunchKernelGGL((k_scale_rows_if_not_one<PRL_F32>), grid, block, 0, (hipStream_t)stream, (char*)x, T, V,
                       row_stride * es, scale);
  else
    hipLaunchKernelGGL((k_scale_rows_if_not_one<PRL_BF16>), grid, block, 0, (hipStream_t)stream, (char*)x, T, V,
                       row_stride * es, scale);
  PRL_CHECK_LAUNCH("scale_rows_if_not_one");
  return PRL_OK;
}
