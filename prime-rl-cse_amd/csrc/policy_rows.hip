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
#include <type_traits>

#include <hip/hip_ext.h>

#include "logprob_dev.h"

// Build-time tunables (defaults = measured best on MI355X; tools/tune_rows.py builds and compares variants)
#ifndef PRL_ROWS_NT_LOAD
#define PRL_ROWS_NT_LOAD 1  // non-temporal hint on the row loads (the logits are read exactly once)
#endif
#ifndef PRL_ROWS_NT_STORE
#define PRL_ROWS_NT_STORE 1  // non-temporal hint on the gradient stores (3 % faster; the 622 MB written exceed the 256 MB Infinity Cache anyway)
#endif
#ifndef PRL_ROWS_STORE_AUX
// cache-policy bits of the gradient stores (gfx942 family: 1 = sc0, 2 = nt, 16 = sc1).  Stores written through
// the L2 at device or system scope (sc1, sc0 + sc1, with and without nt), so that the end of the kernel finds no
// dirty lines to write back, measured 221.7-223.5 us against 222.8 us: no difference.
#define PRL_ROWS_STORE_AUX (PRL_ROWS_NT_STORE ? 2 : 0)
#endif
#ifndef PRL_ROWS_XCD_BLOCKS
#define PRL_ROWS_XCD_BLOCKS 1  // 1: each of the 8 XCDs works through one contiguous eighth of the rows
#endif

// Two other ways of going from one row to the next were built and measured against the default (a fresh workgroup
// per row), T = 2048, V = 151936, every token with a gradient, interleaved random-order rounds in one process:
//   fresh workgroup per row (default)                                    218.6 us
//   PRL_ROWS_WALK=1: one workgroup per compute unit walks over its rows,
//     the next row's loads issued right behind this row's last store     223.5 us
//   PRL_ROWS_WALK=1 PRL_ROWS_INTERLEAVE=1: the same, the next row's chunk
//     j asked for right behind the store of this row's chunk j (reads and
//     writes mixed inside a compute unit, waits counted by hand)         240.4 us
// Both pass the parity tests; neither is compiled unless its flag is set.  Walking saves the workgroup turnover but
// deals the rows out statically; mixing reads into the write stream delays both: a wave's loads and stores retire
// through one in-order counter, so the next row's first chunk is not usable before this row's stores are
// acknowledged, and the sums pass, which a fresh workgroup runs under its own loads, then starts late.
#ifndef PRL_ROWS_WALK
#define PRL_ROWS_WALK 0
#endif
#ifndef PRL_ROWS_INTERLEAVE
#define PRL_ROWS_INTERLEAVE 0  // needs PRL_ROWS_WALK=1
#endif

#ifndef PRL_ROWS_TRACE
#define PRL_ROWS_TRACE 0  // measurement aid: 1 writes phase durations (in 10 ns ticks) over logprob / entropy / lse
#endif

namespace prl {

constexpr int kRowThreads = 1024;
constexpr int kRowWaves = kRowThreads / kWave;
constexpr int kRowMaxChunks = 19;  // 16-byte chunks per lane: 19 * 1024 * 8 = 155,648 bf16 logits
constexpr int kTailSlabs = 4;      // the built register-array sizes are at most this far apart

// The row lives in registers as packed bf16 pairs and is unpacked once per pass over it.  Left
// alone, the compiler keeps the unpacked floats of one pass for the next, which doubles the
// register need and spills; this empty statement makes a packed chunk opaque, so that every pass
// unpacks again (one ALU operation per element, no memory traffic).  It is also where the wait
// for the chunk's load lands, so work on chunk j starts while later chunks are still in flight.
__device__ __forceinline__ void touch(u32x4& v) { asm volatile("" : "+v"(v)); }
// The same, tied to a value the previous chunk's arithmetic produced, so that the waits of later
// chunks cannot be moved up in front of earlier chunks' work.
__device__ __forceinline__ void touch_after(u32x4& v, float& prev) { asm volatile("" : "+v"(v), "+v"(prev)); }

// Workgroup-uniform reads of one byte / one bf16 as aligned 32-bit words.  Sub-word loads have no
// scalar form on this architecture; as vector loads they would queue behind the row loads in flight
// and make the row wait for all of them.  The word read never leaves the aligned 4 bytes that
// hold the requested element.  `words` is the 4-byte aligned address at or below the array, `mis`
// the array's offset from it.
__device__ __forceinline__ uint32_t uniform_byte(const uint32_t* __restrict__ words, int mis, int64_t i) {
  const int64_t k = i + mis;
  return (words[k >> 2] >> ((uint32_t)(k & 3) * 8u)) & 0xffu;
}
__device__ __forceinline__ float uniform_bf16(const uint32_t* __restrict__ words, int64_t i) {
  const uint32_t w = words[i >> 1];
  return (i & 1) ? bf16_hi(w) : bf16_lo(w);
}

// 16-byte buffer store with a scalar slab offset.  On gfx950 the store still reads its data registers
// for a moment after it has issued: a vector ALU write to them in the very next instruction reached
// memory instead of the intended data (seen as the first 8 bytes of a chunk replaced by an unrelated
// 64-bit move).  The compiler's hazard table exempts exactly this form (data wider than 64 bits WITH a
// register soffset), so the wait states are put in by hand and fenced against rescheduling.
template <typename R>
__device__ __forceinline__ void store_chunk(const u32x4& v, R rsrc, uint32_t voffset, int soffset) {
  __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voffset, soffset, PRL_ROWS_STORE_AUX);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 1");  // two wait states
  __builtin_amdgcn_sched_barrier(0);
}

// One workgroup per row.  (Measured alternatives, T = 2048, V = 151936: workgroups that stay on a
// compute unit and refill each chunk's registers with their next row while the gradient is written
// were 8 % slower, because loads and stores retire through one in-order counter and the next row's
// first chunk then waits for this row's stores; workgroups that simply walk over several rows, one
// or two per compute unit, were no faster than fresh workgroups and cost registers; pulling a later
// row towards the caches with one dword per line cost 25 %.)
// GRAD = false: the row is only scored (logprob, entropy, lse); `dlogits` is not touched and may be NULL.
template <int NCH, bool GRAD = true>
__global__ __launch_bounds__(kRowThreads) void k_policy_rows(
    const char* __restrict__ logits, int64_t T, int64_t V, int64_t in_stride_bytes, const int64_t* __restrict__ ids,
    const float* __restrict__ old, const float* __restrict__ adv, const uint32_t* __restrict__ mask_words,
    int mask_mis, int skip_masked, float inv_temp, float clip, float inv_scale, float* __restrict__ lp_out,
    float* __restrict__ ent_out, float* __restrict__ lse_out, char* __restrict__ dlogits, int64_t out_stride_bytes,
    const int64_t* __restrict__ positions, int64_t n_rows) {
  __shared__ float red_m[kRowWaves], red_s[kRowWaves], red_w[kRowWaves];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
#if PRL_ROWS_XCD_BLOCKS
  // workgroups go to the XCDs round-robin (workgroup b runs on XCD b % 8); gridDim.x is a multiple of 8
  // (Every XCD starting its eighth at another place, so that the rows the eight XCDs work on at a time do not lie a
  // multiple of 64 KB apart, changes nothing: 222.1-222.5 against 220.4-221.2 us.)
  const int64_t row = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
#else
  const int64_t row = blockIdx.x;  // row of the logits and of the gradient
#endif
  if (row >= n_rows) return;
#if PRL_ROWS_TRACE
  const uint64_t tick0 = wall_clock64();
#endif
  // The token this row predicts.  Whole packed row: row r holds the logits at position r (n_rows == T).  Gathered
  // rows (`positions` given): row r holds the logits the model produced at position positions[r] of a packed row
  // of T tokens; token 0 and the tokens no row predicts are the caller's business.
  const int64_t at = positions ? positions[row] : row;  // position of the packed row this row of logits belongs to
  const int64_t t = (int64_t)((uint64_t)at + 1u);       // (not used when `at` is out of range, see below)
  const int nchunk = (int)(V >> 3);
  const float c2 = inv_temp * kLog2e;

  // Addressing: buffer instructions with the row as the buffer.  A lane's NCH loads (and stores)
  // share one 32-bit offset register, the slab offset is a scalar, and the hardware drops stores and
  // returns zeros for loads beyond the row's V * 2 bytes, so there are no bounds branches.
  const uint32_t lane_off = (uint32_t)tid * 16u;
  constexpr int kSlab = kRowThreads * 16;
  constexpr int kRsrcFlags = 0x00020000;  // raw buffer, 32-bit data format (gfx9 family descriptor word 3)
  const int row_bytes = (int)(V * 2);
  const auto out_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(dlogits + row * out_stride_bytes, 0, row_bytes, kRsrcFlags);
  auto zero_row = [&]() {
    if constexpr (GRAD) {
      const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < NCH; ++j) store_chunk(z, out_rsrc, lane_off, j * kSlab);
    }
  };

  if (positions && (at < 0 || at >= T - 1)) {  // a position outside the packed row: nothing is read, nothing is emitted
    zero_row();
    return;
  }
  if (t >= T) {
    // the last row predicts nothing; its workgroup also emits token 0, which no row predicts
    if (tid == 0) {
      if (skip_masked && uniform_byte(mask_words, mask_mis, 0) == 0)
        emit_skipped(0, lp_out, ent_out, lse_out);
      else
        emit_zero_row(V, 0, lp_out, ent_out, lse_out);
    }
    zero_row();
    return;
  }
  const bool m = uniform_byte(mask_words, mask_mis, t) != 0;
  if (skip_masked && !m) {  // a token nobody looks at: no read, outputs 0, gradient 0
    if (tid == 0) emit_skipped(t, lp_out, ent_out, lse_out);
    zero_row();
    return;
  }

  // 1. the whole row into registers, issued in chunk order: the arithmetic consumes them in that order
  const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(logits + row * in_stride_bytes), 0,
                                                         row_bytes, kRsrcFlags);
  u32x4 buf[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    buf[j] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, lane_off, j * kSlab, PRL_ROWS_NT_LOAD ? 2 : 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  const int64_t label = ids[t];
  float x_label = __builtin_nanf("");
  if (label >= 0 && label < V)
    x_label = uniform_bf16(reinterpret_cast<const uint32_t*>(logits) + row * (in_stride_bytes >> 2), label);
  const float old_t = old[t], adv_t = adv[t];

  // Chunk j for arithmetic: waits for its load.  The row ends in one of the last kTailSlabs slabs (the
  // host picks the smallest NCH that holds the row); there, lanes past the end get the most negative
  // finite bf16 instead of the zeros the hardware returned.  j is a constant once the loops are
  // unrolled, so the other slabs pay nothing for this.
  auto chunk = [&](int j, float& prev) -> u32x4 {
    touch_after(buf[j], prev);
    u32x4 v = buf[j];
    if (j >= NCH - kTailSlabs) {
      const bool in = tid < nchunk - j * kRowThreads;
      v.x = in ? v.x : 0xff7fff7fu;
      v.y = in ? v.y : 0xff7fff7fu;
      v.z = in ? v.z : 0xff7fff7fu;
      v.w = in ? v.w : 0xff7fff7fu;
    }
    return v;
  };
  // Per-wave sums of 2^(x c2 - M) and x 2^(x c2 - M) over the registers, then merged over the 16 waves.
  // Every wave uses a reference point M of its own, so no wave waits for another before it starts; the
  // merge rescales each wave's sums to the largest M.
  auto sums = [&](float M, float& M_out, float& s_out, float& w_out) {
    float es[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ws[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      float x[8];
      decode16_raw<PRL_BF16>(chunk(j, ws[3]), x);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float e = fast_exp2(fmaf(x[i], c2, -M));
        es[i & 3] += e;
        ws[i & 3] = fmaf(x[i], e, ws[i & 3]);
      }
    }
    const float s = wave_sum((es[0] + es[2]) + (es[1] + es[3]));
    const float w = wave_sum((ws[0] + ws[2]) + (ws[1] + ws[3]));
    if (lane == 0) {
      red_m[wid] = M;
      red_s[wid] = s;
      red_w[wid] = w;
    }
    __syncthreads();
    float Mx = red_m[0];
#pragma unroll
    for (int i = 1; i < kRowWaves; ++i) Mx = fmaxf(Mx, red_m[i]);
    float st = 0.0f, wt = 0.0f;
#pragma unroll
    for (int i = 0; i < kRowWaves; ++i) {
      const float sc = fast_exp2(red_m[i] - Mx);
      st = fmaf(red_s[i], sc, st);
      wt = fmaf(red_w[i], sc, wt);
    }
    M_out = Mx;
    s_out = st;
    w_out = wt;
  };

  // 2. Reference point of a wave's exponentials: the maximum over its lanes' first chunks (512 logits).
  //    Any M <= max + ~100/c2 gives the same sums up to rounding; an M so far below the true maximum
  //    that a sum overflows is caught below and replaced by the exact maximum of the row.
  float mx = kNegBig;
  {
    float x[8];
    decode16_raw<PRL_BF16>(chunk(0, mx), x);
#pragma unroll
    for (int i = 0; i < 8; ++i) mx = fmaxf(mx, x[i]);
  }
  float M, s, w;
#if PRL_ROWS_TRACE
  const uint64_t tick1 = wall_clock64();  // this wave's first chunk has arrived
#endif
  sums(wave_max(mx) * c2, M, s, w);  // c2 > 0: the largest logit gives the largest scaled logit
#if PRL_ROWS_TRACE
  const uint64_t tick2 = wall_clock64();  // all chunks arrived and summed, in every wave
#endif
  // overflow of either sum (x 2^(..) overflows before 2^(..) does), or a NaN / -inf sample: exact two-pass
  // soft-max.  A NaN in w alone (0 * -inf from a masked-out vocabulary entry) is the reference's result too.
  if (!(s <= 3.0e38f) || !(s > 0.0f) || fabsf(w) > 3.0e38f) {
    __syncthreads();                     // every wave has read the first round's sums before they are rewritten
    mx = kNegBig;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      float x[8];
      decode16_raw<PRL_BF16>(chunk(j, mx), x);
#pragma unroll
      for (int i = 0; i < 8; ++i) mx = fmaxf(mx, x[i]);
    }
    mx = wave_max(mx);
    if (lane == 0) red_m[wid] = mx;
    __syncthreads();
    mx = red_m[0];
#pragma unroll
    for (int i = 1; i < kRowWaves; ++i) mx = fmaxf(mx, red_m[i]);
    __syncthreads();  // red_m is rewritten by the sums below
    sums(mx * c2, M, s, w);
  }

  // 3. the token's values and its loss gradient (same expressions as the loss kernels, so that the g
  //    they report is the g used here)
  const float lse2 = M + fast_log2(s);
  const float lse = lse2 * kLn2;
  const float lp = x_label * inv_temp - lse;
  if (tid == 0) {
    if (lse_out) lse_out[t] = lse;
    if (lp_out) lp_out[t] = lp;
    if (ent_out) ent_out[t] = lse - (w / s) * inv_temp;
  }
  if constexpr (!GRAD) return;
  const float r = expf(lp - old_t);
  const float through = (r <= clip) ? -(inv_scale * adv_t) : 0.0f;
  const float g = m ? through * r : 0.0f;
  const float gi = g * inv_temp;
  if (gi == 0.0f) {  // masked, clipped or zero advantage: nothing to learn from this token
    zero_row();
    return;
  }

  // 4. The gradient row from the registers.  Away from the label, d = -gi * 2^(x c2 - lse2)
  //    = -+2^(x c2 - lse2 + log2|gi|): the factor goes into the exponent and the sign onto the packed
  //    pair, which leaves one fma and one exp per element.
  const float shift2 = fast_log2(fabsf(gi)) - lse2;
  const uint32_t sign2 = gi > 0.0f ? 0x80008000u : 0u;  // d has the sign of -gi (a NaN gi gives NaN exponents)
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    touch(buf[j]);
    float x[8], d[8];
    decode16_raw<PRL_BF16>(buf[j], x);  // lanes past the row's end work on zeros; their stores are dropped
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = fast_exp2(fmaf(x[i], c2, shift2));
    u32x4 v;
    v.x = pack_bf16(d[0], d[1]) | sign2;
    v.y = pack_bf16(d[2], d[3]) | sign2;
    v.z = pack_bf16(d[4], d[5]) | sign2;
    v.w = pack_bf16(d[6], d[7]) | sign2;
    store_chunk(v, out_rsrc, lane_off, j * kSlab);
  }
#if PRL_ROWS_TRACE
  if (tid == 0) {
    lp_out[t] = (float)(tick1 - tick0);
    ent_out[t] = (float)(tick2 - tick1);
    lse_out[t] = (float)(wall_clock64() - tick2);
  }
#endif
  // the label's element: gi * (1 - p), written by the lane that stored its chunk, after that store
  if (label >= 0 && label < V && tid == (int)((label >> 3) & (kRowThreads - 1))) {
    const float p = fast_exp2(fmaf(x_label, c2, -lse2));
    reinterpret_cast<uint16_t*>(dlogits + row * out_stride_bytes)[label] = one_bf16(fmaf(-gi, p, gi));
  }
}

#if PRL_ROWS_WALK && !PRL_ROWS_INTERLEAVE
// The same row arithmetic with the workgroups walking: a launch has one workgroup per compute unit (a row's
// registers leave room for no second one), and each walks over its share of the rows.  What that buys: the
// loads of the next row are issued right behind the last store of this row's gradient, so the memory system
// goes from writing to reading without waiting for a workgroup to end and another to start.  The loads are
// never interleaved with the stores (a row's gradient is written in one stretch, the next row is then read in
// one stretch): the compute units run through the same phases at about the same time, and pure read and pure
// write stretches stream faster than a mix.  Rows are dealt out statically: XCD x takes the x-th eighth of the
// rows, its workgroups take every (workgroups per XCD)-th row of that eighth.
template <int NCH>
__global__ __launch_bounds__(kRowThreads) void k_policy_rows_walk(
    const char* __restrict__ logits, int64_t T, int64_t V, int64_t in_stride_bytes, const int64_t* __restrict__ ids,
    const float* __restrict__ old, const float* __restrict__ adv, const uint32_t* __restrict__ mask_words,
    int mask_mis, int skip_masked, float inv_temp, float clip, float inv_scale, float* __restrict__ lp_out,
    float* __restrict__ ent_out, float* __restrict__ lse_out, char* __restrict__ dlogits, int64_t out_stride_bytes) {
  __shared__ float red_m[kRowWaves], red_s[kRowWaves], red_w[kRowWaves];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
  // (the host checks T < 2^31: rows are counted in 32 bits, which keeps the loop's state in few scalar registers)
  const int nrows = (int)T;
  const int per_xcd = (int)(gridDim.x >> 3);                // gridDim.x is a multiple of 8
  const int rows_per_xcd = (nrows + 7) >> 3;
  const int xcd_first = (int)(blockIdx.x & 7) * rows_per_xcd;
  int xcd_last = xcd_first + rows_per_xcd;                  // one past this XCD's last row
  if (xcd_last > nrows) xcd_last = nrows;
  int row = xcd_first + (int)(blockIdx.x >> 3);
  if (row >= xcd_last) return;

  const int nchunk = (int)(V >> 3);
  const float c2 = inv_temp * kLog2e;
  const uint32_t lane_off = (uint32_t)tid * 16u;
  constexpr int kSlab = kRowThreads * 16;
  constexpr int kRsrcFlags = 0x00020000;
  const int row_bytes = (int)(V * 2);

  // whether a row's logits are read at all: not the last row (it predicts nothing) and not a row whose token
  // nobody looks at
  auto row_is_read = [&](int r) -> bool {
    if (r + 1 >= nrows) return false;
    return !(skip_masked && uniform_byte(mask_words, mask_mis, r + 1) == 0);
  };
  u32x4 buf[NCH];
  // The slab offsets j * kSlab go into a scalar register each.  Left alone, the compiler keeps all of them
  // alive around the loop (19 registers for the loads, more for the stores) and spills; a zero it cannot see
  // through makes it form each offset where it is used (one scalar add).
  auto opaque_zero = []() -> int {
    int z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    return z;
  };
  auto issue_loads = [&](int r) {
    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(logits + (int64_t)r * in_stride_bytes),
                                                           0, row_bytes, kRsrcFlags);
    const int z = opaque_zero();
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      buf[j] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, lane_off, z + j * kSlab, PRL_ROWS_NT_LOAD ? 2 : 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  bool have = row_is_read(row);
  if (have) issue_loads(row);
  while (true) {
    const int next = row + per_xcd;
    const bool more = next < xcd_last;
    const bool next_have = more && row_is_read(next);
    const int t = row + 1;
    const auto out_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(dlogits + (int64_t)row * out_stride_bytes, 0, row_bytes, kRsrcFlags);
    // One path for every kind of row, so that the row's registers have one life around the loop: a row that is
    // not read (or whose token turns out to have no gradient) goes through the gradient pass with gi = 0 and
    // stores zeros; what that pass computes from whatever the registers hold is thrown away.
    float gi = 0.0f, lse2 = 0.0f, x_label = 0.0f;
    int64_t label = -1;
    if (!have) {
      if (tid == 0) {
        if (t >= nrows) {  // the last row predicts nothing; its workgroup also emits token 0, which no row predicts
          if (skip_masked && uniform_byte(mask_words, mask_mis, 0) == 0)
            emit_skipped(0, lp_out, ent_out, lse_out);
          else
            emit_zero_row(V, 0, lp_out, ent_out, lse_out);
        } else {
          emit_skipped(t, lp_out, ent_out, lse_out);
        }
      }
    } else {
      const bool m = uniform_byte(mask_words, mask_mis, t) != 0;
      label = ids[t];
      x_label = __builtin_nanf("");
      if (label >= 0 && label < V)
        x_label = uniform_bf16(reinterpret_cast<const uint32_t*>(logits) + (int64_t)row * (in_stride_bytes >> 2), label);
      const float old_t = old[t], adv_t = adv[t];
      auto chunk = [&](int j, float& prev) -> u32x4 {
        touch_after(buf[j], prev);
        u32x4 v = buf[j];
        if (j >= NCH - kTailSlabs) {
          const bool in = tid < nchunk - j * kRowThreads;
          v.x = in ? v.x : 0xff7fff7fu;
          v.y = in ? v.y : 0xff7fff7fu;
          v.z = in ? v.z : 0xff7fff7fu;
          v.w = in ? v.w : 0xff7fff7fu;
        }
        return v;
      };
      auto sums = [&](float M, float& M_out, float& s_out, float& w_out) {
        float es[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ws[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
          float x[8];
          decode16_raw<PRL_BF16>(chunk(j, ws[3]), x);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float e = fast_exp2(fmaf(x[i], c2, -M));
            es[i & 3] += e;
            ws[i & 3] = fmaf(x[i], e, ws[i & 3]);
          }
        }
        const float s = wave_sum((es[0] + es[2]) + (es[1] + es[3]));
        const float w = wave_sum((ws[0] + ws[2]) + (ws[1] + ws[3]));
        if (lane == 0) {
          red_m[wid] = M;
          red_s[wid] = s;
          red_w[wid] = w;
        }
        __syncthreads();
        float Mx = red_m[0];
#pragma unroll
        for (int i = 1; i < kRowWaves; ++i) Mx = fmaxf(Mx, red_m[i]);
        float st = 0.0f, wt = 0.0f;
#pragma unroll
        for (int i = 0; i < kRowWaves; ++i) {
          const float sc = fast_exp2(red_m[i] - Mx);
          st = fmaf(red_s[i], sc, st);
          wt = fmaf(red_w[i], sc, wt);
        }
        M_out = Mx;
        s_out = st;
        w_out = wt;
      };
      float mx = kNegBig;
      {
        float x[8];
        decode16_raw<PRL_BF16>(chunk(0, mx), x);
#pragma unroll
        for (int i = 0; i < 8; ++i) mx = fmaxf(mx, x[i]);
      }
      float M, s, w;
      sums(wave_max(mx) * c2, M, s, w);
      if (!(s <= 3.0e38f) || !(s > 0.0f) || fabsf(w) > 3.0e38f) {  // see k_policy_rows: exact two-pass soft-max
        __syncthreads();
        mx = kNegBig;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
          float x[8];
          decode16_raw<PRL_BF16>(chunk(j, mx), x);
#pragma unroll
          for (int i = 0; i < 8; ++i) mx = fmaxf(mx, x[i]);
        }
        mx = wave_max(mx);
        if (lane == 0) red_m[wid] = mx;
        __syncthreads();
        mx = red_m[0];
#pragma unroll
        for (int i = 1; i < kRowWaves; ++i) mx = fmaxf(mx, red_m[i]);
        __syncthreads();
        sums(mx * c2, M, s, w);
      }
      __syncthreads();  // the next row's sums rewrite red_*: every wave has read this row's first
      lse2 = M + fast_log2(s);
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
      gi = g * inv_temp;
    }
    {
      const bool live = gi != 0.0f;  // (a NaN counts as a gradient, as in the reference)
      const float shift2 = fast_log2(fabsf(gi)) - lse2;
      const uint32_t sign2 = gi > 0.0f ? 0x80008000u : 0u;
      const uint32_t keep = live ? 0xffffffffu : 0u;
      const int z0 = opaque_zero();
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        touch(buf[j]);
        float x[8], d[8];
        decode16_raw<PRL_BF16>(buf[j], x);
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = fast_exp2(fmaf(x[i], c2, shift2));
        u32x4 v;
        v.x = (pack_bf16(d[0], d[1]) | sign2) & keep;
        v.y = (pack_bf16(d[2], d[3]) | sign2) & keep;
        v.z = (pack_bf16(d[4], d[5]) | sign2) & keep;
        v.w = (pack_bf16(d[6], d[7]) | sign2) & keep;
        store_chunk(v, out_rsrc, lane_off, z0 + j * kSlab);
      }
      if (live && label >= 0 && label < V && tid == (int)((label >> 3) & (kRowThreads - 1))) {
        const float p = fast_exp2(fmaf(x_label, c2, -lse2));
        reinterpret_cast<uint16_t*>(dlogits + (int64_t)row * out_stride_bytes)[label] = one_bf16(fmaf(-gi, p, gi));
      }
    }
    if (!more) break;
    if (next_have) issue_loads(next);  // right behind this row's last store
    row = next;
    have = next_have;
  }
}

#endif  // PRL_ROWS_WALK && !PRL_ROWS_INTERLEAVE

#if PRL_ROWS_WALK && PRL_ROWS_INTERLEAVE
// Walking workgroups with reads and writes mixed: while a row's gradient is written chunk by chunk, chunk j of
// the workgroup's NEXT row is asked for into the registers chunk j has just left, so a compute unit has loads and
// stores in flight together and starts summing the next row as soon as its gradient pass is over.
//
// The compiler's wait-count pass cannot describe this (it merges the prologue, where 19 loads are in flight, with
// the loop, where 19 stores and 19 loads alternate, and waits for half of everything before the first chunk), so
// the loads are issued from inline assembly, which it does not track, and every wait is written by hand.  Vector
// memory operations of a wave retire through ONE in-order counter on this architecture: "at most N operations
// outstanding" means everything but the youngest N has completed.  Behind the load of chunk j the wave issues
// 18 - j loads (first row) or 2 * (18 - j) stores and loads (later rows); operations a wave may or may not issue
// on top (the three per-token results, the label's element) only make a wait stricter.
template <int NCH>
__global__ __launch_bounds__(kRowThreads) void k_policy_rows_mix(
    const char* __restrict__ logits, int64_t T, int64_t V, int64_t in_stride_bytes, const int64_t* __restrict__ ids,
    const float* __restrict__ old, const float* __restrict__ adv, const uint32_t* __restrict__ mask_words,
    int mask_mis, int skip_masked, float inv_temp, float clip, float inv_scale, float* __restrict__ lp_out,
    float* __restrict__ ent_out, float* __restrict__ lse_out, char* __restrict__ dlogits, int64_t out_stride_bytes) {
  __shared__ float red_m[kRowWaves], red_s[kRowWaves], red_w[kRowWaves];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
  const int nrows = (int)T;
  const int per_xcd = (int)(gridDim.x >> 3);
  const int rows_per_xcd = (nrows + 7) >> 3;
  const int xcd_first = (int)(blockIdx.x & 7) * rows_per_xcd;
  int xcd_last = xcd_first + rows_per_xcd;
  if (xcd_last > nrows) xcd_last = nrows;
  int row = xcd_first + (int)(blockIdx.x >> 3);
  if (row >= xcd_last) return;

  const int nchunk = (int)(V >> 3);
  const float c2 = inv_temp * kLog2e;
  const uint32_t lane_off = (uint32_t)tid * 16u;
  constexpr int kSlab = kRowThreads * 16;
  constexpr int kRsrcFlags = 0x00020000;
  const int row_bytes = (int)(V * 2);

  auto row_is_read = [&](int r) -> bool {
    if (r + 1 >= nrows) return false;
    return !(skip_masked && uniform_byte(mask_words, mask_mis, r + 1) == 0);
  };
  auto opaque_zero = []() -> int {
    int z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    return z;
  };
  // the row's buffer descriptor as four scalar words (what make_buffer_rsrc builds for the intrinsics)
  auto descriptor = [&](const char* base) -> u32x4 {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    u32x4 d;
    d.x = (uint32_t)a;
    d.y = (uint32_t)(a >> 32) & 0xffffu;
    d.z = (uint32_t)row_bytes;
    d.w = (uint32_t)kRsrcFlags;
    return d;
  };
  u32x4 buf[NCH];
  auto load_chunk = [&](int j, const u32x4& rsrc, int soffset) {
    // (no "memory" clobber: with one the compiler has to assume that the logits change under it and reads the
    // label's element with a vector load, whose wait drains every load in flight; the position among the stores
    // is pinned by the scheduling barriers on both sides)
    __builtin_amdgcn_sched_barrier(0);
#if PRL_ROWS_NT_LOAD
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "=v"(buf[j]) : "v"(lane_off), "s"(rsrc), "s"(soffset));
#else
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(buf[j]) : "v"(lane_off), "s"(rsrc), "s"(soffset));
#endif
    __builtin_amdgcn_sched_barrier(0);
  };

  bool have = row_is_read(row);
  int first_i = 1;  // 1: the row in the registers was asked for by the prologue (only loads behind each load)
  if (have) {
    const u32x4 rsrc = descriptor(logits + (int64_t)row * in_stride_bytes);
    const int z = opaque_zero();
#pragma unroll
    for (int j = 0; j < NCH; ++j) load_chunk(j, rsrc, z + j * kSlab);
  }
  while (true) {
    // Values derived from the lane's index (tail predicates, the label's owner, ...) are cheap; left to itself the
    // compiler computes them once in front of the loop and keeps every one of them in a register around it,
    // which the row's 76 registers leave no room for.  An index it cannot see through is recomputed per row.
    int tid_l = tid;
    asm volatile("" : "+v"(tid_l));
    int64_t V_l = V;
    asm volatile("" : "+s"(V_l));
    const int next = row + per_xcd;
    const bool more = next < xcd_last;
    const bool next_have = more && row_is_read(next);
    const int t = row + 1;
    const auto out_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(dlogits + (int64_t)row * out_stride_bytes, 0, row_bytes, kRsrcFlags);
    float gi = 0.0f, lse2 = 0.0f, x_label = 0.0f;
    int64_t label = -1;
    if (!have) {
      if (tid_l == 0) {
        if (t >= nrows) {
          if (skip_masked && uniform_byte(mask_words, mask_mis, 0) == 0)
            emit_skipped(0, lp_out, ent_out, lse_out);
          else
            emit_zero_row(V_l, 0, lp_out, ent_out, lse_out);
        } else {
          emit_skipped(t, lp_out, ent_out, lse_out);
        }
      }
    } else {
      const bool m = uniform_byte(mask_words, mask_mis, t) != 0;
      label = ids[t];
      x_label = __builtin_nanf("");
      if (label >= 0 && label < V) {
        // the label's logit through the scalar cache, spelled out: the compiler no longer proves that the logits
        // stay unchanged (their address goes into the assembly loads) and would use a vector load, whose wait
        // drains every row load in flight
        const uint32_t* word = reinterpret_cast<const uint32_t*>(logits) + (int64_t)row * (in_stride_bytes >> 2) + (label >> 1);
        uint32_t pair;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pair) : "s"(word));
        x_label = (label & 1) ? bf16_hi(pair) : bf16_lo(pair);
      }
      const float old_t = old[t], adv_t = adv[t];
      // Chunk j for arithmetic: waits until at most 18 - j (row asked for by the prologue: only loads behind
      // each load) or 2 * (18 - j) (row asked for during a gradient pass: a store and a load alternate)
      // operations of this wave are outstanding.  One copy of the code for both: a scalar branch around the wait.
      auto chunk = [&](int j, float& prev) -> u32x4 {
        asm volatile(
            "s_cmp_eq_u32 %2, 0\n\t"
            "s_cbranch_scc1 1f\n\t"
            "s_waitcnt vmcnt(%3)\n\t"
            "s_branch 2f\n"
            "1:\n\t"
            "s_waitcnt vmcnt(%4)\n"
            "2:"
            : "+v"(buf[j]), "+v"(prev)
            : "s"(first_i), "n"(NCH - 1 - j), "n"(2 * (NCH - 1 - j))
            : "scc");
        u32x4 v = buf[j];
        if (j >= NCH - kTailSlabs) {
          const bool in = tid_l < nchunk - j * kRowThreads;
          v.x = in ? v.x : 0xff7fff7fu;
          v.y = in ? v.y : 0xff7fff7fu;
          v.z = in ? v.z : 0xff7fff7fu;
          v.w = in ? v.w : 0xff7fff7fu;
        }
        return v;
      };
      auto sums = [&](float M, float& M_out, float& s_out, float& w_out) {
        float es[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ws[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
          float x[8];
          decode16_raw<PRL_BF16>(chunk(j, ws[3]), x);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float e = fast_exp2(fmaf(x[i], c2, -M));
            es[i & 3] += e;
            ws[i & 3] = fmaf(x[i], e, ws[i & 3]);
          }
        }
        const float s = wave_sum((es[0] + es[2]) + (es[1] + es[3]));
        const float w = wave_sum((ws[0] + ws[2]) + (ws[1] + ws[3]));
        if (lane == 0) {
          red_m[wid] = M;
          red_s[wid] = s;
          red_w[wid] = w;
        }
        __syncthreads();
        float Mx = red_m[0];
#pragma unroll
        for (int i = 1; i < kRowWaves; ++i) Mx = fmaxf(Mx, red_m[i]);
        float st = 0.0f, wt = 0.0f;
#pragma unroll
        for (int i = 0; i < kRowWaves; ++i) {
          const float sc = fast_exp2(red_m[i] - Mx);
          st = fmaf(red_s[i], sc, st);
          wt = fmaf(red_w[i], sc, wt);
        }
        M_out = Mx;
        s_out = st;
        w_out = wt;
      };
      float mx = kNegBig;
      float M, s, w;
      {
        float x[8];
        decode16_raw<PRL_BF16>(chunk(0, mx), x);
#pragma unroll
        for (int i = 0; i < 8; ++i) mx = fmaxf(mx, x[i]);
      }
      sums(wave_max(mx) * c2, M, s, w);
      if (!(s <= 3.0e38f) || !(s > 0.0f) || fabsf(w) > 3.0e38f) {  // see k_policy_rows: exact two-pass soft-max
        __syncthreads();
        mx = kNegBig;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
          float x[8];
          decode16_raw<PRL_BF16>(chunk(j, mx), x);
#pragma unroll
          for (int i = 0; i < 8; ++i) mx = fmaxf(mx, x[i]);
        }
        mx = wave_max(mx);
        if (lane == 0) red_m[wid] = mx;
        __syncthreads();
        mx = red_m[0];
#pragma unroll
        for (int i = 1; i < kRowWaves; ++i) mx = fmaxf(mx, red_m[i]);
        __syncthreads();
        sums(mx * c2, M, s, w);  // (its waits are met already)
      }
      __syncthreads();  // the next row's sums rewrite red_*: every wave has read this row's first
      lse2 = M + fast_log2(s);
      const float lse = lse2 * kLn2;
      const float lp = x_label * inv_temp - lse;
      if (tid_l == 0) {
        if (lse_out) lse_out[t] = lse;
        if (lp_out) lp_out[t] = lp;
        if (ent_out) ent_out[t] = lse - (w / s) * inv_temp;
      }
      const float r = expf(lp - old_t);
      const float through = (r <= clip) ? -(inv_scale * adv_t) : 0.0f;
      const float g = m ? through * r : 0.0f;
      gi = g * inv_temp;
    }
    {
      const bool live = gi != 0.0f;
      const float shift2 = fast_log2(fabsf(gi)) - lse2;
      const uint32_t sign2 = gi > 0.0f ? 0x80008000u : 0u;
      const uint32_t keep = live ? 0xffffffffu : 0u;
      const int z0 = opaque_zero();
      const u32x4 next_rsrc = descriptor(logits + (int64_t)(next_have ? next : row) * in_stride_bytes);
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        touch(buf[j]);
        float x[8], d[8];
        decode16_raw<PRL_BF16>(buf[j], x);
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = fast_exp2(fmaf(x[i], c2, shift2));
        u32x4 v;
        v.x = (pack_bf16(d[0], d[1]) | sign2) & keep;
        v.y = (pack_bf16(d[2], d[3]) | sign2) & keep;
        v.z = (pack_bf16(d[4], d[5]) | sign2) & keep;
        v.w = (pack_bf16(d[6], d[7]) | sign2) & keep;
        store_chunk(v, out_rsrc, lane_off, z0 + j * kSlab);
        // chunk j of the next row into the registers chunk j of this row has just left (the uniform branch
        // keeps the pattern "one store, one load" whenever there is a next row to read)
        if (next_have)
          load_chunk(j, next_rsrc, z0 + j * kSlab);
        else
          asm volatile("" : "=v"(buf[j]));  // (dead either way: one life per chunk for the register allocator)
      }
      if (live && label >= 0 && label < V && tid_l == (int)((label >> 3) & (kRowThreads - 1))) {
        const float p = fast_exp2(fmaf(x_label, c2, -lse2));
        reinterpret_cast<uint16_t*>(dlogits + (int64_t)row * out_stride_bytes)[label] = one_bf16(fmaf(-gi, p, gi));
      }
    }
    if (!more) break;
    row = next;
    have = next_have;
    first_i = 0;
  }
}

#endif  // PRL_ROWS_WALK && PRL_ROWS_INTERLEAVE

// ---------------------------------------------------------------- sequence mode in one pass
//
// With one ratio per sample (src/prime_rl/trainer/rl/loss.py:92-96, 107-108) a token's gradient needs the summed
// log-ratio of its whole sample, i.e. the log-probabilities of up to a few hundred other rows.  Those rows are being
// worked on at the same time: workgroups start in the order of their index, one per compute unit, so the
// workgroups of a sample of up to ~250 tokens are all resident together.  Each publishes its token's masked
// log-ratio as an 8-byte {tag, value} word (a write-through store at device scope, the data is its own flag),
// then one wave re-reads the words of the workgroup's sample until all carry the tag, sums them in a fixed order
// and the row's gradient is written from the registers as in token mode: the logits are read once.
//
// Nothing here relies on the start order for CORRECTNESS: the wait is bounded (`spin_limit` sweeps), and a workgroup
// that gives up, or whose sample is longer than `max_wait_len`, writes no gradient and marks its row in `pending`;
// the caller then runs prl_logprob_bwd_selected over the marked rows (a second read of those rows only).  A
// workgroup publishes before it waits and waits for nothing but published words, so every workgroup ends.
__device__ __forceinline__ void publish_log_ratio(unsigned long long* slots, int64_t t, float lr) {
  __hip_atomic_store(slots + t, (1ull << 32) | (unsigned long long)__float_as_uint(lr), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

template <int NCH>
__global__ __launch_bounds__(kRowThreads) void k_policy_rows_seq(
    const char* __restrict__ logits, int64_t T, int64_t V, int64_t in_stride_bytes, const int64_t* __restrict__ ids,
    const float* __restrict__ old, const float* __restrict__ adv, const uint8_t* __restrict__ mask,
    const uint32_t* __restrict__ mask_words, int mask_mis, int skip_masked,
    float inv_temp, float clip, float inv_scale, int length_norm, const int32_t* __restrict__ seg_id,
    const int64_t* __restrict__ seg_start, unsigned long long* slots, uint8_t* __restrict__ pending,
    int max_wait_len, int spin_limit, float* __restrict__ lp_out, float* __restrict__ ent_out,
    float* __restrict__ lse_out, char* __restrict__ dlogits, int64_t out_stride_bytes) {
  __shared__ float red_m[kRowWaves], red_s[kRowWaves], red_w[kRowWaves];
  __shared__ float sample_sums[4];  // ok, sum of log-ratios, sum of advantages, count (all over unmasked tokens)
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
  const int64_t row = blockIdx.x;  // launch order IS row order here: a sample's rows start together
  if (row >= T) return;
  const int64_t t = row + 1;
  const int nchunk = (int)(V >> 3);
  const float c2 = inv_temp * kLog2e;
  const uint32_t lane_off = (uint32_t)tid * 16u;
  constexpr int kSlab = kRowThreads * 16;
  constexpr int kRsrcFlags = 0x00020000;
  const int row_bytes = (int)(V * 2);
  const auto out_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(dlogits + row * out_stride_bytes, 0, row_bytes, kRsrcFlags);
  auto zero_row = [&]() {
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < NCH; ++j) store_chunk(z, out_rsrc, lane_off, j * kSlab);
  };

  // token 0, which no row predicts (it is scored against all-zero logits), belongs to the first sample: the first
  // workgroup emits and publishes it, so that nobody waits for the last
  if (row == 0 && tid == 0) {
    const bool m0 = uniform_byte(mask_words, mask_mis, 0) != 0;
    float lr0 = 0.0f;
    if (skip_masked && !m0) {
      emit_skipped(0, lp_out, ent_out, lse_out);
    } else {
      emit_zero_row(V, 0, lp_out, ent_out, lse_out);
      const float lp0 = 0.0f - logf((float)V);
      lr0 = m0 ? lp0 - old[0] : 0.0f;
    }
    publish_log_ratio(slots, 0, lr0);
  }
  if (t >= T) {  // the last row predicts nothing
    zero_row();
    if (tid == 0) pending[row] = 0;
    return;
  }
  const bool m = uniform_byte(mask_words, mask_mis, t) != 0;  // (a scalar read: nothing the row's loads queue behind)
  if (skip_masked && !m) {  // a token nobody looks at: no read, outputs 0, log-ratio 0, gradient 0
    if (tid == 0) {
      emit_skipped(t, lp_out, ent_out, lse_out);
      publish_log_ratio(slots, t, 0.0f);
      pending[row] = 0;
    }
    zero_row();
    return;
  }

  const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(logits + row * in_stride_bytes), 0,
                                                         row_bytes, kRsrcFlags);
  u32x4 buf[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    buf[j] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, lane_off, j * kSlab, PRL_ROWS_NT_LOAD ? 2 : 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  const int64_t label = ids[t];
  float x_label = __builtin_nanf("");
  if (label >= 0 && label < V)
    x_label = uniform_bf16(reinterpret_cast<const uint32_t*>(logits) + row * (in_stride_bytes >> 2), label);
  const float old_t = old[t];
  // the token's sample (the table was made by the launch before this one)
  const int64_t sample = seg_id[t];
  const int64_t first = seg_start[sample], last = seg_start[sample + 1];

  auto chunk = [&](int j, float& prev) -> u32x4 {
    touch_after(buf[j], prev);
    u32x4 v = buf[j];
    if (j >= NCH - kTailSlabs) {
      const bool in = tid < nchunk - j * kRowThreads;
      v.x = in ? v.x : 0xff7fff7fu;
      v.y = in ? v.y : 0xff7fff7fu;
      v.z = in ? v.z : 0xff7fff7fu;
      v.w = in ? v.w : 0xff7fff7fu;
    }
    return v;
  };
  auto sums = [&](float M, float& M_out, float& s_out, float& w_out) {
    float es[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ws[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      float x[8];
      decode16_raw<PRL_BF16>(chunk(j, ws[3]), x);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float e = fast_exp2(fmaf(x[i], c2, -M));
        es[i & 3] += e;
        ws[i & 3] = fmaf(x[i], e, ws[i & 3]);
      }
    }
    const float s = wave_sum((es[0] + es[2]) + (es[1] + es[3]));
    const float w = wave_sum((ws[0] + ws[2]) + (ws[1] + ws[3]));
    if (lane == 0) {
      red_m[wid] = M;
      red_s[wid] = s;
      red_w[wid] = w;
    }
    __syncthreads();
    float Mx = red_m[0];
#pragma unroll
    for (int i = 1; i < kRowWaves; ++i) Mx = fmaxf(Mx, red_m[i]);
    float st = 0.0f, wt = 0.0f;
#pragma unroll
    for (int i = 0; i < kRowWaves; ++i) {
      const float sc = fast_exp2(red_m[i] - Mx);
      st = fmaf(red_s[i], sc, st);
      wt = fmaf(red_w[i], sc, wt);
    }
    M_out = Mx;
    s_out = st;
    w_out = wt;
  };
  float mx = kNegBig;
  {
    float x[8];
    decode16_raw<PRL_BF16>(chunk(0, mx), x);
#pragma unroll
    for (int i = 0; i < 8; ++i) mx = fmaxf(mx, x[i]);
  }
  float M, s, w;
  sums(wave_max(mx) * c2, M, s, w);
  if (!(s <= 3.0e38f) || !(s > 0.0f) || fabsf(w) > 3.0e38f) {  // see k_policy_rows: exact two-pass soft-max
    __syncthreads();
    mx = kNegBig;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      float x[8];
      decode16_raw<PRL_BF16>(chunk(j, mx), x);
#pragma unroll
      for (int i = 0; i < 8; ++i) mx = fmaxf(mx, x[i]);
    }
    mx = wave_max(mx);
    if (lane == 0) red_m[wid] = mx;
    __syncthreads();
    mx = red_m[0];
#pragma unroll
    for (int i = 1; i < kRowWaves; ++i) mx = fmaxf(mx, red_m[i]);
    __syncthreads();
    sums(mx * c2, M, s, w);
  }
  const float lse2 = M + fast_log2(s);
  const float lse = lse2 * kLn2;
  const float lp = x_label * inv_temp - lse;
  if (tid == 0) {
    if (lse_out) lse_out[t] = lse;
    if (lp_out) lp_out[t] = lp;
    if (ent_out) ent_out[t] = lse - (w / s) * inv_temp;
    publish_log_ratio(slots, t, m ? lp - old_t : 0.0f);  // before any waiting
  }
  if (!m) {  // scored for the statistics, but masked out of the loss: no gradient whatever the sample's ratio is
    zero_row();
    if (tid == 0) pending[row] = 0;
    return;
  }
  if (last - first > (int64_t)max_wait_len) {  // too long to be resident together: the caller's second pass does it
    if (tid == 0) pending[row] = 1;
    return;
  }

  // the sample's sums, once every one of its tokens has been published: one wave sweeps the words, lane i over the
  // tokens first + i, first + i + 64, ... (the order in which the loss kernel adds them up, so that the gradient
  // factor used here is the one it reports)
  if (wid == 0) {
    bool ok = false;
    float s_lr = 0.0f;
    for (int spin = 0; spin <= spin_limit; ++spin) {
      bool all = true;
      s_lr = 0.0f;
      for (int64_t i = first + lane; i < last; i += kWave) {
        const unsigned long long x = __hip_atomic_load(slots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        all = all && (x >> 32) == 1ull;
        s_lr += __uint_as_float((uint32_t)x);
      }
      if (__all(all ? 1 : 0)) {
        ok = true;
        break;
      }
      if (spin >= 8) __builtin_amdgcn_s_sleep(16);  // the sample's rows end within microseconds of each other
    }
    float s_adv = 0.0f, s_cnt = 0.0f;
    for (int64_t i = first + lane; i < last; i += kWave) {
      if (mask[i] != 0) {
        s_adv += adv[i];
        s_cnt += 1.0f;
      }
    }
    s_lr = wave_sum(s_lr);
    s_adv = wave_sum(s_adv);
    s_cnt = wave_sum(s_cnt);
    if (lane == 0) {
      sample_sums[0] = ok ? 1.0f : 0.0f;
      sample_sums[1] = s_lr;
      sample_sums[2] = s_adv;
      sample_sums[3] = s_cnt;
    }
  }
  __syncthreads();
  if (sample_sums[0] == 0.0f) {  // gave up waiting: the caller's second pass writes this row
    if (tid == 0) pending[row] = 1;
    return;
  }
  // dloss/dlogprob of an unmasked token of this sample (the expressions of the loss kernels, in their order)
  const float cnt = fmaxf(sample_sums[3], 1.0f);
  const float av = length_norm ? sample_sums[1] / cnt : sample_sums[1];
  const float r = expf(av > 10.0f ? 10.0f : av);
  const float g_c = -((inv_scale / cnt) * sample_sums[2]);
  const float g_r = (r <= clip) ? g_c : 0.0f;
  const float g_b = g_r * r;
  const float g_a = (av <= 10.0f) ? g_b : 0.0f;
  const float g = length_norm ? g_a / cnt : g_a;
  const float gi = g * inv_temp;
  if (tid == 0) pending[row] = 0;
  if (gi == 0.0f) {
    zero_row();
    return;
  }
  const float shift2 = fast_log2(fabsf(gi)) - lse2;
  const uint32_t sign2 = gi > 0.0f ? 0x80008000u : 0u;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    touch(buf[j]);
    float x[8], d[8];
    decode16_raw<PRL_BF16>(buf[j], x);
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = fast_exp2(fmaf(x[i], c2, shift2));
    u32x4 v;
    v.x = pack_bf16(d[0], d[1]) | sign2;
    v.y = pack_bf16(d[2], d[3]) | sign2;
    v.z = pack_bf16(d[4], d[5]) | sign2;
    v.w = pack_bf16(d[6], d[7]) | sign2;
    store_chunk(v, out_rsrc, lane_off, j * kSlab);
  }
  if (label >= 0 && label < V && tid == (int)((label >> 3) & (kRowThreads - 1))) {
    const float p = fast_exp2(fmaf(x_label, c2, -lse2));
    reinterpret_cast<uint16_t*>(dlogits + row * out_stride_bytes)[label] = one_bf16(fmaf(-gi, p, gi));
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

// Compute units of the current device (asked once per device; 256 on MI355X).
static int compute_units() {
  static int cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) return -1;
    cached[dev] = n;
  }
  return cached[dev];
}

}  // namespace prl

namespace prl {
int segments_impl(const int64_t* pos, int64_t T, int32_t* seg_id, int64_t* seg_start, int32_t* n_seg,
                  unsigned long long* clear, int64_t n_clear, void* stream);
}

using namespace prl;

// Events for the next launch of the row kernel from this thread (prl_policy_rows_time_next): the launch then goes
// through hipExtLaunchKernelGGL, which stamps the two events with the kernel's own start and end on the GPU.
static thread_local hipEvent_t g_time_start = nullptr, g_time_stop = nullptr;

extern "C" int prl_policy_rows_time_next(void* start_event, void* stop_event) {
  clear_error();
  PRL_REQUIRE((start_event == nullptr) == (stop_event == nullptr),
              "policy_rows_time_next: pass two events, or two NULLs to take the request back");
  g_time_start = (hipEvent_t)start_event;
  g_time_stop = (hipEvent_t)stop_event;
  return PRL_OK;
}

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
#if PRL_ROWS_XCD_BLOCKS
  const dim3 grid((unsigned)((T + 7) / 8 * 8)), block(kRowThreads);  // workgroups past the last row return at once
#else
  const dim3 grid((unsigned)T), block(kRowThreads);
#endif
  const int mask_mis = (int)(reinterpret_cast<uintptr_t>(mask) & 3);
  const uint32_t* mask_words = reinterpret_cast<const uint32_t*>(mask - mask_mis);
  hipStream_t st = (hipStream_t)stream;
#if PRL_ROWS_WALK
  int cus = compute_units();
  if (cus <= 0) {
    set_error("policy_rows_fwd_bwd: cannot read the device's number of compute units");
    return PRL_ELAUNCH;
  }
  int64_t nwg = (T + 7) / 8 * 8;
  if (nwg > cus / 8 * 8) nwg = cus / 8 * 8;  // one workgroup per compute unit, the same number on every XCD
  const dim3 wgrid((unsigned)nwg);
#if PRL_ROWS_INTERLEAVE
#define PRL_WALK_KERNEL k_policy_rows_mix
#else
#define PRL_WALK_KERNEL k_policy_rows_walk
#endif
#define PRL_ROWS(N)                                                                                               \
  hipLaunchKernelGGL((PRL_WALK_KERNEL<N>), wgrid, block, 0, st, (const char*)logits, T, V, row_stride * 2, ids, \
                     old, adv, mask_words, mask_mis, skip_masked ? 1 : 0, inv_temp, clip_ratio, 1.0f / scale,     \
                     logprob, entropy, lse, (char*)dlogits, out_row_stride * 2)
#else
#define PRL_ROWS_ARGS                                                                                              \
  (const char*)logits, T, V, (int64_t)(row_stride * 2), ids, old, adv, mask_words, mask_mis, (int)(skip_masked ? 1 : 0), \
      inv_temp, clip_ratio, (float)(1.0f / scale), logprob, entropy, lse, (char*)dlogits, (int64_t)(out_row_stride * 2), \
      (const int64_t*)nullptr, T
#define PRL_ROWS(N)                                                                                                \
  do {                                                                                                             \
    if (g_time_start && g_time_stop)                                                                               \
      hipExtLaunchKernelGGL((k_policy_rows<N>), grid, block, 0, st, g_time_start, g_time_stop, 0, PRL_ROWS_ARGS);  \
    else                                                                                                           \
      hipLaunchKernelGGL((k_policy_rows<N>), grid, block, 0, st, PRL_ROWS_ARGS);                                   \
  } while (0)
#endif
  // sizes at most kTailSlabs = 4 apart, so that the row always ends in the last 4 slabs of its kernel
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
#ifdef PRL_ROWS_ARGS
#undef PRL_ROWS_ARGS
#endif
  g_time_start = g_time_stop = nullptr;  // one launch per request
  PRL_CHECK_LAUNCH("policy_rows_fwd_bwd");
  return PRL_OK;
}

namespace prl {
// logprob, entropy and lse of a packed row before gathered rows are scored: 0 everywhere (a token no row predicts
// is not scored), except token 0, which is scored against an all-zero row when its mask is set
// (src/prime_rl/trainer/rl/loss.py:51-58).  A kernel, not a memset, so that a caller may capture the call.
__global__ __launch_bounds__(256) void k_unscored_tokens(float* __restrict__ lp, float* __restrict__ ent,
                                                         float* __restrict__ lse, const uint8_t* __restrict__ mask,
                                                         int64_t T, int64_t V) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  if (t == 0 && mask[0] != 0)
    emit_zero_row(V, 0, lp, ent, lse);
  else
    emit_skipped(t, lp, ent, lse);
}
}  // namespace prl

extern "C" int prl_policy_rows_gathered_fwd_bwd(const void* logits, int logits_dtype, int64_t R, int64_t V,
                                                int64_t row_stride, const int64_t* positions, int64_t T,
                                                const int64_t* ids, const float* adv, const float* old,
                                                const uint8_t* mask, float temperature, float clip_ratio,
                                                float loss_scale, float* logprob, float* entropy, float* lse,
                                                void* dlogits, int out_dtype, int64_t out_row_stride, void* stream) {
  clear_error();
#if PRL_ROWS_WALK
  set_error("policy_rows_gathered_fwd_bwd: not built into a library with walking workgroups (PRL_ROWS_WALK)");
  return PRL_EINVAL;
#else
  PRL_REQUIRE(T >= 1 && T < ((int64_t)1 << 31) && R >= 0 && R < ((int64_t)1 << 31),
              "policy_rows_gathered_fwd_bwd: need 1 <= T < 2^31 tokens and 0 <= R < 2^31 rows");
  PRL_REQUIRE(ids && adv && old && mask && logprob, "policy_rows_gathered_fwd_bwd: ids, adv, old, mask and logprob must not be NULL");
  PRL_REQUIRE(temperature > 0.0f && temperature - temperature == 0.0f,
              "policy_rows_gathered_fwd_bwd: temperature must be finite and positive");
  PRL_REQUIRE(V >= 8, "policy_rows_gathered_fwd_bwd: V must be at least 8");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_unscored_tokens, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, logprob, entropy, lse,
                     mask, T, V);
  PRL_CHECK_LAUNCH("policy_rows_gathered_fwd_bwd (unscored tokens)");
  if (R == 0) return PRL_OK;
  if (!dlogits) {  // rows are only scored: the gradient's type and stride play no part
    out_dtype = PRL_BF16;
    out_row_stride = row_stride;
  }
  PRL_REQUIRE(prl_policy_rows_supported(logits_dtype, R, V, row_stride, out_dtype, out_row_stride),
              "policy_rows_gathered_fwd_bwd: shape or dtype not supported (needs bf16 in and out, V a multiple of 8 "
              "and at most %d, row strides multiples of 8)",
              kRowMaxChunks * kRowThreads * 8);
  PRL_REQUIRE(logits && positions, "policy_rows_gathered_fwd_bwd: logits and positions must not be NULL");
  PRL_REQUIRE(((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(dlogits)) & 15) == 0,
              "policy_rows_gathered_fwd_bwd: logits and dlogits must be 16-byte aligned");
  const float scale = loss_scale > 1.0f ? loss_scale : 1.0f;  // max(loss_scale, 1)
  const float inv_temp = 1.0f / temperature;
  const int need = chunks_needed(V);
#if PRL_ROWS_XCD_BLOCKS
  const dim3 grid((unsigned)((R + 7) / 8 * 8)), block(kRowThreads);
#else
  const dim3 grid((unsigned)R), block(kRowThreads);
#endif
  const int mask_mis = (int)(reinterpret_cast<uintptr_t>(mask) & 3);
  const uint32_t* mask_words = reinterpret_cast<const uint32_t*>(mask - mask_mis);
  // skip_masked = 1: a listed row whose token is masked out is not read; it gets a zero gradient row
#define PRL_ROWS_GATHERED_ARGS                                                                                  \
  (const char*)logits, T, V, (int64_t)(row_stride * 2), ids, old, adv, mask_words, mask_mis, 1, inv_temp,      \
      clip_ratio, (float)(1.0f / scale), logprob, entropy, lse, (char*)dlogits, (int64_t)(out_row_stride * 2), \
      positions, R
#define PRL_ROWS(N)                                                                                           \
  do {                                                                                                        \
    if (dlogits)                                                                                              \
      hipLaunchKernelGGL((k_policy_rows<N, true>), grid, block, 0, st, PRL_ROWS_GATHERED_ARGS);               \
    else                                                                                                      \
      hipLaunchKernelGGL((k_policy_rows<N, false>), grid, block, 0, st, PRL_ROWS_GATHERED_ARGS);              \
  } while (0)
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
#undef PRL_ROWS_GATHERED_ARGS
  PRL_CHECK_LAUNCH("policy_rows_gathered_fwd_bwd");
  return PRL_OK;
#endif
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

// Workspace of prl_policy_rows_seq_fwd_bwd: the published words (8 bytes per token), the sample of every token
// (4 bytes), the samples' start offsets (8 bytes per token + 8) and their number.
static size_t seq_ws_slots(int64_t T) { return (size_t)T * 8; }
static size_t seq_ws_seg_start(int64_t T) { return seq_ws_slots(T) + (((size_t)T * 4 + 15) & ~(size_t)15); }
static size_t seq_ws_n_seg(int64_t T) { return seq_ws_seg_start(T) + ((size_t)T + 1) * 8; }

extern "C" size_t prl_policy_rows_seq_workspace_bytes(int64_t T) { return T > 0 ? seq_ws_n_seg(T) + 16 : 0; }

extern "C" int prl_policy_rows_seq_fwd_bwd(const void* logits, int logits_dtype, int64_t T, int64_t V,
                                           int64_t row_stride, const int64_t* ids, const int64_t* pos,
                                           const float* adv, const float* old, const uint8_t* mask, int skip_masked,
                                           float temperature, int length_norm, float clip_ratio, float loss_scale,
                                           float* logprob, float* entropy, float* lse, void* dlogits, int out_dtype,
                                           int64_t out_row_stride, uint8_t* pending, int max_wait_len, int spin_limit,
                                           void* workspace, size_t workspace_bytes, void* stream) {
  clear_error();
  PRL_REQUIRE(prl_policy_rows_supported(logits_dtype, T, V, row_stride, out_dtype, out_row_stride),
              "policy_rows_seq_fwd_bwd: shape or dtype not supported; ask prl_policy_rows_supported first");
  PRL_REQUIRE(logits && ids && pos && adv && old && mask && logprob && lse && dlogits && pending && workspace,
              "policy_rows_seq_fwd_bwd: logits, ids, pos, adv, old, mask, logprob, lse (the second pass reads it), "
              "dlogits, pending and workspace must not be NULL");
  PRL_REQUIRE(((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(dlogits)) & 15) == 0,
              "policy_rows_seq_fwd_bwd: logits and dlogits must be 16-byte aligned");
  PRL_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0 &&
                  workspace_bytes >= prl_policy_rows_seq_workspace_bytes(T),
              "policy_rows_seq_fwd_bwd: workspace too small or not 16-byte aligned (need %zu bytes)",
              prl_policy_rows_seq_workspace_bytes(T));
  PRL_REQUIRE(temperature > 0.0f && temperature - temperature == 0.0f,
              "policy_rows_seq_fwd_bwd: temperature must be finite and positive");
  PRL_REQUIRE(max_wait_len >= 0 && spin_limit >= 0 && spin_limit <= (1 << 20),
              "policy_rows_seq_fwd_bwd: max_wait_len must be >= 0 and spin_limit in [0, 2^20]");
  char* ws = (char*)workspace;
  unsigned long long* slots = (unsigned long long*)ws;
  int32_t* seg_id = (int32_t*)(ws + seq_ws_slots(T));
  int64_t* seg_start = (int64_t*)(ws + seq_ws_seg_start(T));
  int32_t* n_seg = (int32_t*)(ws + seq_ws_n_seg(T));
  // the samples' table, and in the same launch the hand-off words back to "not published"
  int rc = segments_impl(pos, T, seg_id, seg_start, n_seg, slots, T, stream);
  if (rc != PRL_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  const float scale = loss_scale > 1.0f ? loss_scale : 1.0f;
  const float inv_temp = 1.0f / temperature;
  const int need = chunks_needed(V);
  const dim3 grid((unsigned)T), block(kRowThreads);
  const int mask_mis = (int)(reinterpret_cast<uintptr_t>(mask) & 3);
  const uint32_t* mask_words = reinterpret_cast<const uint32_t*>(mask - mask_mis);
#define PRL_ROWS_SEQ(N)                                                                                               \
  hipLaunchKernelGGL((k_policy_rows_seq<N>), grid, block, 0, st, (const char*)logits, T, V, row_stride * 2, ids, old, \
                     adv, mask, mask_words, mask_mis, skip_masked ? 1 : 0, inv_temp, clip_ratio, 1.0f / scale,         \
                     length_norm ? 1 : 0, seg_id,                                                                      \
                     seg_start, slots, pending, max_wait_len, spin_limit, logprob, entropy, lse, (char*)dlogits,      \
                     out_row_stride * 2)
  if (need <= 2)
    PRL_ROWS_SEQ(2);
  else if (need <= 4)
    PRL_ROWS_SEQ(4);
  else if (need <= 8)
    PRL_ROWS_SEQ(8);
  else if (need <= 12)
    PRL_ROWS_SEQ(12);
  else if (need <= 16)
    PRL_ROWS_SEQ(16);
  else
    PRL_ROWS_SEQ(19);
#undef PRL_ROWS_SEQ
  PRL_CHECK_LAUNCH("policy_rows_seq_fwd_bwd");
  return PRL_OK;
}
