// RMSNorm forward and backward over rows of a [rows, H] activation, one launch each
// (plus a small column-sum finish for the weight gradient).
//
// The transformer blocks the trainer runs (src/prime_rl/trainer/model.py:71-88 loads
// them from transformers) spell the normalisation as seven element-wise and reduction
// kernels forward and about fifteen backward, 113 times per micro-batch for a 28-layer
// Qwen3 (input, post-attention, query and key norms); at T = 2048 every one of them is
// a 4-7 us launch.  The reference's own switch for this is `model.impl = "liger_kernel"`
// (src/prime_rl/trainer/model.py:74-75, Triton kernels); this is that switch's counterpart
// here.  Arithmetic follows the module being replaced:
//     y = w * cast(x32 * rsqrt(mean(x32^2) + eps))        x32 = float(x), cast to x's dtype
// HBM-bound: x read once, y written once forward; x, gy read and gx written backward.
#include "logprob_dev.h"

namespace prl {

constexpr int kNormThreads = 256;
#ifndef PRL_NORM_MAX_BLOCKS
#define PRL_NORM_MAX_BLOCKS 256  // backward: also the number of partial rows the weight-gradient finish adds up
#endif
constexpr int kNormMaxBlocks = PRL_NORM_MAX_BLOCKS;
#ifndef PRL_NORM_BWD_ROWS
#define PRL_NORM_BWD_ROWS 1  // rows a row slot of the backward keeps in flight (tools/tune_norm_rows.py: 2 or 4 are no faster at H = 1024 and a quarter slower at 2048)
#endif
constexpr int kNormFwdMaxBlocks = 1024;  // forward: no partial sums, more workgroups only help (tools/tune_norm.py)

template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LPR);
  return v;
}

__device__ __forceinline__ float round_to(float v, int dt) { return dt == PRL_BF16 ? bf16_one(one_bf16(v)) : v; }

template <int DT>
__device__ __forceinline__ void load_vec(const char* p, int64_t v, float* out) {
  u32x4 raw = reinterpret_cast<const u32x4*>(p)[v];
  decode16_raw<DT>(raw, out);
}

// Row `row` of x.  Plain rows have one pitch; with x_group > 0 the rows come in groups of x_group (pitch
// x_stride inside a group, x_group_stride from group to group): the heads of a token inside a wider row, as
// the query / key blocks of one stacked q/k/v projection are.
__device__ __forceinline__ const char* x_row(const char* x, int64_t row, int64_t x_stride, int64_t x_group,
                                             int64_t x_group_stride) {
  if (x_group <= 0) return x + row * x_stride;
  const int64_t grp = row / x_group;
  return x + grp * x_group_stride + (row - grp * x_group) * x_stride;
}

// LPR lanes share a row (16 for short rows such as a 128-wide head, 64 otherwise); a lane owns the
// 16-byte vectors li, li + LPR, ... of its row, NV of them at most.
template <int DT, int LPR, int NV>
__global__ __launch_bounds__(kNormThreads) void k_rmsnorm_fwd(const char* __restrict__ x, int64_t x_stride,
                                                              const char* __restrict__ w, char* __restrict__ y,
                                                              int64_t y_stride, float* __restrict__ rstd,
                                                              int64_t rows, int H, float eps, int64_t x_group,
                                                              int64_t x_group_stride, const char* __restrict__ res,
                                                              int64_t res_stride, char* __restrict__ sum,
                                                              int64_t sum_stride) {
  constexpr int VE = DT == PRL_F32 ? 4 : 8;
  constexpr int RPB = kNormThreads / LPR;  // rows per workgroup and step
  const int li = threadIdx.x % LPR, slot = threadIdx.x / LPR;
  const int nvec = H / VE;
  float wv[NV][VE];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int v = li + k * LPR;
    if (v < nvec) load_vec<DT>(w, v, wv[k]);
  }
  for (int64_t row = (int64_t)blockIdx.x * RPB + slot; row < rows; row += (int64_t)gridDim.x * RPB) {
    const char* xp = x_row(x, row, x_stride, x_group, x_group_stride);
    float xv[NV][VE];
    float ss = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int v = li + k * LPR;
      if (v < nvec) {
        load_vec<DT>(xp, v, xv[k]);
        if (res) {  // the row normalised is x + res, rounded to the storage type as a separate addition would
          float rv[VE];
          load_vec<DT>(res + row * res_stride, v, rv);
#pragma unroll
          for (int i = 0; i < VE; ++i) xv[k][i] = round_to(xv[k][i] + rv[i], DT);
          store_unit<DT, VE>(sum + row * sum_stride, v, xv[k]);
        }
#pragma unroll
        for (int i = 0; i < VE; ++i) ss = fmaf(xv[k][i], xv[k][i], ss);
      }
    }
    ss = row_sum<LPR>(ss);
    const float rs = rsqrtf(ss / (float)H + eps);
    if (li == 0 && rstd) rstd[row] = rs;
    char* yp = y + row * y_stride;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int v = li + k * LPR;
      if (v < nvec) {
        float o[VE];
#pragma unroll
        for (int i = 0; i < VE; ++i) o[i] = wv[k][i] * round_to(xv[k][i] * rs, DT);
        store_unit<DT, VE>(yp, v, o);
      }
    }
  }
}

// gx = rstd * (gy w - xh * mean(gy w xh)) [+ gadd],  xh = x * rstd;  per-workgroup column sums of gy * cast(xh)
// go to gw_part[blockIdx.x, :].  A row slot keeps RU rows in flight (all their loads issued before the first
// is used); measured, one row is as fast as two or four: without the column-sum finish the kernel moves
// 100 MB in ~20 us already.  `gadd`, when given, is a gradient that reaches
// x along another path (the residual connection around the normalised branch): the result is what a separate
// addition of the rounded gx would give.
template <int DT, int LPR, int NV, int RU>
__global__ __launch_bounds__(kNormThreads) void k_rmsnorm_bwd(const char* __restrict__ x, int64_t x_stride,
                                                              const char* __restrict__ w,
                                                              const float* __restrict__ rstd,
                                                              const char* __restrict__ gy, int64_t gy_stride,
                                                              const char* __restrict__ gadd, int64_t gadd_stride,
                                                              char* __restrict__ gx, int64_t gx_stride,
                                                              float* __restrict__ gw_part, int64_t rows, int H,
                                                              int64_t x_group, int64_t x_group_stride) {
  constexpr int VE = DT == PRL_F32 ? 4 : 8;
  constexpr int RPB = kNormThreads / LPR;
  extern __shared__ float lds[];  // RPB * H floats: the row slots' column sums
  const int li = threadIdx.x % LPR, slot = threadIdx.x / LPR;
  const int nvec = H / VE;
  const float inv_h = 1.0f / (float)H;
  float wv[NV][VE], acc[NV][VE];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int v = li + k * LPR;
    if (v < nvec) load_vec<DT>(w, v, wv[k]);
#pragma unroll
    for (int i = 0; i < VE; ++i) acc[k][i] = 0.0f;
  }
  const int64_t step = (int64_t)gridDim.x * RPB;
  for (int64_t row0 = (int64_t)blockIdx.x * RPB + slot; row0 < rows; row0 += step * RU) {
    u32x4 rx[RU][NV], rg[RU][NV], ra[RU][NV];
    float rs[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int64_t row = row0 + u * step < rows ? row0 + u * step : row0;  // past the end: row0 again, unused
      const char* xp = x_row(x, row, x_stride, x_group, x_group_stride);
      const char* gp = gy + row * gy_stride;
      rs[u] = rstd[row];
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int v = li + k * LPR;
        if (v < nvec) {
          rx[u][k] = reinterpret_cast<const u32x4*>(xp)[v];
          rg[u][k] = reinterpret_cast<const u32x4*>(gp)[v];
          if (gadd) ra[u][k] = reinterpret_cast<const u32x4*>(gadd + row * gadd_stride)[v];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int64_t row = row0 + u * step;
      if (row < rows) {
        float xh[NV][VE], gw_[NV][VE];
        float dot = 0.0f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
          const int v = li + k * LPR;
          if (v < nvec) {
            float g[VE];
            decode16_raw<DT>(rx[u][k], xh[k]);
            decode16_raw<DT>(rg[u][k], g);
#pragma unroll
            for (int i = 0; i < VE; ++i) {
              xh[k][i] *= rs[u];
              acc[k][i] = fmaf(g[i], round_to(xh[k][i], DT), acc[k][i]);
              gw_[k][i] = g[i] * wv[k][i];
              dot = fmaf(gw_[k][i], xh[k][i], dot);
            }
          }
        }
        dot = row_sum<LPR>(dot) * inv_h;
        char* op = gx + row * gx_stride;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
          const int v = li + k * LPR;
          if (v < nvec) {
            float o[VE];
#pragma unroll
            for (int i = 0; i < VE; ++i) o[i] = rs[u] * (gw_[k][i] - xh[k][i] * dot);
            if (gadd) {
              float av[VE];
              decode16_raw<DT>(ra[u][k], av);
#pragma unroll
              for (int i = 0; i < VE; ++i) {
                float r = round_to(o[i], DT);
                if (DT == PRL_F32) asm volatile("" : "+v"(r));  // a value of its own: the addition is not fused into the product
                o[i] = r + av[i];
              }
            }
            store_unit<DT, VE>(op, v, o);
          }
        }
      }
    }
  }
  // column sums: row slots -> LDS -> one row per workgroup
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int v = li + k * LPR;
    if (v < nvec) {
#pragma unroll
      for (int i = 0; i < VE; ++i) lds[slot * H + v * VE + i] = acc[k][i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += kNormThreads) {
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < RPB; ++r) s += lds[r * H + c];
    gw_part[(int64_t)blockIdx.x * H + c] = s;
  }
}

// out[c] = sum over the partial rows of part[p, c], in a fixed order.  A workgroup owns 32 columns; its
// 32 groups of 32 threads each take every 32nd partial row (loads of 128 contiguous bytes), then the
// groups' sums are added through LDS.
template <int DT>
__global__ __launch_bounds__(1024) void k_colsum_finish(const float* __restrict__ part, int nparts, int H,
                                                        char* __restrict__ out) {
  __shared__ float lds[32][33];
  const int lc = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + lc;
  float s = 0.0f;
  if (c < H)
    for (int p = g; p < nparts; p += 32) s += part[(int64_t)p * H + c];
  lds[g][lc] = s;
  __syncthreads();
  if (g == 0 && c < H) {
    float t = lds[0][lc];
#pragma unroll
    for (int k = 1; k < 32; ++k) t += lds[k][lc];
    store_elem<DT>(out, c, t);
  }
}

static int norm_blocks(int64_t rows, int lpr, int cap = kNormMaxBlocks) {
  const int64_t rpb = kNormThreads / lpr;
  const int64_t want = (rows + rpb - 1) / rpb;
  return (int)(want < cap ? (want > 0 ? want : 1) : cap);
}
// Backward: every workgroup leaves one partial row for the weight-gradient finish, so few workgroups are
// better for short inputs (tools/tune_norm.py, 2048-token rows); long inputs (16 K-token joined rows) need
// more of them in flight to reach the memory rate: at least 8 row steps per workgroup, between the tuned cap
// and the forward's.
static int norm_bwd_blocks(int64_t rows, int lpr) {
  const int64_t rpb = kNormThreads / lpr;
  const int64_t want = (rows + rpb - 1) / rpb;
  int64_t cap = want / 8;
  if (cap < kNormMaxBlocks) cap = kNormMaxBlocks;
  if (cap > kNormFwdMaxBlocks) cap = kNormFwdMaxBlocks;
  return norm_blocks(rows, lpr, (int)cap);
}
static int norm_lpr(int64_t H, int ve) { return H / ve <= 16 ? 16 : 64; }
static int norm_nv(int64_t H, int ve, int lpr) { return (int)((H / ve + lpr - 1) / lpr); }

}  // namespace prl

using namespace prl;

extern "C" int prl_rmsnorm_supported(int dtype, int64_t H) {
  if (dtype != PRL_F32 && dtype != PRL_BF16) return 0;
  const int ve = dtype == PRL_F32 ? 4 : 8;
  if (H < ve || H % ve) return 0;
  const int lpr = norm_lpr(H, ve);
  return norm_nv(H, ve, lpr) <= 16 && H * (int64_t)(kNormThreads / lpr) * 4 <= 64 * 1024 ? 1 : 0;
}

extern "C" size_t prl_rmsnorm_bwd_workspace_bytes(int dtype, int64_t rows, int64_t H) {
  const int ve = dtype == PRL_F32 ? 4 : 8;
  return (size_t)norm_bwd_blocks(rows, norm_lpr(H, ve)) * (size_t)H * sizeof(float);
}

#define PRL_NORM_DISPATCH(KERNEL, ...)                                    \
  do {                                                                    \
    if (dtype == PRL_BF16) {                                              \
      if (lpr == 16) KERNEL(PRL_BF16, 16, 1, __VA_ARGS__);                \
      else if (nv <= 1) KERNEL(PRL_BF16, 64, 1, __VA_ARGS__);             \
      else if (nv <= 2) KERNEL(PRL_BF16, 64, 2, __VA_ARGS__);             \
      else if (nv <= 4) KERNEL(PRL_BF16, 64, 4, __VA_ARGS__);             \
      else if (nv <= 8) KERNEL(PRL_BF16, 64, 8, __VA_ARGS__);             \
      else KERNEL(PRL_BF16, 64, 16, __VA_ARGS__);                         \
    } else {                                                              \
      if (lpr == 16) KERNEL(PRL_F32, 16, 1, __VA_ARGS__);                 \
      else if (nv <= 1) KERNEL(PRL_F32, 64, 1, __VA_ARGS__);              \
      else if (nv <= 2) KERNEL(PRL_F32, 64, 2, __VA_ARGS__);              \
      else if (nv <= 4) KERNEL(PRL_F32, 64, 4, __VA_ARGS__);              \
      else if (nv <= 8) KERNEL(PRL_F32, 64, 8, __VA_ARGS__);              \
      else KERNEL(PRL_F32, 64, 16, __VA_ARGS__);                          \
    }                                                                     \
  } while (0)

static int norm_check(const char* who, int dtype, int64_t rows, int64_t H, const void* a, const void* b, const void* c,
                      int64_t s0, int64_t s1) {
  PRL_REQUIRE(rows >= 0, "%s: rows must be >= 0", who);
  PRL_REQUIRE(prl_rmsnorm_supported(dtype, H), "%s: unsupported dtype %d or row length %lld", who, dtype, (long long)H);
  PRL_REQUIRE(s0 >= H && s1 >= H, "%s: row strides must be >= H", who);
  const int es = dtype == PRL_F32 ? 4 : 2;
  PRL_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
                (uintptr_t)(s0 * es) | (uintptr_t)(s1 * es)) & 15) == 0,
              "%s: pointers and row strides must be 16-byte aligned", who);
  return PRL_OK;
}

static int norm_group_check(const char* who, int dtype, int64_t H, int64_t x_row_stride, int64_t x_group,
                            int64_t x_group_stride) {
  if (x_group <= 0) return PRL_OK;
  const int es = dtype == PRL_F32 ? 4 : 2;
  PRL_REQUIRE(x_group_stride >= (x_group - 1) * x_row_stride + H && ((x_group_stride * es) & 15) == 0,
              "%s: the group stride must cover a group of rows and be 16-byte aligned", who);
  return PRL_OK;
}

static int rmsnorm_fwd_impl(const char* who, const void* x, int64_t x_row_stride, const void* res,
                            int64_t res_row_stride, void* sum, int64_t sum_row_stride, const void* w, void* y,
                            int64_t y_row_stride, float* rstd, int dtype, int64_t rows, int64_t H, float eps,
                            int64_t x_group, int64_t x_group_stride, void* stream) {
  int rc = norm_check(who, dtype, rows, H, x, w, y, x_row_stride, y_row_stride);
  if (rc != PRL_OK) return rc;
  rc = norm_group_check(who, dtype, H, x_row_stride, x_group, x_group_stride);
  if (rc != PRL_OK) return rc;
  if (res) {
    rc = norm_check(who, dtype, rows, H, res, sum, sum, res_row_stride, sum_row_stride);
    if (rc != PRL_OK) return rc;
  }
  if (rows == 0) return PRL_OK;
  PRL_REQUIRE(x && w && y, "%s: x, w and y must not be NULL", who);
  const int es = dtype == PRL_F32 ? 4 : 2, ve = 16 / es;
  const int lpr = norm_lpr(H, ve), nv = norm_nv(H, ve, lpr);
  const dim3 grid((unsigned)norm_blocks(rows, lpr, kNormFwdMaxBlocks)), block(kNormThreads);
  hipStream_t st = (hipStream_t)stream;
#define PRL_FWD(DT, L, N, ...)                                                                                     \
  hipLaunchKernelGGL((k_rmsnorm_fwd<DT, L, N>), grid, block, 0, st, (const char*)x, x_row_stride * es, (const char*)w, \
                     (char*)y, y_row_stride * es, rstd, rows, (int)H, eps, x_group, x_group_stride * es,          \
                     (const char*)res, res_row_stride * es, (char*)sum, sum_row_stride * es)
  PRL_NORM_DISPATCH(PRL_FWD, 0);
#undef PRL_FWD
  PRL_CHECK_LAUNCH(who);
  return PRL_OK;
}

extern "C" int prl_rmsnorm_fwd(const void* x, int64_t x_row_stride, const void* w, void* y, int64_t y_row_stride,
                               float* rstd, int dtype, int64_t rows, int64_t H, float eps, int64_t x_group,
                               int64_t x_group_stride, void* stream) {
  clear_error();
  return rmsnorm_fwd_impl("rmsnorm_fwd", x, x_row_stride, nullptr, 0, nullptr, 0, w, y, y_row_stride, rstd, dtype, rows,
                          H, eps, x_group, x_group_stride, stream);
}

extern "C" int prl_add_rmsnorm_fwd(const void* x, int64_t x_row_stride, const void* res, int64_t res_row_stride,
                                   void* sum, int64_t sum_row_stride, const void* w, void* y, int64_t y_row_stride,
                                   float* rstd, int dtype, int64_t rows, int64_t H, float eps, void* stream) {
  clear_error();
  PRL_REQUIRE(rows == 0 || (res && sum), "add_rmsnorm_fwd: res and sum must not be NULL");
  return rmsnorm_fwd_impl("add_rmsnorm_fwd", x, x_row_stride, res, res_row_stride, sum, sum_row_stride, w, y,
                          y_row_stride, rstd, dtype, rows, H, eps, 0, 0, stream);
}

static int rmsnorm_bwd_impl(const char* who, const void* x, int64_t x_row_stride, const void* w, const float* rstd,
                            const void* gy, int64_t gy_row_stride, const void* gadd, int64_t gadd_row_stride, void* gx,
                            int64_t gx_row_stride, void* gw, int dtype, int64_t rows, int64_t H, void* workspace,
                            size_t workspace_bytes, int64_t x_group, int64_t x_group_stride, void* stream) {
  int rc = norm_check(who, dtype, rows, H, x, gy, gx, x_row_stride, gy_row_stride);
  if (rc != PRL_OK) return rc;
  rc = norm_group_check(who, dtype, H, x_row_stride, x_group, x_group_stride);
  if (rc != PRL_OK) return rc;
  if (gadd) {
    rc = norm_check(who, dtype, rows, H, gadd, gadd, gadd, gadd_row_stride, gadd_row_stride);
    if (rc != PRL_OK) return rc;
  }
  PRL_REQUIRE(gx_row_stride >= H && ((gx_row_stride * (dtype == PRL_F32 ? 4 : 2)) & 15) == 0,
              "%s: gx row stride must be >= H and 16-byte aligned", who);
  PRL_REQUIRE(x && w && rstd && gy && gx && gw, "%s: x, w, rstd, gy, gx and gw must not be NULL", who);
  PRL_REQUIRE((reinterpret_cast<uintptr_t>(w) & 15) == 0, "%s: w must be 16-byte aligned", who);
  const int es = dtype == PRL_F32 ? 4 : 2, ve = 16 / es;
  const int lpr = norm_lpr(H, ve), nv = norm_nv(H, ve, lpr);
  const int nblk = norm_bwd_blocks(rows, lpr);
  PRL_REQUIRE(workspace && workspace_bytes >= (size_t)nblk * H * sizeof(float),
              "%s: workspace of %zu bytes needed (prl_rmsnorm_bwd_workspace_bytes)", who,
              (size_t)nblk * H * sizeof(float));
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  const dim3 grid((unsigned)nblk), block(kNormThreads);
  const size_t lds = (size_t)(kNormThreads / lpr) * H * sizeof(float);
  // two rows in flight per row slot while their raw vectors fit the registers (rows of up to 4 vectors per lane)
#define PRL_BWD(DT, L, N, ...)                                                                                      \
  hipLaunchKernelGGL((k_rmsnorm_bwd<DT, L, N, (N <= 4 ? PRL_NORM_BWD_ROWS : 1)>), grid, block, lds, st, (const char*)x, \
                     x_row_stride * es, (const char*)w, rstd, (const char*)gy, gy_row_stride * es, (const char*)gadd, \
                     gadd_row_stride * es, (char*)gx, gx_row_stride * es, part, rows, (int)H, x_group,             \
                     x_group_stride * es)
  PRL_NORM_DISPATCH(PRL_BWD, 0);
#undef PRL_BWD
  PRL_CHECK_LAUNCH(who);
  const dim3 fgrid((unsigned)((H + 31) / 32));
  if (dtype == PRL_BF16)
    hipLaunchKernelGGL((k_colsum_finish<PRL_BF16>), fgrid, dim3(1024), 0, st, part, nblk, (int)H, (char*)gw);
  else
    hipLaunchKernelGGL((k_colsum_finish<PRL_F32>), fgrid, dim3(1024), 0, st, part, nblk, (int)H, (char*)gw);
  PRL_CHECK_LAUNCH(who);
  return PRL_OK;
}

extern "C" int prl_rmsnorm_bwd(const void* x, int64_t x_row_stride, const void* w, const float* rstd, const void* gy,
                               int64_t gy_row_stride, void* gx, int64_t gx_row_stride, void* gw, int dtype,
                               int64_t rows, int64_t H, void* workspace, size_t workspace_bytes, int64_t x_group,
                               int64_t x_group_stride, void* stream) {
  clear_error();
  return rmsnorm_bwd_impl("rmsnorm_bwd", x, x_row_stride, w, rstd, gy, gy_row_stride, nullptr, 0, gx, gx_row_stride, gw,
                          dtype, rows, H, workspace, workspace_bytes, x_group, x_group_stride, stream);
}

extern "C" int prl_add_rmsnorm_bwd(const void* x, int64_t x_row_stride, const void* w, const float* rstd,
                                   const void* gy, int64_t gy_row_stride, const void* gadd, int64_t gadd_row_stride,
                                   void* gx, int64_t gx_row_stride, void* gw, int dtype, int64_t rows, int64_t H,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  clear_error();
  PRL_REQUIRE(gadd, "add_rmsnorm_bwd: gadd must not be NULL");
  return rmsnorm_bwd_impl("add_rmsnorm_bwd", x, x_row_stride, w, rstd, gy, gy_row_stride, gadd, gadd_row_stride, gx,
                          gx_row_stride, gw, dtype, rows, H, workspace, workspace_bytes, 0, 0, stream);
}

// ---------------------------------------------------------------- rotary position embedding
//
// out = x * cos + rotate_half(x) * sin over the last dimension D of query and key, both in one launch:
//   out[i]       = x[i] cos[i]             - x[i + D/2] sin[i]              i <  D/2
//   out[i + D/2] = x[i + D/2] cos[i + D/2] + x[i] sin[i + D/2]
// and, with `inverse`, the transpose of that map (what the backward applies to the incoming gradient).
// The transformers blocks spell it as ~14 small kernels forward and ~20 backward per layer
// (slices, negation, concatenation, four products, two sums); the reference's switch for a fused
// version is again `model.impl = "liger_kernel"` (src/prime_rl/trainer/model.py:74-75).
namespace prl {

struct RopeTensor {
  const char* x;
  char* out;
  int64_t xb, xh, xt;  // element strides of batch, head, token in x (the last dimension is contiguous)
  int64_t ob, oh, ot;  // the same for out
  int64_t heads;
};

template <int DT>
__global__ __launch_bounds__(256) void k_rope(RopeTensor q, RopeTensor k, const char* __restrict__ cs,
                                              const char* __restrict__ sn, int64_t cb, int64_t ct, int64_t B,
                                              int64_t T, int D, int inverse) {
  constexpr int ES = DT == PRL_F32 ? 4 : 2;
  constexpr int VE = 16 / ES;
  const int lpr = D / (2 * VE);  // lanes per row: each owns one 16-byte vector of either half
  const int rpb = 256 / lpr;
  const int li = threadIdx.x % lpr, slot = threadIdx.x / lpr;
  if (slot >= rpb) return;
  const int64_t qrows = B * q.heads * T, rows = qrows + B * k.heads * T;
  for (int64_t r = (int64_t)blockIdx.x * rpb + slot; r < rows; r += (int64_t)gridDim.x * rpb) {
    const RopeTensor& a = r < qrows ? q : k;
    int64_t i = r < qrows ? r : r - qrows;
    // row order (batch, token, head): neighbouring rows share cos / sin and, in the usual layout, memory
    const int64_t h = i % a.heads;
    i /= a.heads;
    const int64_t t = i % T, b = i / T;
    const char* xp = a.x + (b * a.xb + h * a.xh + t * a.xt) * ES;
    char* op = a.out + (b * a.ob + h * a.oh + t * a.ot) * ES;
    const char* cp = cs + (b * cb + t * ct) * ES;
    const char* sp = sn + (b * cb + t * ct) * ES;
    float x1[VE], x2[VE], c1[VE], c2[VE], s1[VE], s2[VE], o1[VE], o2[VE];
    load_vec<DT>(xp, li, x1);
    load_vec<DT>(xp, li + lpr, x2);
    load_vec<DT>(cp, li, c1);
    load_vec<DT>(cp, li + lpr, c2);
    load_vec<DT>(sp, li, s1);
    load_vec<DT>(sp, li + lpr, s2);
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      if (!inverse) {
        o1[e] = fmaf(x1[e], c1[e], -(x2[e] * s1[e]));
        o2[e] = fmaf(x2[e], c2[e], x1[e] * s2[e]);
      } else {
        o1[e] = fmaf(x1[e], c1[e], x2[e] * s2[e]);
        o2[e] = fmaf(x2[e], c2[e], -(x1[e] * s1[e]));
      }
    }
    store_unit<DT, VE>(op, li, o1);
    store_unit<DT, VE>(op, li + lpr, o2);
  }
}

}  // namespace prl

extern "C" int prl_rope_supported(int dtype, int64_t D) {
  if (dtype != PRL_F32 && dtype != PRL_BF16) return 0;
  const int ve = dtype == PRL_F32 ? 4 : 8;
  return D >= 2 * ve && D % (2 * ve) == 0 && D / (2 * ve) <= 256 && 256 % (D / (2 * ve)) == 0 ? 1 : 0;
}

extern "C" int prl_rope(const void* q, void* q_out, const int64_t* q_strides, int64_t q_heads, const void* k, void* k_out,
                        const int64_t* k_strides, int64_t k_heads, const void* cos_, const void* sin_,
                        int64_t cs_batch_stride, int64_t cs_token_stride, int dtype, int64_t B, int64_t T, int64_t D,
                        int inverse, void* stream) {
  clear_error();
  PRL_REQUIRE(B >= 0 && T >= 0 && q_heads >= 0 && k_heads >= 0, "rope: sizes must be >= 0");
  PRL_REQUIRE(prl_rope_supported(dtype, D), "rope: unsupported dtype %d or head size %lld", dtype, (long long)D);
  if (B * T * (q_heads + k_heads) == 0) return PRL_OK;
  PRL_REQUIRE(q && q_out && k && k_out && cos_ && sin_ && q_strides && k_strides,
              "rope: tensors and stride arrays (host, 6 entries: batch, head, token of the input, then of the output) "
              "must not be NULL");
  const int es = dtype == PRL_F32 ? 4 : 2;
  uintptr_t bits = reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(q_out) | reinterpret_cast<uintptr_t>(k) |
                   reinterpret_cast<uintptr_t>(k_out) | reinterpret_cast<uintptr_t>(cos_) |
                   reinterpret_cast<uintptr_t>(sin_) | (uintptr_t)(cs_batch_stride * es) |
                   (uintptr_t)(cs_token_stride * es);
  for (int i = 0; i < 6; ++i) bits |= (uintptr_t)(q_strides[i] * es) | (uintptr_t)(k_strides[i] * es);
  PRL_REQUIRE((bits & 15) == 0, "rope: pointers and strides must be 16-byte aligned");
  prl::RopeTensor tq = {(const char*)q, (char*)q_out, q_strides[0], q_strides[1], q_strides[2],
                        q_strides[3],   q_strides[4], q_strides[5], q_heads};
  prl::RopeTensor tk = {(const char*)k, (char*)k_out, k_strides[0], k_strides[1], k_strides[2],
                        k_strides[3],   k_strides[4], k_strides[5], k_heads};
  const int ve = 16 / es;
  const int64_t rpb = 256 / (D / (2 * ve));
  const int64_t rows = B * T * (q_heads + k_heads);
  int64_t nblk = (rows + rpb - 1) / rpb;
  if (nblk > 4096) nblk = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PRL_BF16)
    hipLaunchKernelGGL((prl::k_rope<PRL_BF16>), dim3((unsigned)nblk), dim3(256), 0, st, tq, tk, (const char*)cos_,
                       (const char*)sin_, cs_batch_stride, cs_token_stride, B, T, (int)D, inverse ? 1 : 0);
  else
    hipLaunchKernelGGL((prl::k_rope<PRL_F32>), dim3((unsigned)nblk), dim3(256), 0, st, tq, tk, (const char*)cos_,
                       (const char*)sin_, cs_batch_stride, cs_token_stride, B, T, (int)D, inverse ? 1 : 0);
  PRL_CHECK_LAUNCH("rope");
  return PRL_OK;
}

// ---------------------------------------------------------------- query / key RMSNorm + rotary embedding
//
// What a Qwen3 attention block does between its q/k/v projection and the attention itself: RMSNorm over
// every head of the query and of the key (head dimension 128), then the rotary embedding of both.  As
// separate launches that is two norms and one rotation forward, and the rotation, two norm backwards, two
// column-sum finishes and a concatenation of the three projection gradients backward.  Here one launch reads
// the query and key blocks out of the stacked projection output [token][q heads | k heads | v heads][128] and
// writes the rotated query and key; the backward reads their gradients and writes the gradient with respect to
// the projection output straight INTO the stacked gradient matrix, so the projection's backward is one product
// without a concatenation.  Same switch in the reference as the other block kernels
// (`model.impl = "liger_kernel"`, src/prime_rl/trainer/model.py:74-75).
namespace prl {

constexpr int kQkD = 128;

// A row is one (token, head); lanes of a row: each owns one 16-byte vector of the first half of the head
// dimension and the matching one of the second half (the rotation pairs element i with i + 64).
template <int DT>
__global__ __launch_bounds__(256) void k_qk_norm_rope_fwd(const char* __restrict__ y, int64_t y_pitch,
                                                          const char* __restrict__ wq, const char* __restrict__ wk,
                                                          const char* __restrict__ cs, const char* __restrict__ sn,
                                                          int64_t ct, char* __restrict__ q_out,
                                                          char* __restrict__ k_out, float* __restrict__ rstd,
                                                          int64_t T, int Hq, int Hk, float eps) {
  constexpr int ES = DT == PRL_F32 ? 4 : 2;
  constexpr int VE = 16 / ES;
  constexpr int LPR = kQkD / (2 * VE);
  constexpr int RPB = 256 / LPR;
  const int li = threadIdx.x % LPR, slot = threadIdx.x / LPR;
  const int H = Hq + Hk;
  const int64_t rows = T * H;
  float wq1[VE], wq2[VE], wk1[VE], wk2[VE];
  load_vec<DT>(wq, li, wq1), load_vec<DT>(wq, li + LPR, wq2);
  load_vec<DT>(wk, li, wk1), load_vec<DT>(wk, li + LPR, wk2);
  for (int64_t r = (int64_t)blockIdx.x * RPB + slot; r < rows; r += (int64_t)gridDim.x * RPB) {
    const int64_t t = r / H;
    const int hh = (int)(r - t * H);
    const bool is_q = hh < Hq;
    const char* xp = y + t * y_pitch + (int64_t)hh * kQkD * ES;
    char* op = is_q ? q_out + (t * Hq + hh) * kQkD * ES : k_out + (t * Hk + (hh - Hq)) * kQkD * ES;
    float x1[VE], x2[VE], c1[VE], c2[VE], s1[VE], s2[VE], o1[VE], o2[VE];
    load_vec<DT>(xp, li, x1), load_vec<DT>(xp, li + LPR, x2);
    load_vec<DT>(cs + t * ct, li, c1), load_vec<DT>(cs + t * ct, li + LPR, c2);
    load_vec<DT>(sn + t * ct, li, s1), load_vec<DT>(sn + t * ct, li + LPR, s2);
    float ss = 0.0f;
#pragma unroll
    for (int e = 0; e < VE; ++e) ss = fmaf(x1[e], x1[e], fmaf(x2[e], x2[e], ss));
    ss = row_sum<LPR>(ss);
    const float rs = rsqrtf(ss / (float)kQkD + eps);
    if (li == 0) rstd[r] = rs;
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      // the norm's output as the module rounds it (weight * cast(x * rstd), cast again), then the rotation
      const float n1 = round_to((is_q ? wq1[e] : wk1[e]) * round_to(x1[e] * rs, DT), DT);
      const float n2 = round_to((is_q ? wq2[e] : wk2[e]) * round_to(x2[e] * rs, DT), DT);
      o1[e] = fmaf(n1, c1[e], -(n2 * s1[e]));
      o2[e] = fmaf(n2, c2[e], n1 * s2[e]);
    }
    store_unit<DT, VE>(op, li, o1);
    store_unit<DT, VE>(op, li + LPR, o2);
  }
}

// part[blockIdx.x][0..127] = this workgroup's column sums for the query norm's weight, [128..255] the key's
template <int DT>
// gq / gk may be the query / key blocks of gy itself: a lane stores exactly the elements it loaded, after it
// loaded them, so the three pointers carry no no-alias promise.
__global__ __launch_bounds__(256) void k_qk_norm_rope_bwd(const char* gq, int64_t gq_ts, int64_t gq_hs,
                                                          const char* gk, int64_t gk_ts, int64_t gk_hs,
                                                          const char* __restrict__ y, int64_t y_pitch,
                                                          const char* __restrict__ wq, const char* __restrict__ wk,
                                                          const float* __restrict__ rstd,
                                                          const char* __restrict__ cs, const char* __restrict__ sn,
                                                          int64_t ct, char* gy, int64_t gy_pitch,
                                                          float* __restrict__ part, int64_t T, int Hq, int Hk) {
  constexpr int ES = DT == PRL_F32 ? 4 : 2;
  constexpr int VE = 16 / ES;
  constexpr int LPR = kQkD / (2 * VE);
  constexpr int RPB = 256 / LPR;
  extern __shared__ float lds[];  // RPB * 256 floats
  const int li = threadIdx.x % LPR, slot = threadIdx.x / LPR;
  const int H = Hq + Hk;
  const int64_t rows = T * H;
  float wq1[VE], wq2[VE], wk1[VE], wk2[VE];
  load_vec<DT>(wq, li, wq1), load_vec<DT>(wq, li + LPR, wq2);
  load_vec<DT>(wk, li, wk1), load_vec<DT>(wk, li + LPR, wk2);
  float aq1[VE], aq2[VE], ak1[VE], ak2[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) aq1[e] = aq2[e] = ak1[e] = ak2[e] = 0.0f;
  for (int64_t r = (int64_t)blockIdx.x * RPB + slot; r < rows; r += (int64_t)gridDim.x * RPB) {
    const int64_t t = r / H;
    const int hh = (int)(r - t * H);
    const bool is_q = hh < Hq;
    const char* gp = is_q ? gq + t * gq_ts + (int64_t)hh * gq_hs : gk + t * gk_ts + (int64_t)(hh - Hq) * gk_hs;
    const char* xp = y + t * y_pitch + (int64_t)hh * kQkD * ES;
    char* op = gy + t * gy_pitch + (int64_t)hh * kQkD * ES;
    float g1[VE], g2[VE], x1[VE], x2[VE], c1[VE], c2[VE], s1[VE], s2[VE], o1[VE], o2[VE];
    load_vec<DT>(gp, li, g1), load_vec<DT>(gp, li + LPR, g2);
    load_vec<DT>(xp, li, x1), load_vec<DT>(xp, li + LPR, x2);
    load_vec<DT>(cs + t * ct, li, c1), load_vec<DT>(cs + t * ct, li + LPR, c2);
    load_vec<DT>(sn + t * ct, li, s1), load_vec<DT>(sn + t * ct, li + LPR, s2);
    const float rs = rstd[r];
    float dot = 0.0f;
    float n1[VE], n2[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      // the rotation's transpose gives the gradient with respect to the norm's output ...
      const float d1 = fmaf(g1[e], c1[e], g2[e] * s2[e]);
      const float d2 = fmaf(g2[e], c2[e], -(g1[e] * s1[e]));
      // ... and the norm's backward follows: xh = x rstd, gx = rstd (d w - xh mean(d w xh))
      x1[e] *= rs, x2[e] *= rs;
      if (is_q) {
        aq1[e] = fmaf(d1, round_to(x1[e], DT), aq1[e]), aq2[e] = fmaf(d2, round_to(x2[e], DT), aq2[e]);
      } else {
        ak1[e] = fmaf(d1, round_to(x1[e], DT), ak1[e]), ak2[e] = fmaf(d2, round_to(x2[e], DT), ak2[e]);
      }
      n1[e] = d1 * (is_q ? wq1[e] : wk1[e]);
      n2[e] = d2 * (is_q ? wq2[e] : wk2[e]);
      dot = fmaf(n1[e], x1[e], fmaf(n2[e], x2[e], dot));
    }
    dot = row_sum<LPR>(dot) * (1.0f / (float)kQkD);
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      o1[e] = rs * (n1[e] - x1[e] * dot);
      o2[e] = rs * (n2[e] - x2[e] * dot);
    }
    store_unit<DT, VE>(op, li, o1);
    store_unit<DT, VE>(op, li + LPR, o2);
  }
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    float* row = lds + slot * 2 * kQkD;
    row[li * VE + e] = aq1[e], row[kQkD / 2 + li * VE + e] = aq2[e];
    row[kQkD + li * VE + e] = ak1[e], row[kQkD + kQkD / 2 + li * VE + e] = ak2[e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * kQkD; c += 256) {
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < RPB; ++k) sum += lds[k * 2 * kQkD + c];
    part[(int64_t)blockIdx.x * 2 * kQkD + c] = sum;
  }
}

static int qk_blocks(int64_t rows, int dtype) {
  const int rpb = 256 / (kQkD / (2 * (dtype == PRL_F32 ? 4 : 8)));
  const int64_t want = (rows + rpb - 1) / rpb;
  int64_t cap = want / 8;  // at least 8 row steps per workgroup, between 256 and 1024 workgroups
  if (cap < 256) cap = 256;
  if (cap > 1024) cap = 1024;
  return (int)(want < cap ? (want > 0 ? want : 1) : cap);
}

}  // namespace prl

extern "C" int prl_qk_norm_rope_supported(int dtype, int64_t head_dim) {
  return (dtype == PRL_F32 || dtype == PRL_BF16) && head_dim == prl::kQkD;
}

extern "C" size_t prl_qk_norm_rope_bwd_workspace_bytes(int dtype, int64_t T, int64_t q_heads, int64_t k_heads) {
  return (size_t)prl::qk_blocks(T * (q_heads + k_heads), dtype) * 2 * prl::kQkD * sizeof(float);
}

static int qk_check(const char* who, int dtype, int64_t T, int64_t Hq, int64_t Hk, int64_t pitch, int64_t ct) {
  PRL_REQUIRE(dtype == PRL_F32 || dtype == PRL_BF16, "%s: unsupported dtype %d", who, dtype);
  PRL_REQUIRE(T >= 0 && Hq >= 1 && Hk >= 1 && Hq + Hk <= 4096, "%s: bad sizes", who);
  const int es = dtype == PRL_F32 ? 4 : 2;
  PRL_REQUIRE(pitch >= (Hq + Hk) * prl::kQkD && (pitch * es) % 16 == 0 && ct >= prl::kQkD && (ct * es) % 16 == 0,
              "%s: row pitches must cover the rows and be 16-byte aligned", who);
  return PRL_OK;
}

extern "C" int prl_qk_norm_rope_fwd(const void* y, int64_t y_pitch, const void* wq, const void* wk, const void* cos_,
                                    const void* sin_, int64_t cs_token_stride, void* q_out, void* k_out, float* rstd,
                                    int dtype, int64_t T, int64_t q_heads, int64_t k_heads, float eps, void* stream) {
  prl::clear_error();
  int rc = qk_check("qk_norm_rope_fwd", dtype, T, q_heads, k_heads, y_pitch, cs_token_stride);
  if (rc != PRL_OK || T == 0) return rc;
  PRL_REQUIRE(y && wq && wk && cos_ && sin_ && q_out && k_out && rstd, "qk_norm_rope_fwd: NULL argument");
  PRL_REQUIRE(((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(wq) | reinterpret_cast<uintptr_t>(wk) |
                reinterpret_cast<uintptr_t>(cos_) | reinterpret_cast<uintptr_t>(sin_) |
                reinterpret_cast<uintptr_t>(q_out) | reinterpret_cast<uintptr_t>(k_out)) & 15) == 0,
              "qk_norm_rope_fwd: tensors must be 16-byte aligned");
  const int es = dtype == PRL_F32 ? 4 : 2;
  const dim3 grid((unsigned)prl::qk_blocks(T * (q_heads + k_heads), dtype) * 4 > 1024
                      ? 1024u
                      : (unsigned)prl::qk_blocks(T * (q_heads + k_heads), dtype) * 4),
      block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PRL_BF16)
    hipLaunchKernelGGL((prl::k_qk_norm_rope_fwd<PRL_BF16>), grid, block, 0, st, (const char*)y, y_pitch * es,
                       (const char*)wq, (const char*)wk, (const char*)cos_, (const char*)sin_, cs_token_stride * es,
                       (char*)q_out, (char*)k_out, rstd, T, (int)q_heads, (int)k_heads, eps);
  else
    hipLaunchKernelGGL((prl::k_qk_norm_rope_fwd<PRL_F32>), grid, block, 0, st, (const char*)y, y_pitch * es,
                       (const char*)wq, (const char*)wk, (const char*)cos_, (const char*)sin_, cs_token_stride * es,
                       (char*)q_out, (char*)k_out, rstd, T, (int)q_heads, (int)k_heads, eps);
  PRL_CHECK_LAUNCH("qk_norm_rope_fwd");
  return PRL_OK;
}

extern "C" int prl_qk_norm_rope_bwd(const void* gq, const int64_t* gq_strides, const void* gk,
                                    const int64_t* gk_strides, const void* y, int64_t y_pitch, const void* wq,
                                    const void* wk, const float* rstd, const void* cos_, const void* sin_,
                                    int64_t cs_token_stride, void* gy, int64_t gy_pitch, void* gw, void* workspace,
                                    size_t workspace_bytes, int dtype, int64_t T, int64_t q_heads, int64_t k_heads,
                                    void* stream) {
  prl::clear_error();
  int rc = qk_check("qk_norm_rope_bwd", dtype, T, q_heads, k_heads, y_pitch, cs_token_stride);
  if (rc != PRL_OK) return rc;
  rc = qk_check("qk_norm_rope_bwd", dtype, T, q_heads, k_heads, gy_pitch, cs_token_stride);
  if (rc != PRL_OK) return rc;
  PRL_REQUIRE(gq && gk && gq_strides && gk_strides && y && wq && wk && rstd && cos_ && sin_ && gy && gw,
              "qk_norm_rope_bwd: NULL argument");
  const int es = dtype == PRL_F32 ? 4 : 2;
  PRL_REQUIRE(((reinterpret_cast<uintptr_t>(gq) | reinterpret_cast<uintptr_t>(gk) | reinterpret_cast<uintptr_t>(y) |
                reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(wq) | reinterpret_cast<uintptr_t>(wk) |
                reinterpret_cast<uintptr_t>(cos_) | reinterpret_cast<uintptr_t>(sin_) |
                (uintptr_t)(gq_strides[0] * es) | (uintptr_t)(gq_strides[1] * es) | (uintptr_t)(gk_strides[0] * es) |
                (uintptr_t)(gk_strides[1] * es)) & 15) == 0,
              "qk_norm_rope_bwd: tensors and strides must be 16-byte aligned");
  const int nblk = prl::qk_blocks(T * (q_heads + k_heads), dtype);
  PRL_REQUIRE(workspace && workspace_bytes >= (size_t)nblk * 2 * prl::kQkD * sizeof(float),
              "qk_norm_rope_bwd: workspace too small (prl_qk_norm_rope_bwd_workspace_bytes)");
  hipStream_t st = (hipStream_t)stream;
  float* part = static_cast<float*>(workspace);
  const int rpb = 256 / (prl::kQkD / (2 * (16 / es)));
  const size_t lds = (size_t)rpb * 2 * prl::kQkD * sizeof(float);
  if (T > 0) {
    if (dtype == PRL_BF16)
      hipLaunchKernelGGL((prl::k_qk_norm_rope_bwd<PRL_BF16>), dim3((unsigned)nblk), dim3(256), lds, st, (const char*)gq,
                         gq_strides[0] * es, gq_strides[1] * es, (const char*)gk, gk_strides[0] * es,
                         gk_strides[1] * es, (const char*)y, y_pitch * es, (const char*)wq, (const char*)wk, rstd,
                         (const char*)cos_, (const char*)sin_, cs_token_stride * es, (char*)gy, gy_pitch * es, part, T,
                         (int)q_heads, (int)k_heads);
    else
      hipLaunchKernelGGL((prl::k_qk_norm_rope_bwd<PRL_F32>), dim3((unsigned)nblk), dim3(256), lds, st, (const char*)gq,
                         gq_strides[0] * es, gq_strides[1] * es, (const char*)gk, gk_strides[0] * es,
                         gk_strides[1] * es, (const char*)y, y_pitch * es, (const char*)wq, (const char*)wk, rstd,
                         (const char*)cos_, (const char*)sin_, cs_token_stride * es, (char*)gy, gy_pitch * es, part, T,
                         (int)q_heads, (int)k_heads);
    PRL_CHECK_LAUNCH("qk_norm_rope_bwd");
  }
  const dim3 fgrid((unsigned)(2 * prl::kQkD / 32));
  if (dtype == PRL_BF16)
    hipLaunchKernelGGL((prl::k_colsum_finish<PRL_BF16>), fgrid, dim3(1024), 0, st, part, T > 0 ? nblk : 0,
                       2 * prl::kQkD, (char*)gw);
  else
    hipLaunchKernelGGL((prl::k_colsum_finish<PRL_F32>), fgrid, dim3(1024), 0, st, part, T > 0 ? nblk : 0,
                       2 * prl::kQkD, (char*)gw);
  PRL_CHECK_LAUNCH("qk_norm_rope_bwd(finish)");
  return PRL_OK;
}

// ---------------------------------------------------------------- SwiGLU
//
// out = silu(gate) * up, and its backward, element-wise over contiguous tensors (the gated MLP of the
// transformer blocks: down_proj(act_fn(gate_proj(x)) * up_proj(x))); the third of the three kernels behind
// the reference's `model.impl = "liger_kernel"` (src/prime_rl/trainer/model.py:74-75).  One launch forward
// (torch: 2), one backward (torch: 5), and the activation's output is not kept for the backward.
namespace prl {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <int DT>
__global__ __launch_bounds__(256) void k_swiglu_fwd(const char* __restrict__ gate, const char* __restrict__ up,
                                                    char* __restrict__ out, int64_t nvec) {
  constexpr int VE = DT == PRL_F32 ? 4 : 8;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
    float g[VE], u[VE], o[VE];
    load_vec<DT>(gate, v, g);
    load_vec<DT>(up, v, u);
#pragma unroll
    for (int i = 0; i < VE; ++i) o[i] = round_to(g[i] * sigmoidf_(g[i]), DT) * u[i];  // torch rounds silu's output first
    store_unit<DT, VE>(out, v, o);
  }
}

template <int DT>
__global__ __launch_bounds__(256) void k_swiglu_bwd(const char* __restrict__ gate, const char* __restrict__ up,
                                                    const char* __restrict__ gout, char* __restrict__ ggate,
                                                    char* __restrict__ gup, int64_t nvec) {
  constexpr int VE = DT == PRL_F32 ? 4 : 8;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
    float g[VE], u[VE], d[VE], dg[VE], du[VE];
    load_vec<DT>(gate, v, g);
    load_vec<DT>(up, v, u);
    load_vec<DT>(gout, v, d);
#pragma unroll
    for (int i = 0; i < VE; ++i) {
      const float s = sigmoidf_(g[i]);
      du[i] = d[i] * round_to(g[i] * s, DT);
      dg[i] = round_to(d[i] * u[i], DT) * (s * (1.0f + g[i] * (1.0f - s)));
    }
    store_unit<DT, VE>(ggate, v, dg);
    store_unit<DT, VE>(gup, v, du);
  }
}

// The same two kernels over ROWS with a pitch: gate and up are column blocks of one wider matrix (the output
// of one product over the concatenated gate / up weights), and so are their gradients.
template <int DT>
__global__ __launch_bounds__(256) void k_swiglu_rows_fwd(const char* __restrict__ gate, const char* __restrict__ up,
                                                         int64_t in_pitch, char* __restrict__ out, int64_t out_pitch,
                                                         int64_t rows, int64_t row_vecs) {
  constexpr int VE = DT == PRL_F32 ? 4 : 8;
  const int64_t nvec = rows * row_vecs;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
    // (32-bit division where it fits: a 64-bit one costs a few dozen instructions per vector)
    const int64_t row = nvec < ((int64_t)1 << 31) ? (int64_t)((uint32_t)v / (uint32_t)row_vecs) : v / row_vecs;
    const int64_t cv = v - row * row_vecs;
    float g[VE], u[VE], o[VE];
    load_vec<DT>(gate + row * in_pitch, cv, g);
    load_vec<DT>(up + row * in_pitch, cv, u);
#pragma unroll
    for (int i = 0; i < VE; ++i) o[i] = round_to(g[i] * sigmoidf_(g[i]), DT) * u[i];
    store_unit<DT, VE>(out + row * out_pitch, cv, o);
  }
}

template <int DT>
__global__ __launch_bounds__(256) void k_swiglu_rows_bwd(const char* __restrict__ gate, const char* __restrict__ up,
                                                         int64_t in_pitch, const char* __restrict__ gout,
                                                         int64_t gout_pitch, char* __restrict__ ggate,
                                                         char* __restrict__ gup, int64_t grad_pitch, int64_t rows,
                                                         int64_t row_vecs) {
  constexpr int VE = DT == PRL_F32 ? 4 : 8;
  const int64_t nvec = rows * row_vecs;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
    const int64_t row = nvec < ((int64_t)1 << 31) ? (int64_t)((uint32_t)v / (uint32_t)row_vecs) : v / row_vecs;
    const int64_t cv = v - row * row_vecs;
    float g[VE], u[VE], d[VE], dg[VE], du[VE];
    load_vec<DT>(gate + row * in_pitch, cv, g);
    load_vec<DT>(up + row * in_pitch, cv, u);
    load_vec<DT>(gout + row * gout_pitch, cv, d);
#pragma unroll
    for (int i = 0; i < VE; ++i) {
      const float s = sigmoidf_(g[i]);
      du[i] = d[i] * round_to(g[i] * s, DT);
      dg[i] = round_to(d[i] * u[i], DT) * (s * (1.0f + g[i] * (1.0f - s)));
    }
    store_unit<DT, VE>(ggate + row * grad_pitch, cv, dg);
    store_unit<DT, VE>(gup + row * grad_pitch, cv, du);
  }
}

}  // namespace prl

// One vector per thread and exit wherever the grid allows it: short-lived waves sweep the operands front to back,
// which streams faster than a capped grid striding over them (6.5 against 5.6-5.8 TB/s for a plain copy).
static constexpr int64_t kSwigluGridCap = (int64_t)1 << 30;

static int swiglu_check(const char* who, int dtype, int64_t n, const void* a, const void* b, const void* c) {
  if (dtype != PRL_F32 && dtype != PRL_BF16) {
    prl::set_error("%s: unsupported dtype code %d", who, dtype);
    return PRL_EDTYPE;
  }
  const int ve = dtype == PRL_F32 ? 4 : 8;
  PRL_REQUIRE(n >= 0 && n % ve == 0, "%s: the element count must be a multiple of %d", who, ve);
  PRL_REQUIRE(n == 0 || (a && b && c), "%s: tensors must not be NULL", who);
  PRL_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0,
              "%s: tensors must be 16-byte aligned", who);
  return PRL_OK;
}

extern "C" int prl_swiglu_fwd(const void* gate, const void* up, void* out, int dtype, int64_t n, void* stream) {
  clear_error();
  int rc = swiglu_check("swiglu_fwd", dtype, n, gate, up, out);
  if (rc != PRL_OK || n == 0) return rc;
  const int64_t nvec = n / (dtype == PRL_F32 ? 4 : 8);
  const int64_t want = (nvec + 255) / 256;
  const dim3 grid((unsigned)(want < kSwigluGridCap ? want : kSwigluGridCap));
  if (dtype == PRL_BF16)
    hipLaunchKernelGGL((prl::k_swiglu_fwd<PRL_BF16>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)gate,
                       (const char*)up, (char*)out, nvec);
  else
    hipLaunchKernelGGL((prl::k_swiglu_fwd<PRL_F32>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)gate,
                       (const char*)up, (char*)out, nvec);
  PRL_CHECK_LAUNCH("swiglu_fwd");
  return PRL_OK;
}

extern "C" int prl_swiglu_bwd(const void* gate, const void* up, const void* gout, void* ggate, void* gup, int dtype,
                              int64_t n, void* stream) {
  clear_error();
  int rc = swiglu_check("swiglu_bwd", dtype, n, gate, up, gout);
  if (rc != PRL_OK) return rc;
  rc = swiglu_check("swiglu_bwd", dtype, n, ggate, gup, gout);
  if (rc != PRL_OK || n == 0) return rc;
  const int64_t nvec = n / (dtype == PRL_F32 ? 4 : 8);
  const int64_t want = (nvec + 255) / 256;
  const dim3 grid((unsigned)(want < kSwigluGridCap ? want : kSwigluGridCap));
  if (dtype == PRL_BF16)
    hipLaunchKernelGGL((prl::k_swiglu_bwd<PRL_BF16>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)gate,
                       (const char*)up, (const char*)gout, (char*)ggate, (char*)gup, nvec);
  else
    hipLaunchKernelGGL((prl::k_swiglu_bwd<PRL_F32>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)gate,
                       (const char*)up, (const char*)gout, (char*)ggate, (char*)gup, nvec);
  PRL_CHECK_LAUNCH("swiglu_bwd");
  return PRL_OK;
}

static int swiglu_rows_check(const char* who, int dtype, int64_t rows, int64_t cols, int n_pitch,
                             const int64_t* pitches, int n_ptr, const void* const* ptrs) {
  if (dtype != PRL_F32 && dtype != PRL_BF16) {
    prl::set_error("%s: unsupported dtype code %d", who, dtype);
    return PRL_EDTYPE;
  }
  const int ve = dtype == PRL_F32 ? 4 : 8;
  PRL_REQUIRE(rows >= 0 && cols >= 0 && cols % ve == 0, "%s: the row length must be a multiple of %d", who, ve);
  for (int i = 0; i < n_pitch; ++i)
    PRL_REQUIRE(pitches[i] >= cols && pitches[i] % ve == 0, "%s: a row pitch is shorter than a row or not a multiple of %d",
                who, ve);
  for (int i = 0; i < n_ptr; ++i)
    PRL_REQUIRE(rows * cols == 0 || (ptrs[i] && (reinterpret_cast<uintptr_t>(ptrs[i]) & 15) == 0),
                "%s: tensors must be 16-byte aligned and not NULL", who);
  return PRL_OK;
}

extern "C" int prl_swiglu_rows_fwd(const void* gate, const void* up, int64_t in_pitch, void* out, int64_t out_pitch,
                                   int dtype, int64_t rows, int64_t cols, void* stream) {
  clear_error();
  const int64_t pitches[2] = {in_pitch, out_pitch};
  const void* ptrs[3] = {gate, up, out};
  int rc = swiglu_rows_check("swiglu_rows_fwd", dtype, rows, cols, 2, pitches, 3, ptrs);
  if (rc != PRL_OK || rows * cols == 0) return rc;
  const int es = dtype == PRL_F32 ? 4 : 2;
  const int64_t row_vecs = cols * es / 16, want = (rows * row_vecs + 255) / 256;
  const dim3 grid((unsigned)(want < kSwigluGridCap ? want : kSwigluGridCap));
  if (dtype == PRL_BF16)
    hipLaunchKernelGGL((prl::k_swiglu_rows_fwd<PRL_BF16>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)gate,
                       (const char*)up, in_pitch * es, (char*)out, out_pitch * es, rows, row_vecs);
  else
    hipLaunchKernelGGL((prl::k_swiglu_rows_fwd<PRL_F32>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)gate,
                       (const char*)up, in_pitch * es, (char*)out, out_pitch * es, rows, row_vecs);
  PRL_CHECK_LAUNCH("swiglu_rows_fwd");
  return PRL_OK;
}

extern "C" int prl_swiglu_rows_bwd(const void* gate, const void* up, int64_t in_pitch, const void* gout,
                                   int64_t gout_pitch, void* ggate, void* gup, int64_t grad_pitch, int dtype,
                                   int64_t rows, int64_t cols, void* stream) {
  clear_error();
  const int64_t pitches[3] = {in_pitch, gout_pitch, grad_pitch};
  const void* ptrs[5] = {gate, up, gout, ggate, gup};
  int rc = swiglu_rows_check("swiglu_rows_bwd", dtype, rows, cols, 3, pitches, 5, ptrs);
  if (rc != PRL_OK || rows * cols == 0) return rc;
  const int es = dtype == PRL_F32 ? 4 : 2;
  const int64_t row_vecs = cols * es / 16, want = (rows * row_vecs + 255) / 256;
  const dim3 grid((unsigned)(want < kSwigluGridCap ? want : kSwigluGridCap));
  if (dtype == PRL_BF16)
    hipLaunchKernelGGL((prl::k_swiglu_rows_bwd<PRL_BF16>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)gate,
                       (const char*)up, in_pitch * es, (const char*)gout, gout_pitch * es, (char*)ggate, (char*)gup,
                       grad_pitch * es, rows, row_vecs);
  else
    hipLaunchKernelGGL((prl::k_swiglu_rows_bwd<PRL_F32>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)gate,
                       (const char*)up, in_pitch * es, (const char*)gout, gout_pitch * es, (char*)ggate, (char*)gup,
                       grad_pitch * es, rows, row_vecs);
  PRL_CHECK_LAUNCH("swiglu_rows_bwd");
  return PRL_OK;
}
