// The three products of the LM head (src/prime_rl/trainer/model.py:251-254 forward, autograd's two products behind
// src/prime_rl/trainer/rl/train.py:310) as hand-written bf16 MFMA kernels for gfx950:
//     logits[T, V]  = hidden[T, H] . weight[V, H]^T
//     dhidden[T, H] = dlogits[T, V] . weight[V, H]
//     dweight[V, H] = dlogits[T, V]^T . hidden[T, H]
// One kernel template serves all three.  It computes C = P . Q for a "K-major" P ([M][K], K contiguous) and a Q
// that is either K-major ([N][K]) or "MN-major" ([K][N], N contiguous, fragments through the transposing LDS
// read), and stores C, C^T or an fp32 partial slab of a K split:
//     forward:  P = hidden,   Q = weight  (K-major, K = H),  C = logits
//     dhidden:  P = dlogits,  Q = weight  (MN-major, K = V), C = dhidden (fp32 slabs summed afterwards when K is split)
//     dweight:  P = hidden^T, Q = dlogits (MN-major, K = T), C^T = dweight (hidden^T is made by a small transpose)
//
// Structure of a workgroup: a 256 x 256 tile of C, K walked in tiles of 64, 8 waves as 2 (rows) x 4 (columns), each
// wave 128 x 64 of C in 32 accumulator tiles of v_mfma_f32_16x16x32_bf16.  A K-tile's operands (2 x 32 KB) come into
// one of two LDS stages by direct-to-LDS loads (no registers on the way); the LDS images are XOR-swizzled through the
// per-lane SOURCE address so that the 16-byte row reads and the transposing reads are free of bank conflicts.  A
// K-tile is four phases (one 64 x 32 quadrant of the wave's output each, 16 MFMAs); a phase is a load section (LDS
// reads for this phase, two direct-to-LDS pieces for the K-tiles ahead) and an MFMA section, separated by barriers,
// and the two waves that share a SIMD run one section apart, so that one computes while the other loads.
#include <stdlib.h>

#include "prl_common.h"
#include <cstring>

namespace prl {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

constexpr int kBM = 256, kBN = 256, kBK = 64;
constexpr int kGemmThreads = 512;
constexpr int kQuarterBytes = 16384;         // 128 rows x 64 k, or 64 k x 128 columns, of bf16
constexpr int kStageBytes = 4 * kQuarterBytes;  // P half 0 | P half 1 | Q half 0 | Q half 1

enum { kOutBf16 = 0, kOutF32Slab = 1, kOutBf16Transposed = 2, kOutF32SlabTransposed = 3 };
constexpr bool transposed_out(int out) { return out == kOutBf16Transposed || out == kOutF32SlabTransposed; }

struct GemmArgs {
  const uint16_t* P;  // [M][K], ldp elements between rows
  const uint16_t* Q;  // K-major: [N][K]; MN-major: [K][N]; ldq elements between rows
  void* C;            // bf16 [M][N] (ldc), bf16 [N][M] (ldc, transposed store), fp32 [nsplit][M][N] (ldc = N) or
                      // fp32 [nsplit][N][M] (ldc = M, transposed store)
  int64_t ldp, ldq, ldc;
  int64_t slab_stride;  // elements between the fp32 slabs of two K splits
  int M, N;
  int ktiles;            // K / 64
  int ktiles_per_split;  // K-tiles one work item walks
  int nsplit;
  int mtiles, ntiles;    // ceil(M / 256), ceil(N / 256)
  int valid_items;       // mtiles * ntiles * nsplit: (tile, K split) pairs, in the order `gemm_item` walks them
  int total_items;       // 256 * ceil(ceil(valid_items / 32) / 8): slots of that order dealt to the workgroups
  int cached_stores;     // bf16 tiles inside the matrix are stored with cacheable instead of non-temporal stores
  unsigned long long* stamps;  // diagnostic builds only (PRL_GEMM_STAMPS): per workgroup 4 phase sums, items, clocks
  unsigned long long* stamps2;  // the same builds: per workgroup, shader cycles in each of a K-tile's 8 sections (wave 0)
};

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* global_void_ptr;

__device__ __forceinline__ void glds16(const uint16_t* src, char* lds_wave_base) {
  // 16 bytes per lane from `src` (per lane) to lds_wave_base + 16 * lane (the base is the same in every lane)
  __builtin_amdgcn_global_load_lds((global_void_ptr)src, (lds_void_ptr)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ bf16x8 lds_frag(const char* p) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
}
// One transposing LDS read: a 16-lane group gets a block of 4 rows x 16 columns of 16-bit elements column-major;
// lane 4q + p of the group supplies the address of row q, columns 4p .. 4p + 3, lane i receives column i of the 4
// rows.  Written as inline assembly on purpose: the compiler treats the builtin for this instruction as a possible
// LDS store and puts a full `s_waitcnt vmcnt(0)` between a direct-to-LDS load in flight and the read, which would
// drain the prefetch in every phase.  An asm load is not counted by the compiler, so `wait_tr_reads` below must stand
// between these reads and the first use of their results.
template <int OFF>
__device__ __forceinline__ u32x2_t lds_read_tr(unsigned lds_addr) {
  u32x2_t r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF) : "memory");
  return r;
}
struct QFrag {  // one 16 x 32 operand fragment as its two 8-byte halves (k 0-3 and 4-7 of the lane's eight)
  u32x2_t lo, hi;
};
__device__ __forceinline__ bf16x8 as_bf16x8(const QFrag& f) {
  const u32x4 v = {f.lo.x, f.lo.y, f.hi.x, f.hi.y};
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// Which tile and K split work item `item` is.  The kernel is persistent: 256 workgroups, one per compute unit,
// workgroup b taking the items b, b + 256, ...  Workgroup b runs on XCD b % 8 (so do all its items), and the 32
// items an XCD's compute units work on at a time are 32 consecutive (tile, K split) pairs of ONE order: split by
// split, and inside a split down strips of four tile columns (8 row tiles x 4 column tiles where the counts
// allow), so that the operand panels they share are read from memory once and from that XCD's L2 afterwards.  The
// order holds no padding: every group of 32 but the last is full, whatever the tile counts (ragged edges, K splits).
// False: beyond the last pair.
__device__ __forceinline__ bool gemm_item(const GemmArgs& a, int item, int& mt, int& nt, int& split) {
  const int xcd = item & 7, slot = item >> 3;
  const int index = (((slot >> 5) * 8 + xcd) << 5) + (slot & 31);
  if (index >= a.valid_items) return false;
  const int per_split = a.mtiles * a.ntiles;
  split = index / per_split;
  const int r = index - split * per_split;
  const int strip = r / (a.mtiles * 4);
  const int within = r - strip * a.mtiles * 4;
  const int width = a.ntiles - 4 * strip < 4 ? a.ntiles - 4 * strip : 4;
  mt = within / width;
  nt = 4 * strip + (within - mt * width);
  return true;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {  // all but the wave's N youngest vector-memory operations are done
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// stores a wave issues per tile BEHIND the next tile's first direct-to-LDS loads (tiles inside the matrix only)
template <int OUT>
constexpr int late_stores() {
  return OUT == kOutBf16 ? 16 : 32;
}

template <bool P_KMAJOR, bool Q_KMAJOR, int OUT>
__global__ void __launch_bounds__(kGemmThreads, 2) k_lm_gemm(const GemmArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * kStageBytes];

#ifdef PRL_GEMM_STAMPS  // diagnostic build: where a workgroup's time goes (100 MHz clock), never in the product.
  // Sums per phase in registers, written when the workgroup ends: a store to memory inside the loop would count
  // among the operations the waits count, and one to LDS makes the compiler wait for every load in flight.
  unsigned long long phase_sum[4] = {0, 0, 0, 0}, stamp_first = wall_clock64(), stamp_last = stamp_first;
  const unsigned long long cycles_first = clock64();  // shader clock: its rate against the 100 MHz clock is the frequency
  int stamped = 0;
  unsigned long long section_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, section_last = 0;  // shader cycles per section of a K-tile
#define PRL_SECTION(k)                                        \
  {                                                           \
    const unsigned long long now = clock64();                 \
    section_sum[k] += now - section_last;                     \
    section_last = now;                                       \
  }
#define PRL_STAMP(i)                                  \
  {                                                   \
    const unsigned long long now = wall_clock64();    \
    phase_sum[i] += now - stamp_last;                 \
    stamp_last = now;                                 \
  }
#else
#define PRL_STAMP(i)
#define PRL_SECTION(k)
#endif
  const int stride = (int)gridDim.x;
  int item = (int)blockIdx.x, mt = 0, nt = 0, split = 0;
  while (item < a.total_items && !gemm_item(a, item, mt, nt, split)) item += stride;
  if (item >= a.total_items) return;
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;

  // ---- staging: per thread four 16-byte pieces of P and four of Q per K-tile
  // K-major P or Q: the quarter is [128 rows][64 k] with 128-byte rows; LDS slot (16 bytes) s = row * 8 + chunk
  // holds the row's chunk  chunk ^ ((row >> 1) & 7).
  // MN-major Q: the quarter is [64 k][128 columns] with 256-byte rows; slot s = krow * 16 + chunk holds the row's
  // chunk  chunk ^ (f(krow) << 1),  f(krow) = (krow & 3) | (((krow >> 3) & 1) << 2).
  // MN-major P: the quarter is two pieces of [64 k][64 columns] with 128-byte rows (the first 64 rows of the wave's
  // part of the tile, which phases 0 and 1 multiply, must be a piece of their own: the loop below waits for the other
  // piece only before phase 2); slot s = krow * 8 + chunk of a piece holds the row's chunk  chunk ^ (g(krow) << 1),
  // g(krow) = ((krow >> 3) & 1) | (((krow >> 1) & 1) << 1).
  const uint16_t* psrc[4];
  const uint16_t* qsrc[4];
  int nk = 0;
  auto set_sources = [&](int tid, int mt_, int nt_, int split_) {
    const int m0 = mt_ * kBM, n0 = nt_ * kBN, kt0 = split_ * a.ktiles_per_split;
    nk = a.ktiles - kt0;
    if (nk > a.ktiles_per_split) nk = a.ktiles_per_split;
    const int chunk = (tid & 7) ^ ((tid >> 4) & 7);
    if (P_KMAJOR) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // j = 2 * half + i
        int row = m0 + 64 * j + (tid >> 3);
        if (row > a.M - 1) row = a.M - 1;
        psrc[j] = a.P + (int64_t)row * a.ldp + (int64_t)kt0 * kBK + 8 * chunk;
      }
    } else {
      const int krow = tid >> 3;
      const int ch = (tid & 7) ^ ((((krow >> 3) & 1) | (((krow >> 1) & 1) << 1)) << 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // j = 2 * half + i; piece i holds columns 64 i .. 64 i + 63 of the half, 64 k rows
        int col = m0 + 64 * j + 8 * ch;
        if (col > a.M - 8) col = a.M - 8;
        psrc[j] = a.P + ((int64_t)kt0 * kBK + krow) * a.ldp + col;
      }
    }
    if (Q_KMAJOR) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int row = n0 + 64 * j + (tid >> 3);
        if (row > a.N - 1) row = a.N - 1;
        qsrc[j] = a.Q + (int64_t)row * a.ldq + (int64_t)kt0 * kBK + 8 * chunk;
      }
    } else {
      const int krow = tid >> 4;
      const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
      const int ch = (tid & 15) ^ (f << 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // j = 2 * half + i; piece i holds k rows 32 i .. 32 i + 31
        int col = n0 + 128 * (j >> 1) + 8 * ch;
        if (col > a.N - 8) col = a.N - 8;
        qsrc[j] = a.Q + ((int64_t)kt0 * kBK + 32 * (j & 1) + krow) * a.ldq + col;
      }
    }
  };
  const int64_t pstep = P_KMAJOR ? (int64_t)kBK : (int64_t)kBK * a.ldp;
  const int64_t qstep = Q_KMAJOR ? (int64_t)kBK : (int64_t)kBK * a.ldq;
  // LDS destination of piece j of this wave inside a stage (the hardware adds 16 * lane)
  char* const wave_dst = smem + wid * 1024;

  // piece j of P / of Q for this thread's next K-tile of that piece, into `stage`
  auto issue_p = [&](int stage, int j) {
    glds16(psrc[j], wave_dst + stage * kStageBytes + j * 8192);
    psrc[j] += pstep;
  };
  auto issue_q = [&](int stage, int j) {
    glds16(qsrc[j], wave_dst + stage * kStageBytes + 2 * kQuarterBytes + j * 8192);
    qsrc[j] += qstep;
  };
  // K-tile 0 of an item into stage 0 (8 loads per thread), and the first two pieces of Q of its K-tile 1 into stage 1
  auto issue_tile0 = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_p(0, j);
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_q(0, j);
  };
  constexpr int kAhead = 2;  // pieces of K-tile 1 requested with K-tile 0: half of Q
  auto issue_q1 = [&]() {
    if (nk > 1) {
      issue_q(1, 0);
      issue_q(1, 1);
    }
  };

  // ---- fragment addresses (bytes inside a stage)
  // P rows of this wave are the half `wr`; row tile rt adds 2048, k step ks flips bit 6 of the offset
  // (an MN-major P: row tile rt is the 16-column block rt & 3 of piece rt >> 2, at (p_off ^ (32 (rt & 3))) + 8192 (rt >> 2))
  const int sw = (lane >> 1) & 7;
  const int p_off = P_KMAJOR
                        ? wr * kQuarterBytes + (lane & 15) * 128 + (((lane >> 4)) ^ sw) * 16
                        : wr * kQuarterBytes + (8 * (lane >> 4) + ((lane & 15) >> 2)) * 128 +
                              ((((lane >> 4) & 1) | ((((lane & 15) >> 3) & 1) << 1)) * 32) + 8 * (lane & 3);
  int q_off[4];
  if (Q_KMAJOR) {
    q_off[0] = 2 * kQuarterBytes + (wc >> 1) * kQuarterBytes + ((wc & 1) * 64 + (lane & 15)) * 128 +
               (((lane >> 4)) ^ sw) * 16;
    q_off[1] = q_off[2] = q_off[3] = 0;
  } else {
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const int f = qq | ((g & 1) << 2);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
      q_off[ct] = 2 * kQuarterBytes + (wc >> 1) * kQuarterBytes + (8 * g + qq) * 256 +
                  (((4 * (wc & 1) + ct) ^ f) * 32) + 8 * pp;
  }
  auto p_frag = [&](int stage, int rt, int ks) -> bf16x8 {
    return lds_frag(smem + stage * kStageBytes + (p_off ^ (ks * 64)) + rt * 2048);
  };
  const unsigned smem_addr = (unsigned)(uintptr_t)(lds_void_ptr)smem;
  auto q_frag_k = [&](int stage, int ct, int ks) -> QFrag {
    const u32x4 v = *reinterpret_cast<const u32x4*>(smem + stage * kStageBytes + (q_off[0] ^ (ks * 64)) + ct * 2048);
    QFrag f;
    f.lo = (u32x2_t){v.x, v.y};
    f.hi = (u32x2_t){v.z, v.w};
    return f;
  };

  f32x4 acc[8][4];
  bf16x8 af[4][2];     // the 64 rows of P this phase multiplies: row tiles 4 h .. 4 h + 3
  QFrag aq[4][2];      // the same for an MN-major P (transposing reads)
  QFrag bx[2][2];      // Q columns 0 .. 31 of the wave (column tiles 0, 1)
  QFrag by[2][2];      // Q columns 32 .. 63 (column tiles 2, 3)

  auto load_a = [&](int stage, int h) {
    if (P_KMAJOR) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[r][ks] = p_frag(stage, 4 * h + r, ks);
    } else {
      const unsigned base = smem_addr + stage * kStageBytes;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned ar = base + (unsigned)((p_off ^ (r * 32)) + h * 8192);
        aq[r][0].lo = lds_read_tr<0>(ar);
        aq[r][0].hi = lds_read_tr<512>(ar);
        aq[r][1].lo = lds_read_tr<4096>(ar);
        aq[r][1].hi = lds_read_tr<4096 + 512>(ar);
      }
    }
  };
  // an MN-major P: behind the barrier that follows a load_a, before the first use of its fragments (see lds_read_tr)
  auto wait_a_reads = [&]() {
    if (!P_KMAJOR) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(aq[0][0].lo), "+v"(aq[0][0].hi), "+v"(aq[0][1].lo), "+v"(aq[0][1].hi), "+v"(aq[1][0].lo),
                     "+v"(aq[1][0].hi), "+v"(aq[1][1].lo), "+v"(aq[1][1].hi), "+v"(aq[2][0].lo), "+v"(aq[2][0].hi),
                     "+v"(aq[2][1].lo), "+v"(aq[2][1].hi), "+v"(aq[3][0].lo), "+v"(aq[3][0].hi), "+v"(aq[3][1].lo),
                     "+v"(aq[3][1].hi)
                   :
                   : "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // column tiles 2 hb, 2 hb + 1 of the wave's Q columns into b
  auto load_b = [&](int stage, int hb, QFrag (&b)[2][2]) {
    if (Q_KMAJOR) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) b[c][ks] = q_frag_k(stage, 2 * hb + c, ks);
    } else {
      const unsigned base = smem_addr + stage * kStageBytes;
      const unsigned a0 = base + q_off[2 * hb], a1 = base + q_off[2 * hb + 1];
      b[0][0].lo = lds_read_tr<0>(a0);
      b[0][0].hi = lds_read_tr<1024>(a0);
      b[0][1].lo = lds_read_tr<8192>(a0);
      b[0][1].hi = lds_read_tr<8192 + 1024>(a0);
      b[1][0].lo = lds_read_tr<0>(a1);
      b[1][0].hi = lds_read_tr<1024>(a1);
      b[1][1].lo = lds_read_tr<8192>(a1);
      b[1][1].hi = lds_read_tr<8192 + 1024>(a1);
    }
  };
  // behind the barrier that follows a load_b: the asm reads have landed before anything uses them
  auto wait_tr_reads = [&](QFrag (&b)[2][2]) {
    if (!Q_KMAJOR) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(b[0][0].lo), "+v"(b[0][0].hi), "+v"(b[0][1].lo), "+v"(b[0][1].hi), "+v"(b[1][0].lo),
                     "+v"(b[1][0].hi), "+v"(b[1][1].lo), "+v"(b[1][1].hi)
                   :
                   : "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // 16 MFMAs: rows 4 h .. 4 h + 3 against column tiles 2 hb, 2 hb + 1.  The operand that runs along the lanes of
  // the result is the one whose index is contiguous in the stored matrix.
  auto mma = [&](int h, int hb, const QFrag (&b)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          f32x4& d = acc[4 * h + r][2 * hb + c];
          const bf16x8 q = as_bf16x8(b[c][ks]);
          const bf16x8 p = P_KMAJOR ? af[r][ks] : as_bf16x8(aq[r][ks]);
          d = transposed_out(OUT) ? mfma16(p, q, d) : mfma16(q, p, d);
        }
    __builtin_amdgcn_s_setprio(0);
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- bf16 [M][N] write-back of the wave's 128 rows x 64 columns, 64 rows at a time through an 8 KB LDS image of
  // the wave's own in stage 1 (stage 0 is already receiving the next item's first K-tile), so that every store
  // instruction writes 8 whole 128-byte rows instead of 16 pieces of 32 bytes.  Row r of the image keeps its 16-byte
  // slot s at s ^ (r & 7): the 8-byte writes of 16 rows and the 16-byte reads of 8 slots of two rows then spread
  // over the banks.  `through_image` leaves half h of the tile in v, 16 bytes per lane and store; `store_rows`
  // writes v out.  The image is the wave's own and the LDS works a wave's operations in order, so the second half
  // needs no barrier behind the first.
  // (These take the lane as an argument: given a value the compiler cannot see through, it does not hoist their
  // address arithmetic, dozens of registers, out of the loop over work items and across the K loop.)
  auto through_image = [&](int lane, int h, int stage, u32x4 (&v)[8]) {
    const int lr = lane & 15, lq = lane >> 4;
    char* img = smem + stage * kStageBytes + wid * 8192;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int r = 16 * r4 + lr;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const f32x4 x = acc[4 * h + r4][ct];
        const u32x2_t o = {pack_bf16(x[0], x[1]), pack_bf16(x[2], x[3])};
        *reinterpret_cast<u32x2_t*>(img + r * 128 + (((2 * ct + (lq >> 1)) ^ (r & 7)) * 16) + 8 * (lq & 1)) = o;
      }
    }
    const char* src = img + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) * 16);
#pragma unroll
    for (int it = 0; it < 8; ++it) v[it] = *reinterpret_cast<const u32x4*>(src + it * 1024);
  };
  // Cacheable stores where the K loop is short (`a.cached_stores`: the logits of a narrow model, +2 % at H = 1024:
  // the write-backs of 256 units drain through the caches instead of meeting at the memory), non-temporal ones
  // otherwise (at H = 3584 and 5120 cacheable stores cost 6-7 %: the output lines push operand panels out).
  auto store_rows = [&](int lane, int m0, int n0, int h, bool inside, const u32x4 (&v)[8]) {
    uint16_t* C = reinterpret_cast<uint16_t*>(a.C);
    const int n = n0 + wc * 64 + 8 * (lane & 7);
    const int mw = m0 + wr * 128 + 64 * h + (lane >> 3);
    uint16_t* dst = C + (int64_t)mw * a.ldc + n;
    if (inside) {
      if (a.cached_stores) {
#pragma unroll
        for (int it = 0; it < 8; ++it) *reinterpret_cast<u32x4*>(dst + (int64_t)(8 * it) * a.ldc) = v[it];
      } else {
#pragma unroll
        for (int it = 0; it < 8; ++it)
          __builtin_nontemporal_store(v[it], reinterpret_cast<u32x4*>(dst + (int64_t)(8 * it) * a.ldc));
      }
    } else {
#pragma unroll
      for (int it = 0; it < 8; ++it)
        if (mw + 8 * it < a.M && n < a.N)
          __builtin_nontemporal_store(v[it], reinterpret_cast<u32x4*>(dst + (int64_t)(8 * it) * a.ldc));
    }
  };
  // the other output forms: straight from the accumulators (their tiles are a small share of these products)
  auto store_direct = [&](int lane, int m0, int n0, int split_, bool inside) {
    const int lr = lane & 15, lq = lane >> 4;
    if (OUT == kOutBf16Transposed) {
      uint16_t* C = reinterpret_cast<uint16_t*>(a.C);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int n = n0 + wc * 64 + 16 * ct + lr;
#pragma unroll
        for (int rt = 0; rt < 8; ++rt) {
          const int m = m0 + wr * 128 + 16 * rt + 4 * lq;
          if (inside || (n < a.N && m < a.M)) {
            const f32x4 x = acc[rt][ct];
            const u32x2_t o = {pack_bf16(x[0], x[1]), pack_bf16(x[2], x[3])};
            *reinterpret_cast<u32x2_t*>(C + (int64_t)n * a.ldc + m) = o;
          }
        }
      }
    } else if (OUT == kOutF32SlabTransposed) {  // a lane's four values are four consecutive m of one n: 16 bytes
      float* C = reinterpret_cast<float*>(a.C) + (int64_t)split_ * a.slab_stride;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int n = n0 + wc * 64 + 16 * ct + lr;
#pragma unroll
        for (int rt = 0; rt < 8; ++rt) {
          const int m = m0 + wr * 128 + 16 * rt + 4 * lq;
          if (inside || (n < a.N && m < a.M)) *reinterpret_cast<f32x4*>(C + (int64_t)n * a.ldc + m) = acc[rt][ct];
        }
      }
    } else {
      float* C = reinterpret_cast<float*>(a.C) + (int64_t)split_ * a.slab_stride;
#pragma unroll
      for (int rt = 0; rt < 8; ++rt) {
        const int m = m0 + wr * 128 + 16 * rt + lr;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const int n = n0 + wc * 64 + 16 * ct + 4 * lq;
          if (inside || (m < a.M && n < a.N)) *reinterpret_cast<f32x4*>(C + (int64_t)m * a.ldc + n) = acc[rt][ct];
        }
      }
    }
  };

  set_sources(tid, mt, nt, split);
  issue_tile0();
  issue_q1();
  int behind = 0;  // stores this wave issued behind the loads of K-tile 0 now in flight (0 or late_stores<OUT>())

  for (;;) {
    PRL_STAMP(0)
    const int m0 = mt * kBM, n0 = nt * kBN, my_split = split;
    const bool inside = m0 + kBM <= a.M && n0 + kBN <= a.N;
    // K-tile 0 has landed: everything but the two pieces of K-tile 1 and the stores issued behind K-tile 0's loads
    if (behind) {
      if (nk > 1) wait_vmcnt<kAhead + late_stores<OUT>()>(); else wait_vmcnt<late_stores<OUT>()>();
    } else {
      if (nk > 1) wait_vmcnt<kAhead>(); else wait_vmcnt<0>();
    }
    bar();
    PRL_STAMP(1)
    if (wid >= 4) bar();  // waves 4-7 run one section behind waves 0-3
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // One K-tile u out of stage S: four phases, each a load section, a barrier, an MFMA section, a barrier.
    //   phase 0: reads a0 (rows 0-63 of the wave) and b0 (columns 0-31);   multiplies (a0, b0)
    //   phase 1: reads b1;                                                 multiplies (a0, b1)
    //   phase 2: reads a1;                                                 multiplies (a1, b1)
    //   phase 3: reads nothing;                                            multiplies (a1, b0)
    // Every load section also issues two of the eight direct-to-LDS pieces a thread contributes per K-tile, as
    // early as the slot they overwrite is dead: Q of tile u+2 goes into THIS stage from phase 3 on (its Q slots
    // were last read in phase 1), the rest of Q and then P of tile u+1 into the other stage in phases 0-2.  A slot
    // counts as dead once every wave has waited for its last reads of it and a barrier has followed.  Pieces are
    // waited for with counted waits (a wave's vector-memory operations retire in issue order), always before a
    // barrier that every reader passes before its first read:
    //   end of phase 3's load section: Q of u+1 and the a0 pieces of P of u+1 (read from phase 0 of u+1 on); what
    //     may stay in flight: the a1 pieces of u+1 and the two pieces of Q of u+2 just issued;
    //   end of phase 1's load section: the a1 pieces of P of u (read in phase 2); in flight: all six younger
    //     pieces.  (In an item's K-tile 0 this wait also covers the previous item's stores, issued a write-back's
    //     time earlier; skipping it there makes the compiler peel the loop's first iteration and spill.)
#define PRL_GEMM_TILE(S)                                                            \
  {                                                                                 \
    const bool n1 = u + 1 < nk, n2 = u + 2 < nk;                                    \
    if (n1) { issue_q((S) ^ 1, 2); issue_q((S) ^ 1, 3); }                           \
    load_a((S), 0);                                                                 \
    load_b((S), 0, bx);                                                             \
    bar();                                                                          \
    PRL_SECTION(1)                                                                  \
    wait_a_reads();                                                                 \
    wait_tr_reads(bx);                                                              \
    mma(0, 0, bx);                                                                  \
    bar();                                                                          \
    PRL_SECTION(2)                                                                  \
    if (n1) { issue_p((S) ^ 1, 0); issue_p((S) ^ 1, 2); }                           \
    load_b((S), 1, by);                                                             \
    if (n1) wait_vmcnt<6>(); else wait_vmcnt<0>();                                  \
    bar();                                                                          \
    PRL_SECTION(3)                                                                  \
    wait_tr_reads(by);                                                              \
    mma(0, 1, by);                                                                  \
    bar();                                                                          \
    PRL_SECTION(4)                                                                  \
    if (n1) { issue_p((S) ^ 1, 1); issue_p((S) ^ 1, 3); }                           \
    load_a((S), 1);                                                                 \
    bar();                                                                          \
    PRL_SECTION(5)                                                                  \
    wait_a_reads();                                                                 \
    mma(1, 1, by);                                                                  \
    bar();                                                                          \
    PRL_SECTION(6)                                                                  \
    if (n2) {                                                                       \
      issue_q((S), 0);                                                              \
      issue_q((S), 1);                                                              \
      wait_vmcnt<4>();                                                              \
    } else if (n1) {                                                                \
      wait_vmcnt<2>();                                                              \
    }                                                                               \
    bar();                                                                          \
    PRL_SECTION(7)                                                                  \
    mma(1, 0, bx);                                                                  \
    bar();                                                                          \
    PRL_SECTION(0)                                                                  \
  }

#ifdef PRL_GEMM_STAMPS
    section_last = clock64();
#endif
    {
      int u = 0;
      for (; u + 1 < nk; ++u) {
        PRL_GEMM_TILE(0)
        ++u;
        PRL_GEMM_TILE(1)
      }
      if (u < nk) PRL_GEMM_TILE(0)
    }
#undef PRL_GEMM_TILE
    if (wid < 4) bar();
    PRL_STAMP(2)

    // ---- the item's write-back, with the next item's first loads already on their way
    int nitem = item + stride, nmt = 0, nnt = 0, nsp = 0;
    while (nitem < a.total_items && !gemm_item(a, nitem, nmt, nnt, nsp)) nitem += stride;
    const bool more = nitem < a.total_items;
    int tid_e = tid;  // the thread id again, opaque (see through_image)
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63;
    if (OUT == kOutBf16) {
      // The next item's first K-tile goes into stage 0 at once (every wave has passed its last reads of both
      // stages), a write-back's time ahead of its first use; the tile leaves through stage 1 meanwhile.
      if (more) {
        set_sources(tid_e, nmt, nnt, nsp);
        issue_tile0();
      }
      u32x4 half[8];
#ifdef PRL_GEMM_SKIP_STORE  // diagnostic build: what the kernel takes without its write-back (results are not written)
      const bool really = a.M < 0;
#else
      const bool really = true;
#endif
      through_image(lane_e, 0, 1, half);
      if (really) store_rows(lane_e, m0, n0, 0, inside, half);
      through_image(lane_e, 1, 1, half);
      if (really) store_rows(lane_e, m0, n0, 1, inside, half);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      bar();  // every wave has read its image back: stage 1 may be overwritten
      if (more) issue_q1();
      // behind the eight loads of the next K-tile 0: 16 stores (a tile inside the matrix; otherwise a number the
      // wait cannot know, and it waits for all of them) and the two loads just issued
      behind = inside && really ? late_stores<OUT>() : 0;
    } else {
      if (inside) {
        if (more) {
          set_sources(tid_e, nmt, nnt, nsp);
          issue_tile0();
          issue_q1();
        }
        store_direct(lane_e, m0, n0, my_split, true);
        behind = late_stores<OUT>();
      } else {
        store_direct(lane_e, m0, n0, my_split, false);
        if (more) {
          set_sources(tid_e, nmt, nnt, nsp);
          issue_tile0();
          issue_q1();
        }
        behind = 0;
      }
    }
    PRL_STAMP(3)
#ifdef PRL_GEMM_STAMPS
    ++stamped;
#endif
    if (!more) break;
    item = nitem;
    mt = nmt;
    nt = nnt;
    split = nsp;
  }
#ifdef PRL_GEMM_STAMPS
  if (a.stamps && threadIdx.x == 0) {
    unsigned long long* out = a.stamps + (size_t)blockIdx.x * 8;
    for (int i = 0; i < 4; ++i) out[i] = phase_sum[i];
    out[4] = (unsigned long long)stamped;
    out[5] = stamp_first;
    out[6] = stamp_last;
    out[7] = clock64() - cycles_first;
    if (a.stamps2) for (int i = 0; i < 8; ++i) a.stamps2[(size_t)blockIdx.x * 8 + i] = section_sum[i];
  }
#endif
}

// out[c][r] = in[r][c] for a bf16 [rows][cols] matrix: 64 x 64 blocks through LDS, 16-byte loads and stores.
// Blocks of rows along grid x (the token count, which has no bound to speak of), blocks of columns along grid y.
__global__ void __launch_bounds__(256) k_transpose_bf16(const uint16_t* __restrict__ in, int64_t ld_in,
                                                       uint16_t* __restrict__ out, int64_t ld_out, int rows, int cols) {
  __shared__ uint16_t tile[64][72];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int id = it * 256 + tid, r = id >> 3, ch = id & 7;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r0 + r < rows && c0 + 8 * ch < cols)
      v = *reinterpret_cast<const u32x4*>(in + (int64_t)(r0 + r) * ld_in + c0 + 8 * ch);
    *reinterpret_cast<u32x4*>(&tile[r][8 * ch]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int id = it * 256 + tid, c = id >> 3, ch = id & 7;
    if (c0 + c < cols && r0 + 8 * ch < rows) {
      uint16_t e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = tile[8 * ch + j][c];
      u32x4 v = {(uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16),
                 (uint32_t)e[4] | ((uint32_t)e[5] << 16), (uint32_t)e[6] | ((uint32_t)e[7] << 16)};
      *reinterpret_cast<u32x4*>(out + (int64_t)(c0 + c) * ld_out + r0 + 8 * ch) = v;
    }
  }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static GemmArgs make_args(const void* P, int64_t ldp, const void* Q, int64_t ldq, void* C, int64_t ldc, int64_t M,
                          int64_t N, int64_t K, int nsplit) {
  GemmArgs a;
  a.P = reinterpret_cast<const uint16_t*>(P);
  a.Q = reinterpret_cast<const uint16_t*>(Q);
  a.C = C;
  a.ldp = ldp;
  a.ldq = ldq;
  a.ldc = ldc;
  a.slab_stride = M * N;
  a.M = (int)M;
  a.N = (int)N;
  a.ktiles = (int)(K / kBK);
  a.ktiles_per_split = (a.ktiles + nsplit - 1) / nsplit;
  a.nsplit = (a.ktiles + a.ktiles_per_split - 1) / a.ktiles_per_split;  // no split without a K-tile
  a.mtiles = (int)((M + kBM - 1) / kBM);
  a.ntiles = (int)((N + kBN - 1) / kBN);
  a.valid_items = a.mtiles * a.ntiles * a.nsplit;
  a.total_items = 256 * ((((a.valid_items + 31) >> 5) + 7) >> 3);
  a.stamps = nullptr;
  a.stamps2 = nullptr;
  a.cached_stores = a.ktiles_per_split <= 16;  // (only the forward product at H <= 1024 stores bf16 tiles that often)
  return a;
}
// One workgroup per compute unit of an MI355X, whatever the product's size: the kernel walks over its work items,
// which `gemm_item` deals out in rounds of 256 (`total_items`), and a workgroup without a valid item ends at once.
static unsigned gemm_grid(const GemmArgs&) { return 256u; }

// K splits for a product of `tiles` output tiles over `ktiles` K-tiles whose output is small and whose K is long
// (the input gradient of the LM head; weight gradients).  The chip works the (tile, split) pairs off 32 per XCD and
// round, so s splits take ceil(ceil(tiles * s / 32) / 8) rounds of ceil(ktiles / s) K-tiles; every split beyond the
// first also costs an fp32 slab of the output written and read back, which is priced in K-tile times PER SLAB of 256
// tiles (`slab_cost`), not relative to the product: a short K does not make a slab cheaper.  Measured: 47.5 for the
// LM head's input gradient (2 % of its 2374 K-tiles per split), 20 for the weight gradients of a decoder block's
// layers over a 16 K-token row.  The cheapest count up to 16 that leaves a split at least 8 K-tiles is taken.
static int k_splits(int64_t tiles, int64_t ktiles, double slab_cost) {
  double best = 1e30;
  int best_s = 1;
  for (int s = 1; s <= 16 && s * 8 <= ktiles; ++s) {
    const int64_t rounds = ((tiles * s + 31) / 32 + 7) / 8;
    const double cost = (double)rounds * (double)((ktiles + s - 1) / s) + (s - 1) * slab_cost * (double)tiles / 256.0;
    if (cost < best * 0.995) {
      best = cost;
      best_s = s;
    }
  }
  return best_s;
}

static int dx_splits(int64_t T, int64_t H, int64_t V) {
  const int64_t tiles = ((T + kBM - 1) / kBM) * ((H + kBN - 1) / kBN);
  const int64_t ktiles = V / kBK;
  if (const char* e = getenv("PRL_LM_HEAD_DX_SPLITS")) {  // measurements only: a fixed split count
    const int s = atoi(e);
    if (s >= 1 && s <= 16 && (int64_t)s * 2 <= ktiles) return s;
  }
  return k_splits(tiles, ktiles, 47.5);
}

// The weight gradient's two forms.  Both operands read along the tokens as they lie (transposing LDS reads for
// both): nothing is copied, but a K-tile takes 4-5 % longer than with a K-major first operand.  Or a transposed
// copy of the narrower operand in memory first, whose cost goes with T x min(H, V) while the product's goes with
// T x H x V: measured at 16384 tokens the copy pays from max(H, V) ~ 16384 on (the LM head, a 7B model's gate + up).
// PRL_LM_HEAD_DW_FORM=both / transpose (measurements only) takes one form everywhere.
static bool dw_reads_both_along_tokens(int64_t H, int64_t V) {
  if (const char* e = getenv("PRL_LM_HEAD_DW_FORM")) {
    if (strcmp(e, "transpose") == 0) return false;
    if (strcmp(e, "both") == 0) return true;
  }
  return (H > V ? H : V) <= 16384;
}

static int dw_splits(int64_t T, int64_t H, int64_t V) {
  const int64_t tiles = ((H + kBM - 1) / kBM) * ((V + kBN - 1) / kBN);
  const int64_t ktiles = T / kBK;
  if (const char* e = getenv("PRL_LM_HEAD_DW_SPLITS")) {  // measurements only: a fixed split count
    const int s = atoi(e);
    if (s >= 1 && s <= 16 && (int64_t)s * 2 <= ktiles) return s;
  }
  return k_splits(tiles, ktiles, 20.0);
}

}  // namespace prl

using namespace prl;

#ifdef PRL_GEMM_STAMPS
static unsigned long long* g_stamps = nullptr;
static unsigned long long* g_stamps2 = nullptr;
extern "C" void prl_lm_head_set_stamps(void* p) { g_stamps = (unsigned long long*)p; }
extern "C" void prl_lm_head_set_section_stamps(void* p) { g_stamps2 = (unsigned long long*)p; }
#endif

extern "C" int prl_lm_head_supported(int dtype, int64_t T, int64_t H, int64_t V) {
  if (!(dtype == PRL_BF16 && T >= 1 && H >= kBK && V >= 8 && H % kBK == 0 && V % kBK == 0 && T % kBK == 0 &&
        T < (1 << 30) && V < (1 << 30) && H < (1 << 22)))  // (H / 64 is a grid's y extent in the transposition)
    return 0;
  // the kernels count (tile, K split) pairs in 32 bits: at most 2^24 tiles per product (16 splits at most)
  const int64_t tt = (T + kBM - 1) / kBM, th = (H + kBM - 1) / kBM, tv = (V + kBM - 1) / kBM;
  return tt * tv <= (1 << 24) && tt * th <= (1 << 24) && th * tv <= (1 << 24);
}

extern "C" size_t prl_lm_head_workspace_bytes(int64_t T, int64_t H, int64_t V) {
  if (T <= 0 || H <= 0 || V <= 0) return 0;
  const int s = dx_splits(T, H, V), sw = dw_splits(T, H, V);
  const size_t slabs = s > 1 ? (size_t)s * (size_t)T * (size_t)H * sizeof(float) : 0;
  // the weight gradient: the narrower of hidden and dlogits transposed, and behind it the fp32 slabs of a K split
  const size_t xt = (size_t)T * (size_t)(V < H ? V : H) * sizeof(uint16_t) +
                    (sw > 1 ? (size_t)sw * (size_t)V * (size_t)H * sizeof(float) : 0);
  return slabs > xt ? slabs : xt;
}

extern "C" int prl_lm_head_fwd(const void* hidden, int64_t hidden_stride, const void* weight, int64_t weight_stride,
                               void* logits, int64_t logits_stride, int64_t T, int64_t H, int64_t V, void* stream) {
  clear_error();
  PRL_REQUIRE(hidden && weight && logits, "prl_lm_head_fwd: null pointer");
  PRL_REQUIRE(prl_lm_head_supported(PRL_BF16, T, H, V), "prl_lm_head_fwd: unsupported shape T=%lld H=%lld V=%lld",
              (long long)T, (long long)H, (long long)V);
  PRL_REQUIRE(aligned16(hidden) && aligned16(weight) && aligned16(logits) && hidden_stride % 8 == 0 &&
                  weight_stride % 8 == 0 && logits_stride % 4 == 0 && hidden_stride >= H && weight_stride >= H &&
                  logits_stride >= V && logits_stride % 8 == 0,
              "prl_lm_head_fwd: pointers must be 16-byte aligned and row strides multiples of 8 elements");
  GemmArgs a = make_args(hidden, hidden_stride, weight, weight_stride, logits, logits_stride, T, V, H, 1);
#ifdef PRL_GEMM_STAMPS
  a.stamps = g_stamps;
  a.stamps2 = g_stamps2;
#endif
  hipLaunchKernelGGL((k_lm_gemm<true, true, kOutBf16>), dim3(gemm_grid(a)), dim3(kGemmThreads), 0, (hipStream_t)stream, a);
  PRL_CHECK_LAUNCH("k_lm_gemm (forward)");
  return PRL_OK;
}

extern "C" int prl_sum_parts(const void* parts, int parts_dtype, int64_t part_stride, int32_t nparts, int64_t n,
                             void* out, void* stream);

extern "C" int prl_lm_head_bwd_input(const void* dlogits, int64_t dlogits_stride, const void* weight,
                                     int64_t weight_stride, void* dhidden, int64_t dhidden_stride, int64_t T, int64_t H,
                                     int64_t V, void* workspace, size_t workspace_bytes, void* stream) {
  clear_error();
  PRL_REQUIRE(dlogits && weight && dhidden, "prl_lm_head_bwd_input: null pointer");
  PRL_REQUIRE(prl_lm_head_supported(PRL_BF16, T, H, V), "prl_lm_head_bwd_input: unsupported shape T=%lld H=%lld V=%lld",
              (long long)T, (long long)H, (long long)V);
  PRL_REQUIRE(aligned16(dlogits) && aligned16(weight) && aligned16(dhidden) && dlogits_stride % 8 == 0 &&
                  weight_stride % 8 == 0 && dhidden_stride % 8 == 0 && dlogits_stride >= V && weight_stride >= H &&
                  dhidden_stride >= H,
              "prl_lm_head_bwd_input: pointers must be 16-byte aligned and row strides multiples of 8 elements");
  const int s = dx_splits(T, H, V);
  if (s == 1) {
    const GemmArgs a = make_args(dlogits, dlogits_stride, weight, weight_stride, dhidden, dhidden_stride, T, H, V, 1);
    hipLaunchKernelGGL((k_lm_gemm<true, false, kOutBf16>), dim3(gemm_grid(a)), dim3(kGemmThreads), 0, (hipStream_t)stream,
                       a);
    PRL_CHECK_LAUNCH("k_lm_gemm (input gradient)");
    return PRL_OK;
  }
  PRL_REQUIRE(dhidden_stride == H, "prl_lm_head_bwd_input: a K-split product needs a contiguous dhidden");
  const size_t need = (size_t)s * (size_t)T * (size_t)H * sizeof(float);
  PRL_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= need,
              "prl_lm_head_bwd_input: workspace of %zu bytes needed (prl_lm_head_workspace_bytes), got %zu", need,
              workspace_bytes);
  const GemmArgs a = make_args(dlogits, dlogits_stride, weight, weight_stride, workspace, H, T, H, V, s);
  hipLaunchKernelGGL((k_lm_gemm<true, false, kOutF32Slab>), dim3(gemm_grid(a)), dim3(kGemmThreads), 0, (hipStream_t)stream,
                     a);
  PRL_CHECK_LAUNCH("k_lm_gemm (input gradient, K split)");
  return prl_sum_parts(workspace, PRL_F32, T * H, a.nsplit, T * H, dhidden, stream);
}

extern "C" int prl_lm_head_bwd_weight(const void* dlogits, int64_t dlogits_stride, const void* hidden,
                                      int64_t hidden_stride, void* dweight, int64_t dweight_stride, int64_t T,
                                      int64_t H, int64_t V, void* workspace, size_t workspace_bytes, void* stream) {
  clear_error();
  PRL_REQUIRE(dlogits && hidden && dweight, "prl_lm_head_bwd_weight: null pointer");
  PRL_REQUIRE(prl_lm_head_supported(PRL_BF16, T, H, V),
              "prl_lm_head_bwd_weight: unsupported shape T=%lld H=%lld V=%lld", (long long)T, (long long)H,
              (long long)V);
  PRL_REQUIRE(aligned16(dlogits) && aligned16(hidden) && aligned16(dweight) && dlogits_stride % 8 == 0 &&
                  hidden_stride % 8 == 0 && dweight_stride % 4 == 0 && dlogits_stride >= V && hidden_stride >= H &&
                  dweight_stride >= H,
              "prl_lm_head_bwd_weight: pointers must be 16-byte aligned and row strides multiples of 8 elements");
  // One operand has to be read along the tokens, so one of the two is transposed into the workspace first: the
  // narrower one.  hidden^T [H][T]: dweight^T = hidden^T . dlogits, stored transposed (the LM head: V >> H).
  // dlogits^T [V][T]: dweight = dlogits^T . hidden, stored as it comes (a layer that narrows: V < H).
  const int s = dw_splits(T, H, V);
  const bool narrow_out = V < H;
  const size_t xt_bytes = (size_t)T * (size_t)(narrow_out ? V : H) * sizeof(uint16_t);
  const size_t need = xt_bytes + (s > 1 ? (size_t)s * (size_t)V * (size_t)H * sizeof(float) : 0);
  PRL_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= need,
              "prl_lm_head_bwd_weight: workspace of %zu bytes needed (prl_lm_head_workspace_bytes), got %zu", need,
              workspace_bytes);
  PRL_REQUIRE(s == 1 || dweight_stride == H, "prl_lm_head_bwd_weight: a K-split product needs a contiguous dweight");
  PRL_REQUIRE(!narrow_out || dweight_stride % 8 == 0,
              "prl_lm_head_bwd_weight: dweight's row stride must be a multiple of 8 elements when V < H");
  float* slabs = reinterpret_cast<float*>(static_cast<char*>(workspace) + xt_bytes);
  if (dw_reads_both_along_tokens(H, V) && dweight_stride % 8 == 0) {
    // both operands as they lie, [T][V] and [T][H], read along the tokens by transposing LDS reads: nothing is
    // transposed in memory, and dweight [V][H] is stored as it comes
    if (s > 1) {
      const GemmArgs a = make_args(dlogits, dlogits_stride, hidden, hidden_stride, slabs, H, V, H, T, s);
      hipLaunchKernelGGL((k_lm_gemm<false, false, kOutF32Slab>), dim3(gemm_grid(a)), dim3(kGemmThreads), 0,
                         (hipStream_t)stream, a);
      PRL_CHECK_LAUNCH("k_lm_gemm (weight gradient, K split)");
      return prl_sum_parts(slabs, PRL_F32, V * H, a.nsplit, V * H, dweight, stream);
    }
    const GemmArgs a = make_args(dlogits, dlogits_stride, hidden, hidden_stride, dweight, dweight_stride, V, H, T, 1);
    hipLaunchKernelGGL((k_lm_gemm<false, false, kOutBf16>), dim3(gemm_grid(a)), dim3(kGemmThreads), 0,
                       (hipStream_t)stream, a);
    PRL_CHECK_LAUNCH("k_lm_gemm (weight gradient)");
    return PRL_OK;
  }
  if (narrow_out) {
    hipLaunchKernelGGL(k_transpose_bf16, dim3((unsigned)((T + 63) / 64), (unsigned)((V + 63) / 64)), dim3(256), 0,
                       (hipStream_t)stream, reinterpret_cast<const uint16_t*>(dlogits), dlogits_stride,
                       reinterpret_cast<uint16_t*>(workspace), T, (int)T, (int)V);
    PRL_CHECK_LAUNCH("k_transpose_bf16");
    if (s > 1) {
      const GemmArgs a = make_args(workspace, T, hidden, hidden_stride, slabs, H, V, H, T, s);
      hipLaunchKernelGGL((k_lm_gemm<true, false, kOutF32Slab>), dim3(gemm_grid(a)), dim3(kGemmThreads), 0,
                         (hipStream_t)stream, a);
      PRL_CHECK_LAUNCH("k_lm_gemm (weight gradient of a narrowing layer, K split)");
      return prl_sum_parts(slabs, PRL_F32, V * H, a.nsplit, V * H, dweight, stream);
    }
    const GemmArgs a = make_args(workspace, T, hidden, hidden_stride, dweight, dweight_stride, V, H, T, 1);
    hipLaunchKernelGGL((k_lm_gemm<true, false, kOutBf16>), dim3(gemm_grid(a)), dim3(kGemmThreads), 0, (hipStream_t)stream,
                       a);
    PRL_CHECK_LAUNCH("k_lm_gemm (weight gradient of a narrowing layer)");
    return PRL_OK;
  }
  hipLaunchKernelGGL(k_transpose_bf16, dim3((unsigned)((T + 63) / 64), (unsigned)((H + 63) / 64)), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<const uint16_t*>(hidden), hidden_stride,
                     reinterpret_cast<uint16_t*>(workspace), T, (int)T, (int)H);
  PRL_CHECK_LAUNCH("k_transpose_bf16");
  if (s > 1) {  // a small weight and many tokens: fp32 slabs of dweight [V][H], one per range of the tokens
    GemmArgs a = make_args(workspace, T, dlogits, dlogits_stride, slabs, V, H, V, T, s);
    a.ldc = H;  // (the slab is stored transposed: rows of dweight)
    hipLaunchKernelGGL((k_lm_gemm<true, false, kOutF32SlabTransposed>), dim3(gemm_grid(a)), dim3(kGemmThreads), 0,
                       (hipStream_t)stream, a);
    PRL_CHECK_LAUNCH("k_lm_gemm (weight gradient, K split)");
    return prl_sum_parts(slabs, PRL_F32, V * H, a.nsplit, V * H, dweight, stream);
  }
  const GemmArgs a = make_args(workspace, T, dlogits, dlogits_stride, dweight, dweight_stride, H, V, T, 1);
  hipLaunchKernelGGL((k_lm_gemm<true, false, kOutBf16Transposed>), dim3(gemm_grid(a)), dim3(kGemmThreads), 0,
                     (hipStream_t)stream, a);
  PRL_CHECK_LAUNCH("k_lm_gemm (weight gradient)");
  return PRL_OK;
}
