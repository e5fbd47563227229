// Store-only experiments behind the row kernel's write rate (what a load-all / store-all structure can reach,
// and why ~256 concurrently written rows stop at 5.6 TB/s where one front-to-back sweep reaches 6.8): eight
// small experiments, one per number.
//     hipcc --offload-arch=gfx950 -O3 -o tools/bin/store_bench tools/store_bench.hip
//     tools/bin/store_bench N

//   1: the row kernel's store shape: workgroup size, cache hint, chunk-major against wave-contiguous layout
//   2: which plain fill pattern reaches the runtime memset's rate
//   3: the row shape with the number of stores a wave keeps in flight limited, next to a 16-bytes-per-thread fill
//   4: does a compact window of concurrently written addresses decide the rate?
//   5: one store per thread and exit, workgroups scattered over S sequential streams; blocks per workgroup
//   6: many streams, every workgroup's 4 KB written as pieces a stride apart inside its stream
//   7: the row shape under other maps from (round, wave, quarter-wave, lane) to the address
//   8: the row shape: does the order in which a unit's 16 waves emit their stores matter?
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)


// ------------------------------------------------------------------ experiment 1: the row kernel's store shape: workgroup size, cache hint, chunk-major against wave-contiguous layout
namespace exp1 {
constexpr int kChunks = 19;
constexpr long kRowVec = 18992;
// LAYOUT 0: chunk-major (chunk j of the workgroup = THREADS consecutive vectors); 1: wave-contiguous (a wave owns 19 KB)
template <int THREADS, bool NT, int LAYOUT>
__global__ __launch_bounds__(THREADS) void k_store(u32x4* __restrict__ out, long rows) {
  constexpr int kPieces = 1024 / THREADS;
  constexpr long kPieceVec = kRowVec / kPieces;
  const long piece = blockIdx.x;
  const long base = (piece / kPieces) * kRowVec + (piece % kPieces) * kPieceVec;
#pragma unroll
  for (int j = 0; j < kChunks; ++j) {
    long u;
    if (LAYOUT == 0) u = (long)j * THREADS + threadIdx.x;
    else u = (long)(threadIdx.x / 64) * (kChunks * 64) + j * 64 + (threadIdx.x % 64);
    u32x4 v = u32x4{(unsigned)u, 1u, 2u, (unsigned)piece};
    if (u < kPieceVec) { if (NT) __builtin_nontemporal_store(v, out + base + u); else out[base + u] = v; }
  }
}
template <bool NT, int UNR>
__global__ __launch_bounds__(256) void k_fill_stride(u32x4* __restrict__ out, long n_vec) {
  const long chunk = (long)UNR * 256;
  for (long base = (long)blockIdx.x * chunk; base < n_vec; base += (long)gridDim.x * chunk) {
    long i = base + threadIdx.x;
#pragma unroll
    for (int q = 0; q < UNR; ++q) if (i + q * 256 < n_vec) { u32x4 v = u32x4{(unsigned)i, 1u, 2u, 3u}; if (NT) __builtin_nontemporal_store(v, out + i + q * 256); else out[i + q * 256] = v; }
  }
}
template <bool NT, int UNR>
__global__ __launch_bounds__(256) void k_read_stride(const u32x4* __restrict__ in, u32x4* __restrict__ out, long n_vec) {
  const long chunk = (long)UNR * 256;
  unsigned acc = 0;
  for (long base = (long)blockIdx.x * chunk; base < n_vec; base += (long)gridDim.x * chunk) {
    long i = base + threadIdx.x;
    u32x4 v[UNR];
#pragma unroll
    for (int q = 0; q < UNR; ++q) v[q] = (i + q * 256 < n_vec) ? (NT ? __builtin_nontemporal_load(in + i + q * 256) : in[i + q * 256]) : u32x4{0,0,0,0};
#pragma unroll
    for (int q = 0; q < UNR; ++q) acc ^= v[q].x ^ v[q].y ^ v[q].z ^ v[q].w;
  }
  if (acc == 0x12345u) out[threadIdx.x] = u32x4{acc, 0, 0, 0};
}
int run() {
  const long rows = 2048, bytes = rows * kRowVec * 16;
  const int sets = 6, rounds = 9;
  std::vector<u32x4*> out(sets);
  for (int i = 0; i < sets; ++i) { CK(hipMalloc(&out[i], bytes)); CK(hipMemset(out[i], i + 1, bytes)); }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const char* names[] = {"1024 thr NT chunk-major", "1024 thr plain chunk-major", "1024 thr NT wave-contig", "256 thr NT chunk-major (quarter rows)",
                         "512 thr NT chunk-major (half rows)", "fill stride NT u4 g2048", "fill stride plain u4 g2048", "fill stride NT u4 g8192", "hipMemsetAsync",
                         "read stride NT u4 g2048", "read stride plain u4 g2048", "read stride NT u8 g4096"};
  const int nv = 12;
  std::vector<float> t[nv];
  for (int r = 0; r < rounds; ++r) for (int k = 0; k < nv; ++k) {
    const int v = (k * 5 + r * 3) % nv;
    CK(hipEventRecord(a));
    for (int i = 0; i < sets; ++i) {
      switch (v) {
        case 0: hipLaunchKernelGGL((k_store<1024, true, 0>), dim3(rows), dim3(1024), 0, 0, out[i], rows); break;
        case 1: hipLaunchKernelGGL((k_store<1024, false, 0>), dim3(rows), dim3(1024), 0, 0, out[i], rows); break;
        case 2: hipLaunchKernelGGL((k_store<1024, true, 1>), dim3(rows), dim3(1024), 0, 0, out[i], rows); break;
        case 3: hipLaunchKernelGGL((k_store<256, true, 0>), dim3(rows * 4), dim3(256), 0, 0, out[i], rows); break;
        case 4: hipLaunchKernelGGL((k_store<512, true, 0>), dim3(rows * 2), dim3(512), 0, 0, out[i], rows); break;
        case 5: hipLaunchKernelGGL((k_fill_stride<true, 4>), dim3(2048), dim3(256), 0, 0, out[i], rows * kRowVec); break;
        case 6: hipLaunchKernelGGL((k_fill_stride<false, 4>), dim3(2048), dim3(256), 0, 0, out[i], rows * kRowVec); break;
        case 7: hipLaunchKernelGGL((k_fill_stride<true, 4>), dim3(8192), dim3(256), 0, 0, out[i], rows * kRowVec); break;
        case 8: CK(hipMemsetAsync(out[i], 7, bytes, 0)); break;
        case 9: hipLaunchKernelGGL((k_read_stride<true, 4>), dim3(2048), dim3(256), 0, 0, out[i], out[(i + 1) % sets], rows * kRowVec); break;
        case 10: hipLaunchKernelGGL((k_read_stride<false, 4>), dim3(2048), dim3(256), 0, 0, out[i], out[(i + 1) % sets], rows * kRowVec); break;
        case 11: hipLaunchKernelGGL((k_read_stride<true, 8>), dim3(4096), dim3(256), 0, 0, out[i], out[(i + 1) % sets], rows * kRowVec); break;
      }
    }
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (r > 0) t[v].push_back(ms / sets);
  }
  for (int v = 0; v < nv; ++v) {
    std::sort(t[v].begin(), t[v].end());
    const float med = t[v][t[v].size() / 2];
    printf("%-40s median %.4f ms  min %.4f ms  %6.0f GB/s\n", names[v], med, t[v][0], bytes / med / 1e6);
  }
  return 0;
}

}  // namespace exp1

// ------------------------------------------------------------------ experiment 2: which plain fill pattern reaches the runtime memset's rate
namespace exp2 {
// Which store pattern reaches what hipMemsetAsync reaches (6.7 TB/s on buffers that are not cache-resident)?
// every thread writes one 16-byte vector per trip, the grid strides over the buffer
template <bool NT, bool CONST>
__global__ __launch_bounds__(256) void k_fill1(u32x4* __restrict__ out, long n_vec) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += stride) {
    u32x4 v = CONST ? u32x4{7u, 7u, 7u, 7u} : u32x4{(unsigned)i, 1u, 2u, (unsigned)(i >> 7)};
    if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
  }
}
template <bool CONST>
__global__ __launch_bounds__(256) void k_fill8(u32x2* __restrict__ out, long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    u32x2 v = CONST ? u32x2{7u, 7u} : u32x2{(unsigned)i, (unsigned)(i >> 7)};
    out[i] = v;
  }
}
int run() {
  const long n_vec = 2048L * 18992, bytes = n_vec * 16;
  const int sets = 6, rounds = 7;
  std::vector<u32x4*> out(sets);
  for (int i = 0; i < sets; ++i) { CK(hipMalloc(&out[i], bytes)); CK(hipMemset(out[i], i + 1, bytes)); }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  struct V { std::string name; int kind; int grid; };
  std::vector<V> vs;
  for (int g : {1024, 2048, 4096, 8192, 16384, 65536}) {
    vs.push_back({"plain varying g" + std::to_string(g), 0, g});
    vs.push_back({"plain const   g" + std::to_string(g), 1, g});
  }
  for (int g : {2048, 8192}) { vs.push_back({"NT varying    g" + std::to_string(g), 2, g}); vs.push_back({"NT const      g" + std::to_string(g), 3, g}); }
  for (int g : {4096, 16384}) { vs.push_back({"8-byte varying g" + std::to_string(g), 4, g}); vs.push_back({"8-byte const  g" + std::to_string(g), 5, g}); }
  vs.push_back({"hipMemsetAsync (byte 7)", 6, 0});
  vs.push_back({"hipMemsetD32Async", 7, 0});
  std::vector<std::vector<float>> t(vs.size());
  for (int r = 0; r < rounds; ++r) for (size_t k = 0; k < vs.size(); ++k) {
    const size_t v = (k * 7 + r * 3) % vs.size();
    CK(hipEventRecord(a));
    for (int i = 0; i < sets; ++i) {
      const int g = vs[v].grid;
      switch (vs[v].kind) {
        case 0: hipLaunchKernelGGL((k_fill1<false, false>), dim3(g), dim3(256), 0, 0, out[i], n_vec); break;
        case 1: hipLaunchKernelGGL((k_fill1<false, true>), dim3(g), dim3(256), 0, 0, out[i], n_vec); break;
        case 2: hipLaunchKernelGGL((k_fill1<true, false>), dim3(g), dim3(256), 0, 0, out[i], n_vec); break;
        case 3: hipLaunchKernelGGL((k_fill1<true, true>), dim3(g), dim3(256), 0, 0, out[i], n_vec); break;
        case 4: hipLaunchKernelGGL((k_fill8<false>), dim3(g), dim3(256), 0, 0, (u32x2*)out[i], n_vec * 2); break;
        case 5: hipLaunchKernelGGL((k_fill8<true>), dim3(g), dim3(256), 0, 0, (u32x2*)out[i], n_vec * 2); break;
        case 6: CK(hipMemsetAsync(out[i], 7, bytes, 0)); break;
        case 7: CK(hipMemsetD32Async((hipDeviceptr_t)out[i], 0x01020304, bytes / 4, 0)); break;
      }
    }
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (r > 0) t[v].push_back(ms / sets);
  }
  for (size_t v = 0; v < vs.size(); ++v) {
    std::sort(t[v].begin(), t[v].end());
    const float med = t[v][t[v].size() / 2];
    printf("%-28s median %.4f ms  min %.4f ms  %6.0f GB/s\n", vs[v].name.c_str(), med, t[v][0], bytes / med / 1e6);
  }
  return 0;
}

}  // namespace exp2

// ------------------------------------------------------------------ experiment 3: the row shape with the number of stores a wave keeps in flight limited, next to a 16-bytes-per-thread fill
namespace exp3 {
// Store-only kernels in the row kernel's shape (one workgroup of 1024 threads writes a 152 K-entry bf16 row as
// 19 x 16 bytes per lane) with the number of stores a wave may have in flight limited, next to a fill that
// writes 16 bytes per thread and ends (what the runtime's memset does).
constexpr int kChunks = 19;
constexpr long kRowVec = 18992;
template <int INFLIGHT, bool NT>
__global__ __launch_bounds__(1024) void k_store(u32x4* __restrict__ out, long rows) {
  const long base = (long)blockIdx.x * kRowVec;
#pragma unroll
  for (int j = 0; j < kChunks; ++j) {
    const long u = (long)j * 1024 + threadIdx.x;
    u32x4 v = u32x4{(unsigned)u, 1u, 2u, (unsigned)blockIdx.x};
    if (u < kRowVec) { if (NT) __builtin_nontemporal_store(v, out + base + u); else out[base + u] = v; }
    if (INFLIGHT == 0) __builtin_amdgcn_s_waitcnt(0x0070 | 0xc000 | 0x0f00);      // vmcnt(0), others untouched
    else if (INFLIGHT == 1) __builtin_amdgcn_s_waitcnt(0x0071 | 0xc000 | 0x0f00);
    else if (INFLIGHT == 2) __builtin_amdgcn_s_waitcnt(0x0072 | 0xc000 | 0x0f00);
    else if (INFLIGHT == 4) __builtin_amdgcn_s_waitcnt(0x0074 | 0xc000 | 0x0f00);
    else if (INFLIGHT == 8) __builtin_amdgcn_s_waitcnt(0x0078 | 0xc000 | 0x0f00);
  }
}
__global__ __launch_bounds__(256) void k_fill_once(u32x4* __restrict__ out, long n_vec) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n_vec) out[i] = u32x4{(unsigned)i, 1u, 2u, (unsigned)(i >> 7)};
}
__global__ __launch_bounds__(1024) void k_fill_once_1024(u32x4* __restrict__ out, long n_vec) {
  const long i = (long)blockIdx.x * 1024 + threadIdx.x;
  if (i < n_vec) out[i] = u32x4{(unsigned)i, 1u, 2u, (unsigned)(i >> 7)};
}
// a row per workgroup, but only `LIVE` of the 16 waves store at a time (turn by turn through a barrier)
template <int GROUPS>
__global__ __launch_bounds__(1024) void k_store_turns(u32x4* __restrict__ out, long rows) {
  const long base = (long)blockIdx.x * kRowVec;
  const int wave = threadIdx.x / 64;
  for (int turn = 0; turn < GROUPS; ++turn) {
    if (wave % GROUPS == turn) {
#pragma unroll
      for (int j = 0; j < kChunks; ++j) {
        const long u = (long)j * 1024 + threadIdx.x;
        u32x4 v = u32x4{(unsigned)u, 1u, 2u, (unsigned)blockIdx.x};
        if (u < kRowVec) out[base + u] = v;
      }
      __builtin_amdgcn_s_waitcnt(0x0070 | 0xc000 | 0x0f00);
    }
    __syncthreads();
  }
}
int run() {
  const long rows = 2048, n_vec = rows * kRowVec, bytes = n_vec * 16;
  const int sets = 6, rounds = 7;
  std::vector<u32x4*> out(sets);
  for (int i = 0; i < sets; ++i) { CK(hipMalloc(&out[i], bytes)); CK(hipMemset(out[i], i + 1, bytes)); }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const char* names[] = {"row, all 19 in flight, plain", "row, 8 in flight", "row, 4 in flight", "row, 2 in flight", "row, 1 in flight",
                         "row, wait after every store", "row, all in flight, NT", "row, 2 in flight, NT", "fill once 256 thr", "fill once 1024 thr",
                         "row, waves in 2 turns", "row, waves in 4 turns", "hipMemsetAsync"};
  const int nv = 13;
  std::vector<float> t[nv];
  for (int r = 0; r < rounds; ++r) for (int k = 0; k < nv; ++k) {
    const int v = (k * 5 + r * 3) % nv;
    CK(hipEventRecord(a));
    for (int i = 0; i < sets; ++i) {
      switch (v) {
        case 0: hipLaunchKernelGGL((k_store<99, false>), dim3(rows), dim3(1024), 0, 0, out[i], rows); break;
        case 1: hipLaunchKernelGGL((k_store<8, false>), dim3(rows), dim3(1024), 0, 0, out[i], rows); break;
        case 2: hipLaunchKernelGGL((k_store<4, false>), dim3(rows), dim3(1024), 0, 0, out[i], rows); break;
        case 3: hipLaunchKernelGGL((k_store<2, false>), dim3(rows), dim3(1024), 0, 0, out[i], rows); break;
        case 4: hipLaunchKernelGGL((k_store<1, false>), dim3(rows), dim3(1024), 0, 0, out[i], rows); break;
        case 5: hipLaunchKernelGGL((k_store<0, false>), dim3(rows), dim3(1024), 0, 0, out[i], rows); break;
        case 6: hipLaunchKernelGGL((k_store<99, true>), dim3(rows), dim3(1024), 0, 0, out[i], rows); break;
        case 7: hipLaunchKernelGGL((k_store<2, true>), dim3(rows), dim3(1024), 0, 0, out[i], rows); break;
        case 8: hipLaunchKernelGGL(k_fill_once, dim3((n_vec + 255) / 256), dim3(256), 0, 0, out[i], n_vec); break;
        case 9: hipLaunchKernelGGL(k_fill_once_1024, dim3((n_vec + 1023) / 1024), dim3(1024), 0, 0, out[i], n_vec); break;
        case 10: hipLaunchKernelGGL((k_store_turns<2>), dim3(rows), dim3(1024), 0, 0, out[i], rows); break;
        case 11: hipLaunchKernelGGL((k_store_turns<4>), dim3(rows), dim3(1024), 0, 0, out[i], rows); break;
        case 12: CK(hipMemsetAsync(out[i], 7, bytes, 0)); break;
      }
    }
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (r > 0) t[v].push_back(ms / sets);
  }
  for (int v = 0; v < nv; ++v) {
    std::sort(t[v].begin(), t[v].end());
    const float med = t[v][t[v].size() / 2];
    printf("%-32s median %.4f ms  min %.4f ms  %6.0f GB/s\n", names[v], med, t[v][0], bytes / med / 1e6);
  }
  return 0;
}

}  // namespace exp3

// ------------------------------------------------------------------ experiment 4: does a compact window of concurrently written addresses decide the rate?
namespace exp4 {
// Store-only patterns, continued: does a compact window of concurrently written addresses decide the rate?
constexpr int kChunks = 19;
constexpr long kRowVec = 18992;
#define WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0070 | 0xc000 | 0x0f00)
// grid-stride fill, one 16-byte vector per thread and trip; WAIT: every wave waits for its store before the next
template <int THREADS, bool WAIT>
__global__ __launch_bounds__(THREADS) void k_fill_stride(u32x4* __restrict__ out, long n_vec) {
  const long stride = (long)gridDim.x * THREADS;
  for (long i = (long)blockIdx.x * THREADS + threadIdx.x; i < n_vec; i += stride) {
    out[i] = u32x4{(unsigned)i, 1u, 2u, (unsigned)(i >> 7)};
    if (WAIT) WAIT_VM0();
  }
}
// a row per workgroup of 1024 threads; ORDER: which 1 KB block of the row wave w writes in round j
//  0: j * 16 + w (16 KB contiguous per round)   1: w * 19 + j (a wave owns 19 KB)   2: j + 19 * w (the blocks of one pass lie 19 KB apart)
//  3: rounds in reverse   4: round order rotated by the row index (rows of neighbouring units start at different offsets)
template <int ORDER>
__global__ __launch_bounds__(1024) void k_row(u32x4* __restrict__ out, long rows) {
  const long base = (long)blockIdx.x * kRowVec;
  const int w = threadIdx.x / 64, lane = threadIdx.x % 64;
#pragma unroll
  for (int jj = 0; jj < kChunks; ++jj) {
    int j = jj;
    if (ORDER == 3) j = kChunks - 1 - jj;
    if (ORDER == 4) j = (jj + (int)(blockIdx.x % kChunks)) % kChunks;
    long blk;
    if (ORDER == 1) blk = w * kChunks + j;
    else if (ORDER == 2) blk = j + kChunks * w;
    else blk = j * 16 + w;
    const long u = blk * 64 + lane;
    if (u < kRowVec) out[base + u] = u32x4{(unsigned)u, 1u, 2u, (unsigned)blockIdx.x};
  }
}
// rows written by 256 workgroups that walk over rows b, b + 256, ... (no workgroup turnover), waiting per row
__global__ __launch_bounds__(1024) void k_row_walk(u32x4* __restrict__ out, long rows) {
  for (long row = blockIdx.x; row < rows; row += gridDim.x) {
    const long base = row * kRowVec;
#pragma unroll
    for (int j = 0; j < kChunks; ++j) {
      const long u = (long)j * 1024 + threadIdx.x;
      if (u < kRowVec) out[base + u] = u32x4{(unsigned)u, 1u, 2u, (unsigned)row};
    }
    WAIT_VM0();
  }
}
int run() {
  const long rows = 2048, n_vec = rows * kRowVec, bytes = n_vec * 16;
  const int sets = 6, rounds = 7;
  std::vector<u32x4*> out(sets);
  for (int i = 0; i < sets; ++i) { CK(hipMalloc(&out[i], bytes)); CK(hipMemset(out[i], i + 1, bytes)); }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const char* names[] = {"stride 256 thr g2048", "stride 256 thr g2048, wait per store", "stride 256 thr g8192, wait per store",
                         "stride 1024 thr g512, wait per store", "stride 1024 thr g2048", "row order 0 (16 KB per round)", "row order 1 (wave owns 19 KB)",
                         "row order 2 (19 KB apart)", "row order 3 (reverse)", "row order 4 (rotated by row)", "row walk, 256 workgroups", "row walk, 512 workgroups",
                         "hipMemsetAsync"};
  const int nv = 13;
  std::vector<float> t[nv];
  for (int r = 0; r < rounds; ++r) for (int k = 0; k < nv; ++k) {
    const int v = (k * 5 + r * 3) % nv;
    CK(hipEventRecord(a));
    for (int i = 0; i < sets; ++i) {
      switch (v) {
        case 0: hipLaunchKernelGGL((k_fill_stride<256, false>), dim3(2048), dim3(256), 0, 0, out[i], n_vec); break;
        case 1: hipLaunchKernelGGL((k_fill_stride<256, true>), dim3(2048), dim3(256), 0, 0, out[i], n_vec); break;
        case 2: hipLaunchKernelGGL((k_fill_stride<256, true>), dim3(8192), dim3(256), 0, 0, out[i], n_vec); break;
        case 3: hipLaunchKernelGGL((k_fill_stride<1024, true>), dim3(512), dim3(1024), 0, 0, out[i], n_vec); break;
        case 4: hipLaunchKernelGGL((k_fill_stride<1024, false>), dim3(2048), dim3(1024), 0, 0, out[i], n_vec); break;
        case 5: hipLaunchKernelGGL((k_row<0>), dim3(rows), dim3(1024), 0, 0, out[i], rows); break;
        case 6: hipLaunchKernelGGL((k_row<1>), dim3(rows), dim3(1024), 0, 0, out[i], rows); break;
        case 7: hipLaunchKernelGGL((k_row<2>), dim3(rows), dim3(1024), 0, 0, out[i], rows); break;
        case 8: hipLaunchKernelGGL((k_row<3>), dim3(rows), dim3(1024), 0, 0, out[i], rows); break;
        case 9: hipLaunchKernelGGL((k_row<4>), dim3(rows), dim3(1024), 0, 0, out[i], rows); break;
        case 10: hipLaunchKernelGGL(k_row_walk, dim3(256), dim3(1024), 0, 0, out[i], rows); break;
        case 11: hipLaunchKernelGGL(k_row_walk, dim3(512), dim3(1024), 0, 0, out[i], rows); break;
        case 12: CK(hipMemsetAsync(out[i], 7, bytes, 0)); break;
      }
    }
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (r > 0) t[v].push_back(ms / sets);
  }
  for (int v = 0; v < nv; ++v) {
    std::sort(t[v].begin(), t[v].end());
    const float med = t[v][t[v].size() / 2];
    printf("%-40s median %.4f ms  min %.4f ms  %6.0f GB/s\n", names[v], med, t[v][0], bytes / med / 1e6);
  }
  return 0;
}

}  // namespace exp4
#undef WAIT_VM0

// ------------------------------------------------------------------ experiment 5: one store per thread and exit, workgroups scattered over S sequential streams; blocks per workgroup
namespace exp5 {
// Store-only, continued: one 16-byte store per thread and exit (what reaches the memset rate), with consecutive
// workgroups scattered over S separate sequential streams; and a loop whose trips write neighbouring blocks.
__global__ __launch_bounds__(256) void k_fill_streams(u32x4* __restrict__ out, long n_blocks, long streams) {
  const long i = blockIdx.x;
  const long per = n_blocks / streams;  // n_blocks is a multiple of every S tried
  const long blk = (i % streams) * per + i / streams;
  out[blk * 256 + threadIdx.x] = u32x4{(unsigned)i, 1u, 2u, (unsigned)threadIdx.x};
}
// K consecutive 4 KB blocks per workgroup, one per trip (short loop, contiguous)
template <int K>
__global__ __launch_bounds__(256) void k_fill_k(u32x4* __restrict__ out, long n_blocks) {
  const long b0 = (long)blockIdx.x * K;
#pragma unroll
  for (int k = 0; k < K; ++k) if (b0 + k < n_blocks) out[(b0 + k) * 256 + threadIdx.x] = u32x4{(unsigned)b0, 1u, 2u, (unsigned)k};
}
int run() {
  const long n_blocks = 2048L * 74 * 1, n_vec = n_blocks * 256, bytes = n_vec * 16;  // 620 MB
  const int sets = 6, rounds = 7;
  std::vector<u32x4*> out(sets);
  for (int i = 0; i < sets; ++i) { CK(hipMalloc(&out[i], bytes)); CK(hipMemset(out[i], i + 1, bytes)); }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const long S[] = {1, 8, 32, 128, 256, 512, 2048};
  const char* names[] = {"once, 1 stream", "once, 8 streams", "once, 32 streams", "once, 128 streams", "once, 256 streams", "once, 512 streams",
                         "once, 2048 streams", "2 blocks per workgroup", "4 blocks per workgroup", "8 blocks per workgroup", "19 blocks per workgroup", "hipMemsetAsync"};
  const int nv = 12;
  std::vector<float> t[nv];
  for (int r = 0; r < rounds; ++r) for (int k = 0; k < nv; ++k) {
    const int v = (k * 5 + r * 3) % nv;
    CK(hipEventRecord(a));
    for (int i = 0; i < sets; ++i) {
      if (v < 7) hipLaunchKernelGGL(k_fill_streams, dim3(n_blocks), dim3(256), 0, 0, out[i], n_blocks, S[v]);
      else if (v == 7) hipLaunchKernelGGL((k_fill_k<2>), dim3((n_blocks + 1) / 2), dim3(256), 0, 0, out[i], n_blocks);
      else if (v == 8) hipLaunchKernelGGL((k_fill_k<4>), dim3((n_blocks + 3) / 4), dim3(256), 0, 0, out[i], n_blocks);
      else if (v == 9) hipLaunchKernelGGL((k_fill_k<8>), dim3((n_blocks + 7) / 8), dim3(256), 0, 0, out[i], n_blocks);
      else if (v == 10) hipLaunchKernelGGL((k_fill_k<19>), dim3((n_blocks + 18) / 19), dim3(256), 0, 0, out[i], n_blocks);
      else CK(hipMemsetAsync(out[i], 7, bytes, 0));
    }
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (r > 0) t[v].push_back(ms / sets);
  }
  for (int v = 0; v < nv; ++v) {
    std::sort(t[v].begin(), t[v].end());
    const float med = t[v][t[v].size() / 2];
    printf("%-28s median %.4f ms  min %.4f ms  %6.0f GB/s\n", names[v], med, t[v][0], bytes / med / 1e6);
  }
  return 0;
}

}  // namespace exp5

// ------------------------------------------------------------------ experiment 6: many streams, every workgroup's 4 KB written as pieces a stride apart inside its stream
namespace exp6 {
// Store-only, continued: many sequential streams (where the write rate is low), but every workgroup writes its
// 4 KB as PIECES that lie a stride apart inside its stream instead of next to each other.  If the pieces of one
// DRAM page sit a fixed stride apart in the address space, then at that stride a workgroup fills whole pages at
// once and the rate of a single stream should come back.
// Everything is a power of two, so that the index arithmetic is shifts (with divisions the kernel, not the
// memory, sets the rate).  lg_streams, lg_piece (16-byte units per piece), lg_tile (workgroups sharing one tile).
__global__ __launch_bounds__(256) void k_fill_pieces(u32x4* __restrict__ out, int lg_blocks, int lg_streams, int lg_piece,
                                                     int lg_tile) {
  const unsigned i = blockIdx.x;
  const unsigned s = i & ((1u << lg_streams) - 1), k = i >> lg_streams;  // stream, and the workgroup's turn inside it
  const int lg_per = lg_blocks - lg_streams;                              // 4 KB blocks per stream
  const unsigned tile = k >> lg_tile, slot = k & ((1u << lg_tile) - 1);
  const unsigned p = threadIdx.x >> lg_piece, within = threadIdx.x & ((1u << lg_piece) - 1);
  // a tile is 2^lg_tile blocks = 2^(lg_tile + 8) units; the workgroup's pieces lie 2^(lg_tile + lg_piece) units apart
  const unsigned long unit = ((unsigned long)s << (lg_per + 8)) + ((unsigned long)tile << (lg_tile + 8)) +
                             ((unsigned long)p << (lg_tile + lg_piece)) + (slot << lg_piece) + within;
  out[unit] = u32x4{i, 1u, 2u, (unsigned)threadIdx.x};
}
int run() {
  const int lg_blocks = 18;  // 262144 blocks of 4 KB = 1 GiB
  const long n_blocks = 1L << lg_blocks, bytes = n_blocks * 4096;
  const int sets = 5, rounds = 7;
  std::vector<u32x4*> out(sets);
  for (int i = 0; i < sets; ++i) { CK(hipMalloc(&out[i], bytes)); CK(hipMemset(out[i], i + 1, bytes)); }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  struct V { int lg_streams, lg_piece, lg_tile; };
  std::vector<V> vs = {{0, 8, 0}, {3, 8, 0}, {5, 8, 0}, {6, 8, 0}, {8, 8, 0}};
  for (int ls : {8, 6})
    for (int lp : {4, 5, 6})           // pieces of 256 B, 512 B, 1 KB
      for (int lt = 1; lt <= 10 && lt <= lg_blocks - ls; ++lt) vs.push_back({ls, lp, lt});
  std::vector<std::vector<float>> t(vs.size());
  for (int r = 0; r < rounds; ++r) for (size_t k = 0; k < vs.size(); ++k) {
    const size_t v = (k * 7 + r * 3) % vs.size();
    CK(hipEventRecord(a));
    for (int i = 0; i < sets; ++i)
      hipLaunchKernelGGL(k_fill_pieces, dim3(n_blocks), dim3(256), 0, 0, out[i], lg_blocks, vs[v].lg_streams, vs[v].lg_piece, vs[v].lg_tile);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (r > 0) t[v].push_back(ms / sets);
  }
  for (size_t v = 0; v < vs.size(); ++v) {
    std::sort(t[v].begin(), t[v].end());
    const float med = t[v][t[v].size() / 2];
    printf("%4d streams, pieces of %4d B every %8ld B: median %.4f ms  %6.0f GB/s\n", 1 << vs[v].lg_streams, 16 << vs[v].lg_piece,
           16L << (vs[v].lg_tile + vs[v].lg_piece), med, bytes / med / 1e6);
  }
  return 0;
}

}  // namespace exp6

// ------------------------------------------------------------------ experiment 7: the row shape under other maps from (round, wave, quarter-wave, lane) to the address
namespace exp7 {
// Store-only, in the row kernel's shape (one workgroup of 1024 threads per 152 K-entry bf16 row, 19 x 16 bytes per
// lane), with other maps from (round j, wave w, quarter-wave q, lane-in-quarter) to the address: which 256-byte
// pieces a wave's store instruction covers, and how far apart they lie.  The first 18 rounds (18432 units of 16
// bytes) are permuted, the row's last 560 units are written in place.
constexpr long kRowVec = 18992;
constexpr int kRounds = 18;
// ORDER names the digits above the 16 units of a piece, least significant first: 0 = q,w,j (the kernel's map today)
template <int ORDER>
__device__ __forceinline__ unsigned map_unit(unsigned j, unsigned w, unsigned q, unsigned l) {
  switch (ORDER) {
    case 0: return l + 16u * (q + 4u * (w + 16u * j));
    case 1: return l + 16u * (q + 4u * (j + kRounds * w));
    case 2: return l + 16u * (w + 16u * (q + 4u * j));
    case 3: return l + 16u * (w + 16u * (j + kRounds * q));
    case 4: return l + 16u * (j + kRounds * (q + 4u * w));
    default: return l + 16u * (j + kRounds * (w + 16u * q));
  }
}
template <int ORDER>
__global__ __launch_bounds__(1024) void k_store(u32x4* __restrict__ out) {
  const long base = (long)blockIdx.x * kRowVec;
  const unsigned w = threadIdx.x >> 6, q = (threadIdx.x >> 4) & 3, l = threadIdx.x & 15;
#pragma unroll
  for (unsigned j = 0; j < kRounds; ++j)
    out[base + map_unit<ORDER>(j, w, q, l)] = u32x4{j, 1u, 2u, (unsigned)blockIdx.x};
  const long u = (long)kRounds * 1024 + threadIdx.x;
  if (u < kRowVec) out[base + u] = u32x4{99u, 1u, 2u, (unsigned)blockIdx.x};
}
// the same maps for a load-all / store-all copy (what the row kernel's skeleton is)
template <int ORDER>
__global__ __launch_bounds__(1024) void k_copy(const u32x4* __restrict__ in, u32x4* __restrict__ out) {
  const long base = (long)blockIdx.x * kRowVec;
  const unsigned w = threadIdx.x >> 6, q = (threadIdx.x >> 4) & 3, l = threadIdx.x & 15;
  u32x4 buf[kRounds + 1];
#pragma unroll
  for (unsigned j = 0; j < kRounds; ++j) buf[j] = __builtin_nontemporal_load(in + base + map_unit<ORDER>(j, w, q, l));
  const long u = (long)kRounds * 1024 + threadIdx.x;
  buf[kRounds] = u < kRowVec ? __builtin_nontemporal_load(in + base + u) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
  for (unsigned j = 0; j < kRounds; ++j) {
    buf[j].x ^= 1u;
    __builtin_nontemporal_store(buf[j], out + base + map_unit<ORDER>(j, w, q, l));
  }
  if (u < kRowVec) __builtin_nontemporal_store(buf[kRounds], out + base + u);
}
int run() {
  const long rows = 2048, bytes = rows * kRowVec * 16;
  const int sets = 4, rounds = 9;
  std::vector<u32x4*> in(sets), out(sets);
  for (int i = 0; i < sets; ++i) { CK(hipMalloc(&in[i], bytes)); CK(hipMalloc(&out[i], bytes)); CK(hipMemset(in[i], i + 1, bytes)); }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const char* names[6] = {"q,w,j (today: 1 KB per wave, 16 KB per round)", "q,j,w (a wave owns 18 KB)", "w,q,j (a wave's pieces 4 KB apart)",
                          "w,j,q (a wave's pieces 72 KB apart)", "j,q,w (rounds 256 B apart)", "j,w,q (rounds 256 B apart, pieces 72 KB apart)"};
  std::vector<float> t[12];
  for (int r = 0; r < rounds; ++r) for (int k = 0; k < 12; ++k) {
    const int v = (k * 5 + r * 3) % 12;
    CK(hipEventRecord(a));
    for (int i = 0; i < sets; ++i) {
      switch (v) {
        case 0: hipLaunchKernelGGL(k_store<0>, dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 1: hipLaunchKernelGGL(k_store<1>, dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 2: hipLaunchKernelGGL(k_store<2>, dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 3: hipLaunchKernelGGL(k_store<3>, dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 4: hipLaunchKernelGGL(k_store<4>, dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 5: hipLaunchKernelGGL(k_store<5>, dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 6: hipLaunchKernelGGL(k_copy<0>, dim3(rows), dim3(1024), 0, 0, in[i], out[i]); break;
        case 7: hipLaunchKernelGGL(k_copy<1>, dim3(rows), dim3(1024), 0, 0, in[i], out[i]); break;
        case 8: hipLaunchKernelGGL(k_copy<2>, dim3(rows), dim3(1024), 0, 0, in[i], out[i]); break;
        case 9: hipLaunchKernelGGL(k_copy<3>, dim3(rows), dim3(1024), 0, 0, in[i], out[i]); break;
        case 10: hipLaunchKernelGGL(k_copy<4>, dim3(rows), dim3(1024), 0, 0, in[i], out[i]); break;
        case 11: hipLaunchKernelGGL(k_copy<5>, dim3(rows), dim3(1024), 0, 0, in[i], out[i]); break;
      }
    }
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (r > 0) t[v].push_back(ms / sets);
  }
  for (int v = 0; v < 12; ++v) {
    std::sort(t[v].begin(), t[v].end());
    const float med = t[v][t[v].size() / 2];
    printf("%-5s %-50s median %.4f ms  %6.0f GB/s\n", v < 6 ? "store" : "copy", names[v % 6], med, (v < 6 ? 1.0 : 2.0) * bytes / med / 1e6);
  }
  return 0;
}

}  // namespace exp7

// ------------------------------------------------------------------ experiment 8: the row shape: does the order in which a unit's 16 waves emit their stores matter?
namespace exp8 {
// Store-only in the row kernel's shape, continued: does the ORDER in which a compute unit's 16 waves emit their
// stores matter?  (a) today's map with a barrier behind every round, so that a unit emits its row as 19 contiguous
// 16 KB blocks one after the other; (b) a wave owning 19 contiguous KB with the waves storing one after the other,
// so that a unit emits its row strictly front to back; (c) today's map, waves in 8 / 16 turns.
constexpr int kChunks = 19;
constexpr long kRowVec = 18992;
#define WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0070 | 0xc000 | 0x0f00)
template <int MODE, bool NT>
__global__ __launch_bounds__(1024) void k_store(u32x4* __restrict__ out) {
  const long base = (long)blockIdx.x * kRowVec;
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
  auto put = [&](long u, unsigned tag) {
    const u32x4 v = u32x4{(unsigned)u, tag, 2u, (unsigned)blockIdx.x};
    if (u < kRowVec) { if (NT) __builtin_nontemporal_store(v, out + base + u); else out[base + u] = v; }
  };
  if (MODE == 0) {  // today: everything at once
#pragma unroll
    for (int j = 0; j < kChunks; ++j) put((long)j * 1024 + threadIdx.x, j);
  } else if (MODE == 1) {  // a barrier behind every round
#pragma unroll
    for (int j = 0; j < kChunks; ++j) {
      put((long)j * 1024 + threadIdx.x, j);
      __syncthreads();
    }
  } else if (MODE == 2) {  // a barrier behind every round, and the round's stores acknowledged before it
#pragma unroll
    for (int j = 0; j < kChunks; ++j) {
      put((long)j * 1024 + threadIdx.x, j);
      WAIT_VM0();
      __syncthreads();
    }
  } else if (MODE == 3) {  // a wave owns 19 KB, waves one after the other (issue order only)
    for (int turn = 0; turn < 16; ++turn) {
      if (wave == turn) {
#pragma unroll
        for (int j = 0; j < kChunks; ++j) put((long)wave * (kChunks * 64) + j * 64 + lane, j);
      }
      __syncthreads();
    }
  } else {  // today's map, waves in (MODE) turns
    for (int turn = 0; turn < MODE; ++turn) {
      if (wave % MODE == turn) {
#pragma unroll
        for (int j = 0; j < kChunks; ++j) put((long)j * 1024 + threadIdx.x, j);
        WAIT_VM0();
      }
      __syncthreads();
    }
  }
}
int run() {
  const long rows = 2048, bytes = rows * kRowVec * 16;
  const int sets = 6, rounds = 7;
  std::vector<u32x4*> out(sets);
  for (int i = 0; i < sets; ++i) { CK(hipMalloc(&out[i], bytes)); CK(hipMemset(out[i], i + 1, bytes)); }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const char* names[] = {"all at once, plain", "all at once, NT", "barrier per round, plain", "barrier per round, NT", "barrier + acknowledged per round, plain",
                         "wave owns 19 KB, one wave after the other, plain", "wave owns 19 KB, one wave after the other, NT", "waves in 8 turns", "waves in 16 turns"};
  const int nv = 9;
  std::vector<float> t[nv];
  for (int r = 0; r < rounds; ++r) for (int k = 0; k < nv; ++k) {
    const int v = (k * 5 + r * 3) % nv;
    CK(hipEventRecord(a));
    for (int i = 0; i < sets; ++i) {
      switch (v) {
        case 0: hipLaunchKernelGGL((k_store<0, false>), dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 1: hipLaunchKernelGGL((k_store<0, true>), dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 2: hipLaunchKernelGGL((k_store<1, false>), dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 3: hipLaunchKernelGGL((k_store<1, true>), dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 4: hipLaunchKernelGGL((k_store<2, false>), dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 5: hipLaunchKernelGGL((k_store<3, false>), dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 6: hipLaunchKernelGGL((k_store<3, true>), dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 7: hipLaunchKernelGGL((k_store<8, false>), dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 8: hipLaunchKernelGGL((k_store<16, false>), dim3(rows), dim3(1024), 0, 0, out[i]); break;
      }
    }
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (r > 0) t[v].push_back(ms / sets);
  }
  for (int v = 0; v < nv; ++v) {
    std::sort(t[v].begin(), t[v].end());
    const float med = t[v][t[v].size() / 2];
    printf("%-52s median %.4f ms  %6.0f GB/s\n", names[v], med, bytes / med / 1e6);
  }
  return 0;
}

}  // namespace exp8
#undef WAIT_VM0

int main(int argc, char** argv) {
  if (argc < 2) {
    printf("usage: %s N, N = 1..8 (see the top of the source)\n", argv[0]);
    return 2;
  }
  const int n = atoi(argv[1]);

  switch (n) {
    case 1: return exp1::run();
    case 2: return exp2::run();
    case 3: return exp3::run();
    case 4: return exp4::run();
    case 5: return exp5::run();
    case 6: return exp6::run();
    case 7: return exp7::run();
    case 8: return exp8::run();
  }
  printf("no experiment %d\n", n);
  return 2;
}
