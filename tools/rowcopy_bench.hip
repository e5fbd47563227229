// How fast can "load a piece of a row into registers, then store it" go, by piece size?  One workgroup holds
// 19 x 16 bytes per lane between its loads and its stores, as the single-pass row kernel does, with 1024, 512
// or 256 threads (a whole 152 K-entry bf16 row, half of one, a quarter): 1, 2 or 4 workgroups per compute unit.
// And a row SHARED by 2 or 4 compute units of one XCD (k_shared): workgroups of 1024 threads that hold a half or a
// quarter of a row (10 or 5 x 16 bytes per lane, so 2 or 4 of them fit a compute unit), the pieces of a row in
// blocks b, b + 8, ... (blocks go to the XCDs round robin, so these share an XCD and start together), or in
// consecutive blocks (K different XCDs).  What such a row kernel could reach before its soft-max sums are exchanged.
// "one piece per unit": the same with 100 KB of (unused) LDS per workgroup, so that a compute unit holds ONE piece and
// the chip works on 128 or 64 rows at a time instead of 256 (fewer concurrent write streams, half / a quarter of
// the bytes in flight).
//   hipcc --offload-arch=gfx950 -O3 -o rowcopy_bench tools/rowcopy_bench.hip && ./rowcopy_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kChunks = 19;
constexpr long kRowVec = 18992;  // 151936 bf16 = 18992 16-byte units

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_piece(const u32x4* __restrict__ in, u32x4* __restrict__ out, long rows) {
  constexpr int kPieces = 1024 / THREADS;
  constexpr long kPieceVec = kRowVec / kPieces;  // 18992 divides by 1, 2 and 4
  const long piece = blockIdx.x;
  if (piece >= rows * kPieces) return;
  const long base = (piece / kPieces) * kRowVec + (piece % kPieces) * kPieceVec;
  u32x4 buf[kChunks];
#pragma unroll
  for (int j = 0; j < kChunks; ++j) {
    const long u = (long)j * THREADS + threadIdx.x;
    buf[j] = u < kPieceVec ? __builtin_nontemporal_load(in + base + u) : u32x4{0u, 0u, 0u, 0u};
  }
#pragma unroll
  for (int j = 0; j < kChunks; ++j) {
    const long u = (long)j * THREADS + threadIdx.x;
    buf[j].x ^= 1u;  // something between the load and the store
    if (u < kPieceVec) __builtin_nontemporal_store(buf[j], out + base + u);
  }
}

// PIECES workgroups of 1024 threads per row; SAME_XCD: a row's pieces are blocks 8 apart inside a group of 8 * PIECES
template <int PIECES, bool SAME_XCD>
__global__ __launch_bounds__(1024) void k_shared(const u32x4* __restrict__ in, u32x4* __restrict__ out, long rows) {
  constexpr int kCh = (kChunks + PIECES - 1) / PIECES;
  constexpr long kPieceVec = kRowVec / PIECES;
  const long b = blockIdx.x;
  long row, piece;
  if (SAME_XCD) {
    const long group = b / (8 * PIECES), within = b % (8 * PIECES);
    row = group * 8 + (within & 7);
    piece = within >> 3;
  } else {
    row = b / PIECES;
    piece = b % PIECES;
  }
  if (row >= rows) return;
  const long base = row * kRowVec + piece * kPieceVec;
  u32x4 buf[kCh];
#pragma unroll
  for (int j = 0; j < kCh; ++j) {
    const long u = (long)j * 1024 + threadIdx.x;
    buf[j] = u < kPieceVec ? __builtin_nontemporal_load(in + base + u) : u32x4{0u, 0u, 0u, 0u};
  }
#pragma unroll
  for (int j = 0; j < kCh; ++j) {
    const long u = (long)j * 1024 + threadIdx.x;
    buf[j].x ^= 1u;
    if (u < kPieceVec) __builtin_nontemporal_store(buf[j], out + base + u);
  }
}

int main() {
  const long rows = 2048, bytes = rows * kRowVec * 16;
  const int sets = 4, rounds = 15;
  std::vector<u32x4*> in(sets), out(sets);
  for (int i = 0; i < sets; ++i) {
    CK(hipMalloc(&in[i], bytes));
    CK(hipMalloc(&out[i], bytes));
    CK(hipMemset(in[i], i + 1, bytes));
  }
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  constexpr int kVariants = 9;
  const char* names[kVariants] = {"whole row, 1024 threads", "half row, 512 threads", "quarter row, 256 threads",
                                  "row over 2 units of an XCD", "row over 4 units of an XCD",
                                  "row over 2 units, any XCD", "row over 4 units, any XCD",
                                  "2 units of an XCD, one piece per unit", "4 units of an XCD, one piece per unit"};
  const int kLdsHog = 100 * 1024;
  CK(hipFuncSetAttribute((const void*)k_shared<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsHog));
  CK(hipFuncSetAttribute((const void*)k_shared<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsHog));
  std::vector<float> t[kVariants];
  for (int r = 0; r < rounds; ++r) {
    for (int v = 0; v < kVariants; ++v) {
      const int which = (v + r) % kVariants;
      CK(hipEventRecord(a));
      for (int i = 0; i < sets; ++i) {
        if (which == 0) hipLaunchKernelGGL(k_piece<1024>, dim3(rows), dim3(1024), 0, 0, in[i], out[i], rows);
        if (which == 1) hipLaunchKernelGGL(k_piece<512>, dim3(rows * 2), dim3(512), 0, 0, in[i], out[i], rows);
        if (which == 2) hipLaunchKernelGGL(k_piece<256>, dim3(rows * 4), dim3(256), 0, 0, in[i], out[i], rows);
        if (which == 3) hipLaunchKernelGGL((k_shared<2, true>), dim3(rows * 2), dim3(1024), 0, 0, in[i], out[i], rows);
        if (which == 4) hipLaunchKernelGGL((k_shared<4, true>), dim3(rows * 4), dim3(1024), 0, 0, in[i], out[i], rows);
        if (which == 5) hipLaunchKernelGGL((k_shared<2, false>), dim3(rows * 2), dim3(1024), 0, 0, in[i], out[i], rows);
        if (which == 6) hipLaunchKernelGGL((k_shared<4, false>), dim3(rows * 4), dim3(1024), 0, 0, in[i], out[i], rows);
        if (which == 7) hipLaunchKernelGGL((k_shared<2, true>), dim3(rows * 2), dim3(1024), kLdsHog, 0, in[i], out[i], rows);
        if (which == 8) hipLaunchKernelGGL((k_shared<4, true>), dim3(rows * 4), dim3(1024), kLdsHog, 0, in[i], out[i], rows);
      }
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      if (r > 1) t[which].push_back(ms / sets);
    }
  }
  for (int v = 0; v < kVariants; ++v) {
    std::sort(t[v].begin(), t[v].end());
    const float med = t[v][t[v].size() / 2];
    printf("%-40s median %.4f ms  min %.4f ms  %6.0f GB/s (read + write)\n", names[v], med, t[v][0], 2.0 * bytes / med / 1e6);
  }
  return 0;
}
