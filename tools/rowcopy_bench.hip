// How fast can "load a piece of a row into registers, then store it" go, by piece size?  One workgroup holds
// 19 x 16 bytes per lane between its loads and its stores, as the single-pass row kernel does, with 1024, 512
// or 256 threads (a whole 152 K-entry bf16 row, half of one, a quarter): 1, 2 or 4 workgroups per compute unit.
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
  const char* names[3] = {"whole row, 1024 threads", "half row, 512 threads", "quarter row, 256 threads"};
  std::vector<float> t[3];
  for (int r = 0; r < rounds; ++r) {
    for (int v = 0; v < 3; ++v) {
      const int which = (v + r) % 3;
      CK(hipEventRecord(a));
      for (int i = 0; i < sets; ++i) {
        if (which == 0) hipLaunchKernelGGL(k_piece<1024>, dim3(rows), dim3(1024), 0, 0, in[i], out[i], rows);
        if (which == 1) hipLaunchKernelGGL(k_piece<512>, dim3(rows * 2), dim3(512), 0, 0, in[i], out[i], rows);
        if (which == 2) hipLaunchKernelGGL(k_piece<256>, dim3(rows * 4), dim3(256), 0, 0, in[i], out[i], rows);
      }
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      if (r > 1) t[which].push_back(ms / sets);
    }
  }
  for (int v = 0; v < 3; ++v) {
    std::sort(t[v].begin(), t[v].end());
    const float med = t[v][t[v].size() / 2];
    printf("%-26s median %.4f ms  min %.4f ms  %6.0f GB/s (read + write)\n", names[v], med, t[v][0], 2.0 * bytes / med / 1e6);
  }
  return 0;
}
