#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
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
int main() {
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
