// Store-only, in the row kernel's shape (one workgroup of 1024 threads per 152 K-entry bf16 row, 19 x 16 bytes per
// lane), with other maps from (round j, wave w, quarter-wave q, lane-in-quarter) to the address: which 256-byte
// pieces a wave's store instruction covers, and how far apart they lie.  The first 18 rounds (18432 units of 16
// bytes) are permuted, the row's last 560 units are written in place.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
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
int main() {
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
