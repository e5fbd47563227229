// Which store pattern reaches what hipMemsetAsync reaches (6.7 TB/s on buffers that are not cache-resident)?
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
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
int main() {
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
