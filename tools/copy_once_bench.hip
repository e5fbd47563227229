// Read + write streaming structures on buffers that are not cache-resident: one 16-byte unit per thread and exit,
// several per thread, or a capped grid striding over the buffer (what a grid-stride loop does).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int UNR, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_copy_once(const u32x4* __restrict__ in, u32x4* __restrict__ out, long n_vec) {
  const long base = (long)blockIdx.x * (UNR * 256) + threadIdx.x;
  u32x4 v[UNR];
#pragma unroll
  for (int q = 0; q < UNR; ++q) v[q] = base + q * 256 < n_vec ? (NTL ? __builtin_nontemporal_load(in + base + q * 256) : in[base + q * 256]) : u32x4{0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < UNR; ++q) {
    v[q].x ^= 1u;
    if (base + q * 256 < n_vec) { if (NTS) __builtin_nontemporal_store(v[q], out + base + q * 256); else out[base + q * 256] = v[q]; }
  }
}
template <int UNR, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_copy_stride(const u32x4* __restrict__ in, u32x4* __restrict__ out, long n_vec) {
  const long chunk = (long)UNR * 256;
  for (long b = (long)blockIdx.x * chunk; b < n_vec; b += (long)gridDim.x * chunk) {
    const long base = b + threadIdx.x;
    u32x4 v[UNR];
#pragma unroll
    for (int q = 0; q < UNR; ++q) v[q] = base + q * 256 < n_vec ? (NTL ? __builtin_nontemporal_load(in + base + q * 256) : in[base + q * 256]) : u32x4{0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
      v[q].x ^= 1u;
      if (base + q * 256 < n_vec) { if (NTS) __builtin_nontemporal_store(v[q], out + base + q * 256); else out[base + q * 256] = v[q]; }
    }
  }
}
int main() {
  const long n_vec = 2048L * 18992, bytes = n_vec * 16;
  const int sets = 4, rounds = 9;
  std::vector<u32x4*> in(sets), out(sets);
  for (int i = 0; i < sets; ++i) { CK(hipMalloc(&in[i], bytes)); CK(hipMalloc(&out[i], bytes)); CK(hipMemset(in[i], i + 1, bytes)); }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const char* names[] = {"once u1 ntl nts", "once u1 ntl plain-store", "once u1 plain plain", "once u2 ntl nts", "once u2 ntl plain-store", "once u4 ntl nts",
                         "once u4 ntl plain-store", "once u8 ntl plain-store", "stride u2 g4096 ntl plain-store", "stride u4 g2048 ntl plain-store",
                         "stride u2 g16384 ntl plain-store", "hipMemcpyAsync DtoD"};
  const int nv = 12;
  std::vector<float> t[nv];
  auto grid = [&](int unr) { return dim3((unsigned)((n_vec + unr * 256 - 1) / (unr * 256))); };
  for (int r = 0; r < rounds; ++r) for (int k = 0; k < nv; ++k) {
    const int v = (k * 5 + r * 3) % nv;
    CK(hipEventRecord(a));
    for (int i = 0; i < sets; ++i) {
      switch (v) {
        case 0: hipLaunchKernelGGL((k_copy_once<1, true, true>), grid(1), dim3(256), 0, 0, in[i], out[i], n_vec); break;
        case 1: hipLaunchKernelGGL((k_copy_once<1, true, false>), grid(1), dim3(256), 0, 0, in[i], out[i], n_vec); break;
        case 2: hipLaunchKernelGGL((k_copy_once<1, false, false>), grid(1), dim3(256), 0, 0, in[i], out[i], n_vec); break;
        case 3: hipLaunchKernelGGL((k_copy_once<2, true, true>), grid(2), dim3(256), 0, 0, in[i], out[i], n_vec); break;
        case 4: hipLaunchKernelGGL((k_copy_once<2, true, false>), grid(2), dim3(256), 0, 0, in[i], out[i], n_vec); break;
        case 5: hipLaunchKernelGGL((k_copy_once<4, true, true>), grid(4), dim3(256), 0, 0, in[i], out[i], n_vec); break;
        case 6: hipLaunchKernelGGL((k_copy_once<4, true, false>), grid(4), dim3(256), 0, 0, in[i], out[i], n_vec); break;
        case 7: hipLaunchKernelGGL((k_copy_once<8, true, false>), grid(8), dim3(256), 0, 0, in[i], out[i], n_vec); break;
        case 8: hipLaunchKernelGGL((k_copy_stride<2, true, false>), dim3(4096), dim3(256), 0, 0, in[i], out[i], n_vec); break;
        case 9: hipLaunchKernelGGL((k_copy_stride<4, true, false>), dim3(2048), dim3(256), 0, 0, in[i], out[i], n_vec); break;
        case 10: hipLaunchKernelGGL((k_copy_stride<2, true, false>), dim3(16384), dim3(256), 0, 0, in[i], out[i], n_vec); break;
        case 11: CK(hipMemcpyAsync(out[i], in[i], bytes, hipMemcpyDeviceToDevice, 0)); break;
      }
    }
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (r > 0) t[v].push_back(ms / sets);
  }
  for (int v = 0; v < nv; ++v) {
    std::sort(t[v].begin(), t[v].end());
    const float med = t[v][t[v].size() / 2];
    printf("%-36s median %.4f ms  %6.0f GB/s (read + write)\n", names[v], med, 2.0 * bytes / med / 1e6);
  }
  return 0;
}
