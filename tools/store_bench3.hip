// Store-only kernels in the row kernel's shape (one workgroup of 1024 threads writes a 152 K-entry bf16 row as
// 19 x 16 bytes per lane) with the number of stores a wave may have in flight limited, next to a fill that
// writes 16 bytes per thread and ends (what the runtime's memset does).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
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
int main() {
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
