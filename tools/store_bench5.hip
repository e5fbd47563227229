// Store-only, continued: one 16-byte store per thread and exit (what reaches the memset rate), with consecutive
// workgroups scattered over S separate sequential streams; and a loop whose trips write neighbouring blocks.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
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
int main() {
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
