// Store-only patterns, continued: does a compact window of concurrently written addresses decide the rate?
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
int main() {
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
