// Which read+write streaming structure gets the most out of HBM on MI355X?
// A plain 16-byte-per-lane copy in several work decompositions, timed with HIP events
// in interleaved rounds.  Build: hipcc --offload-arch=gfx950 -O3 tools/copybench.hip -o tools/bin/copybench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool NTL, bool NTS>
__device__ __forceinline__ void mv(const u32x4* s, u32x4* d) {
  u32x4 v = NTL ? __builtin_nontemporal_load(s) : *s;
  if (NTS) __builtin_nontemporal_store(v, d); else *d = v;
}

// (1) one workgroup per row of `row_vec` vectors
template <bool NTL, bool NTS, int UNR>
__global__ __launch_bounds__(256) void k_rows(const u32x4* in, u32x4* out, long row_vec) {
  const u32x4* s = in + (long)blockIdx.x * row_vec;
  u32x4* d = out + (long)blockIdx.x * row_vec;
  long i = threadIdx.x;
  for (; i + (UNR - 1) * 256 < row_vec; i += UNR * 256) {
    u32x4 v[UNR];
#pragma unroll
    for (int q = 0; q < UNR; ++q) v[q] = NTL ? __builtin_nontemporal_load(s + i + q * 256) : s[i + q * 256];
#pragma unroll
    for (int q = 0; q < UNR; ++q) { if (NTS) __builtin_nontemporal_store(v[q], d + i + q * 256); else d[i + q * 256] = v[q]; }
  }
  for (; i < row_vec; i += 256) mv<NTL, NTS>(s + i, d + i);
}

// (2) grid-stride over chunks of UNR*256 vectors: at any moment the grid works on one contiguous window
template <bool NTL, bool NTS, int UNR>
__global__ __launch_bounds__(256) void k_stride(const u32x4* in, u32x4* out, long n_vec) {
  const long chunk = (long)UNR * 256;
  for (long base = (long)blockIdx.x * chunk; base < n_vec; base += (long)gridDim.x * chunk) {
    u32x4 v[UNR];
    long i = base + threadIdx.x;
#pragma unroll
    for (int q = 0; q < UNR; ++q) if (i + q * 256 < n_vec) v[q] = NTL ? __builtin_nontemporal_load(in + i + q * 256) : in[i + q * 256];
#pragma unroll
    for (int q = 0; q < UNR; ++q) if (i + q * 256 < n_vec) { if (NTS) __builtin_nontemporal_store(v[q], out + i + q * 256); else out[i + q * 256] = v[q]; }
  }
}

struct Variant { const char* name; void (*run)(const u32x4*, u32x4*, long, long, hipStream_t); };

template <bool NTL, bool NTS, int UNR>
void run_rows(const u32x4* in, u32x4* out, long rows, long row_vec, hipStream_t st) {
  hipLaunchKernelGGL((k_rows<NTL, NTS, UNR>), dim3(rows), dim3(256), 0, st, in, out, row_vec);
}
template <bool NTL, bool NTS, int UNR, int GRID>
void run_stride(const u32x4* in, u32x4* out, long rows, long row_vec, hipStream_t st) {
  hipLaunchKernelGGL((k_stride<NTL, NTS, UNR>), dim3(GRID), dim3(256), 0, st, in, out, rows * row_vec);
}
void run_memcpy(const u32x4* in, u32x4* out, long rows, long row_vec, hipStream_t st) {
  CK(hipMemcpyAsync(out, in, rows * row_vec * 16, hipMemcpyDeviceToDevice, st));
}

int main(int argc, char** argv) {
  const long rows = 2048, row_vec = 151936 * 2 / 16;  // one micro-batch of bf16 logits
  const long bytes = rows * row_vec * 16;
  u32x4 *in, *out;
  CK(hipMalloc(&in, bytes));
  CK(hipMalloc(&out, bytes));
  CK(hipMemset(in, 1, bytes));
  CK(hipMemset(out, 0, bytes));
  std::vector<Variant> vs = {
      {"memcpyDtoD", run_memcpy},
      {"rows ntl1 nts0 u4", run_rows<true, false, 4>},
      {"rows ntl1 nts1 u4", run_rows<true, true, 4>},
      {"rows ntl0 nts0 u4", run_rows<false, false, 4>},
      {"rows ntl1 nts0 u8", run_rows<true, false, 8>},
      {"stride ntl1 nts0 u4 g2048", run_stride<true, false, 4, 2048>},
      {"stride ntl1 nts1 u4 g2048", run_stride<true, true, 4, 2048>},
      {"stride ntl0 nts0 u4 g2048", run_stride<false, false, 4, 2048>},
      {"stride ntl1 nts0 u1 g2048", run_stride<true, false, 1, 2048>},
      {"stride ntl1 nts0 u2 g2048", run_stride<true, false, 2, 2048>},
      {"stride ntl1 nts0 u8 g2048", run_stride<true, false, 8, 2048>},
      {"stride ntl1 nts0 u4 g1024", run_stride<true, false, 4, 1024>},
      {"stride ntl1 nts0 u4 g4096", run_stride<true, false, 4, 4096>},
      {"stride ntl1 nts0 u4 g512", run_stride<true, false, 4, 512>},
      {"stride ntl1 nts0 u2 g4096", run_stride<true, false, 2, 4096>},
      {"stride ntl1 nts0 u1 g8192", run_stride<true, false, 1, 8192>},
  };
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  std::vector<std::vector<float>> t(vs.size());
  for (int round = 0; round < 25; ++round) {
    for (size_t i = 0; i < vs.size(); ++i) {
      CK(hipEventRecord(a, st));
      vs[i].run(in, out, rows, row_vec, st);
      CK(hipEventRecord(b, st));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      if (round >= 3) t[i].push_back(ms);
    }
  }
  for (size_t i = 0; i < vs.size(); ++i) {
    std::sort(t[i].begin(), t[i].end());
    float med = t[i][t[i].size() / 2];
    printf("%-28s median %.4f ms  min %.4f ms  %7.0f GB/s (read+write)\n", vs[i].name, med, t[i][0], 2.0 * bytes / med / 1e6);
  }
  return 0;
}
