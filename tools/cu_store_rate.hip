// How fast can ONE compute unit write 128 KB (a 256 x 256 bf16 tile of a GEMM's output) when G compute units do
// it at the same time?  One workgroup of 512 threads per unit writes `iters` tiles (whole 128-byte rows of a
// row-major matrix with a 304 KB row pitch, 16 bytes per lane), each wave 16 store instructions per tile,
// waiting for its stores after every tile.  Prints microseconds per tile for G = 1 .. 256.
//     hipcc --offload-arch=gfx950 -O3 -o tools/bin/cu_store_rate tools/cu_store_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool NT>
__global__ __launch_bounds__(512) void k_tiles(u32x4* out, long pitch_vec, int iters, long tiles_per_row) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wr = wid >> 2, wc = wid & 3;
  const u32x4 v = {(unsigned)threadIdx.x, 1u, 2u, 3u};
  for (int it = 0; it < iters; ++it) {
    const long tile = (long)blockIdx.x + (long)it * gridDim.x;
    const long m0 = (tile / tiles_per_row) * 256, n0 = (tile % tiles_per_row) * 256;  // elements
    u32x4* base = out + (m0 + wr * 128 + (lane >> 3)) * pitch_vec + (n0 + wc * 64) / 8 + (lane & 7);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (NT) __builtin_nontemporal_store(v, base + (long)(8 * j) * pitch_vec);
      else base[(long)(8 * j) * pitch_vec] = v;
    }
    __builtin_amdgcn_s_waitcnt(0x0070 | 0xc000 | 0x0f00);  // vmcnt(0)
  }
}

int main() {
  const long V = 151936, T = 16384, pitch_vec = V / 8, tiles_per_row = V / 256;
  u32x4* out;
  CK(hipMalloc(&out, (size_t)T * V * 2));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int nt = 0; nt < 2; ++nt)
    for (int g : {1, 8, 32, 64, 128, 256}) {
      const int iters = 64;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        if (nt) hipLaunchKernelGGL(k_tiles<true>, dim3(g), dim3(512), 0, 0, out, pitch_vec, iters, tiles_per_row);
        else hipLaunchKernelGGL(k_tiles<false>, dim3(g), dim3(512), 0, 0, out, pitch_vec, iters, tiles_per_row);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep == 2)
          printf("%s stores, %3d units: %.2f us per 128 KB tile per unit (%.0f GB/s per unit, %.2f TB/s in all)\n",
                 nt ? "non-temporal" : "plain", g, ms * 1e3 / iters, 131072.0 / (ms * 1e-3 / iters) / 1e9,
                 131072.0 * g / (ms * 1e-3 / iters) / 1e12);
      }
    }
  return 0;
}
