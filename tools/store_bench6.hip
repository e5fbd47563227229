// Store-only, continued: many sequential streams (where the write rate is low), but every workgroup writes its
// 4 KB as PIECES that lie a stride apart inside its stream instead of next to each other.  If the pieces of one
// DRAM page sit a fixed stride apart in the address space, then at that stride a workgroup fills whole pages at
// once and the rate of a single stream should come back.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
// Everything is a power of two, so that the index arithmetic is shifts (with divisions the kernel, not the
// memory, sets the rate).  lg_streams, lg_piece (16-byte units per piece), lg_tile (workgroups sharing one tile).
__global__ __launch_bounds__(256) void k_fill_pieces(u32x4* __restrict__ out, int lg_blocks, int lg_streams, int lg_piece,
                                                     int lg_tile) {
  const unsigned i = blockIdx.x;
  const unsigned s = i & ((1u << lg_streams) - 1), k = i >> lg_streams;  // stream, and the workgroup's turn inside it
  const int lg_per = lg_blocks - lg_streams;                              // 4 KB blocks per stream
  const unsigned tile = k >> lg_tile, slot = k & ((1u << lg_tile) - 1);
  const unsigned p = threadIdx.x >> lg_piece, within = threadIdx.x & ((1u << lg_piece) - 1);
  // a tile is 2^lg_tile blocks = 2^(lg_tile + 8) units; the workgroup's pieces lie 2^(lg_tile + lg_piece) units apart
  const unsigned long unit = ((unsigned long)s << (lg_per + 8)) + ((unsigned long)tile << (lg_tile + 8)) +
                             ((unsigned long)p << (lg_tile + lg_piece)) + (slot << lg_piece) + within;
  out[unit] = u32x4{i, 1u, 2u, (unsigned)threadIdx.x};
}
int main() {
  const int lg_blocks = 18;  // 262144 blocks of 4 KB = 1 GiB
  const long n_blocks = 1L << lg_blocks, bytes = n_blocks * 4096;
  const int sets = 5, rounds = 7;
  std::vector<u32x4*> out(sets);
  for (int i = 0; i < sets; ++i) { CK(hipMalloc(&out[i], bytes)); CK(hipMemset(out[i], i + 1, bytes)); }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  struct V { int lg_streams, lg_piece, lg_tile; };
  std::vector<V> vs = {{0, 8, 0}, {3, 8, 0}, {5, 8, 0}, {6, 8, 0}, {8, 8, 0}};
  for (int ls : {8, 6})
    for (int lp : {4, 5, 6})           // pieces of 256 B, 512 B, 1 KB
      for (int lt = 1; lt <= 10 && lt <= lg_blocks - ls; ++lt) vs.push_back({ls, lp, lt});
  std::vector<std::vector<float>> t(vs.size());
  for (int r = 0; r < rounds; ++r) for (size_t k = 0; k < vs.size(); ++k) {
    const size_t v = (k * 7 + r * 3) % vs.size();
    CK(hipEventRecord(a));
    for (int i = 0; i < sets; ++i)
      hipLaunchKernelGGL(k_fill_pieces, dim3(n_blocks), dim3(256), 0, 0, out[i], lg_blocks, vs[v].lg_streams, vs[v].lg_piece, vs[v].lg_tile);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (r > 0) t[v].push_back(ms / sets);
  }
  for (size_t v = 0; v < vs.size(); ++v) {
    std::sort(t[v].begin(), t[v].end());
    const float med = t[v][t[v].size() / 2];
    printf("%4d streams, pieces of %4d B every %8ld B: median %.4f ms  %6.0f GB/s\n", 1 << vs[v].lg_streams, 16 << vs[v].lg_piece,
           16L << (vs[v].lg_tile + vs[v].lg_piece), med, bytes / med / 1e6);
  }
  return 0;
}
