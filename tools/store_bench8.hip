// Store-only in the row kernel's shape, continued: does the ORDER in which a compute unit's 16 waves emit their
// stores matter?  (a) today's map with a barrier behind every round, so that a unit emits its row as 19 contiguous
// 16 KB blocks one after the other; (b) a wave owning 19 contiguous KB with the waves storing one after the other,
// so that a unit emits its row strictly front to back; (c) today's map, waves in 8 / 16 turns.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kChunks = 19;
constexpr long kRowVec = 18992;
#define WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0070 | 0xc000 | 0x0f00)
template <int MODE, bool NT>
__global__ __launch_bounds__(1024) void k_store(u32x4* __restrict__ out) {
  const long base = (long)blockIdx.x * kRowVec;
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
  auto put = [&](long u, unsigned tag) {
    const u32x4 v = u32x4{(unsigned)u, tag, 2u, (unsigned)blockIdx.x};
    if (u < kRowVec) { if (NT) __builtin_nontemporal_store(v, out + base + u); else out[base + u] = v; }
  };
  if (MODE == 0) {  // today: everything at once
#pragma unroll
    for (int j = 0; j < kChunks; ++j) put((long)j * 1024 + threadIdx.x, j);
  } else if (MODE == 1) {  // a barrier behind every round
#pragma unroll
    for (int j = 0; j < kChunks; ++j) {
      put((long)j * 1024 + threadIdx.x, j);
      __syncthreads();
    }
  } else if (MODE == 2) {  // a barrier behind every round, and the round's stores acknowledged before it
#pragma unroll
    for (int j = 0; j < kChunks; ++j) {
      put((long)j * 1024 + threadIdx.x, j);
      WAIT_VM0();
      __syncthreads();
    }
  } else if (MODE == 3) {  // a wave owns 19 KB, waves one after the other (issue order only)
    for (int turn = 0; turn < 16; ++turn) {
      if (wave == turn) {
#pragma unroll
        for (int j = 0; j < kChunks; ++j) put((long)wave * (kChunks * 64) + j * 64 + lane, j);
      }
      __syncthreads();
    }
  } else {  // today's map, waves in (MODE) turns
    for (int turn = 0; turn < MODE; ++turn) {
      if (wave % MODE == turn) {
#pragma unroll
        for (int j = 0; j < kChunks; ++j) put((long)j * 1024 + threadIdx.x, j);
        WAIT_VM0();
      }
      __syncthreads();
    }
  }
}
int main() {
  const long rows = 2048, bytes = rows * kRowVec * 16;
  const int sets = 6, rounds = 7;
  std::vector<u32x4*> out(sets);
  for (int i = 0; i < sets; ++i) { CK(hipMalloc(&out[i], bytes)); CK(hipMemset(out[i], i + 1, bytes)); }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const char* names[] = {"all at once, plain", "all at once, NT", "barrier per round, plain", "barrier per round, NT", "barrier + acknowledged per round, plain",
                         "wave owns 19 KB, one wave after the other, plain", "wave owns 19 KB, one wave after the other, NT", "waves in 8 turns", "waves in 16 turns"};
  const int nv = 9;
  std::vector<float> t[nv];
  for (int r = 0; r < rounds; ++r) for (int k = 0; k < nv; ++k) {
    const int v = (k * 5 + r * 3) % nv;
    CK(hipEventRecord(a));
    for (int i = 0; i < sets; ++i) {
      switch (v) {
        case 0: hipLaunchKernelGGL((k_store<0, false>), dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 1: hipLaunchKernelGGL((k_store<0, true>), dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 2: hipLaunchKernelGGL((k_store<1, false>), dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 3: hipLaunchKernelGGL((k_store<1, true>), dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 4: hipLaunchKernelGGL((k_store<2, false>), dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 5: hipLaunchKernelGGL((k_store<3, false>), dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 6: hipLaunchKernelGGL((k_store<3, true>), dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 7: hipLaunchKernelGGL((k_store<8, false>), dim3(rows), dim3(1024), 0, 0, out[i]); break;
        case 8: hipLaunchKernelGGL((k_store<16, false>), dim3(rows), dim3(1024), 0, 0, out[i]); break;
      }
    }
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (r > 0) t[v].push_back(ms / sets);
  }
  for (int v = 0; v < nv; ++v) {
    std::sort(t[v].begin(), t[v].end());
    const float med = t[v][t[v].size() / 2];
    printf("%-52s median %.4f ms  %6.0f GB/s\n", names[v], med, bytes / med / 1e6);
  }
  return 0;
}
