// The row kernel's load-all / store-all skeleton with a limit on how many workgroups may be in their store phase at
// a time (a counter in device memory: a workgroup takes a permit before its stores and gives it back once they are
// acknowledged).  Fewer rows written at a time means fewer write streams, which a plain fill rewards; does the
// skeleton gain, with the other compute units loading meanwhile?
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kChunks = 19;
constexpr long kRowVec = 18992;
__global__ __launch_bounds__(1024) void k_copy(const u32x4* __restrict__ in, u32x4* __restrict__ out, int* permits, int limit) {
  const long base = (long)blockIdx.x * kRowVec;
  u32x4 buf[kChunks];
  int took = 0;  // (thread 0 only)
#pragma unroll
  for (int j = 0; j < kChunks; ++j) {
    const long u = (long)j * 1024 + threadIdx.x;
    buf[j] = u < kRowVec ? __builtin_nontemporal_load(in + base + u) : u32x4{0u, 0u, 0u, 0u};
  }
  if (limit > 0) {
    unsigned keep = 0;
#pragma unroll
    for (int j = 0; j < kChunks; ++j) keep ^= buf[j].x;  // the loads have arrived
    if (threadIdx.x == 0 && keep != 0x7fffffffu) {
      // A permit is taken by compare-and-swap on the count of holders, so waiting workgroups never change the
      // count (an add-then-undo would: with 200 waiting, the count is above the limit all the time and nobody
      // gets in).  Every holder gives its permit back after a bounded time, and after 20000 looks a workgroup
      // stores without one, so every workgroup ends whatever the others do.
      for (int tries = 0; tries < 20000; ++tries) {
        const int cur = __hip_atomic_load(permits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur < limit && atomicCAS(permits, cur, cur + 1) == cur) {
          took = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < kChunks; ++j) {
    const long u = (long)j * 1024 + threadIdx.x;
    buf[j].x ^= 1u;
    if (u < kRowVec) __builtin_nontemporal_store(buf[j], out + base + u);
  }
  if (limit > 0) {
    __builtin_amdgcn_s_waitcnt(0x0070 | 0xc000 | 0x0f00);  // this wave's stores acknowledged
    __syncthreads();
    if (threadIdx.x == 0 && took) atomicSub(permits, 1);
  }
}
int main() {
  const long rows = 2048, bytes = rows * kRowVec * 16;
  const int sets = 4, rounds = 5;
  std::vector<u32x4*> in(sets), out(sets);
  for (int i = 0; i < sets; ++i) { CK(hipMalloc(&in[i], bytes)); CK(hipMalloc(&out[i], bytes)); CK(hipMemset(in[i], i + 1, bytes)); }
  int* permits; CK(hipMalloc(&permits, 256)); CK(hipMemset(permits, 0, 256));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int limits[] = {0, 16, 32, 48, 64, 96, 128, 192};
  const int nv = 8;
  std::vector<float> t[nv];
  for (int r = 0; r < rounds; ++r) for (int k = 0; k < nv; ++k) {
    const int v = (k * 5 + r * 3) % nv;
    CK(hipEventRecord(a));
    for (int i = 0; i < sets; ++i) hipLaunchKernelGGL(k_copy, dim3(rows), dim3(1024), 0, 0, in[i], out[i], permits, limits[v]);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (r > 0) t[v].push_back(ms / sets);
  }
  int left = -1; CK(hipMemcpy(&left, permits, 4, hipMemcpyDeviceToHost));
  for (int v = 0; v < nv; ++v) {
    std::sort(t[v].begin(), t[v].end());
    const float med = t[v][t[v].size() / 2];
    printf("at most %3d workgroups storing: median %.4f ms  %6.0f GB/s (read + write)\n", limits[v], med, 2.0 * bytes / med / 1e6);
  }
  printf("permits out at the end: %d\n", left);
  return 0;
}
