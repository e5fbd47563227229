// Does the memory system move a row-sized load-all / store-all stream faster when the compute units work in
// step?  Left alone, about half of the 256 units read while the other half write, so every memory channel sees
// a fine mix of reads and writes.  Here every workgroup (one 152 K-entry bf16 row in the registers of 1024
// threads, as in the single-pass row kernel) waits at its start until the device's constant-rate clock reaches
// a multiple of `period` ticks (unless it is at most `window` ticks late), so all units read together and then
// write together.  Also timed: the same structure reading only, and writing only.
//   hipcc --offload-arch=gfx950 -O3 -o phase_gate_bench tools/phase_gate_bench.hip && ./phase_gate_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kChunks = 19;
constexpr long kRowVec = 18992;  // 151936 bf16 = 18992 16-byte units

__device__ inline void gate(unsigned period, unsigned window) {
  if (period == 0) return;
  const unsigned long long t = wall_clock64();
  const unsigned r = (unsigned)(t % period);
  if (r < window) return;
  const unsigned long long target = t - r + period;  // at most one period away: every wave leaves this loop
  while (wall_clock64() < target) __builtin_amdgcn_s_sleep(1);
}

// mode 0: load the row, store the row; 1: load only; 2: store only
template <int MODE>
__global__ __launch_bounds__(1024) void k_row(const u32x4* __restrict__ in, u32x4* __restrict__ out, long rows,
                                              unsigned period, unsigned window, unsigned store_at) {
  const long row = blockIdx.x;
  if (row >= rows) return;
  const long base = row * kRowVec;
  gate(period, window);
  u32x4 buf[kChunks];
#pragma unroll
  for (int j = 0; j < kChunks; ++j) {
    const long u = (long)j * 1024 + threadIdx.x;
    if (MODE != 2) buf[j] = u < kRowVec ? __builtin_nontemporal_load(in + base + u) : u32x4{0u, 0u, 0u, 0u};
    else buf[j] = u32x4{(unsigned)u, 1u, 2u, 3u};
  }
  if (MODE == 1) {
    unsigned acc = 0;
#pragma unroll
    for (int j = 0; j < kChunks; ++j) acc ^= buf[j].x ^ buf[j].y ^ buf[j].z ^ buf[j].w;
    if (acc == 0x12345u) out[base + threadIdx.x] = buf[0];  // never true for the test pattern
    return;
  }
  if (MODE == 0 && store_at != 0) {
    // the stores wait for the half-period mark as well (only when the loads have come in before it)
    unsigned keep = 0;
#pragma unroll
    for (int j = 0; j < kChunks; ++j) keep ^= buf[j].x;  // forces the loads to have arrived
    const unsigned long long t = wall_clock64();
    const unsigned r = (unsigned)(t % period);
    if (r < store_at && keep != 0x7fffffffu) {
      const unsigned long long target = t - r + store_at;
      while (wall_clock64() < target) __builtin_amdgcn_s_sleep(1);
    }
  }
#pragma unroll
  for (int j = 0; j < kChunks; ++j) {
    const long u = (long)j * 1024 + threadIdx.x;
    buf[j].x ^= 1u;
    if (u < kRowVec) __builtin_nontemporal_store(buf[j], out + base + u);
  }
}

int main(int argc, char** argv) {
  const long rows = 2048, bytes = rows * kRowVec * 16;
  const int sets = 4, rounds = 9;
  std::vector<u32x4*> in(sets), out(sets);
  for (int i = 0; i < sets; ++i) {
    CK(hipMalloc(&in[i], bytes));
    CK(hipMalloc(&out[i], bytes));
    CK(hipMemset(in[i], i + 1, bytes));
  }
  int rate_khz = 0;
  CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
  printf("wall clock %d kHz\n", rate_khz);
  const double ticks_per_us = rate_khz / 1000.0;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  struct Variant { int mode; double period_us, window_us, store_us; };
  std::vector<Variant> vs = {{0, 0, 0, 0}, {1, 0, 0, 0}, {2, 0, 0, 0}};
  for (double p = 18; p <= 28.01; p += 1.0) vs.push_back({0, p, 1.5, 0});
  for (double p = 20; p <= 26.01; p += 2.0) vs.push_back({0, p, 1.5, p * 0.5});
  for (double p = 20; p <= 26.01; p += 2.0) vs.push_back({0, p, 4.0, 0});
  std::vector<std::vector<float>> t(vs.size());
  for (int r = 0; r < rounds; ++r) {
    for (size_t k = 0; k < vs.size(); ++k) {
      const size_t v = (k * 7 + r * 3) % vs.size();
      const unsigned period = (unsigned)(vs[v].period_us * ticks_per_us), window = (unsigned)(vs[v].window_us * ticks_per_us),
                     store_at = (unsigned)(vs[v].store_us * ticks_per_us);
      CK(hipEventRecord(a));
      for (int i = 0; i < sets; ++i) {
        if (vs[v].mode == 0) hipLaunchKernelGGL(k_row<0>, dim3(rows), dim3(1024), 0, 0, in[i], out[i], rows, period, window, store_at);
        if (vs[v].mode == 1) hipLaunchKernelGGL(k_row<1>, dim3(rows), dim3(1024), 0, 0, in[i], out[i], rows, period, window, store_at);
        if (vs[v].mode == 2) hipLaunchKernelGGL(k_row<2>, dim3(rows), dim3(1024), 0, 0, in[i], out[i], rows, period, window, store_at);
      }
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      if (r > 0) t[v].push_back(ms / sets);
    }
  }
  for (size_t v = 0; v < vs.size(); ++v) {
    std::sort(t[v].begin(), t[v].end());
    const float med = t[v][t[v].size() / 2];
    const double moved = (vs[v].mode == 0 ? 2.0 : 1.0) * bytes;
    printf("mode %d period %5.1f us window %4.1f us stores at %5.1f us: median %.4f ms  min %.4f ms  %6.0f GB/s\n", vs[v].mode,
           vs[v].period_us, vs[v].window_us, vs[v].store_us, med, t[v][0], moved / med / 1e6);
  }
  return 0;
}
