// Micro-benchmark: throughput of CSR-row-shaped gathers through one CU's vector memory path.
// Each wave reads "rows" of RUN consecutive elements at pseudo-random row starts inside a window,
// 64 lanes covering 64/lanes_per_row rows per instruction, BYTES per lane per load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int BYTES, int RUN, int ELEM>   // RUN elements of ELEM bytes per row, BYTES loaded per lane
__global__ __launch_bounds__(64) void k(const char* __restrict__ base, size_t windowBytes, int iters, double* out, int contiguous) {
  const int lane = threadIdx.x;
  constexpr int EPL = BYTES / ELEM;                 // elements per lane per load
  constexpr int LPR = (RUN + EPL - 1) / EPL;        // lanes per row
  constexpr int RPI = 64 / LPR;                     // rows per instruction
  const int r = lane / LPR, t = lane % LPR;
  unsigned long long seed = (blockIdx.x * 1315423911u) ^ 0x9E3779B97F4A7C15ull;
  double acc = 0;
  size_t nrows = 1; while (nrows * 2 * (RUN * ELEM) <= windowBytes) nrows *= 2;   // power of two: mask, no division
  const size_t mask = nrows - 1;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      seed = seed * 6364136223846793005ull + 1442695040888963407ull;
      size_t row = contiguous ? (((size_t)(blockIdx.x * 977 + it * 8 + u) * RPI + r) & mask) : (((seed >> 20) + (size_t)r * 7919) & mask);
      const char* p = base + row * (RUN * ELEM) + (size_t)t * BYTES;
      if (r < RPI && t * EPL < RUN) {
        if constexpr (BYTES == 4) { acc += *(const float*)p; }
        else if constexpr (BYTES == 8) { float2 v; __builtin_memcpy(&v, p, 8); acc += v.x + v.y; }
        else { float4 v; __builtin_memcpy(&v, p, 16); acc += v.x + v.y + v.z + v.w; }
      }
    }
  }
  if (acc == 123.456) out[0] = acc;
}
template <int BYTES, int RUN, int ELEM>
void run(const char* d, size_t window, const char* name, int contiguous) {
  double* out; hipMalloc(&out, 8);
  const int grid = 256 * 32, iters = 200;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<BYTES, RUN, ELEM>), dim3(grid), dim3(64), 0, 0, d, window, 10, out, contiguous);
  hipEventRecord(a); hipLaunchKernelGGL((k<BYTES, RUN, ELEM>), dim3(grid), dim3(64), 0, 0, d, window, iters, out, contiguous); hipEventRecord(b);
  hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
  constexpr int EPL = BYTES / ELEM; constexpr int LPR = (RUN + EPL - 1) / EPL; constexpr int RPI = 64 / LPR;
  double bytes = (double)grid * iters * 8 * RPI * RUN * ELEM;
  double instr = (double)grid * iters * 8;
  printf("%-44s window %8zu KB: %8.1f GB/s useful, %6.1f cycles/instr/CU (2.1GHz), %.2f ms\n", name, window >> 10, bytes / ms / 1e6,
         ms * 1e-3 * 2.1e9 / (instr / 256), ms);
}
int main() {
  size_t big = 1ull << 30; char* d; hipMalloc(&d, big); hipMemset(d, 0, big);
  for (size_t w : {(size_t)16 << 20}) {
    run<4, 27, 4>(d, w, "cols: 4B/lane, runs of 27 ints", 0);
    run<8, 27, 4>(d, w, "cols: 8B/lane (2 ints, 4B-aligned), runs of 27", 0);
    run<16, 27, 4>(d, w, "cols: 16B/lane (4 ints, 4B-aligned), runs of 27", 0);
    run<8, 27, 8>(d, w, "vals: 8B/lane, runs of 27 doubles", 0);
    run<16, 27, 8>(d, w, "vals: 16B/lane (2 doubles, 8B-aligned), runs of 27", 0);
    run<16, 28, 8>(d, w, "vals: 16B/lane, runs of 28 doubles (16B-aligned)", 0);
  }
  return 0;
}
