// Does workgroup b of a kernel land on XCD (b + off) % 8 with ONE offset for the whole grid, whatever ran before it?
// (k_class_numeric deals super-runs to "XCD blockIdx.x & 7": the label does not matter, a uniform rotation does.)
// Launches a small kernel of G0 workgroups, then the probed grid of 3072 x 64 lanes, for several G0, same stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* xcc) {
  unsigned x = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
  if (threadIdx.x == 0 && xcc) xcc[blockIdx.x] = x & 0xf;
  float a = threadIdx.x; for (int i = 0; i < 4000; ++i) a = a * 1.0001f + 0.5f; if (a == 123.f && xcc) xcc[0] = -1;
}
int main() {
  const int nb = 3072;
  int* dx; hipMalloc(&dx, nb * 4);
  std::vector<int> x(nb);
  for (int threads : {64, 512})
  for (int g0 : {0, 1, 3, 5, 8, 13, 125, 256, 4096, 4099}) {
    for (int rep = 0; rep < 3; ++rep) {
      if (g0) hipLaunchKernelGGL(k, dim3(g0), dim3(threads), 0, 0, (int*)nullptr);
      hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, dx);
      hipMemcpy(x.data(), dx, nb * 4, hipMemcpyDeviceToHost);
      int best = 0, bestOff = 0;
      for (int off = 0; off < 8; ++off) { int c = 0; for (int b = 0; b < nb; ++b) c += x[b] == (b + off) % 8; if (c > best) { best = c; bestOff = off; } }
      int hist[8] = {0}; for (int b = 0; b < nb; ++b) hist[x[b] & 7]++;
      printf("before: %4d wg x %3d lanes  rep %d: best offset %d explains %d of %d; per-XCD counts:", g0, threads, rep, bestOff, best, nb);
      for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
      printf("\n");
    }
  }
  return 0;
}
