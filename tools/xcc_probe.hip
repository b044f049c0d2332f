// Which XCD does block b of a 64-thread-per-block grid land on?  (performance-only knowledge)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* xcc, int* cu) {
  unsigned x = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
  unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
  if (threadIdx.x == 0) { xcc[blockIdx.x] = x & 0xf; cu[blockIdx.x] = hw; }
  // burn a little time so that blocks overlap
  float a = threadIdx.x; for (int i = 0; i < 2000; ++i) a = a * 1.0001f + 0.5f; if (a == 123.f) xcc[0] = -1;
}
int main() {
  for (int threads : {64, 256}) {
    const int nb = 8192;
    int *dx, *dc; hipMalloc(&dx, nb * 4); hipMalloc(&dc, nb * 4);
    hipLaunchKernelGGL(k, dim3(nb), dim3(threads), 0, 0, dx, dc);
    std::vector<int> x(nb), c(nb);
    hipMemcpy(x.data(), dx, nb * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, nb * 4, hipMemcpyDeviceToHost);
    int match = 0; int hist[16] = {0};
    for (int b = 0; b < nb; ++b) { match += (x[b] == (b % 8)); hist[x[b] & 15]++; }
    printf("threads=%d: xcc == b%%8 for %d of %d blocks; hist:", threads, match, nb);
    for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
    printf("\n first 32 xcc:"); for (int b = 0; b < 32; ++b) printf(" %d", x[b]);
    printf("\n blocks 4096..4127:"); for (int b = 4096; b < 4128; ++b) printf(" %d", x[b]);
    printf("\n hw_id first 8: "); for (int b = 0; b < 64; b+=8) printf(" %08x", c[b]); printf("\n");
  }
  return 0;
}
