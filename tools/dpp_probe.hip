// Verifies the semantics of the DPP / permlane-swap cross-lane helpers on real gfx950 hardware.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../benchmark_spgemm_using_csr_amd/csrc/bhs_wave.hip.h"
using namespace bhs;
__global__ void k(unsigned* o) {
  int lane = threadIdx.x;
  unsigned x = lane * 3 + 1;
  o[lane] = lane_xor<1>(x, lane);
  o[64 + lane] = lane_xor<2>(x, lane);
  o[128 + lane] = lane_xor<4>(x, lane);
  o[192 + lane] = lane_xor<8>(x, lane);
  o[256 + lane] = lane_xor<16>(x, lane);
  o[320 + lane] = lane_xor<32>(x, lane);
  o[384 + lane] = (unsigned)wave_incl_scan_dpp((int)x);
  o[448 + lane] = (unsigned)wave_sum_dpp((int)x);
  unsigned long long z = ((unsigned long long)(lane * 7 + 5) << 32) | (unsigned)(1000 - lane);
  unsigned long long w = lane_xor64<16>(z, lane);
  o[512 + lane] = (unsigned)(w >> 32); o[576 + lane] = (unsigned)w;
}
int main() {
  unsigned *d, h[640];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  const int xs[6] = {1, 2, 4, 8, 16, 32};
  for (int t = 0; t < 6; ++t)
    for (int l = 0; l < 64; ++l) if (h[t * 64 + l] != (unsigned)((l ^ xs[t]) * 3 + 1)) { if (bad < 10) printf("xor%d lane %d got %u\n", xs[t], l, h[t*64+l]); ++bad; }
  unsigned acc = 0;
  for (int l = 0; l < 64; ++l) { acc += l * 3 + 1; if (h[384 + l] != acc) { if (bad < 10) printf("scan lane %d got %u want %u\n", l, h[384+l], acc); ++bad; } }
  for (int l = 0; l < 64; ++l) if (h[448 + l] != acc) { if (bad < 10) printf("sum lane %d got %u want %u\n", l, h[448+l], acc); ++bad; }
  for (int l = 0; l < 64; ++l) { int p = l ^ 16; if (h[512 + l] != (unsigned)(p * 7 + 5) || h[576 + l] != (unsigned)(1000 - p)) ++bad; }
  printf(bad ? "DPP PROBE FAILED (%d)\n" : "DPP PROBE OK\n", bad);
  return bad != 0;
}
