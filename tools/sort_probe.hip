// Hardware check of wave_flip_sort_u32 (bhs_wave.hip.h) against std::sort.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I benchmark_spgemm_using_csr_amd/csrc tools/sort_probe.hip -o /tmp/sort_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
#include "bhs_wave.hip.h"

template <int E, int GW>
__global__ void k_sort(unsigned* data, int nblocks)
{
    const int lane = threadIdx.x & 63;
    unsigned* d = data + (size_t)blockIdx.x * 64 * E;
    unsigned x[E];
#pragma unroll
    for (int e = 0; e < E; ++e) x[e] = d[lane * E + e];
    bhs::wave_flip_sort_u32<E, GW>(x, lane);
#pragma unroll
    for (int e = 0; e < E; ++e) d[lane * E + e] = x[e];
}

template <int E, int GW>
int run(const char* name, int mode)
{
    const int nb = 2048, n = nb * 64 * E;
    std::vector<unsigned> h(n), ref;
    std::mt19937 rng(1234 + E * 7 + GW);
    for (auto& v : h) {
        if (mode == 0) v = rng();
        else if (mode == 1) v = rng() % 7;            // many duplicates
        else v = 0xffffffffu - (rng() % 3);           // near the padding value
    }
    ref = h;
    const int seg = GW * E;
    for (int i = 0; i < n; i += seg) std::sort(ref.begin() + i, ref.begin() + i + seg);
    unsigned* d;
    hipMalloc(&d, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k_sort<E, GW>), dim3(nb), dim3(64), 0, 0, d, nb);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    hipFree(d);
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += h[i] != ref[i];
    printf("%-14s mode %d: %s (%d mismatches of %d)\n", name, mode, bad ? "FAIL" : "ok", bad, n);
    return bad != 0;
}

int main()
{
    int f = 0;
    for (int mode = 0; mode < 3; ++mode) {
        f += run<1, 64>("E=1 GW=64", mode);
        f += run<2, 64>("E=2 GW=64", mode);
        f += run<4, 64>("E=4 GW=64", mode);
        f += run<8, 64>("E=8 GW=64", mode);
        f += run<16, 64>("E=16 GW=64", mode);
        f += run<1, 16>("E=1 GW=16", mode);
        f += run<4, 16>("E=4 GW=16", mode);
    }
    printf(f ? "SORT PROBE FAILED\n" : "SORT PROBE PASSED\n");
    return f;
}
