// vmm_alloc.hip -- measurement helper: a virtually contiguous device buffer whose physical chunks are mapped in SHUFFLED order
// (hipMemCreate per chunk, hipMemMap at a permuted place): does the ring kernel's time follow how scattered C's pages are?
//   hipcc -O2 -shared -fPIC -o gpurun_variants/libvmm.so tools/vmm_alloc.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>
extern "C" void* vmm_alloc(size_t bytes, size_t chunk, unsigned seed, int shuffle)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) { printf("granularity failed\n"); return nullptr; }
    if (chunk < gran) chunk = gran;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t n = (bytes + chunk - 1) / chunk, total = n * chunk;
    void* va = nullptr;
    if (hipMemAddressReserve(&va, total, 0, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); return nullptr; }
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    for (size_t i = 0; i < n; ++i)
        if (hipMemCreate(&h[i], chunk, &prop, 0) != hipSuccess) { printf("create %zu failed\n", i); return nullptr; }
    std::vector<size_t> perm(n);
    for (size_t i = 0; i < n; ++i) perm[i] = i;
    if (shuffle) { std::mt19937 g(seed); std::shuffle(perm.begin(), perm.end(), g); }
    for (size_t i = 0; i < n; ++i)
        if (hipMemMap((char*)va + perm[i] * chunk, chunk, 0, h[i], 0) != hipSuccess) { printf("map %zu failed\n", i); return nullptr; }
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(va, total, &acc, 1) != hipSuccess) { printf("access failed\n"); return nullptr; }
    printf("vmm_alloc: %zu chunks of %zu bytes (granularity %zu), shuffle %d\n", n, chunk, gran, shuffle);
    return va;
}
