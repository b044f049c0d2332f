// Does a wave's DS instruction wait behind its own LDS-direct load (global_load_lds) that is still in flight?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/glds_order_probe.hip -o gpurun_variants/glds_order_probe
// One wave per CU.  Per iteration: [a load of 16 bytes per lane from a fresh place in a big array: MODE 0 none, 1
// global_load_lds into LDS area A, 2 global_load_dwordx4 into registers, 3 a 8-byte store]; then a ds_read_b64 of LDS
// area B and s_waitcnt lgkmcnt(0); the cycles from before the load to behind that wait, and to behind a vmcnt(0).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

template <int MODE>
__global__ __launch_bounds__(64) void k_probe(const double* src, double* dst, long long n, unsigned long long* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) double buf[];               // A: [0, 128), B: [1024, 1088)
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) buf[i] = 1.0;
    __syncthreads();
    unsigned long long tLds = 0, tAll = 0;
    double sink = 0.0;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 r = {0.f, 0.f, 0.f, 0.f};
    const unsigned bAddr = 1024 * 8 + lane * 8;
    for (int it = 0; it < iters; ++it) {
        const long long at = (((long long)blockIdx.x * iters + it) * 4099 * 128) % (n - 256) + lane * 2;
        __builtin_amdgcn_s_waitcnt(0x0F70);
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (MODE == 1) __builtin_amdgcn_global_load_lds((glb_void*)(src + at), (lds_void*)&buf[0], 16, 0, 0);
        if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(src + at) : "memory");
        if (MODE == 3) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst + at), "v"(sink) : "memory");
        double v;
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(bAddr) : "memory");
        const unsigned long long t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_readcyclecounter();
        sink += v + r.x;
        tLds += t1 - t0;
        tAll += t2 - t0;
    }
    if (lane == 0) { atomicAdd(&out[0], tLds); atomicAdd(&out[1], tAll); }
    if (sink == 1234.5) dst[0] = sink;
}

template <int MODE>
void run(const char* name, const double* src, double* dst, long long n, unsigned long long* out, int cus)
{
    const int iters = 2000;
    (void)hipMemset(out, 0, 16);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(cus), dim3(64), 2048 * 8, 0, src, dst, n, out, iters);
    unsigned long long h[2];
    (void)hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("%-44s ds_read + lgkmcnt(0): %7.0f cycles   ... + vmcnt(0): %7.0f cycles\n", name, (double)h[0] / iters / cus, (double)h[1] / iters / cus);
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const long long n = 1ll << 28;                                            // 2 GB of doubles
    double *src, *dst;
    unsigned long long* out;
    (void)hipMalloc(&src, n * 8); (void)hipMalloc(&dst, n * 8); (void)hipMalloc(&out, 16);
    (void)hipMemset(src, 0, n * 8);
    for (int cus : {1, p.multiProcessorCount}) {
        printf("%d workgroups of one wave\n", cus);
        run<0>("no memory instruction", src, dst, n, out, cus);
        run<1>("global_load_lds_dwordx4 (LDS-direct)", src, dst, n, out, cus);
        run<2>("global_load_dwordx4 (to registers)", src, dst, n, out, cus);
        run<3>("global_store_dwordx2 sc1", src, dst, n, out, cus);
    }
    return 0;
}
