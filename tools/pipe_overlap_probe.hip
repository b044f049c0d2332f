// Do a CU's LDS pipe and its vector-memory pipe work side by side?  16 one-wave workgroups per CU; a wave is an "LDS wave"
// (ds_read_b64 / ds_write_b64 in a loop), a "store wave" (global_store_dwordx4 nt, streaming), a "load wave"
// (global_load_lds_dwordx4, streaming) or a "VALU wave" (dependent fma chain), by its block number.  Time of the mixes against the
// parts alone.   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pipe_overlap_probe.hip -o gpurun_variants/pipe_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

// role: 0 idle, 1 LDS, 2 store, 3 LDS-direct load, 4 VALU
__global__ __launch_bounds__(64) void k_mix(int roleEven, int roleOdd, int itersLds, int itersMem, int itersValu, double* buf, long long perWave, double* sink)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];      // 8 KB
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) lds[i] = 1.0;
    __syncthreads();
    const int role = (blockIdx.x >> 3) & 1 ? roleOdd : roleEven;      // (blocks b, b + 8, .. share an XCD; alternate per CU slot)
    double acc = 0.0;
    if (role == 1) {
        const unsigned a = (unsigned)(((lane * 0x9E3779B1u) >> 22) & 1023) * 8;
        for (int it = 0; it < itersLds; ++it) {
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                double v;
                asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"((a + u * 640) & 8191) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                acc += v;
            }
#pragma unroll
            for (int u = 0; u < 6; ++u) asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(lane * 8 + u * 512)), "v"(acc) : "memory");
        }
    } else if (role == 2) {
        double* p = buf + (long long)blockIdx.x * perWave;
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2 v = {1.0, 2.0};
        for (int it = 0; it < itersMem; ++it) {
            double* q = p + ((long long)it * 128) % (perWave - 128) + lane * 2;
            asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(q), "v"(v) : "memory");
        }
    } else if (role == 3) {
        const double* p = buf + (long long)blockIdx.x * perWave;
        for (int it = 0; it < itersMem; ++it) {
            const double* q = p + ((long long)it * 128) % (perWave - 128) + lane * 2;
            __builtin_amdgcn_global_load_lds((glb_void*)q, (lds_void*)&lds[(it & 3) * 128], 16, 0, 0);
            if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (role == 4) {
        double x = lane * 1e-9;
        for (int it = 0; it < itersValu; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) x = __builtin_fma(x, 1.0000001, 1e-12);
        }
        acc = x;
    }
    if (acc == 1234.5678) sink[0] = acc;
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, grid = cus * 16;
    const long long perWave = 1 << 16;                                 // doubles per wave: 512 KB; 4096 waves: 2 GB
    double *buf, *sink;
    (void)hipMalloc(&buf, (size_t)grid * perWave * 8);
    (void)hipMalloc(&sink, 64);
    (void)hipMemset(buf, 0, (size_t)grid * perWave * 8);
    auto run = [&](const char* name, int re, int ro, int il, int im, int iv) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        hipLaunchKernelGGL(k_mix, dim3(grid), dim3(64), 8192, 0, re, ro, il / 8 + 1, im / 8 + 1, iv / 8 + 1, buf, perWave, sink);
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k_mix, dim3(grid), dim3(64), 8192, 0, re, ro, il, im, iv, buf, perWave, sink);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        printf("%-58s %7.3f ms\n", name, ms);
    };
    const int IL = 3000, IM = 6000, IV = 4000;
    run("8 LDS waves per CU (the others idle)", 1, 0, IL, IM, IV);
    run("8 store waves per CU", 2, 0, IL, IM, IV);
    run("8 LDS-direct load waves per CU", 3, 0, IL, IM, IV);
    run("8 VALU waves per CU", 4, 0, IL, IM, IV);
    run("8 LDS + 8 store waves", 1, 2, IL, IM, IV);
    run("8 LDS + 8 LDS-direct load waves", 1, 3, IL, IM, IV);
    run("8 LDS + 8 VALU waves", 1, 4, IL, IM, IV);
    run("8 store + 8 VALU waves", 2, 4, IL, IM, IV);
    run("8 store + 8 LDS-direct load waves", 2, 3, IL, IM, IV);
    run("16 LDS waves", 1, 1, IL, IM, IV);
    run("16 store waves", 2, 2, IL, IM, IV);
    return 0;
}
