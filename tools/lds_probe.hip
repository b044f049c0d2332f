// Cost of LDS wave instructions on gfx950, in CU cycles per instruction, at 8 waves per SIMD on every CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lds_probe.hip -o gpurun_variants/lds_probe
// What the numeric class kernel needs to know: a ds_add_f64 costs ~33 cycles whatever the number of active lanes
// (round 2); what do plain b64 writes with few active lanes cost, b64 / b128 reads, and b32 reads?
#include <hip/hip_runtime.h>
#include <cstdio>

enum { RD64, RD64_RAND, RD32, RD128, WR64, WR64_8, WR64_2, WR128, ADD64, ADD64_8, ADD64_2, ADD64_SAME, RD64_BCAST, NMODES };
static const char* kNames[NMODES] = {
    "ds_read_b64, 64 lanes, consecutive", "ds_read_b64, 64 lanes, scattered (hash)", "ds_read_b32, 64 lanes, consecutive",
    "ds_read_b128, 64 lanes, consecutive", "ds_write_b64, 64 lanes, consecutive", "ds_write_b64, 8 lanes active (exec)",
    "ds_write_b64, 2 lanes active (exec)", "ds_write_b128, 64 lanes, consecutive", "ds_add_f64, 64 lanes, distinct",
    "ds_add_f64, 8 lanes active", "ds_add_f64, 2 lanes active", "ds_add_f64, 64 lanes, 16 distinct addresses",
    "ds_read_b64, 64 lanes, 4 distinct addresses"};

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(double* out, int iter)
{
    __shared__ __attribute__((aligned(16))) double lds[2048];                  // 16 KB per block, 8 blocks per CU
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = 0.0;
    __syncthreads();
    unsigned addr = (unsigned)(size_t)(&lds[wv * 512]) ;                      // LDS byte address of the wave's 8 KB
    addr = (unsigned)__builtin_amdgcn_readfirstlane((int)addr);
    unsigned a;
    if (MODE == RD64_RAND) a = addr + ((lane * 0x9E3779B1u >> 22) & 511) * 8;
    else if (MODE == RD128 || MODE == WR128) a = addr + lane * 16;
    else if (MODE == RD32) a = addr + lane * 4;
    else if (MODE == ADD64_SAME) a = addr + (lane & 15) * 8;
    else if (MODE == RD64_BCAST) a = addr + (lane & 3) * 8;
    else a = addr + lane * 8;
    const double one = 1.0;
    double sink = 0.0;
    const bool act = (MODE == WR64_8 || MODE == ADD64_8) ? lane < 8 : (MODE == WR64_2 || MODE == ADD64_2) ? lane < 2 : true;
    for (int i = 0; i < iter; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int off = (u & 7) * 512;                                    // (immediate offsets: no address VALU)
            if (MODE == RD64 || MODE == RD64_RAND || MODE == RD64_BCAST) {
                double v;
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(off));
                if (u == 15) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); sink += v; }
            } else if (MODE == RD32) {
                float v;
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(off));
                if (u == 15) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); sink += v; }
            } else if (MODE == RD128) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                f4 v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"((u & 3) * 1024));
                if (u == 15) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); sink += v.x; }
            } else if (MODE == WR128) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                f4 v = {1.f, 2.f, 3.f, 4.f};
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a), "v"(v), "n"((u & 3) * 1024) : "memory");
            } else if (MODE == WR64 || MODE == WR64_8 || MODE == WR64_2) {
                if (act) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a), "v"(one), "n"(off) : "memory");
            } else {
                if (act) asm volatile("ds_add_f64 %0, %1 offset:%2" ::"v"(a), "v"(one), "n"(off) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (sink == 12345.678) out[threadIdx.x] = sink + lds[lane];
}

template <int MODE>
void run(double* out, const hipDeviceProp_t& p)
{
    const int iter = 2000, blocks = p.multiProcessorCount * 8;              // 8 blocks x 4 waves = 8 waves per SIMD
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, iter);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double perCU = (double)iter * 16 * 4 * 8;
    printf("%-48s %7.3f ms  %6.2f CU cycles per wave instruction\n", kNames[MODE], ms, ms * 1e-3 * p.clockRate * 1e3 / perCU);
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs, %.2f GHz\n", p.name, p.multiProcessorCount, p.clockRate * 1e-6);
    double* out;
    (void)hipMalloc(&out, 4096);
    run<RD64>(out, p); run<RD64_RAND>(out, p); run<RD64_BCAST>(out, p); run<RD32>(out, p); run<RD128>(out, p);
    run<WR64>(out, p); run<WR64_8>(out, p); run<WR64_2>(out, p); run<WR128>(out, p);
    run<ADD64>(out, p); run<ADD64_8>(out, p); run<ADD64_2>(out, p); run<ADD64_SAME>(out, p);
    return 0;
}
