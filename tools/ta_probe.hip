// Texture-addresser cost of buffer loads whose lanes mostly need nothing (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ta_probe.hip -o /tmp/ta_probe
// Every wave issues ITER x 16 independent buffer_load_dwordx2 from a small, cache-resident array, 8 waves per SIMD on
// every CU.  Modes: 0 all 64 lanes read; 1 14 lanes read, the other 50 carry an offset outside the buffer;
// 2 14 lanes read, the others are switched off in the exec mask; 3 27 lanes read (exec mask); 4 / 5: 16 / 4 bytes per lane.
// Printed: CU cycles per wave-level load instruction.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(const double* base, unsigned bytes, double* out, int iter)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, bytes, 0x00020000);
    const int lane = threadIdx.x & 63;
    const int active = MODE == 0 ? 64 : (MODE == 3 ? 27 : 14);
    int voff = (lane % 27) * 8;
    if (MODE == 1 && lane >= active) voff = -1;
    double acc = 0.0;
    for (int i = 0; i < iter; ++i) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int soff = ((i * 16 + u) * 216 + (blockIdx.x & 63) * 4096) & 0xFFFF8;
            v[u] = 0.0;
            if (MODE == 4) {
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                const v4u q = __builtin_amdgcn_raw_buffer_load_b128(r, voff * 2, soff, 0);
                v[u] = __builtin_bit_cast(double, ((unsigned long long)q.y << 32) | q.x) + __builtin_bit_cast(double, ((unsigned long long)q.w << 32) | q.z);
            } else if (MODE == 5) {
                v[u] = (double)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
            } else if (MODE < 2 || lane < active)
                v[u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
    }
    if (acc == 12345.678) out[threadIdx.x] = acc;
}

template <int MODE>
void run(const char* name, const double* d, unsigned bytes, double* out)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int iter = 2000, blocks = p.multiProcessorCount * 8;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(256), 0, 0, d, bytes, out, 10);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(256), 0, 0, d, bytes, out, iter);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double loadsPerCU = (double)iter * 16 * 4 * 8;            // waves per block 4, blocks per CU 8
    printf("%-44s %.3f ms  %.2f CU cycles per wave load (at %.2f GHz)\n", name, ms,
           ms * 1e-3 * p.clockRate * 1e3 / loadsPerCU, p.clockRate * 1e-6);
}

int main()
{
    double *d, *out;
    const unsigned bytes = 1u << 20;
    (void)hipMalloc(&d, bytes); (void)hipMemset(d, 0, bytes);
    (void)hipMalloc(&out, 4096);
    run<0>("all 64 lanes read", d, bytes, out);
    run<1>("14 lanes read, 50 outside the buffer", d, bytes, out);
    run<2>("14 lanes read, 50 off in exec", d, bytes, out);
    run<3>("27 lanes read, 37 off in exec", d, bytes, out);
    run<4>("all 64 lanes read 16 bytes (dwordx4)", d, bytes, out);
    run<5>("all 64 lanes read 4 bytes (dword)", d, bytes, out);
    return 0;
}
