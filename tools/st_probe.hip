// Measurement helper: how many L1 -> L2 write requests do 8-byte and 16-byte stores per lane make?
//   hipcc --offload-arch=gfx950 -O3 -o st_probe st_probe.hip ; rocprofv3 --kernel-trace --pmc TCP_TCC_WRITE_REQ_sum -- ./st_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void st_b64(double* p, long n) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = 1.0; }
__global__ void st_b128(double2* p, long n) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n / 2) p[i] = make_double2(1.0, 2.0); }
__global__ void st_b128_sc1(double2* p, long n)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f v = {1.f, 2.f, 3.f, 4.f};
    if (i < n / 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p + i), "v"(v) : "memory");
}
__global__ void st_b64_sc1(double* p, long n)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double v = 1.0;
    if (i < n) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p + i), "v"(v) : "memory");
}
__global__ void st_b32(int* p, long n) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = 1; }
__global__ void st_b64x2(int2* p, long n) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n / 2) p[i] = make_int2(1, 2); }
int main()
{
    const long n = 1L << 26;
    double* d; hipMalloc(&d, n * 8 + 64);
    for (int rep = 0; rep < 2; ++rep) {
        st_b64<<<n / 256, 256>>>(d, n);
        st_b128<<<n / 512, 256>>>((double2*)d, n);
        st_b64_sc1<<<n / 256, 256>>>(d, n);
        st_b128_sc1<<<n / 512, 256>>>((double2*)d, n);
        st_b32<<<n / 256, 256>>>((int*)d, n);
        st_b64x2<<<n / 512, 256>>>((int2*)d, n);
        // misaligned by 8 bytes: every lane's 16 bytes straddle a 16-byte boundary
        st_b128<<<n / 512, 256>>>((double2*)(d + 1), n);
    }
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
