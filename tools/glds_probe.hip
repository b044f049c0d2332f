// global_load_lds_dwordx4 on gfx950: does it take a source address that is only 8-byte aligned, partial exec masks,
// and what lands where?   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/glds_probe.hip -o gpurun_variants/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

__global__ __launch_bounds__(64) void k_probe(const double* src, double* out, int shift, int nact)
{
    __shared__ __attribute__((aligned(16))) double buf[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) buf[i] = -1.0;
    __syncthreads();
    // two instructions: elements [shift, shift + 128) and [shift + 128, shift + 256) -> buf[0, 256), lanes >= nact off in the second
    const double* g0 = src + shift + lane * 2;
    __builtin_amdgcn_global_load_lds((glb_void*)g0, (lds_void*)&buf[0], 16, 0, 0);
    if (lane < nact) __builtin_amdgcn_global_load_lds((glb_void*)(g0 + 128), (lds_void*)&buf[128], 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = buf[i];
}

int main()
{
    const int n = 4096;
    std::vector<double> h(n);
    for (int i = 0; i < n; ++i) h[i] = i;
    double *d, *o;
    (void)hipMalloc(&d, n * 8); (void)hipMalloc(&o, 256 * 8);
    (void)hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    for (int shift : {0, 1, 3, 16, 17}) {
        for (int nact : {64, 10}) {
            hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, o, shift, nact);
            std::vector<double> r(256);
            hipError_t e = hipMemcpy(r.data(), o, 256 * 8, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int i = 0; i < 256; ++i) {
                const double want = (i < 128 || (i - 128) / 2 < nact) ? shift + i : -1.0;
                bad += r[i] != want;
            }
            printf("shift %2d (source %s-byte aligned), %2d lanes in the 2nd load: %s (%d wrong; r[0]=%g r[1]=%g r[128]=%g r[255]=%g) err=%d\n",
                   shift, shift % 2 ? "8" : "16", nact, bad ? "MISMATCH" : "ok", bad, r[0], r[1], r[128], r[255], (int)e);
        }
    }
    return 0;
}
