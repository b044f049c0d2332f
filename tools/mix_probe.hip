// mix_probe.hip -- what the memory system sustains for the ring kernel's MIX of traffic: 3 parts read, 5 parts written (1.9 GB in, 3.2 GB
// out), streaming, next to plain copy (1:1), read-only and write-only.   hipcc -O3 --offload-arch=gfx950 -o /tmp/mix tools/mix_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
// every workgroup walks its own contiguous piece: R vectors read per W vectors written
template <int R, int W, bool NT>
__global__ __launch_bounds__(256) void k_mix(const f4* __restrict__ in, f4* __restrict__ out, long long unitsPerBlock)
{
    const long long u0 = (long long)blockIdx.x * unitsPerBlock;
    f4 acc = {0, 0, 0, 0};
    for (long long u = 0; u < unitsPerBlock; ++u) {
        const long long b = (u0 + u) * 256 + threadIdx.x;
#pragma unroll
        for (int r = 0; r < R; ++r) acc += in[((u0 + u) * R + r) * 256 + threadIdx.x];
#pragma unroll
        for (int w = 0; w < W; ++w) {
            f4 v = acc + (float)w;
            if (NT) __builtin_nontemporal_store(v, &out[((u0 + u) * W + w) * 256 + threadIdx.x]);
            else out[((u0 + u) * W + w) * 256 + threadIdx.x] = v;
        }
    }
}
template <int R, int W, bool NT>
void run(const char* name, f4* in, f4* out, long long bytesTotal)
{
    const int blocks = 256 * 16;
    const long long unitBytes = 256LL * 16 * (R + W);
    const long long units = bytesTotal / unitBytes / blocks;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_mix<R, W, NT>), dim3(blocks), dim3(256), 0, 0, in, out, units);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it && ms < best) best = ms;
    }
    const double moved = (double)units * blocks * unitBytes;
    printf("%-28s %5.2f GB read %5.2f GB written  %.3f ms  %.0f GB/s\n", name, moved * R / (R + W) / 1e9, moved * W / (R + W) / 1e9, best, moved / best / 1e6);
}
int main()
{
    const long long cap = 6LL << 30;
    f4 *in, *out; CK(hipMalloc(&in, cap)); CK(hipMalloc(&out, cap));
    CK(hipMemset(in, 0, cap)); CK(hipMemset(out, 0, cap));
    run<1, 1, false>("copy 1:1", in, out, 5100000000LL);
    run<1, 1, true>("copy 1:1, nt stores", in, out, 5100000000LL);
    run<3, 5, false>("mix 3:5", in, out, 5100000000LL);
    run<3, 5, true>("mix 3:5, nt stores", in, out, 5100000000LL);
    run<1, 0, false>("read only", in, out, 5100000000LL);
    run<0, 1, false>("write only", in, out, 5100000000LL);
    run<0, 1, true>("write only, nt", in, out, 5100000000LL);
    return 0;
}
