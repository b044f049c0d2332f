// Issue-rate probe for the VALU instructions on the per-product path (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/valu_rate_probe.hip -o /tmp/valu_rate_probe
// Every kernel runs ITER x 32 independent copies of one instruction per lane (8 chains x 4), on
// 4 waves per SIMD of every CU; the figure printed is SIMD cycles per wave64 instruction.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAINS 8
#define REP4(x) x x x x

#define PROBE(NAME, DECL, ASM, CONSTRAINT)                                                          \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, int iter, unsigned k)               \
    {                                                                                               \
        DECL v[CHAINS];                                                                             \
        for (int c = 0; c < CHAINS; ++c) v[c] = (DECL)(threadIdx.x + c + 1);                        \
        for (int i = 0; i < iter; ++i) {                                                            \
            _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) {                                    \
                REP4(asm volatile(ASM : "+" CONSTRAINT(v[c]) : "s"(k), "v"(threadIdx.x));)          \
            }                                                                                       \
        }                                                                                           \
        unsigned acc = 0;                                                                           \
        for (int c = 0; c < CHAINS; ++c) acc += (unsigned)v[c];                                     \
        if (acc == 0x12345678u) out[threadIdx.x] = acc;                                             \
    }

PROBE(k_add_u32, unsigned, "v_add_u32 %0, %1, %0", "v")
PROBE(k_mul_lo_u32, unsigned, "v_mul_lo_u32 %0, %0, %1", "v")
PROBE(k_mul_hi_u32, unsigned, "v_mul_hi_u32 %0, %0, %1", "v")
PROBE(k_mul_u32_u24, unsigned, "v_mul_u32_u24 %0, %0, %1", "v")
PROBE(k_mad_u32_u24, unsigned, "v_mad_u32_u24 %0, %0, %1, %2", "v")
PROBE(k_lshl_add_u32, unsigned, "v_lshl_add_u32 %0, %0, 2, %2", "v")
PROBE(k_bfe_u32, unsigned, "v_bfe_u32 %0, %0, 3, 9", "v")
PROBE(k_mbcnt_lo, unsigned, "v_mbcnt_lo_u32_b32 %0, %0, %2", "v")
PROBE(k_bcnt, unsigned, "v_bcnt_u32_b32 %0, %0, %2", "v")
PROBE(k_cmp_e64, unsigned, "v_cmp_ne_u32_e64 s[20:21], %0, %2\n v_add_u32 %0, 1, %0", "v")
PROBE(k_mul_f64, double, "v_mul_f64 %0, %0, %0", "v")
PROBE(k_fma_f64, double, "v_fma_f64 %0, %0, %0, %0", "v")
PROBE(k_add_f64, double, "v_add_f64 %0, %0, %0", "v")
PROBE(k_mul_f32, float, "v_mul_f32 %0, %0, %0", "v")
PROBE(k_mad_u64_u32, unsigned long long, "v_mad_u64_u32 %0, s[20:21], %1, %2, %0", "v")
PROBE(k_min_u32_dpp, unsigned, "v_min_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0x5", "v")
PROBE(k_mov_dpp, unsigned, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "v")
PROBE(k_cndmask, unsigned, "v_cndmask_b32 %0, %0, %2, vcc", "v")

template <typename K>
void run(const char* name, K kern, int instPerIter)
{
    int dev = 0;
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, dev);
    const int cus = p.multiProcessorCount, iter = 4096;
    unsigned* out;
    (void)hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    // 4 blocks of 256 threads per CU = 4 waves per SIMD
    hipLaunchKernelGGL(kern, dim3(cus * 4), dim3(256), 0, 0, out, 16, 3u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(cus * 4), dim3(256), 0, 0, out, iter, 3u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double clk = p.clockRate * 1e3;                               // Hz
    const double instPerSimd = 4.0 * iter * CHAINS * 4 * instPerIter;   // 4 waves per SIMD
    printf("%-18s %7.3f ms  %.2f cycles per wave instruction (clock %.0f MHz)\n", name, ms,
           ms * 1e-3 * clk / instPerSimd, clk / 1e6);
    (void)hipFree(out);
}

int main()
{
    run("v_add_u32", k_add_u32, 1);
    run("v_mul_lo_u32", k_mul_lo_u32, 1);
    run("v_mul_hi_u32", k_mul_hi_u32, 1);
    run("v_mul_u32_u24", k_mul_u32_u24, 1);
    run("v_mad_u32_u24", k_mad_u32_u24, 1);
    run("v_lshl_add_u32", k_lshl_add_u32, 1);
    run("v_bfe_u32", k_bfe_u32, 1);
    run("v_mbcnt_lo", k_mbcnt_lo, 1);
    run("v_bcnt_u32", k_bcnt, 1);
    run("v_cmp_e64+add", k_cmp_e64, 2);
    run("v_mul_f64", k_mul_f64, 1);
    run("v_fma_f64", k_fma_f64, 1);
    run("v_add_f64", k_add_f64, 1);
    run("v_mul_f32", k_mul_f32, 1);
    run("v_mad_u64_u32", k_mad_u64_u32, 1);
    run("v_min_u32_dpp", k_min_u32_dpp, 1);
    run("v_mov_b32_dpp", k_mov_dpp, 1);
    run("v_cndmask_b32", k_cndmask, 1);
    return 0;
}
