// class_tile_probe.hip -- k_class_tile against k_class_fused on a poisson27pt grid: rows left without a class, and whether the
// two kernels cut the rows into the same classes (B pass, then A pass on B's classes).
//   hipcc -O2 --offload-arch=gfx950 -I benchmark_spgemm_using_csr_amd/csrc -o /tmp/ctp tools/class_tile_probe.hip && /tmp/ctp 51
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define BHS_TILE_DEBUG 1
#include "bhs_kernels.hip.h"
#include "bhs_row_wg.hip.h"
#include "bhs_row_wave.hip.h"
#include "bhs_class.hip.h"
#include "bhs_class_fused.hip.h"
#include "bhs_class_tile.hip.h"
using namespace bhs;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 51;
    const int m = n * n * n;
    std::vector<int> rp(m + 1, 0), col;
    for (int z = 0; z < n; ++z) for (int y = 0; y < n; ++y) for (int x = 0; x < n; ++x) {
        const int r = (z * n + y) * n + x;
        for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
            const int X = x + dx, Y = y + dy, Z = z + dz;
            if (X < 0 || Y < 0 || Z < 0 || X >= n || Y >= n || Z >= n) continue;
            col.push_back((Z * n + Y) * n + X);
        }
        rp[r + 1] = (int)col.size();
    }
    if (argc > 2) {                                               // a band matrix with the grid's size: the same 27 offsets in every row (no heads but the pieces' first rows)
        col.clear();
        for (int r = 0; r < m; ++r) {
            for (int k = -13; k <= 13; ++k) { const long long c = (long long)r + (long long)k * 37; if (c >= 0 && c < m) col.push_back((int)c); }
            rp[r + 1] = (int)col.size();
        }
    }
    const long long nnz = (long long)col.size();
    int *dRp, *dRj, *dClsF, *dClsT, *dClsAF, *dClsAT, *dStats;
    unsigned long long* dTab;
    CK(hipMalloc(&dRp, (m + 1) * 4)); CK(hipMalloc(&dRj, nnz * 4 + 64));
    CK(hipMalloc(&dClsF, m * 4)); CK(hipMalloc(&dClsT, m * 4)); CK(hipMalloc(&dClsAF, m * 4)); CK(hipMalloc(&dClsAT, m * 4));
    CK(hipMalloc(&dStats, CS_INTS * 4)); CK(hipMalloc(&dTab, kClassSlots * 8));
    CK(hipMemcpy(dRp, rp.data(), (m + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dRj, col.data(), nnz * 4, hipMemcpyHostToDevice));
    auto run = [&](bool tile, bool isA, const int* cb, int* out) {
        CK(hipMemset(dStats, 0, CS_INTS * 4)); CK(hipMemset(dTab, 0xFF, kClassSlots * 8)); CK(hipMemset(out, 0xEE, m * 4));
        const int piece = tile ? 504 : 512;
        if (tile) {
            const long long per = (long long)(kClassTileBlock / 64) * piece;
            if (isA) hipLaunchKernelGGL((k_class_tile<true, 8, 4>), dim3((unsigned)((m + per - 1) / per)), dim3(kClassTileBlock), 0, 0, m, dRp, dRj, cb, out, dTab, dStats, nnz, piece, (const int*)nullptr);
            else hipLaunchKernelGGL((k_class_tile<false, 8, 4>), dim3((unsigned)((m + per - 1) / per)), dim3(kClassTileBlock), 0, 0, m, dRp, dRj, cb, out, dTab, dStats, nnz, piece, (const int*)nullptr);
        } else {
            const long long per = (long long)(kClassHeadsBlock / 64) * piece;
            if (isA) hipLaunchKernelGGL((k_class_fused<true, 8, 4>), dim3((unsigned)((m + per - 1) / per)), dim3(kClassHeadsBlock), 0, 0, m, dRp, dRj, cb, out, dTab, dStats, nnz, piece, (const int*)nullptr, 1);
            else hipLaunchKernelGGL((k_class_fused<false, 8, 4>), dim3((unsigned)((m + per - 1) / per)), dim3(kClassHeadsBlock), 0, 0, m, dRp, dRj, cb, out, dTab, dStats, nnz, piece, (const int*)nullptr, 1);
        }
        CK(hipDeviceSynchronize());
        std::vector<int> st(CS_INTS), o(m);
        {   // timed once more (tables as they are: every class is found at its first probe now -- the steady state of a block's cache is what counts)
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            const int piece2 = tile ? 567 : 512;
            if (tile) { const long long per = (long long)(kClassTileBlock / 64) * piece2;
                if (isA) hipLaunchKernelGGL((k_class_tile<true, 8, 4>), dim3((unsigned)((m + per - 1) / per)), dim3(kClassTileBlock), 0, 0, m, dRp, dRj, cb, out, dTab, dStats, nnz, piece2, (const int*)nullptr);
                else hipLaunchKernelGGL((k_class_tile<false, 8, 4>), dim3((unsigned)((m + per - 1) / per)), dim3(kClassTileBlock), 0, 0, m, dRp, dRj, cb, out, dTab, dStats, nnz, piece2, (const int*)nullptr);
            } else { const long long per = (long long)(kClassHeadsBlock / 64) * piece2;
                if (isA) hipLaunchKernelGGL((k_class_fused<true, 8, 4>), dim3((unsigned)((m + per - 1) / per)), dim3(kClassHeadsBlock), 0, 0, m, dRp, dRj, cb, out, dTab, dStats, nnz, piece2, (const int*)nullptr, 1);
                else hipLaunchKernelGGL((k_class_fused<false, 8, 4>), dim3((unsigned)((m + per - 1) / per)), dim3(kClassHeadsBlock), 0, 0, m, dRp, dRj, cb, out, dTab, dStats, nnz, piece2, (const int*)nullptr, 1);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("      [%.1f us]  ", ms * 1e3);
        }
        CK(hipMemcpy(st.data(), dStats, CS_INTS * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(o.data(), out, m * 4, hipMemcpyDeviceToHost));
        int neg = 0, firstNeg = -1;
        std::map<int, int> cnt;
        for (int i = 0; i < m; ++i) { if (o[i] < 0) { ++neg; if (firstNeg < 0) firstNeg = i; } cnt[o[i]]++; }
        printf("%s %s: flags %d heads %d, rows without a class %d (first %d), distinct classes %zu\n", tile ? "tile " : "fused", isA ? "A" : "B", st[CS_FLAGS], st[CS_HEADS], neg, firstNeg, cnt.size());
        return o;
    };
    auto same_partition = [&](const std::vector<int>& a, const std::vector<int>& b) {
        std::map<int, int> ab, ba; int bad = 0, firstBad = -1;
        for (int i = 0; i < m; ++i) {
            auto x = ab.find(a[i]); if (x == ab.end()) ab[a[i]] = b[i]; else if (x->second != b[i]) { ++bad; if (firstBad < 0) firstBad = i; }
            auto y = ba.find(b[i]); if (y == ba.end()) ba[b[i]] = a[i]; else if (y->second != a[i]) { ++bad; if (firstBad < 0) firstBad = i; }
        }
        printf("   partitions differ at %d rows (first %d: fused %d tile %d)\n", bad, firstBad, firstBad >= 0 ? a[firstBad] : 0, firstBad >= 0 ? b[firstBad] : 0);
    };
    auto bf = run(false, false, nullptr, dClsF);
    auto bt = run(true, false, nullptr, dClsT);
    same_partition(bf, bt);
    { int d[512]; CK(hipMemcpyFromSymbol(d, HIP_SYMBOL(g_tileDbg), sizeof(d)));
      for (int l = 0; l < 64; ++l) printf("lane %2d head %d cls %5d incl %8x len %2d c0 %6d prevlane %3d ok %d my %4d | fused %d tile %d\n", l, d[l], d[64+l], d[128+l], d[192+l], d[256+l], d[320+l], d[384+l], d[448+l], l ? bf[l-1] : -9, l ? bt[l-1] : -9); }
    auto af = run(false, true, dClsF, dClsAF);
    auto at = run(true, true, dClsF, dClsAT);
    same_partition(af, at);
    return 0;
}
