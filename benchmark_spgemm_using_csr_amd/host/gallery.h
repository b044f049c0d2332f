// gallery.h — Poisson stencil matrices in CSR (pattern + unit values), replacing
// the reference driver's calls to cusp::gallery::poisson{5,9,7,27}pt
// (SpGEMM_cuda/main.cu:30-53).  Grid index = x + nx*(y + ny*z); rows sorted.
// Same construction as the Python gallery (benchmark_spgemm_using_csr_amd/gallery.py).
#ifndef BHSPARSE_AMD_GALLERY_H
#define BHSPARSE_AMD_GALLERY_H
#include <cstdint>
#include <string>
#include <vector>

struct CsrHost {
    int num_rows = 0, num_cols = 0, num_entries = 0;
    std::vector<int> row_offsets, column_indices;
    std::vector<double> values;
};

inline bool gallery_poisson(const std::string &name, int nx, int ny, int nz, CsrHost &A)
{
    std::vector<int> dx, dy, dz;
    auto add = [&](int a, int b, int c) { dx.push_back(a); dy.push_back(b); dz.push_back(c); };
    if (name == "poisson5pt") { nz = 1; add(0,-1,0); add(-1,0,0); add(0,0,0); add(1,0,0); add(0,1,0); }
    else if (name == "poisson9pt") { nz = 1; for (int b = -1; b <= 1; ++b) for (int a = -1; a <= 1; ++a) add(a,b,0); }
    else if (name == "poisson7pt") { add(0,0,-1); add(0,-1,0); add(-1,0,0); add(0,0,0); add(1,0,0); add(0,1,0); add(0,0,1); }
    else if (name == "poisson27pt") { for (int c = -1; c <= 1; ++c) for (int b = -1; b <= 1; ++b) for (int a = -1; a <= 1; ++a) add(a,b,c); }
    else return false;
    const long long m = (long long)nx * ny * nz;
    A.num_rows = A.num_cols = (int)m;
    A.row_offsets.assign(m + 1, 0);
    A.column_indices.clear();
    A.column_indices.reserve((size_t)m * dx.size());
    for (int z = 0; z < nz; ++z)
        for (int y = 0; y < ny; ++y)
            for (int x = 0; x < nx; ++x) {
                const long long row = x + (long long)nx * (y + (long long)ny * z);
                for (size_t t = 0; t < dx.size(); ++t) {
                    const int X = x + dx[t], Y = y + dy[t], Z = z + dz[t];
                    if (X < 0 || X >= nx || Y < 0 || Y >= ny || Z < 0 || Z >= nz) continue;
                    A.column_indices.push_back((int)(X + (long long)nx * (Y + (long long)ny * Z)));
                }
                A.row_offsets[row + 1] = (int)A.column_indices.size();
            }
    A.num_entries = (int)A.column_indices.size();
    A.values.assign(A.num_entries, 1.0);
    return true;
}

// deterministic stand-in for `rand()%9+1` with srand(time(NULL)) (main.cu:79-94):
// 1 + (lcg(seed, i) % 9), identical to gallery.fill_values in the Python package
inline void fill_values(std::vector<double> &v, uint64_t seed = 20140519ull, uint64_t offset = 0)
{
    for (size_t i = 0; i < v.size(); ++i) {
        const uint64_t x = (offset + i + seed) * 6364136223846793005ull + 1442695040888963407ull;
        v[i] = (double)(1 + ((x >> 33) % 9));
    }
}
#endif
