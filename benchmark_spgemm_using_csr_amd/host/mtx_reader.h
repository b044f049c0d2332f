// mtx_reader.h — Matrix Market coordinate reader -> CSR, written from the MM
// format specification.  Covers what the reference's two loaders accept
// (cusp::io::read_matrix_market_file at SpGEMM_cuda/main.cu:57-60 and the mmio
// based loader at SpGEMM_opencl/main.cpp:55-208): real / integer / pattern /
// complex (real part kept), general / symmetric / skew-symmetric / hermitian
// (mirrored entries expanded).  Unlike the OpenCL loader it column-sorts every
// row (the reference's CUDA driver does that in a second step, main.cu:62-64),
// which the long-row kernels rely on.
#ifndef BHSPARSE_AMD_MTX_READER_H
#define BHSPARSE_AMD_MTX_READER_H
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <string>
#include <vector>

#include "gallery.h"

inline int read_matrix_market(const char *path, CsrHost &A, std::string *errmsg = nullptr)
{
    auto fail = [&](const std::string &m) { if (errmsg) *errmsg = m; return -1; };
    FILE *f = fopen(path, "r");
    if (!f) return fail(std::string("cannot open ") + path);
    char line[1 << 12];
    if (!fgets(line, sizeof(line), f)) { fclose(f); return fail("empty file"); }
    std::string banner(line);
    for (auto &c : banner) c = (char)tolower((unsigned char)c);
    std::istringstream bs(banner);
    std::string tag, object, format, field, symmetry;
    bs >> tag >> object >> format >> field >> symmetry;
    if (tag != "%%matrixmarket" || object != "matrix" || format != "coordinate") {
        fclose(f);
        return fail("only '%%MatrixMarket matrix coordinate' files are supported");
    }
    const bool pattern = field == "pattern", complex_ = field == "complex";
    if (!(pattern || complex_ || field == "real" || field == "integer" || field == "double")) { fclose(f); return fail("unknown field " + field); }
    const bool sym = symmetry == "symmetric", skew = symmetry == "skew-symmetric", herm = symmetry == "hermitian";
    if (!(sym || skew || herm || symmetry == "general")) { fclose(f); return fail("unknown symmetry " + symmetry); }
    do { if (!fgets(line, sizeof(line), f)) { fclose(f); return fail("missing size line"); } } while (line[0] == '%' || line[0] == '\n');
    long long M, N, NZ;
    if (sscanf(line, "%lld %lld %lld", &M, &N, &NZ) != 3) { fclose(f); return fail("bad size line"); }
    struct Ent { int r, c; double v; };
    std::vector<Ent> e;
    e.reserve((size_t)NZ * ((sym || skew || herm) ? 2 : 1));
    for (long long t = 0; t < NZ; ++t) {
        if (!fgets(line, sizeof(line), f)) { fclose(f); return fail("unexpected end of file"); }
        char *p = line;
        const long long r = strtoll(p, &p, 10), c = strtoll(p, &p, 10);
        double v = 1.0;
        if (!pattern) v = strtod(p, &p);          // complex: real part kept, imaginary ignored
        if (r < 1 || r > M || c < 1 || c > N) { fclose(f); return fail("index out of range"); }
        e.push_back({(int)(r - 1), (int)(c - 1), v});
        if ((sym || skew || herm) && r != c) e.push_back({(int)(c - 1), (int)(r - 1), skew ? -v : v});
    }
    fclose(f);
    std::stable_sort(e.begin(), e.end(), [](const Ent &a, const Ent &b) { return a.r != b.r ? a.r < b.r : a.c < b.c; });
    A.num_rows = (int)M; A.num_cols = (int)N; A.num_entries = (int)e.size();
    A.row_offsets.assign(M + 1, 0);
    A.column_indices.resize(e.size());
    A.values.resize(e.size());
    for (size_t i = 0; i < e.size(); ++i) { A.row_offsets[e[i].r + 1]++; A.column_indices[i] = e[i].c; A.values[i] = e[i].v; }
    for (long long i = 0; i < M; ++i) A.row_offsets[i + 1] += A.row_offsets[i];
    return 0;
}
#endif
