// mtx_reader.h — Matrix Market coordinate reader -> CSR, written from the MM
// format specification.  Covers what the reference's two loaders accept
// (cusp::io::read_matrix_market_file at SpGEMM_cuda/main.cu:57-60 and the mmio
// based loader at SpGEMM_opencl/main.cpp:55-208): real / integer / pattern /
// complex (real part kept), general / symmetric / skew-symmetric / hermitian
// (mirrored entries expanded).  Unlike the OpenCL loader it column-sorts every
// row (the reference's CUDA driver does that in a second step, main.cu:62-64),
// which the long-row kernels rely on.
// Round 6: the file is read in one piece and its lines are parsed by all host threads (a SuiteSparse file of a few million
// entries used to spend seconds in fgets / strtod / one std::stable_sort over all entries -- more than the multiply it feeds
// by three orders of magnitude); entries go to their rows by a counting sort (file order kept inside a row: duplicates stay
// in the order the file has them) and every row is column-sorted by the threads, rows dealt round robin in blocks.
#ifndef BHSPARSE_AMD_MTX_READER_H
#define BHSPARSE_AMD_MTX_READER_H
#include <algorithm>
#include <atomic>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "gallery.h"

namespace bhs_mtx {
struct Ent { int r, c; double v; };
inline int host_threads()
{
    if (const char *e = getenv("BHS_HOST_THREADS")) { const int t = atoi(e); if (t > 0) return t; }
    const unsigned hc = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(hc ? hc : 1u, 32u));
}
template <typename F>
inline void parallel_for(int parts, F &&f)
{
    if (parts <= 1) { f(0); return; }
    std::vector<std::thread> th;
    th.reserve(parts - 1);
    for (int t = 1; t < parts; ++t) th.emplace_back([&f, t]() { f(t); });
    f(0);
    for (auto &x : th) x.join();
}
}  // namespace bhs_mtx

// sort_rows: column-sort every row (default).  false: the rows keep the file's order -- for callers that hand them to
// bhs_csr_sort_indices_device (include/bhsparse_hip.h) instead.
inline int read_matrix_market(const char *path, CsrHost &A, std::string *errmsg = nullptr, bool sort_rows = true)
{
    using bhs_mtx::Ent;
    auto fail = [&](const std::string &m) { if (errmsg) *errmsg = m; return -1; };
    FILE *f = fopen(path, "rb");
    if (!f) return fail(std::string("cannot open ") + path);
    std::vector<char> buf;
    {
        if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return fail("cannot seek"); }
        const long sz = ftell(f);
        if (sz < 0) { fclose(f); return fail("cannot tell the file's size"); }
        rewind(f);
        buf.resize((size_t)sz + 1);
        const size_t got = fread(buf.data(), 1, (size_t)sz, f);
        fclose(f);
        buf.resize(got + 1);
        buf[got] = '\n';                                   // (a last line without a newline still ends)
    }
    const char *p = buf.data(), *end = buf.data() + buf.size();
    auto next_line = [&](const char *q) { const char *n = (const char *)memchr(q, '\n', (size_t)(end - q)); return n ? n + 1 : end; };
    if (buf.size() <= 1) return fail("empty file");
    const char *l1 = next_line(p);
    std::string banner(p, l1);
    for (auto &c : banner) c = (char)tolower((unsigned char)c);
    std::istringstream bs(banner);
    std::string tag, object, format, field, symmetry;
    bs >> tag >> object >> format >> field >> symmetry;
    if (tag != "%%matrixmarket" || object != "matrix" || format != "coordinate")
        return fail("only '%%MatrixMarket matrix coordinate' files are supported");
    const bool pattern = field == "pattern", complex_ = field == "complex";
    if (!(pattern || complex_ || field == "real" || field == "integer" || field == "double")) return fail("unknown field " + field);
    const bool sym = symmetry == "symmetric", skew = symmetry == "skew-symmetric", herm = symmetry == "hermitian";
    if (!(sym || skew || herm || symmetry == "general")) return fail("unknown symmetry " + symmetry);
    p = l1;
    for (;;) {                                             // comments and blank lines in front of the size line
        if (p >= end) return fail("missing size line");
        if (*p != '%' && *p != '\n' && *p != '\r') break;
        p = next_line(p);
    }
    long long M, N, NZ;
    {
        const char *l = next_line(p);
        std::string sl(p, l);
        if (sscanf(sl.c_str(), "%lld %lld %lld", &M, &N, &NZ) != 3) return fail("bad size line");
        p = l;
    }
    if (M < 0 || N < 0 || NZ < 0 || M > 0x7ffffffeLL || N > 0x7ffffffeLL) return fail("bad size line");
    // ---- the entry lines, in as many pieces as there are threads (cut at line ends), each piece parsed by its thread
    const int T = (int)std::max<long long>(1, std::min<long long>(bhs_mtx::host_threads(), NZ / 65536 + 1));
    std::vector<const char *> cut(T + 1);
    cut[0] = p; cut[T] = end;
    for (int t = 1; t < T; ++t) {
        const char *q = p + (size_t)((end - p) / T) * t;
        cut[t] = q >= end ? end : next_line(q);
    }
    for (int t = 1; t <= T; ++t) cut[t] = std::max(cut[t], cut[t - 1]);
    const bool mirror = sym || skew || herm;
    std::vector<std::vector<Ent>> part(T);
    std::vector<long long> lines(T, 0);
    std::atomic<int> bad(0);                               // 1 index out of range, 2 malformed line
    bhs_mtx::parallel_for(T, [&](int t) {
        std::vector<Ent> &e = part[t];
        e.reserve((size_t)(NZ / T + 16) * (mirror ? 2 : 1));
        const char *q = cut[t], *qe = cut[t + 1];
        long long n = 0;
        while (q < qe) {
            const char *le = (const char *)memchr(q, '\n', (size_t)(qe - q));
            if (!le) le = qe;
            while (q < le && (*q == ' ' || *q == '\t' || *q == '\r')) ++q;
            if (q < le) {
                char *r = nullptr;
                const long long rr = strtoll(q, &r, 10);
                if (r == q) { bad = 2; break; }
                const char *q2 = r;
                const long long cc = strtoll(q2, &r, 10);
                if (r == q2) { bad = 2; break; }
                double v = 1.0;
                if (!pattern) v = strtod(r, &r);          // complex: real part kept, imaginary ignored
                if (rr < 1 || rr > M || cc < 1 || cc > N) { bad = 1; break; }
                e.push_back({(int)(rr - 1), (int)(cc - 1), v});
                if (mirror && rr != cc) e.push_back({(int)(cc - 1), (int)(rr - 1), skew ? -v : v});
                ++n;
            }
            q = le + 1;
        }
        lines[t] = n;
    });
    if (bad == 1) return fail("index out of range");
    if (bad == 2) return fail("malformed entry line");
    long long seen = 0;
    for (int t = 0; t < T; ++t) seen += lines[t];
    if (seen < NZ) return fail("unexpected end of file");
    // (entries beyond the announced count are ignored, as a reader that stops after NZ lines ignores them)
    if (seen > NZ) {
        long long extra = seen - NZ;
        for (int t = T - 1; t >= 0 && extra > 0; --t)
            while (extra > 0 && !part[t].empty() && lines[t] > 0) {
                const Ent last = part[t].back();
                part[t].pop_back();
                if (mirror && !part[t].empty() && last.r != last.c && part[t].back().r == last.c && part[t].back().c == last.r) part[t].pop_back();
                --lines[t]; --extra;
            }
    }
    size_t total = 0;
    for (int t = 0; t < T; ++t) total += part[t].size();
    if (total > 0x7fffffffULL) return fail("more entries than int32 index_type holds");
    // ---- to rows: counting sort (stable: file order inside a row -- thread t's piece of the file lies in front of thread
    // t + 1's), every thread counts and scatters its own piece; then every row by column
    A.num_rows = (int)M; A.num_cols = (int)N; A.num_entries = (int)total;
    A.row_offsets.assign((size_t)M + 1, 0);
    A.column_indices.resize(total);
    A.values.resize(total);
    {
        std::vector<std::vector<int>> at(T);                 // at[t][r]: entries of row r in piece t, then where piece t's first entry of row r goes
        bhs_mtx::parallel_for(T, [&](int t) {
            at[t].assign((size_t)M, 0);
            for (const Ent &x : part[t]) at[t][x.r]++;
        });
        for (long long i = 0; i < M; ++i) {
            int rowTotal = 0;
            for (int t = 0; t < T; ++t) rowTotal += at[t][i];
            A.row_offsets[i + 1] = A.row_offsets[i] + rowTotal;
        }
        bhs_mtx::parallel_for(T, [&](int t) {              // (rows [lo, hi) of every piece's table: exclusive sums across the pieces)
            const long long lo = M * t / T, hi = M * (t + 1) / T;
            for (long long i = lo; i < hi; ++i) {
                int run = A.row_offsets[i];
                for (int u = 0; u < T; ++u) { const int c = at[u][i]; at[u][i] = run; run += c; }
            }
        });
        bhs_mtx::parallel_for(T, [&](int t) {
            for (const Ent &x : part[t]) { const int k = at[t][x.r]++; A.column_indices[k] = x.c; A.values[k] = x.v; }
            std::vector<Ent>().swap(part[t]);
            std::vector<int>().swap(at[t]);
        });
    }
    if (sort_rows && total > 0) {
        const int TS = (int)std::max<long long>(1, std::min<long long>(bhs_mtx::host_threads(), M / 4096 + 1));
        std::atomic<long long> nextBlock(0);
        bhs_mtx::parallel_for(TS, [&](int) {
            std::vector<std::pair<int, double>> tmp;
            for (;;) {
                const long long b0 = nextBlock.fetch_add(1024);
                if (b0 >= M) break;
                const long long b1 = std::min<long long>(M, b0 + 1024);
                for (long long i = b0; i < b1; ++i) {
                    const int s0 = A.row_offsets[i], s1 = A.row_offsets[i + 1];
                    bool sorted = true;
                    for (int k = s0 + 1; k < s1 && sorted; ++k) sorted = A.column_indices[k - 1] <= A.column_indices[k];
                    if (sorted) continue;
                    tmp.resize((size_t)(s1 - s0));
                    for (int k = s0; k < s1; ++k) tmp[k - s0] = {A.column_indices[k], (double)A.values[k]};
                    std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<int, double> &a, const std::pair<int, double> &b) { return a.first < b.first; });
                    for (int k = s0; k < s1; ++k) { A.column_indices[k] = tmp[k - s0].first; A.values[k] = tmp[k - s0].second; }
                }
            }
        });
    }
    return 0;
}
#endif
