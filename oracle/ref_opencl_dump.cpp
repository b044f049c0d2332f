// TEST INFRASTRUCTURE (oracle side) - not part of the product library.
//
// A small dump driver around the REFERENCE's own OpenCL implementation of the path
// (/root/reference/SpGEMM_opencl/{bhsparse,bhsparse_opencl,basiccl}.cpp, compiled UNMODIFIED from where
// they lie by oracle/Makefile's `_ref` target; nothing of the reference is copied into this repository).
// It runs one C = A * B through the reference's public class in the order its own driver does
// (SpGEMM_opencl/main.cpp:233-262: initPlatform -> initData -> warmup x3 -> spgemm -> get_nnzC -> get_C)
// and writes C to a flat binary file, so that the CPU restatement in ref_spgemm_oracle.c can be pinned
// against outputs of the reference itself (tests/golden/ref_opencl_*.npz, tests/test_oracle.py).
//
// Deliberately NOT called: free_mem() / freePlatform() - the reference's OpenCL free_mem frees
// never-allocated host pointers in non-host-mem mode (bhsparse_opencl.cpp:1217-1244); the process
// leaves through _exit once C is on disk.
//
// The reference opens its six .cl kernel files by bare name (bhsparse_opencl.cpp:108-120), so the binary
// has to run from a directory that holds them: oracle/_ref/ (git-ignored), staged there by the recipe.
//
// usage: ref_opencl_spgemm <in.bin> <out.bin | - | =>  ("-": no dump, only the reference's own timing lines --
//                                                          bench.py's `reference_opencl` leg reads "SpGEMM time";
//                                                          "=": no dump, C is fetched and its digests are printed --
//                                                          the same sums as make_ref_golden.digest_of, bench.py and
//                                                          tests/test_full_size_gpu.py compare the HIP result with them)
//   in.bin : int32 m,k,n,nnzA,nnzB | rowPtrA[m+1] colIndA[nnzA] | rowPtrB[k+1] colIndB[nnzB] | f64 valA[nnzA] valB[nnzB]
//   out.bin: int32 nnzC | rowPtrC[m+1] colIndC[nnzC] | f64 valC[nnzC]
#include "bhsparse.h"

#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <vector>

static bool rd(FILE *f, void *p, size_t bytes) { return bytes == 0 || fread(p, 1, bytes, f) == bytes; }
static bool wr(FILE *f, const void *p, size_t bytes) { return bytes == 0 || fwrite(p, 1, bytes, f) == bytes; }

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
    FILE *fi = fopen(argv[1], "rb");
    if (!fi) { perror(argv[1]); return 2; }
    int hdr[5];
    if (!rd(fi, hdr, sizeof hdr)) { fprintf(stderr, "short header\n"); return 2; }
    const int m = hdr[0], k = hdr[1], n = hdr[2], nnzA = hdr[3], nnzB = hdr[4];
    // +1 element everywhere: the reference creates cl buffers of exactly nnz elements and a zero-byte
    // clCreateBuffer fails; callers never send empty matrices here, the slack only keeps data() non-null.
    std::vector<int> rpA(m + 1), ciA(nnzA + 1), rpB(k + 1), ciB(nnzB + 1), rpC(m + 1, 0);
    std::vector<double> vA(nnzA + 1), vB(nnzB + 1);
    bool ok = rd(fi, rpA.data(), sizeof(int) * (m + 1)) && rd(fi, ciA.data(), sizeof(int) * nnzA) &&
              rd(fi, rpB.data(), sizeof(int) * (k + 1)) && rd(fi, ciB.data(), sizeof(int) * nnzB) &&
              rd(fi, vA.data(), sizeof(double) * nnzA) && rd(fi, vB.data(), sizeof(double) * nnzB);
    fclose(fi);
    if (!ok) { fprintf(stderr, "short input\n"); return 2; }

    bool platforms[NUM_PLATFORMS];
    for (int i = 0; i < NUM_PLATFORMS; i++) platforms[i] = false;
    platforms[BHSPARSE_OPENCL] = true;                       // main.cpp:330-337 (-opencl)

    bhsparse *bh = new bhsparse();
    int err = bh->initPlatform(platforms);
    if (err != BHSPARSE_SUCCESS) { fprintf(stderr, "initPlatform error %d\n", err); return 10; }
    err = bh->initData(m, k, n, nnzA, vA.data(), rpA.data(), ciA.data(),
                       nnzB, vB.data(), rpB.data(), ciB.data(), rpC.data(), false);
    if (err != BHSPARSE_SUCCESS) { fprintf(stderr, "initData error %d\n", err); return 11; }
    for (int i = 0; i < 3; i++) {
        err = bh->warmup();
        if (err != BHSPARSE_SUCCESS) { fprintf(stderr, "warmup error %d\n", err); return 12; }
    }
    err = bh->spgemm();
    if (err != BHSPARSE_SUCCESS) { fprintf(stderr, "spgemm error %d\n", err); return 13; }
    const int nnzC = bh->get_nnzC();
    if (argv[2][0] == '-' && argv[2][1] == 0) {
        printf("ref_opencl_spgemm: m=%d k=%d n=%d nnzA=%d nnzB=%d -> nnzC=%d\n", m, k, n, nnzA, nnzB, nnzC);
        fflush(stdout);
        _exit(0);
    }
    std::vector<int> ciC((size_t)nnzC + 1);
    std::vector<double> vC((size_t)nnzC + 1);
    err = bh->get_C(ciC.data(), vC.data());                  // also re-reads rowPtrC (bhsparse_opencl.cpp:1182-1185)
    if (err != BHSPARSE_SUCCESS) { fprintf(stderr, "get_C error %d\n", err); return 14; }

    if (argv[2][0] == '=' && argv[2][1] == 0) {
        // digests of the reference's C as it left the device (rows are not re-sorted: `rows_sorted` says whether
        // every row was ascending); integer-valued inputs keep every sum below 2^53, i.e. exact in any order
        unsigned long long sumRp = 0, wsumCol = 0;
        double sumVal = 0.0, wsumVal = 0.0;
        int sorted = 1;
        for (int i = 0; i <= m; i++) sumRp += (unsigned long long)rpC[i];
        for (int i = 0; i < m; i++)
            for (int p = rpC[i] + 1; p < rpC[i + 1]; p++) if (ciC[p] <= ciC[p - 1]) sorted = 0;
        for (long long p = 0; p < nnzC; p++) {
            const unsigned long long w = (unsigned long long)p % 8191ull + 1ull;
            wsumCol += (unsigned long long)ciC[p] * w;
            sumVal += vC[p];
            wsumVal += vC[p] * (double)w;
        }
        printf("ref_opencl_digest: nnzC=%d sum_rowptr=%llu wsum_col=%llu sum_val=%.17g wsum_val=%.17g rows_sorted=%d\n",
               nnzC, sumRp, wsumCol, sumVal, wsumVal, sorted);
        printf("ref_opencl_spgemm: m=%d k=%d n=%d nnzA=%d nnzB=%d -> nnzC=%d\n", m, k, n, nnzA, nnzB, nnzC);
        fflush(stdout);
        _exit(0);
    }
    FILE *fo = fopen(argv[2], "wb");
    if (!fo) { perror(argv[2]); return 2; }
    ok = wr(fo, &nnzC, sizeof(int)) && wr(fo, rpC.data(), sizeof(int) * (m + 1)) &&
         wr(fo, ciC.data(), sizeof(int) * (size_t)nnzC) && wr(fo, vC.data(), sizeof(double) * (size_t)nnzC);
    if (fclose(fo) != 0 || !ok) { fprintf(stderr, "short output\n"); return 2; }
    printf("ref_opencl_spgemm: m=%d k=%d n=%d nnzA=%d nnzB=%d -> nnzC=%d\n", m, k, n, nnzA, nnzB, nnzC);
    fflush(stdout);
    _exit(0);
}
