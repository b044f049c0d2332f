// csr_sort.h — per-row sort of (column, value) pairs by column, in place.
// Host helper with the signature of ref_spgemm::csr_sort_indices
// (SpGEMM_cuda/ref_spgemm.h:37-62), which the reference's driver applies to
// Matrix Market inputs before the multiply (main.cu:62-64): rows of B must be
// column-sorted for the long-row kernels to take their fast path.  Stable.
#ifndef BHSPARSE_AMD_CSR_SORT_H
#define BHSPARSE_AMD_CSR_SORT_H
#include <algorithm>
#include <utility>
#include <vector>

template <class I, class T>
void csr_sort_indices(const I n_row, const I Ap[], I Aj[], T Ax[])
{
    std::vector<std::pair<I, T> > temp;
    for (I i = 0; i < n_row; i++) {
        const I row_start = Ap[i], row_end = Ap[i + 1];
        bool sorted = true;
        for (I jj = row_start + 1; jj < row_end; jj++)
            if (Aj[jj - 1] > Aj[jj]) { sorted = false; break; }
        if (sorted) continue;
        temp.clear();
        for (I jj = row_start; jj < row_end; jj++) temp.push_back(std::make_pair(Aj[jj], Ax[jj]));
        std::stable_sort(temp.begin(), temp.end(),
                         [](const std::pair<I, T> &a, const std::pair<I, T> &b) { return a.first < b.first; });
        for (I jj = row_start, n = 0; jj < row_end; jj++, n++) { Aj[jj] = temp[n].first; Ax[jj] = temp[n].second; }
    }
}
#endif
