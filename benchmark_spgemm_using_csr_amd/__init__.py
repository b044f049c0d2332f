"""MI355X-native CSR SpGEMM (C = A*B, int32 indices, fp64 values) behind the
bhSPARSE class API.  The compute path is hand-written HIP for gfx950 in
csrc/ behind the C-ABI of include/bhsparse_hip.h; this package is the host-side
mirror of the reference's `bhsparse` interface plus input helpers."""
from .facade import (bhsparse, BhsparseError, BHSPARSE_SUCCESS, BHSPARSE_HIP,  # noqa: F401
                     BHSPARSE_CUDA, BHSPARSE_OPENCL, NUM_PLATFORMS, spgemm_csr)
