#!/bin/bash
# The kernels that were built, measured and left out of the product library (bhs_row_span.hip.h, bhs_row_tiny.hip.h, the ring
# kernel's ablation copy tools/lab/bhs_class_ring_lab.hip.h): a lab build (-DBHS_LAB=1) and their parity tests against it.
#   tools/lab_tests.sh            on a GPU box (gpurun -- tools/lab_tests.sh); the build itself needs no GPU
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
[ -f gpurun_variants/lab.so ] || tools/build_variants.sh lab="-DBHS_RING_LAB=1" > /dev/null
BHSPARSE_HIP_LIB=$ROOT/gpurun_variants/lab.so python -m pytest tests/test_parity_gpu.py -m gpu -q \
  -k "rows_accumulated_over_their_column_span or rows_of_at_most_32_products_in_registers or row_class_path or mixed" 2>&1 | tail -5
