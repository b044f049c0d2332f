#!/bin/bash
# builds measurement variants of the library: tools/build_variants.sh 0 4 36 ...  -> gpurun_variants/abl<mask>.so
cd "$(dirname "$0")/../benchmark_spgemm_using_csr_amd/csrc"
mkdir -p ../../gpurun_variants
for m in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fvisibility=hidden -DBHS_ABL=$m $EXTRA -shared -o ../../gpurun_variants/abl$m$SUFFIX.so bhsparse_hip.hip &
done
wait
ls -la ../../gpurun_variants
