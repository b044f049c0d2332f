#!/bin/bash
# builds measurement variants of the library:
#   tools/build_variants.sh name1="-DFLAG=.. -DFLAG2=.." name2="..."   -> gpurun_variants/<name>.so
cd "$(dirname "$0")/../benchmark_spgemm_using_csr_amd/csrc"
mkdir -p ../../gpurun_variants
n=0
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fvisibility=hidden -DBHS_LAB=1 $flags -shared -o ../../gpurun_variants/$name.so bhsparse_hip.hip &
  n=$((n+1)); if [ $((n % 3)) -eq 0 ]; then wait; fi
done
wait
ls -la ../../gpurun_variants/*.so
