#!/bin/bash
# PMC passes focused on the vector memory pipe (TA / TCP / TD) and address translation; two counters of a
# block per pass (more exceeds the hardware), every pass under its own timeout
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; out=$ROOT/gpurun_out/prof_ta; rm -rf $out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-extra"
i=0
for set in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum"; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o p --output-format csv -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $out/p$i.err || echo "pass $i ($set) failed"
done
ls $out
