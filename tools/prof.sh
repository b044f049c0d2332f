#!/bin/bash
# usage: tools_prof.sh <tag> [bench args...]   -- rocprofv3 stats + PMC passes of bench.py, outputs under gpurun_out/prof_<tag>/
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
out=$ROOT/gpurun_out/prof_$tag
cd /tmp
mkdir -p $out
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-general --no-reference $@"
timeout 120 rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 $ROOT/bench.py $ARGS > $out/bench_stats.json 2> $out/stats.err
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $out/pmc1 -o p --output-format csv -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $out/pmc1.err
timeout 120 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM -d $out/pmc2 -o p --output-format csv -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $out/pmc2.err
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc3 -o p --output-format csv -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $out/pmc3.err
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $out/pmc4 -o p --output-format csv -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $out/pmc4.err
# memory-side request counters of the L2 by request size (FETCH_SIZE tallies every request at 64 bytes: it under-counts
# by up to 2x on gfx950, where a request can be 128 bytes): read bytes = 128 x RDREQ_128B + 64 x RDREQ_64B + 32 x RDREQ_32B;
# the second pass: write requests by size, and how many of either kind went on to DRAM (the rest hit the Infinity Cache)
timeout 120 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $out/pmc5 -o p --output-format csv -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $out/pmc5.err
timeout 120 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum -d $out/pmc6 -o p --output-format csv -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $out/pmc6.err
find $out -name "*.csv" | head -30
