#!/bin/bash
# the parts of tools/r06_final.sh that tools/r06_refresh.sh leaves out, on the round's final build: the general pipeline under the
# profiler, mixed mode at 160^3 and its kernels one by one, the web-graph timeline, the results table, the R-MAT profile
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
BENCH_LIB_OPTS=class_path=0,wave_first=0,lane_first=0,direct_bins=0 timeout 600 bash tools/prof.sh r06general > gpurun_out/r06_final_prof_general.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_r06general > gpurun_out/r06_final_general_pmc_summary.txt 2>&1
timeout 600 python tools/mixed_case.py 160 clean,p0.1,long > gpurun_out/r06_final_mixed_160.txt 2>&1
timeout 300 bash tools/prof_mixed.sh r06mixed 128 p0.1 > gpurun_out/r06_final_mixed_kernels.txt 2>&1
timeout 300 bash tools/timeline.sh r06web python3 $R/tools/run_case.py weblike > gpurun_out/r06_final_weblike_timeline.txt 2>&1
timeout 1500 python tools/suite_table.py > gpurun_out/r06_final_suite_table.md 2> gpurun_out/r06_final_suite_table.err; cut -d'|' -f2,9,11,14,17,18 gpurun_out/r06_final_suite_table.md
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r06rmat/stats -o s --output-format csv -- python3 $R/tools/run_suite_case.py rmat_s20 > $R/gpurun_out/r06_final_rmat_run.txt 2> $R/gpurun_out/prof_r06rmat_stats.err )
