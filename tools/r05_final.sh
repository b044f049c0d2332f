#!/bin/bash
# round 5's measurement set, one box (gpurun --timeout 5400 -- 'bash tools/r05_final.sh'): tests, profiles (stats + PMC passes), traffic, pipes of the ring kernel, bench, results table, the R-MAT profile
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r05_final_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r05_final_pytest.log
timeout 1500 bash tools/prof.sh r05final > gpurun_out/r05_final_prof.log 2>&1
python tools/hbm_traffic.py gpurun_out/prof_r05final > gpurun_out/r05_final_traffic.txt 2>&1; head -24 gpurun_out/r05_final_traffic.txt | tail -10
python tools/pmc_summary.py gpurun_out/prof_r05final > gpurun_out/r05_final_pmc_summary.txt 2>&1
cp profiles/hbm_traffic.json gpurun_out/r05_final_hbm_traffic.json
BHS_OPTS=class_numeric=2 timeout 900 bash tools/pmc_full.sh r05final > gpurun_out/r05_final_pipes.txt 2>&1
timeout 900 python bench.py > gpurun_out/r05_final_bench.json 2> gpurun_out/r05_final_bench.err; tail -c 400 gpurun_out/r05_final_bench.json
# the general pipeline under the profiler
BENCH_LIB_OPTS=class_path=0,wave_first=0,lane_first=0,direct_bins=0 timeout 600 bash tools/prof.sh r05general > gpurun_out/r05_final_prof_general.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_r05general > gpurun_out/r05_final_general_pmc_summary.txt 2>&1
timeout 1500 python tools/suite_table.py > gpurun_out/r05_final_suite_table.md 2> gpurun_out/r05_final_suite_table.err; cut -d'|' -f2,9,11,14,17,18 gpurun_out/r05_final_suite_table.md
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r05rmat/stats -o s --output-format csv -- python3 $R/tools/run_suite_case.py rmat_s20 > $R/gpurun_out/r05_final_rmat_run.txt 2> $R/gpurun_out/prof_r05rmat_stats.err )
find gpurun_out/prof_r05rmat -name "*kernel_stats.csv" | head -2; tail -4 gpurun_out/r05_final_rmat_run.txt | cut -c1-300
timeout 120 python tools/box_probe.py > gpurun_out/r05_final_box_probe.txt 2>&1; tail -3 gpurun_out/r05_final_box_probe.txt
