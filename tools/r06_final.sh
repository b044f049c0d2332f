#!/bin/bash
# round 6's measurement set, one box (gpurun --timeout 5400 -- 'bash tools/r06_final.sh'): GPU tests, profiles (stats + PMC passes),
# traffic, the ring kernel's pipes, a default bench line, the general pipeline's profile, the mixed-mode cases one by one,
# the lab build's tests, the results table, the web-graph timeline, the box probe -- what profiles/r06_* is copied from
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06_final_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r06_final_pytest.log
timeout 1500 bash tools/prof.sh r06final > gpurun_out/r06_final_prof.log 2>&1
python tools/hbm_traffic.py gpurun_out/prof_r06final > gpurun_out/r06_final_traffic.txt 2>&1; head -24 gpurun_out/r06_final_traffic.txt | tail -10
python tools/pmc_summary.py gpurun_out/prof_r06final > gpurun_out/r06_final_pmc_summary.txt 2>&1
cp profiles/hbm_traffic.json gpurun_out/r06_final_hbm_traffic.json
BHS_OPTS=class_numeric=2 timeout 900 bash tools/pmc_full.sh r06final > gpurun_out/r06_final_pipes.txt 2>&1
timeout 900 python bench.py > gpurun_out/r06_final_bench.json 2> gpurun_out/r06_final_bench.err; tail -c 400 gpurun_out/r06_final_bench.json
# the general pipeline under the profiler
BENCH_LIB_OPTS=class_path=0,wave_first=0,lane_first=0,direct_bins=0 timeout 600 bash tools/prof.sh r06general > gpurun_out/r06_final_prof_general.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_r06general > gpurun_out/r06_final_general_pmc_summary.txt 2>&1
# mixed mode: clean / 0.1 % / 1 % / one long row, each against the general pipeline's digest; the kernels of a mixed multiply one by one
timeout 600 python tools/mixed_case.py 128 clean,p0.1,p1,long > gpurun_out/r06_final_mixed_128.txt 2>&1; grep -v amdgpu gpurun_out/r06_final_mixed_128.txt | grep -v "{"
timeout 600 python tools/mixed_case.py 160 clean,p0.1,long > gpurun_out/r06_final_mixed_160.txt 2>&1
timeout 300 bash tools/prof_mixed.sh r06mixed 128 p0.1 > gpurun_out/r06_final_mixed_kernels.txt 2>&1
timeout 300 bash tools/timeline.sh r06web python3 $R/tools/run_case.py weblike > gpurun_out/r06_final_weblike_timeline.txt 2>&1
timeout 900 bash tools/lab_tests.sh > gpurun_out/r06_final_lab_tests.txt 2>&1; tail -2 gpurun_out/r06_final_lab_tests.txt
timeout 1500 python tools/suite_table.py > gpurun_out/r06_final_suite_table.md 2> gpurun_out/r06_final_suite_table.err; cut -d'|' -f2,9,11,14,17,18 gpurun_out/r06_final_suite_table.md
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r06rmat/stats -o s --output-format csv -- python3 $R/tools/run_suite_case.py rmat_s20 > $R/gpurun_out/r06_final_rmat_run.txt 2> $R/gpurun_out/prof_r06rmat_stats.err )
timeout 120 python tools/box_probe.py > gpurun_out/r06_final_box_probe.txt 2>&1; tail -3 gpurun_out/r06_final_box_probe.txt
