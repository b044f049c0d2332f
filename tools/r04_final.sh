#!/bin/bash
# round 4's final measurement set, one box (gpurun --timeout 5400 -- 'bash tools/r04_final.sh'): tests, profiles (stats + PMC passes), traffic, pipes, bench (with traffic), tables, probes
R=$GRAFT_REPO_ROOT
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r04_final_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r04_final_pytest.log
timeout 1500 bash tools/prof.sh r04final > gpurun_out/r04_final_prof.log 2>&1
python tools/hbm_traffic.py gpurun_out/prof_r04final > gpurun_out/r04_final_traffic.txt 2>&1; head -24 gpurun_out/r04_final_traffic.txt | tail -10
python tools/pmc_summary.py gpurun_out/prof_r04final > gpurun_out/r04_final_pmc_summary.txt 2>&1
cp profiles/hbm_traffic.json gpurun_out/r04_final_hbm_traffic.json
timeout 900 bash tools/pmc_full.sh r04final > gpurun_out/r04_final_pipes.txt 2>&1
timeout 900 python bench.py > gpurun_out/r04_final_bench.json 2> gpurun_out/r04_final_bench.err; tail -c 400 gpurun_out/r04_final_bench.json
timeout 300 python tools/probe_spread.py > gpurun_out/r04_final_spread.txt 2>&1
[ -x gpurun_variants/xcc_seq_probe ] || /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -I benchmark_spgemm_using_csr_amd/csrc -o gpurun_variants/xcc_seq_probe tools/xcc_seq_probe.hip
./gpurun_variants/xcc_seq_probe > gpurun_out/r04_final_xcc_seq.txt 2>&1
timeout 1500 python tools/suite_table.py > gpurun_out/r04_final_suite_table.md 2> gpurun_out/r04_final_suite_table.err; cut -d'|' -f2,9,11,14,17,18 gpurun_out/r04_final_suite_table.md
timeout 200 python tools/probe_super.py 160 0,64 > gpurun_out/r04_final_super.txt 2>&1; timeout 200 python tools/probe_super.py 96 0,64 >> gpurun_out/r04_final_super.txt 2>&1; timeout 200 python tools/probe_super.py 128 0,64 >> gpurun_out/r04_final_super.txt 2>&1; tail -3 gpurun_out/r04_final_super.txt
# the R-MAT graph (column-window kernels): kernel stats and one counter pass
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r04rmat/stats -o s --output-format csv -- python3 $R/tools/run_suite_case.py rmat_s20 > $R/gpurun_out/r04_final_rmat_run.txt 2> $R/gpurun_out/prof_r04rmat_stats.err
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/prof_r04rmat/pmc1 -o p --output-format csv -- python3 $R/tools/run_suite_case.py rmat_s20 > /dev/null 2> $R/gpurun_out/prof_r04rmat_pmc1.err )
python tools/pmc_summary.py gpurun_out/prof_r04rmat > gpurun_out/r04_final_rmat_pmc_summary.txt 2>&1
find gpurun_out/prof_r04rmat -name "*kernel_stats.csv" | head -2; tail -4 gpurun_out/r04_final_rmat_run.txt | cut -c1-300
