#!/bin/bash
# after the last source change of the round: the GPU tests, the profile + traffic passes and a default bench line of the FINAL build, one box
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06_final_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r06_final_pytest.log
BHS_TEST_OPTS=ring_dynamic=1 timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "row_class_path or mixed or ranges" > gpurun_out/r06_final_pytest_ring_dynamic.log 2>&1; grep -E "passed|failed" gpurun_out/r06_final_pytest_ring_dynamic.log
timeout 1500 bash tools/prof.sh r06final > gpurun_out/r06_final_prof.log 2>&1
python tools/hbm_traffic.py gpurun_out/prof_r06final > gpurun_out/r06_final_traffic.txt 2>&1
python tools/pmc_summary.py gpurun_out/prof_r06final > gpurun_out/r06_final_pmc_summary.txt 2>&1
cp profiles/hbm_traffic.json gpurun_out/r06_final_hbm_traffic.json
BHS_OPTS=class_numeric=2 timeout 900 bash tools/pmc_full.sh r06final > gpurun_out/r06_final_pipes.txt 2>&1
timeout 900 python bench.py > gpurun_out/r06_final_bench.json 2> gpurun_out/r06_final_bench.err; tail -c 300 gpurun_out/r06_final_bench.json
timeout 600 python tools/mixed_case.py 128 clean,p0.1,p1,long > gpurun_out/r06_final_mixed_128.txt 2>&1; grep -v amdgpu gpurun_out/r06_final_mixed_128.txt | grep -v "{"
timeout 120 python tools/box_probe.py > gpurun_out/r06_final_box_probe.txt 2>&1; tail -3 gpurun_out/r06_final_box_probe.txt
