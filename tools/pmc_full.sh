#!/bin/bash
# all pipes of the class numeric kernel: SQ instruction / wait counters, LDS, TA / TCP / TD (run_case.py p27_128)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; out=$ROOT/gpurun_out/pmcf_$1; rm -rf $out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
[ -n "$2" ] && export BHSPARSE_HIP_LIB=$ROOT/gpurun_variants/$2.so
export BHS_WARM=1; export BHS_OPTS=${BHS_OPTS:-class_numeric=1}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUSY_max" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $set -d $out/pmc$i -o p --output-format csv -- python3 $ROOT/tools/run_case.py p27_128 > /dev/null 2> $out/p$i.err || echo "pass $i ($set) failed"
done
cd $ROOT; python3 tools/pmc_summary.py $out | grep -A40 "k_class_numeric\|k_class_ring" | grep -v "^   _" | head -48
