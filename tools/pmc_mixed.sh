#!/bin/bash
# usage: pmc_mixed.sh <tag> <n> <case> <kernel regex> -- PMC counters of one kernel of tools/mixed_case.py (separate passes, kernel trace only)
tag=$1; n=${2:-128}; cases=${3:-p0.1}; kern=${4:-k_class_tile}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
out=$ROOT/gpurun_out/pmc_$tag
cd /tmp; mkdir -p $out; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_FLAT SQ_INSTS_GDS"; do
  i=$((i+1))
  BHS_NOGEN=1 timeout 300 rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o p --output-format csv -- python3 $ROOT/tools/mixed_case.py $n $cases > /dev/null 2> $out/p$i.err
done
python3 - $out "$kern" <<'PY'
import csv, sys, glob, re, collections
out, kern = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(out + "/p*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if re.search(kern, r["Kernel_Name"]):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k)
        for c, v in sorted(d.items()):
            print("    %-24s launches %3d  mean %14.0f" % (c, len(v), sum(v) / len(v)))
PY
