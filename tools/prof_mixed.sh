#!/bin/bash
# usage: prof_mixed.sh <tag> [n] [cases]  -- rocprofv3 kernel trace of tools/mixed_case.py (round 6: the mixed mode's kernels one by one)
tag=$1; n=${2:-128}; cases=${3:-p0.1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
out=$ROOT/gpurun_out/prof_$tag
cd /tmp; mkdir -p $out; export TMPDIR=/tmp
BHS_NOGEN=1 timeout 300 rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 $ROOT/tools/mixed_case.py $n $cases > $out/run.txt 2> $out/stats.err
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print("%-90s calls %6s avg %10.1f ns  total %6.2f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]), float(r["Percentage"])))
PY
cat $out/run.txt | tail -4
