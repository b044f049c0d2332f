#!/bin/bash
# usage: timeline.sh <tag> <command...>  -- rocprofv3 kernel trace of a command; prints the LAST multiply's kernels as a timeline
# (start offset, duration, queue) -- what overlaps what on the side streams
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
out=$ROOT/gpurun_out/tl_$tag
cd /tmp; mkdir -p $out; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $out/t -o t --output-format csv -- "$@" > $out/run.txt 2> $out/err.txt
f=$(find $out/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last multiply: from the last k_class_reset / k_upper_bound / memset-start backwards -- take the kernels after the last gap > 200 us... simpler: last 60 kernels
ks = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -70:]
# find the start of the last multiply: the last kernel whose name has k_upper_bound or k_class_reset or k_row_lane<..false
start = 0
for i, r in enumerate(ks):
    n = r["Kernel_Name"]
    if "k_upper_bound<" in n or "k_class_reset" in n: start = i
ks = ks[start:]
t0 = int(ks[0]["Start_Timestamp"])
for r in ks:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f us  +%8.1f us  q%-3s %s" % (s / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:110]))
print("multiply: %.1f us" % ((int(ks[-1]["End_Timestamp"]) - t0) / 1e3))
PY
