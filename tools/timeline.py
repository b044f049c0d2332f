#!/usr/bin/env python3
"""The kernels of the LAST multiply of a `rocprofv3 --kernel-trace` run as a timeline (start offset, duration, queue):
which bins of the general pipeline run side by side and what the critical path is.
    python tools/timeline.py gpurun_out/trace_x [first-kernel-substring]"""
import csv, glob, os, sys
d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "k_upper_bound"
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
# the last multiply starts at the last `first` kernel that follows a gap of other kernels
s = idx[-1]
while s - 1 in idx: s -= 1
t0 = int(rows[s]["Start_Timestamp"])
for r in rows[s:]:
    n = r["Kernel_Name"].replace("void bhs::", "").replace("bhs::", "").split("(")[0][:52]
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(a - t0) / 1e3:9.1f} us  +{(b - a) / 1e3:8.1f}  ->{(b - t0) / 1e3:9.1f}  q{r['Queue_Id']} wgs {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):>7d} lds {r['LDS_Block_Size']:>6s}  {n}")
