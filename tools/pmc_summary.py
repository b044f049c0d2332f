#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: mean counter value per launch for bhs:: kernels."""
import csv, sys, glob, collections, re
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/pmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "bhs::" not in k: continue
        k = re.sub(r"\(.*", "", k).replace("void bhs::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["_vgpr"] = [float(r["VGPR_Count"])]; agg[k]["_lds"] = [float(r["LDS_Block_Size"])]; agg[k]["_grid"]=[float(r["Grid_Size"])]
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
