#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/prof.sh.

    python tools/hbm_traffic.py gpurun_out/prof_<tag> [workload]

HBM bytes per launch = (FETCH_SIZE + WRITE_SIZE) KB x 1024, mean over the launches of each bhs:: kernel, keyed by the
name bhs_get_kernel_stats uses, and stamped with the digest of the device sources so that bench.py only quotes it
for the build it was measured on."""
import collections, csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchmark_spgemm_using_csr_amd import _lib

prof = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "p27_weak"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(prof + "/pmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "bhs::" in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))


def stat_name(k):
    mt = re.search(r"k_row_wave<(\d+), \d+, (true|false)", k)
    if mt:
        return "%s_wave<%s>" % ("numeric" if mt.group(2) == "true" else "symbolic", mt.group(1))
    if "k_class_numeric" in k:
        return "numeric_class"
    if "k_class_rows" in k:
        return None                      # (two launches, rows of B and of A, under one statistics name)
    if "k_num_rank" in k:
        return "numeric_rank"
    if "k_sym_sorted" in k:
        return "symbolic_sorted"
    mt = re.search(r"k_row_lane<(\d+), (true|false)", k)
    if mt:
        return "numeric_lane" if mt.group(2) == "true" else "symbolic_lane"
    return None


out = {}
for k, d in agg.items():
    n = stat_name(k)
    if n and d.get("FETCH_SIZE") and d.get("WRITE_SIZE"):
        fetch = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"])
        write = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        out[n] = int((fetch + write) * 1024)
path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
doc = {"_comment": "HBM bytes per launch = (FETCH_SIZE + WRITE_SIZE) KB x 1024, separate --pmc passes (tools/prof.sh). "
                   "Calibration on this access pattern (round 1): k_upper_bound's 4 B/lane streaming read of 223 MB reads "
                   "FETCH_SIZE = 222 MB and k_fill_queues' 16 B/lane stores of 33.6 MB read WRITE_SIZE = 33.7 MB, i.e. 1:1 for "
                   "these widths, so the guide's x2 correction (16 B/lane streaming reads) is not applied.",
       "build": _lib.source_digest(), workload: out}
json.dump(doc, open(path, "w"), indent=1)
print(json.dumps(doc, indent=1))
