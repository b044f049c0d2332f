#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the memory-side PMC passes of tools/prof.sh.

    python tools/hbm_traffic.py gpurun_out/prof_<tag> [workload]

Raw figures per launch of every bhs:: kernel: FETCH_SIZE / WRITE_SIZE (KB x 1024) and the L2's request counters
(TCC_EA0_RDREQ / _32B, TCC_EA0_WRREQ / _64B).  The guide warns that FETCH_SIZE under-counts on gfx950 (128-byte
requests tallied at 64), and round 2's file was contradicted by its own passes (k_check_sorted read 146 MB
where it must read >= 231 MB).  So the read side is CALIBRATED PER RUN on passes whose byte count is known exactly:
  k_check_sorted     reads colIndB + rowPtrB once                      4 * nnzB + 4 * (k + 1)
  k_class_heads<B>   reads colIndB + rowPtrB once                      4 * nnzB + 4 * (k + 1)
read factor = known / FETCH_SIZE of those kernels (mean).  WRITE_SIZE is taken as it is; the file records how it
compares with the one write count that is known exactly (the class path's numeric kernel writes every entry of C once:
12 * nnzC).  The file states the factor; `numeric_class` etc. are the calibrated bytes per launch (bench.py's
roofline.traffic), stamped with the digest of the device sources so that bench.py only quotes them for that build."""
import collections, csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchmark_spgemm_using_csr_amd import _lib

prof = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "p27_weak"
WANT = ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(prof + "/pmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "bhs::" in r["Kernel_Name"] and r["Counter_Name"] in WANT:
            agg[re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void bhs::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
bench = json.load(open(os.path.join(prof, "bench_stats.json")))
cfg = bench["config"]
nnzB, k, nnzC = cfg["nnzA_total"], cfg["m"], cfg["nnzC"]          # (C = A^2: B = A)


def stat_name(kk):
    mt = re.search(r"k_row_wave<(\d+), \d+, (true|false)", kk)
    if mt:
        return "%s_wave<%s>" % ("numeric" if mt.group(2) == "true" else "symbolic", mt.group(1))
    if "k_class_numeric" in kk:
        return "numeric_class"
    mt = re.search(r"k_row_lane<(\d+), (true|false)", kk)
    if mt:
        return "numeric_lane" if mt.group(2) == "true" else "symbolic_lane"
    return None


mean = lambda v: sum(v) / len(v) if v else None
raw = {}
for kk, d in sorted(agg.items()):
    e = {c: mean(d.get(c)) for c in WANT}
    e["fetch_bytes"] = e["FETCH_SIZE"] * 1024 if e["FETCH_SIZE"] is not None else None
    e["write_bytes"] = e["WRITE_SIZE"] * 1024 if e["WRITE_SIZE"] is not None else None
    if e["TCC_EA0_RDREQ_sum"] is not None and e["TCC_EA0_RDREQ_32B_sum"] is not None:
        e["rdreq_bytes_64_32"] = 64 * (e["TCC_EA0_RDREQ_sum"] - e["TCC_EA0_RDREQ_32B_sum"]) + 32 * e["TCC_EA0_RDREQ_32B_sum"]
    if e["TCC_EA0_WRREQ_sum"] is not None and e["TCC_EA0_WRREQ_64B_sum"] is not None:
        e["wrreq_bytes_64_32"] = 64 * e["TCC_EA0_WRREQ_64B_sum"] + 32 * (e["TCC_EA0_WRREQ_sum"] - e["TCC_EA0_WRREQ_64B_sum"])
    raw[kk] = e
known_read = 4 * nnzB + 4 * (k + 1)
cal_r = [known_read / e["fetch_bytes"] for kk, e in raw.items()
         if e["fetch_bytes"] and ("k_check_sorted" in kk or "k_class_heads<false" in kk)]
chk_w = [e["write_bytes"] / (12.0 * nnzC) for kk, e in raw.items() if e["write_bytes"] and "k_class_numeric" in kk]
read_factor = mean(cal_r) if cal_r else 1.0
write_factor = 1.0
out = {}
for kk, e in raw.items():
    n = stat_name(kk)
    if n and e["fetch_bytes"] is not None and e["write_bytes"] is not None:
        out[n] = int(e["fetch_bytes"] * read_factor + e["write_bytes"] * write_factor)
        out[n + " (read, write)"] = [int(e["fetch_bytes"] * read_factor), int(e["write_bytes"] * write_factor)]
path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
doc = {"_comment": "calibrated HBM bytes per launch (tools/hbm_traffic.py): FETCH_SIZE x read_factor + WRITE_SIZE x write_factor; "
                   "factors from this same run's kernels of known traffic (see `calibration`); `raw` keeps every counter",
       "build": _lib.source_digest(),
       "calibration": {"read_factor": read_factor, "write_factor": write_factor, "known_read_bytes_colIndB_rowPtrB": known_read,
                       "read_factor_per_kernel": cal_r, "write_size_over_known_12_nnzC_numeric_class": chk_w},
       workload: out, "raw": raw}
json.dump(doc, open(path, "w"), indent=1)
print(json.dumps({kk: v for kk, v in doc.items() if kk != "raw"}, indent=1))
for kk, e in raw.items():
    print("%-60s fetch %8.1f MB  write %8.1f MB  rdreq(64/32) %s MB  wrreq(64/32) %s MB" % (
        kk[:60], (e["fetch_bytes"] or 0) / 1e6, (e["write_bytes"] or 0) / 1e6,
        "%.1f" % (e["rdreq_bytes_64_32"] / 1e6) if e.get("rdreq_bytes_64_32") else "-",
        "%.1f" % (e["wrreq_bytes_64_32"] / 1e6) if e.get("wrreq_bytes_64_32") else "-"))
