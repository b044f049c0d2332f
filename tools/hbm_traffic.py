#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the memory-side PMC passes of tools/prof.sh.

    python tools/hbm_traffic.py gpurun_out/prof_<tag> [workload]

Bytes per launch of every bhs:: kernel that left / entered the XCDs' L2s on the memory side, from the request
counters BY SIZE: reads = 128 x TCC_EA0_RDREQ_128B + 64 x TCC_EA0_RDREQ_64B + 32 x TCC_EA0_RDREQ_32B, writes = 64 x
TCC_EA0_WRREQ_64B + 32 x (TCC_EA0_WRREQ - TCC_EA0_WRREQ_64B).  FETCH_SIZE is NOT used for the figure: it tallies every
read request at 64 bytes and so reads 0.5-0.65 x of what a streaming pass moves on gfx950 (the guide says
the same).  The script checks the size-resolved figure where the truth is known exactly --
  k_check_sorted, k_class_heads<false>   read colIndB + rowPtrB once:   4 * nnzB + 4 * (k + 1) bytes
  k_class_numeric*                        writes every entry of C once:  12 * nnzC bytes
-- and records the ratios under `check` (all within a few per cent of 1 or the file says so).  *_DRAM: requests that
went on to HBM (the others were served by the Infinity Cache).  Stamped with the digest of the device sources:
bench.py quotes `roofline.traffic` only for the build it was measured on."""
import collections, csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchmark_spgemm_using_csr_amd import _lib

prof = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "p27_weak"
WANT = ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum",
        "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_DRAM_sum")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(prof + "/pmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "bhs::" in r["Kernel_Name"] and r["Counter_Name"] in WANT:
            agg[re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void bhs::", "").replace("bhs::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
bench = json.load(open(os.path.join(prof, "bench_stats.json")))
cfg = bench["config"]
nnzB, k, nnzC = cfg["nnzA_total"], cfg["m"], cfg["nnzC"]          # (C = A^2: B = A)


def stat_name(kk):
    mt = re.search(r"k_row_wave<(\d+), \d+, (true|false)", kk)
    if mt:
        return "%s_wave<%s>" % ("numeric" if mt.group(2) == "true" else "symbolic", mt.group(1))
    if "k_class_numeric" in kk or "k_class_ring" in kk:
        return "numeric_class"
    mt = re.search(r"k_row_lane<(\d+), (true|false)", kk)
    if mt:
        return "numeric_lane" if mt.group(2) == "true" else "symbolic_lane"
    return None


mean = lambda v: sum(v) / len(v) if v else None
raw, out, check = {}, {}, {}
for kk, d in sorted(agg.items()):
    e = {c: mean(d.get(c)) for c in WANT}
    g = lambda c: e.get(c) or 0.0
    have_r = e["TCC_EA0_RDREQ_128B_sum"] is not None
    have_w = e["TCC_EA0_WRREQ_sum"] is not None
    e["read_bytes"] = 128 * g("TCC_EA0_RDREQ_128B_sum") + 64 * g("TCC_EA0_RDREQ_64B_sum") + 32 * g("TCC_EA0_RDREQ_32B_sum") if have_r else None
    e["write_bytes"] = 64 * g("TCC_EA0_WRREQ_64B_sum") + 32 * (g("TCC_EA0_WRREQ_sum") - g("TCC_EA0_WRREQ_64B_sum")) if have_w else None
    e["fetch_size_bytes"] = e["FETCH_SIZE"] * 1024 if e["FETCH_SIZE"] is not None else None
    e["write_size_bytes"] = e["WRITE_SIZE"] * 1024 if e["WRITE_SIZE"] is not None else None
    if e["TCC_EA0_RDREQ_sum"]:
        e["reads_to_dram_fraction"] = g("TCC_EA0_RDREQ_DRAM_sum") / e["TCC_EA0_RDREQ_sum"] if e["TCC_EA0_RDREQ_DRAM_sum"] is not None else None
    raw[kk] = e
    n = stat_name(kk)
    if n and have_r and have_w:
        out[n] = int(e["read_bytes"] + e["write_bytes"])
        out[n + " (read, write)"] = [int(e["read_bytes"]), int(e["write_bytes"])]
    if have_r and ("k_check_sorted" in kk or "k_class_heads<false" in kk or "k_class_fused<false" in kk or "k_class_tile<false" in kk):
        check[kk + ": read / (4 nnzB + 4 (k + 1))"] = round(e["read_bytes"] / (4.0 * nnzB + 4.0 * (k + 1)), 4)
        if e["fetch_size_bytes"]:
            check[kk + ": FETCH_SIZE / same"] = round(e["fetch_size_bytes"] / (4.0 * nnzB + 4.0 * (k + 1)), 4)
    if have_w and ("k_class_numeric" in kk or "k_class_ring" in kk):
        check[kk + ": written / (12 nnzC)"] = round(e["write_bytes"] / (12.0 * nnzC), 4)
path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
doc = {"_comment": "memory-side bytes per launch (tools/hbm_traffic.py): 128/64/32-byte read requests and 64/32-byte write requests "
                   "of the L2s, separate --pmc passes (tools/prof.sh); `check` compares them with passes of exactly known traffic; "
                   "`raw` keeps every counter incl. FETCH_SIZE / WRITE_SIZE",
       "build": _lib.source_digest(), "check": check, workload: out, "raw": raw}
json.dump(doc, open(path, "w"), indent=1)
print(json.dumps({kk: v for kk, v in doc.items() if kk != "raw"}, indent=1))
for kk, e in raw.items():
    print("%-44s read %8.1f MB (FETCH_SIZE %8.1f)  write %8.1f MB (WRITE_SIZE %8.1f)  reads that went to DRAM: %s" % (
        kk[:44], (e["read_bytes"] or 0) / 1e6, (e["fetch_size_bytes"] or 0) / 1e6, (e["write_bytes"] or 0) / 1e6,
        (e["write_size_bytes"] or 0) / 1e6, "%.2f" % e["reads_to_dram_fraction"] if e.get("reads_to_dram_fraction") is not None else "-"))
