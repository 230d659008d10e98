#!/usr/bin/env python
"""Per-kernel HBM/fabric traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, csv output).
gfx950 corrections (MI355X_MICROARCH.md section HBM): FETCH_SIZE counts 64 B per 128-B request -> doubled; both counters
are in KiB.  Calibrated here on layernorm_kernel (reads 4 B, writes 2 B per element: 554 / 277 MB at 135168 x 1024).
usage: tools/pmc_traffic.py <dir with FETCH_SIZE_counter_collection.csv, WRITE_SIZE_counter_collection.csv> > profiles/x.json"""
import collections, csv, json, re, sys
d = sys.argv[1]
out = collections.OrderedDict()
for c, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
    for r in csv.DictReader(open(f"{d}/{c}_counter_collection.csv")):
        m = re.search(r"(gemm256_kernel<\d, \d>|gemm_kernel<\d, \d>|attn_kernel<\d>|layernorm_kernel<[^>]*>|lbs_vertex_kernel)", r["Kernel_Name"])
        if m:
            e = out.setdefault(m.group(1), {"FETCH_SIZE": [], "WRITE_SIZE": []})
            e[c].append(float(r["Counter_Value"]) * 1024 * mult)
res = {}
for k, e in out.items():
    rd, wr = sum(e["FETCH_SIZE"]) / max(len(e["FETCH_SIZE"]), 1), sum(e["WRITE_SIZE"]) / max(len(e["WRITE_SIZE"]), 1)
    res[k] = {"launches": len(e["FETCH_SIZE"]), "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr), "total_bytes_per_launch": round(rd + wr)}
gl = [(v["launches"], v["total_bytes_per_launch"]) for k, v in res.items() if k.startswith("gemm")]
if gl:
    res["_gemm_avg_bytes_per_launch"] = round(sum(n * b for n, b in gl) / sum(n for n, _ in gl))
print(json.dumps(res, indent=1))
