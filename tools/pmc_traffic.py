#!/usr/bin/env python
"""Per-kernel HBM/fabric traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, csv output).
gfx950 corrections (MI355X_MICROARCH.md section HBM): FETCH_SIZE counts 64 B per 128-B request -> doubled; both counters
are in KiB.  Calibrated here on layernorm_kernel (reads 4 B, writes 2 B per element: 554 / 277 MB at 135168 x 1024).
If <dir>/SQ_counter_collection.csv exists (a third pass: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE) it adds per kernel: matrix-pipe busy fraction =
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), the effective shader clock = (GRBM_GUI_ACTIVE / 8) / duration,
and the LDS bank-conflict share of LDS-active cycles.
usage: tools/pmc_traffic.py <dir with FETCH_SIZE_counter_collection.csv, WRITE_SIZE_counter_collection.csv> > profiles/x.json"""
import os
import collections, csv, json, re, sys
d = sys.argv[1]
out = collections.OrderedDict()
for c, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
    for r in csv.DictReader(open(f"{d}/{c}_counter_collection.csv")):
        m = re.search(r"(gemm256_kernel<\d, \d>|gemm_kernel<\d, \d>|attn_kernel<[^>]*>|layernorm_kernel<[^>]*>|lbs_vertex_kernel)", r["Kernel_Name"])
        if m:
            e = out.setdefault(m.group(1), {"FETCH_SIZE": [], "WRITE_SIZE": []})
            e[c].append(float(r["Counter_Value"]) * 1024 * mult)
res = {}
for k, e in out.items():
    rd, wr = sum(e["FETCH_SIZE"]) / max(len(e["FETCH_SIZE"]), 1), sum(e["WRITE_SIZE"]) / max(len(e["WRITE_SIZE"]), 1)
    res[k] = {"launches": len(e["FETCH_SIZE"]), "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr), "total_bytes_per_launch": round(rd + wr)}
sq_path = f"{d}/SQ_counter_collection.csv"
if os.path.isfile(sq_path):
    sq = collections.OrderedDict()
    for r in csv.DictReader(open(sq_path)):
        m = re.search(r"(gemm256_kernel<\d, \d>|gemm_kernel<\d, \d>|attn_kernel<[^>]*>|layernorm_kernel<[^>]*>|lbs_vertex_kernel)", r["Kernel_Name"])
        if m:
            e = sq.setdefault(m.group(1), collections.defaultdict(list))
            e[r["Counter_Name"]].append(float(r["Counter_Value"]))
            e["_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, e in sq.items():
        avg = lambda c: sum(e[c]) / max(len(e[c]), 1)
        cyc = avg("GRBM_GUI_ACTIVE") / 8.0
        res.setdefault(k, {}).update({
            "mfma_busy_frac": round(avg("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc), 4) if cyc else None,
            "shader_clock_ghz": round(cyc / avg("_ns"), 3) if e["_ns"] else None,
            "avg_duration_ms_under_pmc": round(avg("_ns") * 1e-6, 4),
            "lds_bank_conflict_frac": round(avg("SQ_LDS_BANK_CONFLICT") / avg("SQ_LDS_IDX_ACTIVE"), 4) if avg("SQ_LDS_IDX_ACTIVE") else 0.0,
            "valu_inst_per_mfma_busy_cycle": round(avg("SQ_ACTIVE_INST_VALU") / avg("SQ_VALU_MFMA_BUSY_CYCLES"), 3) if avg("SQ_VALU_MFMA_BUSY_CYCLES") else None})
gl = [(v["launches"], v["total_bytes_per_launch"]) for k, v in res.items() if k.startswith("gemm")]
if gl:
    res["_gemm_avg_bytes_per_launch"] = round(sum(n * b for n, b in gl) / sum(n for n, _ in gl))
print(json.dumps(res, indent=1))
