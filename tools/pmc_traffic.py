#!/usr/bin/env python
"""Per-kernel HBM/fabric traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, csv output).
gfx950 corrections (MI355X_MICROARCH.md section HBM): FETCH_SIZE counts 64 B per 128-B request -> doubled; both counters
are in KiB.  Calibrated here on layernorm_kernel (reads 4 B, writes 2 B per element: 554 / 277 MB at 135168 x 1024).
If <dir>/SQ_counter_collection.csv exists (a third pass: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE) it adds per kernel: matrix-pipe busy fraction =
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), the effective shader clock = (GRBM_GUI_ACTIVE / 8) / duration,
and the LDS bank-conflict share of LDS-active cycles.
<dir>/lbs/ (optional) holds the same passes over tools/lbs_bench.py 160: the SMPL-X layer's kernels at P = 160 are taken from
there (inside the forward the layer runs at other person counts).  The summary records the hash of the sources the measured
libmhmr.so was built from (`_source_hash`, compiled into the library; plus `_lib_sha16`, the file's own sha256 prefix); bench.py reports
`traffic` from it only when `_source_hash` matches the library it is running.
usage: tools/pmc_traffic.py <dir with FETCH_SIZE_counter_collection.csv, WRITE_SIZE_counter_collection.csv> > profiles/rNN_pmc.json"""
import collections, csv, glob, hashlib, json, os, re, sys
d = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAT = re.compile(r"(gemm256_kernel<[^>]*>|gemm_kernel<\d, \d>|cls_linear_kernel<[^>]*>|ln_stats_kernel|attn_kernel<[^>]*>|attn16_kernel<[^>]*>|attn64_kernel<[^>]*>|layernorm_kernel<[^>]*>|lbs_vertex_kernel|lbs_pose_kernel|lbs_extra_joints_kernel)")


def find(dirname, counter):
    hits = [p for p in glob.glob(f"{dirname}/**/{counter}_counter_collection.csv", recursive=True) if "/lbs/" not in p[len(dirname):] or dirname.rstrip("/").endswith("lbs")]
    return hits[0] if hits else None


def traffic(dirname, keep):
    out = collections.OrderedDict()
    for c, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        path = find(dirname, c)
        if not path:
            continue
        for r in csv.DictReader(open(path)):
            m = PAT.search(r["Kernel_Name"])
            if m and keep(m.group(1)):
                e = out.setdefault(m.group(1), {"FETCH_SIZE": [], "WRITE_SIZE": []})
                e[c].append(float(r["Counter_Value"]) * 1024 * mult)
    res = {}
    for k, e in out.items():
        rd, wr = sum(e["FETCH_SIZE"]) / max(len(e["FETCH_SIZE"]), 1), sum(e["WRITE_SIZE"]) / max(len(e["WRITE_SIZE"]), 1)
        res[k] = {"launches": len(e["FETCH_SIZE"]), "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr), "total_bytes_per_launch": round(rd + wr)}
    return res


def sq(dirname, keep, res):
    path = find(dirname, "SQ")
    if not path:
        return
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        m = PAT.search(r["Kernel_Name"])
        if m and keep(m.group(1)):
            e = acc.setdefault(m.group(1), collections.defaultdict(list))
            e[r["Counter_Name"]].append(float(r["Counter_Value"]))
            e["_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, e in acc.items():
        avg = lambda c: sum(e[c]) / max(len(e[c]), 1)
        cyc = avg("GRBM_GUI_ACTIVE") / 8.0
        res.setdefault(k, {}).update({
            "mfma_busy_frac": round(avg("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc), 4) if cyc else None,
            "shader_clock_ghz": round(cyc / avg("_ns"), 3) if e["_ns"] else None,
            "avg_duration_ms_under_pmc": round(avg("_ns") * 1e-6, 4),
            "lds_bank_conflict_frac": round(avg("SQ_LDS_BANK_CONFLICT") / avg("SQ_LDS_IDX_ACTIVE"), 4) if avg("SQ_LDS_IDX_ACTIVE") else 0.0,
            "valu_inst_per_mfma_busy_cycle": round(avg("SQ_ACTIVE_INST_VALU") / avg("SQ_VALU_MFMA_BUSY_CYCLES"), 3) if avg("SQ_VALU_MFMA_BUSY_CYCLES") else None})


is_lbs = lambda k: k.startswith("lbs_")
lbs_dir = os.path.join(d, "lbs")
have_lbs_dir = os.path.isdir(lbs_dir)
res = traffic(d, (lambda k: not is_lbs(k)) if have_lbs_dir else (lambda k: True))
sq(d, (lambda k: not is_lbs(k)) if have_lbs_dir else (lambda k: True), res)
if have_lbs_dir:
    r2 = traffic(lbs_dir, is_lbs)
    sq(lbs_dir, is_lbs, r2)
    for k, v in r2.items():
        v["persons"] = 160
    res.update(r2)
gl = [(v["launches"], v["total_bytes_per_launch"]) for k, v in res.items() if k.startswith("gemm") and "launches" in v]
if gl:
    res["_gemm_avg_bytes_per_launch"] = round(sum(n * b for n, b in gl) / sum(n for n, _ in gl))
# one attention CALL = the main kernel + the (normally empty) fallback launch behind it
al = sorted(((v["launches"], v["launches"] * v["total_bytes_per_launch"]) for k, v in res.items() if k.startswith(("attn_kernel", "attn16_kernel", "attn64_kernel")) and "launches" in v), reverse=True)
if al:
    res["_attention_bytes_per_call"] = round(sum(b for _, b in al) / al[0][0])
with open(os.path.join(ROOT, "multi_hmr_amd", "csrc", "libmhmr.so"), "rb") as f:
    blob = f.read()
res["_lib_sha16"] = hashlib.sha256(blob).hexdigest()[:16]
# the hash of the SOURCES the measured library was built from (compiled into it): what bench.py matches, machine-independent
m = re.search(rb"MHMR_SOURCE_HASH=([0-9a-f]{16})", blob)
res["_source_hash"] = m.group(1).decode() if m else None
print(json.dumps(res, indent=1))
