#!/usr/bin/env python
"""Per-kernel wave-time split from one rocprofv3 --pmc pass (csv) over
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES:
share of wave-cycles issuing / waiting at s_waitcnt or a barrier / stalled with an instruction ready, VALU instructions per wave.
usage: tools/pmc_wait.py <counter_collection.csv> > profiles/rNN_attention_wait_pmc.json"""
import collections, csv, json, re, sys
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(attn_kernel<[^>]*>|gemm256_kernel<[^>]*>|cls_linear_kernel<[^>]*>|lbs_vertex_kernel)", r["Kernel_Name"])
    if m:
        acc.setdefault(m.group(1), collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, e in acc.items():
    a = lambda c: sum(e[c]) / max(len(e[c]), 1)
    wc = a("SQ_WAVE_CYCLES")
    if not wc:
        continue
    out[k] = {"launches": len(e["SQ_WAVE_CYCLES"]), "active_frac": round(a("SQ_ACTIVE_INST_ANY") / wc, 4), "wait_any_frac": round(a("SQ_WAIT_ANY") / wc, 4),
              "wait_inst_any_frac": round(a("SQ_WAIT_INST_ANY") / wc, 4),
              "valu_inst_per_wave": round(a("SQ_ACTIVE_INST_VALU") / a("SQ_WAVES"), 1) if a("SQ_WAVES") else None,
              "mfma_busy_frac": round(a("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * a("GRBM_GUI_ACTIVE") / 8.0), 4) if a("GRBM_GUI_ACTIVE") else None}
print(json.dumps(out, indent=1))
