#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2 'rocpd' SQLite) kernel trace: per kernel name -> calls, total/avg/min/max ms, %.
Groups GEMM dispatches additionally by grid size so the per-epilogue launches can be told apart.
usage: tools/rocpd_stats.py results.db [--by-grid] > profiles/summary.txt"""
import sqlite3
import subprocess
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    by_grid = "--by-grid" in sys.argv
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select * from kernels").fetchall()
    ix = {c: i for i, c in enumerate(cols)}
    name_c = "name" if "name" in ix else [c for c in cols if "name" in c][0]
    stats = defaultdict(list)
    for r in rows:
        nm = r[ix[name_c]]
        key = nm
        if by_grid:
            key = f"{nm} grid={r[ix['grid_x']] if 'grid_x' in ix else r[ix.get('grid_size', 0)]}"
        stats[key].append((r[ix["end"]] - r[ix["start"]]) * 1e-6)
    total = sum(sum(v) for v in stats.values())
    names = list(stats)
    dem = subprocess.run(["c++filt"], input="\n".join(n.split(" grid=")[0] for n in names), capture_output=True, text=True).stdout.split("\n")
    print(f"{'kernel':90s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>9s} {'min_ms':>9s} {'max_ms':>9s} {'%':>6s}")
    for (k, v), d in sorted(zip(stats.items(), dem), key=lambda t: -sum(t[0][1])):
        label = d.replace("(anonymous namespace)::", "").replace("void ", "")
        label = label.split("(")[0] + (" grid=" + k.split(" grid=")[1] if " grid=" in k else "")
        print(f"{label[:90]:90s} {len(v):6d} {sum(v):10.3f} {sum(v)/len(v):9.4f} {min(v):9.4f} {max(v):9.4f} {100*sum(v)/total:6.2f}")
    print(f"{'TOTAL':90s} {sum(len(v) for v in stats.values()):6d} {total:10.3f}")


if __name__ == "__main__":
    main()
