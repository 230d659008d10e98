#!/usr/bin/env python
"""Kernel-trace digest of `bench.py --steps N --no-extras --no-cpu-baseline` under `rocprofv3 --kernel-trace --output-format csv -d OUT`:
the last `forwards` forwards of the trace (found by the attention launches: 24 per ViT-L forward) -- span, busy time, idle time between
consecutive kernels, overlap (a kernel starting before its predecessor ended), and per-kernel averages.
   python tools/trace_gaps.py OUT [forwards=3] [attention launches per forward=24]"""
import csv
import glob
import os
import sys


def main(out, forwards=3, per_fwd=24):
    rows = []
    for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    att = [i for i, r in enumerate(rows) if "attn16_kernel" in r[2]]
    assert len(att) >= forwards * per_fwd, (len(att), forwards * per_fwd)
    first = att[-forwards * per_fwd]
    # start of that forward: walk back to the im2col kernel
    while first > 0 and "im2col" not in rows[first][2]:
        first -= 1
    win = rows[first:]
    span = (win[-1][1] - win[0][0]) / 1e3
    busy = sum(e - s for s, e, _ in win) / 1e3
    gaps = sum(max(win[i][0] - win[i - 1][1], 0) for i in range(1, len(win))) / 1e3
    over = sum(max(win[i - 1][1] - win[i][0], 0) for i in range(1, len(win))) / 1e3
    nover = sum(1 for i in range(1, len(win)) if win[i][0] < win[i - 1][1])
    print(f"{len(win)} kernels = {len(win) / forwards:.1f} per forward; span {span / forwards:.1f} us per forward, sum of durations {busy / forwards:.1f}, "
          f"idle between kernels {gaps / forwards:.1f} ({gaps / max(len(win) - 1, 1):.2f} per boundary), overlap {over / forwards:.1f} us in {nover / forwards:.1f} boundaries per forward")
    agg = {}
    for s, e, n in win:
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")
        n = (n.split("(")[0] if not n.startswith("_Z") else n)[:100]
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    print(f"{'kernel':100s} {'calls/fwd':>9s} {'us/fwd':>10s} {'avg us':>9s}")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
        print(f"{n:100s} {c / forwards:9.1f} {t / forwards:10.1f} {t / c:9.2f}")


if __name__ == "__main__":
    main(sys.argv[1], *(int(a) for a in sys.argv[2:]))
