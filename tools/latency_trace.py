#!/usr/bin/env python
"""Where a batch-1 forward's time goes: GPU-busy share and the kernels of one forward, from a rocprofv3 kernel trace.

  rocprofv3 --kernel-trace --output-format csv -d OUT -- python tools/latency_trace.py run multiHMR_672_S 100 [graph]
  python tools/latency_trace.py parse OUT 100

`run` builds the model (seeded weights), warms up, sleeps 0.3 s (the marker `parse` looks for) and calls demo.forward_model `reps` times
(batch 1, inference mode, 4 persons; `graph`: forward_model(use_graph=True)).  `parse` takes the kernels after the last idle gap > 0.1 s:
kernels per forward, the sum of their durations per forward, the idle time between consecutive kernels per forward (launch gaps: the
forward is a chain of dependent launches on one stream), and the kernels ranked by their share."""
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(name, reps, graph):
    import torch
    import torch.nn.functional as F
    import bench
    import synthetic
    from multi_hmr_amd import demo
    dev = torch.device("cuda", 0)
    backbone, S = {n: (b, s) for n, b, s in bench.LATENCY_MODELS}[name]
    model = bench.build_model(backbone, S, "f16", synthetic.make_smplx_data(0), synthetic.make_mean_params(0), dev)
    x, K, idx = bench.make_inputs(1, S, 4, 0, dev)
    s = model(x, idx=idx, K=K, is_training=True)["scores"][..., 0]
    m = F.max_pool2d(s[:, None], 3, stride=1, padding=1)[:, 0]
    surv = torch.sort(s[m == s], descending=True).values
    thr = float(0.5 * (surv[3] + surv[4]))
    call = lambda: demo.forward_model(model, x, K, det_thresh=thr, nms_kernel_size=3, use_graph=graph)
    for _ in range(5):
        humans = call()
    torch.cuda.synchronize()
    time.sleep(0.3)
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    print(f"{name} {'graph' if graph else 'eager'}: {len(humans)} persons, {1e3 * (time.perf_counter() - t0) / reps:.3f} ms per forward under the tracer")


def parse(out, reps):
    files = glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)
    assert files, "no *kernel_trace.csv under " + out
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - rows[i - 1][1] > 100_000_000:
            cut = i
    win = rows[cut:]
    span = (win[-1][1] - win[0][0]) / 1e3
    busy = sum(e - s for s, e, _ in win) / 1e3
    gaps = sum(max(win[i][0] - win[i - 1][1], 0) for i in range(1, len(win))) / 1e3
    print(f"{len(win)} kernels in the timed window = {len(win) / reps:.1f} per forward; window {span / reps:.1f} us per forward: "
          f"kernels {busy / reps:.1f} us ({100 * busy / span:.1f} %), idle between kernels {gaps / reps:.1f} us ({100 * gaps / span:.1f} %) "
          f"= {gaps / max(len(win) - 1, 1):.2f} us per launch boundary")
    agg = {}
    for s, e, n in win:
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")
        n = (n.split("(")[0] if not n.startswith("_Z") else n)[:90]
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    print(f"{'kernel':90s} {'calls/fwd':>9s} {'us/fwd':>9s} {'avg us':>8s} {'share':>6s}")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"{n:90s} {c / reps:9.1f} {t / reps:9.1f} {t / c:8.2f} {100 * t / busy:5.1f}%")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]), len(sys.argv) > 4 and sys.argv[4] == "graph")
    else:
        parse(sys.argv[2], int(sys.argv[3]))
