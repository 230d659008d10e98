"""-m "not gpu": the N > 1 CONTROL FLOW of bench.py under gloo, on CPU, with a stand-in model (MHMR_BENCH_STUB=1).

What runs is bench.py's own code: the per-step asynchronous collation (`collate.allgather_persons_async`, waited one step later),
drain + barrier on both sides of the timed region, the all_reduce(MAX) of the clock, rank 0's extra passes while the other ranks
wait at the barrier behind them, the single JSON line of rank 0, destroy_process_group.  A hang or a rank-divergent collective in
that flow is found here, not on the 8-GPU node (which the builder never sees).  Both launch forms are covered: the driver's
`python -m torch.distributed.run ... bench.py --gpus N` and the bare `python bench.py --gpus N` that starts its own ranks."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--steps", "3", "--warmup", "2", "--batch", "4", "--persons", "3", "--img-size", "224"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _check(proc, n):
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, proc.stdout          # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["stub"] is True and d["n_gpus"] == n and d["steps"] == 3 and d["warmup"] == 2
    assert d["config"]["global_batch"] == 4 * n and d["scaling"] == "weak" and d["value"] > 0
    mg = d["multi_gpu"]
    assert mg["max_rank_compute_ms_per_step"] >= 0 and "exposed_collation_ms_per_step" in mg
    assert abs(d["value"] - 4 * n * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-2      # value = all ranks' images / max-over-ranks time


@pytest.mark.parametrize("n", [2, 3])
def test_bench_control_flow_under_the_drivers_launch_form(n):
    env = dict(os.environ, MHMR_BENCH_STUB="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + ARGS
    _check(subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300), n)


def test_bench_starts_its_own_ranks_in_stub_mode():
    env = dict(os.environ, MHMR_BENCH_STUB="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + ARGS
    _check(subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300), 2)
