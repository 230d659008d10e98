#!/bin/bash
# Round-5 session M: the person tables in one zeroed block (one fill launch instead of five per inference forward): every test that runs
# the inference path, smoke(), and the latency legs.  No library change.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05m
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 700 python -m pytest tests/test_gpu_graph.py tests/test_gpu_model.py tests/test_demo_config1.py tests/test_gpu_x3.py tests/test_gpu_multi.py tests/test_anny_model.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "== pytest (inference-path tests): rc $?" > $OUT/summary.txt
tail -4 $OUT/pytest.log >> $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids >> $OUT/summary.txt
timeout 400 python bench.py --only-latency > $OUT/latency.json 2> $OUT/latency.err
echo "== latency rc $?" >> $OUT/summary.txt
python - >> $OUT/summary.txt <<PY
import json
d = json.load(open("$OUT/latency.json"))
for k, v in d.items():
    if isinstance(v, dict):
        print(k, {n: v[n] for n in ("ms", "gpu_ms", "graph_ms", "graph_replay_ms", "graph_equals_eager", "persons")})
PY
cat $OUT/summary.txt
