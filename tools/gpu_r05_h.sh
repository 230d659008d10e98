#!/bin/bash
# Round-5 session H: the hipGraph replay of the inference forward (multi_hmr_amd/graphed.py): bit equality with the eager forward, then the
# batch-1 latency of both (bench.py --only-latency: eager `ms` beside `graph_ms`, same process, same box).  No library change (source hash as before).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05h
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_gpu_graph.py tests/test_gpu_model.py -q -p no:cacheprovider -k "graph or fixed_capacity or inference_mode" > $OUT/pytest_graph.log 2>&1
echo "== pytest graph: rc $?" > $OUT/summary.txt
tail -15 $OUT/pytest_graph.log >> $OUT/summary.txt
timeout 400 python bench.py --only-latency > $OUT/latency.json 2> $OUT/latency.err
echo "== latency rc $?" >> $OUT/summary.txt
cat $OUT/latency.json >> $OUT/summary.txt
tail -5 $OUT/latency.err >> $OUT/summary.txt
cat $OUT/summary.txt
