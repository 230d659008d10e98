#!/bin/bash
# Round-5 session J (after the one-block output allocation; the traces again with kernel names): where a batch-1 forward's time goes (rocprofv3 kernel trace of 100 forwards: GPU-busy share, launch gaps, kernels
# ranked), eager and replayed from the hipGraph; then the latency legs again with the replay-only figure.  No library change.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05j
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_graph.py tests/test_gpu_model.py tests/test_demo_config1.py -q -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "== pytest graph + model + demo: rc $?" > $OUT/summary.txt
tail -4 $OUT/pytest.log >> $OUT/summary.txt
for M in multiHMR_672_S multiHMR_896_L; do
  for MODE in eager graph; do
    D=$OUT/trace_${M}_${MODE}
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python tools/latency_trace.py run $M 100 $MODE > $OUT/run_${M}_${MODE}.log 2>&1
    echo "== $M $MODE (rc $?)" >> $OUT/summary.txt
    grep "ms per forward" $OUT/run_${M}_${MODE}.log >> $OUT/summary.txt
    python tools/latency_trace.py parse $D 100 >> $OUT/summary.txt 2>&1
    rm -rf $D
  done
done
timeout 400 python bench.py --only-latency > $OUT/latency.json 2> $OUT/latency.err
echo "== latency rc $?" >> $OUT/summary.txt
cat $OUT/latency.json >> $OUT/summary.txt
cat $OUT/summary.txt
