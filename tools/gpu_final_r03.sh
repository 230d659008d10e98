#!/bin/bash
# final validation of the round on the GPU box: the whole -m gpu suite, smoke(), the default bench command
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03f}; export TAG
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log > $OUT/summary.txt
grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log >> $OUT/summary.txt
cp gpurun_out/parity_fullsize.json $OUT/ 2>/dev/null
# the 128x128 kernel everywhere (debug switch): the ViT-L golden must still hold
MHMR_GEMM128=1 timeout 600 python -m pytest tests/test_gpu_model.py -q -k "golden and vitl_224" -p no:cacheprovider 2>&1 | tail -2 >> $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $OUT/summary.txt 2>&1
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python - >> $OUT/summary.txt <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/%s/bench.json" % __import__("os").environ.get("TAG", "r03f")))
    for k in ("value", "ms_per_step", "mfma_utilisation_whole_forward", "roofline", "roofline_attention", "lbs", "inference_mode", "parity", "cpu_baseline", "other_precision", "configs", "lbs_small_batches"):
        print(k, json.dumps(d.get(k))[:600])
except Exception as e:
    print("bench parse failed", e)
PY
tail -5 $OUT/bench.err >> $OUT/summary.txt
cat $OUT/summary.txt
