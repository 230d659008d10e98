#!/bin/bash
# Second final validation of round 5 (after the Python-side changes of sessions H-J: hipGraph wrapper, one-block head outputs; the library
# is the one of tools/gpu_final_r05.sh, same source hash, so its rocprofv3 / PMC evidence stands): the whole -m gpu suite, smoke(), the
# default bench command, the bench under torch.distributed.run.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05y}; export TAG
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
rm -f gpurun_out/parity_fullsize.json
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log > $OUT/summary.txt
grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log >> $OUT/summary.txt
cp gpurun_out/parity_fullsize.json $OUT/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $OUT/summary.txt 2>&1
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python - >> $OUT/summary.txt <<'PY'
import json, os
try:
    d = json.load(open("gpurun_out/%s/bench.json" % os.environ.get("TAG", "r05y")))
    for k in ("value", "ms_per_step", "mfma_utilisation_whole_forward", "source_hash", "roofline", "roofline_attention", "lbs", "ms_per_person_lbs", "inference_mode", "parity", "cpu_baseline", "other_precision", "configs", "latency_b1"):
        print(k, json.dumps(d.get(k))[:1500])
except Exception as e:
    print("bench parse failed", e)
PY
tail -3 $OUT/bench.err >> $OUT/summary.txt
echo "== bench under torch.distributed.run, RCCL group of one" >> $OUT/summary.txt
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400 >> $OUT/summary.txt
cat $OUT/summary.txt
