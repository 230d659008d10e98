#!/bin/bash
# Final validation of round 5 on the GPU box: the whole -m gpu suite, smoke(), the default bench command, then the round's rocprofv3
# evidence (tools/collect_profiles.sh) and the two-stream soak -- all on ONE build (the library says which sources it was made from).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05z}; export TAG
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
rm -f gpurun_out/parity_fullsize.json
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log > $OUT/summary.txt
grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log >> $OUT/summary.txt
cp gpurun_out/parity_fullsize.json $OUT/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $OUT/summary.txt 2>&1
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python - >> $OUT/summary.txt <<'PY'
import json, os
try:
    d = json.load(open("gpurun_out/%s/bench.json" % os.environ.get("TAG", "r05z")))
    for k in ("value", "ms_per_step", "mfma_utilisation_whole_forward", "source_hash", "backbone_image_blocks", "roofline", "roofline_attention", "lbs", "ms_per_person_lbs", "inference_mode", "parity", "cpu_baseline", "other_precision", "configs", "latency_b1"):
        print(k, json.dumps(d.get(k))[:900])
except Exception as e:
    print("bench parse failed", e)
PY
tail -3 $OUT/bench.err >> $OUT/summary.txt
echo "== two-stream soak" >> $OUT/summary.txt
MHMR_SOAK_BATCH=8 MHMR_SPLIT=2 REPS=200 timeout 500 python tools/two_stream_check.py 24 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tail -3 >> $OUT/summary.txt
echo "== the opt-in paths on this build (MHMR_LO8=1: fp8 low-half ranges; MHMR_LBS_FUSED=1: one-launch SMPL-X layer)" >> $OUT/summary.txt
MHMR_LO8=1 timeout 300 python -m pytest tests/test_gpu_parity_fullsize.py -q -p no:cacheprovider -k "vitl_672_full-f16 or vitl_896_full-f16" 2>&1 | tail -1 >> $OUT/summary.txt
MHMR_LBS_FUSED=1 timeout 300 python -m pytest tests/test_gpu_model.py -q -p no:cacheprovider -k "training_mode or inference_mode" 2>&1 | tail -1 >> $OUT/summary.txt
echo "== bench under torch.distributed.run, RCCL group of one" >> $OUT/summary.txt
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400 >> $OUT/summary.txt
echo "== x3 cost" >> $OUT/summary.txt
timeout 300 python tools/x3_bench.py 8 >> $OUT/summary.txt 2>/dev/null
echo "== profiles" >> $OUT/summary.txt
timeout 1500 bash tools/collect_profiles.sh r05 > $OUT/collect.log 2>&1
tail -12 $OUT/collect.log >> $OUT/summary.txt
cat $OUT/summary.txt
