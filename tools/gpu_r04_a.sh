#!/bin/bash
# Round-4 session A on the GPU box (one gpurun call): the new device-side bookkeeping / capacity / split tests, then whole-forward A/B of
# the backbone image blocks on side streams (MHMR_SPLIT) and of the column-group widths (MHMR_COLGROUP=a,b).  -> gpurun_out/$TAG
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04a}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== tests" > $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_multi.py tests/test_gpu_fullsize.py -x -q -m gpu -p no:cacheprovider > $OUT/pytest_model.log 2>&1; tail -5 $OUT/pytest_model.log >> $OUT/summary.txt
echo "== bench A/B (20 steps, f16): value ms/step gemmTF attnTF" >> $OUT/summary.txt
i=0
for cfg in "MHMR_SPLIT=1" "MHMR_SPLIT=2" "MHMR_SPLIT=4" "MHMR_SPLIT=1 MHMR_COLGROUP=4,8" "MHMR_SPLIT=1 MHMR_COLGROUP=8,8" "MHMR_SPLIT=1 MHMR_COLGROUP=8,4" "MHMR_SPLIT=2 MHMR_COLGROUP=4,8" "MHMR_SPLIT=1" "MHMR_SPLIT=2"; do
  i=$((i+1))
  env $cfg timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_$i.json 2> $OUT/bench_$i.err
  echo "$cfg: $(python -c "import json,sys; d=json.load(open('$OUT/bench_$i.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'], d['roofline_attention']['achieved'])" 2>&1 | tail -1)" >> $OUT/summary.txt
done
echo "== full default line with MHMR_SPLIT=2 (inference leg, other configs)" >> $OUT/summary.txt
MHMR_SPLIT=2 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_full_split2.json 2> $OUT/bench_full_split2.err
python - >> $OUT/summary.txt 2>&1 <<PY
import json
d = json.load(open("$OUT/bench_full_split2.json"))
print("headline", d["value"], d["ms_per_step"])
print("inference", d.get("inference_mode"))
print("lbs", d.get("lbs", {}).get("layer_ms"), d.get("lbs_small_batches"))
print("other_precision", d.get("other_precision"))
for c in d.get("configs", []):
    print(c["config"], c["value"], c["ms_per_step"], c["mfma_utilisation_whole_forward"])
PY
cat $OUT/summary.txt
