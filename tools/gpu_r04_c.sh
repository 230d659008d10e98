#!/bin/bash
# Round-4 session C: full-size parity re-run (max-norm gate at the measured level), where the inference-mode host share goes
# (MHMR_TRACE_HOST), config 5 A/B of the split backbone and of the LayerNorm fold without the row map.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04c}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== parity" > $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity_fullsize.py -q -m gpu -p no:cacheprovider > $OUT/pytest_parity.log 2>&1; tail -4 $OUT/pytest_parity.log >> $OUT/summary.txt
cp gpurun_out/parity_fullsize.json $OUT/ 2>/dev/null
echo "== inference host share" >> $OUT/summary.txt
timeout 600 python tools/infer_host_trace.py >> $OUT/summary.txt 2> $OUT/infer.err
echo "== cfg5 A/B: value ms/step util" >> $OUT/summary.txt
i=0
for cfg in "MHMR_SPLIT=1 MHMR_LNFOLD_ALLROWS=0" "MHMR_SPLIT=2 MHMR_LNFOLD_ALLROWS=0" "MHMR_SPLIT=1" "MHMR_SPLIT=2"; do
  i=$((i+1))
  env $cfg timeout 400 python bench.py --img-size 1288 --batch 8 --persons 20 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/cfg5_$i.json 2> $OUT/cfg5_$i.err
  echo "$cfg: $(python -c "import json; d=json.load(open('$OUT/cfg5_$i.json')); print(d['value'], d['ms_per_step'], d['mfma_utilisation_whole_forward'])" 2>&1 | tail -1)" >> $OUT/summary.txt
done
echo "== cfg2 A/B" >> $OUT/summary.txt
for cfg in "MHMR_SPLIT=1" "MHMR_SPLIT=2"; do
  i=$((i+1))
  env $cfg timeout 400 python bench.py --backbone dinov2_vits14 --img-size 672 --batch 16 --persons 8 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/cfg2_$i.json 2> $OUT/cfg2_$i.err
  echo "$cfg: $(python -c "import json; d=json.load(open('$OUT/cfg2_$i.json')); print(d['value'], d['ms_per_step'], d['mfma_utilisation_whole_forward'])" 2>&1 | tail -1)" >> $OUT/summary.txt
done
cat $OUT/summary.txt
