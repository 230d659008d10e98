#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03m; mkdir -p $OUT; cd $R
for cfg in "MHMR_STAGGER_PCT=100" "MHMR_STAGGER_PCT=50" "MHMR_STAGGER_PCT=0" "MHMR_STAGGER_PCT=100" "MHMR_STAGGER_PCT=50" "MHMR_STAGGER_PCT=25"; do
  echo "$cfg: $(env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_attention']['achieved'])")" | tee -a $OUT/ab_stagger.txt
done
