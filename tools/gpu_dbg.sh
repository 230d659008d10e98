#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03k; mkdir -p $OUT; cd $R
for cfg in "MHMR_DESCEND=0" "MHMR_DESCEND=1" "MHMR_DESCEND=0" "MHMR_DESCEND=1"; do
  echo "$cfg: $(env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_attention']['achieved'])")" | tee -a $OUT/ab_descend.txt
done
