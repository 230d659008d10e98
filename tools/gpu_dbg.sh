#!/bin/bash
# attention forms on the no-SLP build: kernel alone and inside the forward
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03i; mkdir -p $OUT; cd $R
timeout 300 python tools/kbench.py --dtype f16 --only attn --variants 0,1,2,3,4,5 --iters 10 2>&1 | grep -v amdgpu.ids > $OUT/kb_attn.txt
cat $OUT/kb_attn.txt
for cfg in "X=1" "MHMR_ATTN_VARIANT=4" "MHMR_ATTN_VARIANT=5" "X=2" "MHMR_ATTN_VARIANT=4"; do
  env $cfg timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $OUT/b.json
  echo "$cfg: $(python -c "import json; d=json.load(open('$OUT/b.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_attention']['achieved'])")" | tee -a $OUT/ab_attn.txt
done
