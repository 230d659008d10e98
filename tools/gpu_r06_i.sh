#!/bin/bash
# Round-6 session I (needs tools/ubench/attention_priority_experiment.patch applied to csrc/attention.hip: variants 10-18 = other s_setprio levels of the
# two matrix sections of attn16_kernel; none beat the shipped one, the code was not kept): kbench + the headline per variant.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r06i}; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
S=$OUT/summary.txt
echo "== attention s_setprio forms (kbench, f16): 6 = both sections at prio 1 (shipped), 10 = none, 11 = score MFMAs only, 12 = PV MFMAs only" > $S
timeout 300 python tools/kbench.py --dtype f16 --only attn --variants 6,13,14,15,16,17,18 --iters 10 2>/dev/null >> $S
timeout 300 python tools/kbench.py --dtype f16 --only attn --variants 6,13,14,15,16,17,18 --iters 10 2>/dev/null >> $S
echo "== in the forward, 20 steps" >> $S
for i in 1 2; do for V in 6 13 14 15 16 17 18; do
  MHMR_ATTN_VARIANT=$V timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/head.json 2> $OUT/head.err
  python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/head.json"))
print("ATTN_VARIANT=$V run $i:", d["value"], d["ms_per_step"])
PY
done; done
cat $S
