#!/bin/bash
# Round-6 session J: the headline with the backbone as two image blocks on two streams (MHMR_SPLIT=2; round 4 measured +0.3 %), re-measured on this round's kernels.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r06j}; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
S=$OUT/summary.txt
echo "== headline, 20 steps, MHMR_SPLIT = 1 / 2 / 4" > $S
for i in 1 2 3; do for V in 1 2 4; do
  MHMR_SPLIT=$V timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/head.json 2> $OUT/head.err
  python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/head.json"))
print("SPLIT=$V run $i:", d["value"], d["ms_per_step"], "blocks", d["backbone_image_blocks"])
PY
done; done
cat $S
