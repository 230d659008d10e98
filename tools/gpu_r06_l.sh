#!/bin/bash
# Round-6 session L: would UNEVEN image blocks pay at config 3 (672^2 ViT-L, 32 images)?  At 672^2 an image is 9 row tiles: blocks whose size is a
# multiple of 7 fill whole rounds of 256 CUs in every linear (28 images: 1008 / 2016 / 4032 tiles = 3.94 / 7.9 / 15.75 rounds; 16 images: 2.25 / 4.5 / 9).
# Probe without new code: the step time at 32 images (two blocks of 16: the shipped rule), and at 28 and 4 images alone (one block each).
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r06l}; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
S=$OUT/summary.txt
echo "== ViT-L 672^2, 8 persons per image, 20 steps: batch / MHMR_SPLIT" > $S
for i in 1 2; do for V in "32 0" "32 1" "28 1" "28 2" "4 1" "14 1" "21 1"; do set -- $V
  MHMR_SPLIT=$2 timeout 300 python bench.py --img-size 672 --batch $1 --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/head.json 2> $OUT/head.err
  python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/head.json"))
print("batch $1 split $2 run $i: %.1f img/s  %.3f ms/step  %.4f ms/image  blocks %s" % (d["value"], d["ms_per_step"], d["ms_per_step"] / $1, d["backbone_image_blocks"]))
PY
done; done
cat $S
