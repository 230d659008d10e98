#!/bin/bash
# Round-6 session G: why is fc1 (N = 4096, GELU) 12 % slower per k tile than Q | K (N = 2048)?  The same shape with a plain / ReLU epilogue,
# and the column-group widths, through tools/kbench.py (token-row map, f16, two rounds).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r06g}; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
S=$OUT/summary.txt
echo "== kbench gemm, rows=map, f16 (default column groups)" > $S
for i in 1 2; do timeout 300 python tools/kbench.py --dtype f16 --only gemm --rows map --iters 20 2>/dev/null >> $S; done
for CG in "0" "8,8" "4,8" "4,16"; do
  echo "== MHMR_COLGROUP=$CG" >> $S
  MHMR_COLGROUP=$CG timeout 300 python tools/kbench.py --dtype f16 --only gemm --rows map --iters 20 2>/dev/null | grep -E "fc1|qk " >> $S
done
cat $S
