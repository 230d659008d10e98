#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03g; mkdir -p $OUT; cd $R
for i in 1 2; do for lib in default build_ab/noslp build_ab/clsnoslp; do
  echo "== $lib" >> $OUT/ab_slp.txt
  python tools/run_with_lib.py $lib bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 >> $OUT/ab_slp.txt
done; done
python - <<'PY'
import json
for l in open("gpurun_out/r03g/ab_slp.txt"):
    if l.startswith("=="): print(l.strip(), end="  ")
    else:
        d = json.loads(l); print(d["value"], d["ms_per_step"])
PY
