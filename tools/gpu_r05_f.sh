#!/bin/bash
# Round-5 session F: tiny batches with all rows through the big GEMMs (no class-row launches): batch-1 latency A/B (interleaved), then the
# tests that ride on the rule (small goldens, inference goldens, batch invariance, the full-size goldens in both row modes).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05f}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== latency_b1 A/B" > $OUT/summary.txt
for i in 1 2; do
  for V in 1 0; do
    MHMR_TINY_ALLROWS=$V timeout 200 python bench.py --only-latency > $OUT/lat_${V}_$i.json 2> $OUT/lat_${V}_$i.err
    python - >> $OUT/summary.txt 2>&1 <<PY
import json
d = json.load(open("$OUT/lat_${V}_$i.json"))
print("TINY_ALLROWS=$V run $i:", {k: (v["ms"], v["gpu_ms"]) for k, v in d.items() if isinstance(v, dict)})
PY
  done
done
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_demo_config1.py tests/test_anny_model.py -q -p no:cacheprovider > $OUT/pytest_model.log 2>&1
echo "== pytest model / fullsize / demo / anny: rc $?" >> $OUT/summary.txt
tail -3 $OUT/pytest_model.log >> $OUT/summary.txt
grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest_model.log | head -20 >> $OUT/summary.txt
rm -f gpurun_out/parity_fullsize.json
timeout 600 python -m pytest tests/test_gpu_parity_fullsize.py -q -p no:cacheprovider -s -k "f16 or auto" > $OUT/pytest_parity.log 2>&1
echo "== pytest parity fullsize: rc $?" >> $OUT/summary.txt
grep -E "^\[parity|passed|failed|^FAILED|^ERROR" $OUT/pytest_parity.log | cut -c1-300 >> $OUT/summary.txt
cp gpurun_out/parity_fullsize.json $OUT/ 2>/dev/null
cat $OUT/summary.txt
