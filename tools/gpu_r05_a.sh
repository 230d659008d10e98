#!/bin/bash
# Round-5 session A: the f16x3 precision mode's kernels and parity (new this round), the full-size goldens incl. ViT-B and the auto rule,
# then the bench line with the new latency_b1 leg (no CPU legs) and what the x3 mode costs.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05a}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
rm -f gpurun_out/parity_fullsize.json
timeout 600 python -m pytest tests/test_gpu_x3.py -q -p no:cacheprovider -s > $OUT/pytest_x3.log 2>&1
echo "== pytest x3: rc $?" > $OUT/summary.txt
grep -E "^\[|passed|failed|^FAILED|^ERROR|Error|assert" $OUT/pytest_x3.log | head -60 >> $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity_fullsize.py -q -p no:cacheprovider -s > $OUT/pytest_parity.log 2>&1
echo "== pytest parity fullsize: rc $?" >> $OUT/summary.txt
grep -E "^\[parity|passed|failed|^FAILED|^ERROR" $OUT/pytest_parity.log | cut -c1-400 >> $OUT/summary.txt
cp gpurun_out/parity_fullsize.json $OUT/ 2>/dev/null
timeout 300 python tools/x3_bench.py 8 > $OUT/x3_bench.json 2> $OUT/x3_bench.err
echo "== x3 cost" >> $OUT/summary.txt; cat $OUT/x3_bench.json >> $OUT/summary.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
echo "== bench rc $?" >> $OUT/summary.txt
python - >> $OUT/summary.txt 2>&1 <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("headline", d["value"], d["ms_per_step"], "gemm", d["roofline"]["achieved"], "attn", d["roofline_attention"]["achieved"])
print("lbs", d.get("ms_per_person_lbs"), d.get("lbs", {}).get("layer_ms"))
for c in d.get("configs", []):
    print(c["config"], c["value"], c["ms_per_step"], c["mfma_utilisation_whole_forward"])
print("latency_b1", json.dumps(d.get("latency_b1"), indent=1))
PY
tail -5 $OUT/bench.err >> $OUT/summary.txt
cat $OUT/summary.txt
