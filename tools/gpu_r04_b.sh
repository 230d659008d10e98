#!/bin/bash
# Round-4 session B on the GPU box: the whole -m gpu suite (device-side person set, capacity inference, split backbone, one-product
# pose correctives in the SMPL-X blend, LayerNorm fold without the row map, hostile goldens, max-norm gate), then the default bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04b}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "== pytest -m gpu" > $OUT/summary.txt
tail -15 $OUT/pytest_gpu.log >> $OUT/summary.txt
grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log >> $OUT/summary.txt
cp gpurun_out/parity_fullsize.json $OUT/ 2>/dev/null
echo "== default bench line (no CPU legs)" >> $OUT/summary.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - >> $OUT/summary.txt 2>&1 <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("headline", d["value"], d["ms_per_step"], "gemm", d["roofline"]["achieved"], "attn", d["roofline_attention"]["achieved"])
print("inference", d.get("inference_mode"))
print("lbs", d.get("lbs"), d.get("lbs_small_batches"))
print("other_precision", d.get("other_precision", {}).get("value"))
for c in d.get("configs", []):
    print(c["config"], c["value"], c["ms_per_step"], c["mfma_utilisation_whole_forward"])
PY
python - >> $OUT/summary.txt 2>&1 <<PY
import json
d = json.load(open("$OUT/parity_fullsize.json"))
for k, v in sorted(d.items()):
    print(k, "worst_rel_l2 %.2e" % v["worst_rel_l2"], "worst_max_norm %.2e" % v.get("worst_max_norm", -1), "fold", v.get("lnfold"), "maxvert_mm %.3f" % v["max_vertex_error_mm"])
    if "hostile" in k:
        print("   rel", {a: float("%.2e" % b) for a, b in v["rel_l2"].items()})
PY
cat $OUT/summary.txt
