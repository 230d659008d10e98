#!/bin/bash
# Round-6 session C: (1) the merged qkv launch + fc1 row map of a batch of one: kernel test, latency A/B; (2) attention experiment forms
# (variants 7 / 8 / 9) against the shipped one; (3) all GPU tests on the new defaults (proj@0-11 low halves, ViT-S back on the 128x128 kernel);
# (4) MHMR_ANYORDER A/B, three alternations; (5) the precision rule's ladder at 896^2 and 1288^2.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06c}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
S=$OUT/summary.txt
echo "== pytest kernels (qkv / attention)" > $S
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -k "qkv or test_attention" > $OUT/pytest_k.log 2>&1; tail -3 $OUT/pytest_k.log >> $S
grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest_k.log | head -20 >> $S
echo "== latency_b1: (QKV_MERGE, FC1_ROWMAP) = (0,0) (1,0) (1,1), two rounds" >> $S
for i in 1 2; do for V in "0 0" "1 0" "1 1"; do set -- $V
  MHMR_QKV_MERGE=$1 MHMR_FC1_ROWMAP=$2 timeout 300 python bench.py --only-latency > $OUT/lat.json 2> $OUT/lat.err
  python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/lat.json"))
print("QKV_MERGE=$1 FC1_ROWMAP=$2 run $i:", {k: (v["ms"], v["gpu_ms"]) for k, v in d.items() if isinstance(v, dict)})
PY
done; done
echo "== attention forms (kbench, f16, interleaved)" >> $S
timeout 300 python tools/kbench.py --dtype f16 --only attn --variants 6,7,8,9 --iters 10 2>/dev/null >> $S
timeout 300 python tools/kbench.py --dtype f16 --only attn --variants 6,7,8,9 --iters 10 --batch 1 2>/dev/null >> $S
echo "== the whole -m gpu suite" >> $S
rm -f gpurun_out/parity_fullsize.json
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log >> $S
grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest_gpu.log | head -30 >> $S
cp gpurun_out/parity_fullsize.json $OUT/ 2>/dev/null
echo "== headline A/B (MHMR_ANYORDER), 20 steps, three alternations" >> $S
for i in 1 2 3; do for V in 0 1; do
  MHMR_ANYORDER=$V timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/head.json 2> $OUT/head.err
  python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/head.json"))
print("ANYORDER=$V run $i:", d["value"], d["ms_per_step"])
PY
done; done
echo "== precision ladder at 896^2" >> $S
PROBE_SIZE=896 timeout 900 python tools/auto_rule_probe.py 0.0 0.3 0.45 0.55 0.65 0.8 2>/dev/null >> $S
echo "== precision ladder at 1288^2" >> $S
PROBE_SIZE=1288 timeout 1200 python tools/auto_rule_probe.py 0.0 0.3 0.45 0.55 0.65 2>/dev/null >> $S
cat $S
