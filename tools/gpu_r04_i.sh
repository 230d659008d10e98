#!/bin/bash
# Round-4 session I: attention variant 6 (MODE 3 arithmetic on 16x16x32 MFMAs) against the shipped form: kernel tests, kernel A/B, forward A/B,
# full-size parity with the variant selected.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04i
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== kernel tests" > $OUT/summary.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention" -p no:cacheprovider 2>&1 | tail -2 >> $OUT/summary.txt
echo "== kernel A/B" >> $OUT/summary.txt
timeout 200 python tools/kbench.py --dtype f16 --only attn --variants 0,6 --iters 10 2>&1 | grep attention >> $OUT/summary.txt
echo "== forward A/B (20 steps): value ms/step gemmTF attnTF" >> $OUT/summary.txt
i=0
for cfg in "MHMR_ATTN_VARIANT=0" "MHMR_ATTN_VARIANT=6" "MHMR_ATTN_VARIANT=0" "MHMR_ATTN_VARIANT=6"; do
  i=$((i+1))
  env $cfg timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_$i.json 2> $OUT/bench_$i.err
  echo "$cfg: $(python -c "import json,sys; d=json.load(open('$OUT/bench_$i.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_attention']['achieved'], d['roofline_attention']['avg_launch_ms'])" 2>&1 | tail -1)" >> $OUT/summary.txt
done
echo "== full-size parity with variant 6" >> $OUT/summary.txt
MHMR_ATTN_VARIANT=6 timeout 900 python -m pytest tests/test_gpu_parity_fullsize.py tests/test_gpu_fullsize.py -q -m gpu -k "f16 or bit or invariance or batch" -p no:cacheprovider 2>&1 | tail -3 >> $OUT/summary.txt
cp gpurun_out/parity_fullsize.json $OUT/parity_fullsize_variant6.json 2>/dev/null
python - >> $OUT/summary.txt <<PY
import json
d = json.load(open("$OUT/parity_fullsize_variant6.json"))
for k, v in sorted(d.items()):
    if k.endswith("f16"): print(k, "worst_rel_l2 %.2e" % v["worst_rel_l2"], "worst_max_norm %.2e" % v["worst_max_norm"])
PY
cat $OUT/summary.txt
