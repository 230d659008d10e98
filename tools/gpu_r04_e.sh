#!/bin/bash
# Round-4 session E: image blocks one after the other on ONE stream (activations of half a batch are MALL-sized), cross-attention after the
# real-work-first grid fix (kernel trace of the heads).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04e}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== tests (heads)" > $OUT/summary.txt
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_anny_hph.py tests/test_anny_model.py -q -m gpu -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log >> $OUT/summary.txt
echo "== bench A/B (20 steps): value ms/step gemmTF attnTF" >> $OUT/summary.txt
i=0
for cfg in "MHMR_SPLIT=1" "MHMR_SPLIT=2 MHMR_SPLIT_SEQ=1" "MHMR_SPLIT=4 MHMR_SPLIT_SEQ=1" "MHMR_SPLIT=1" "MHMR_SPLIT=2 MHMR_SPLIT_SEQ=1"; do
  i=$((i+1))
  env $cfg timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_$i.json 2> $OUT/bench_$i.err
  echo "$cfg: $(python -c "import json,sys; d=json.load(open('$OUT/bench_$i.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'], d['roofline_attention']['achieved'])" 2>&1 | tail -1)" >> $OUT/summary.txt
done
echo "== kernel trace of the headline (heads after the cross-attention fix)" >> $OUT/summary.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o headline --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --only-headline-kernels > $OUT/trace_headline.json 2> $OUT/trace_err.txt
cd $R
python - >> $OUT/summary.txt 2>&1 <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/headline_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("hph_", "linear_f32", "lbs_", "person_groups", "detect", "layernorm_f32")):
        print(n.replace("(anonymous namespace)::", "")[:60], r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3))
PY
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/summary.txt
