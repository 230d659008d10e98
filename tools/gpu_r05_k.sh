#!/bin/bash
# Round-5 session K: image blocks n = 1 / 2 / 4 (/ 8) for configs 2, 3, 5 (is the automatic rule's "two" right?), and this round's
# kernel trace of config 2.  No library change.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05k
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 400 python tools/split_probe.py > $OUT/split_probe.txt 2> $OUT/split_probe.err
echo "== split probe rc $?" > $OUT/summary.txt
cat $OUT/split_probe.txt >> $OUT/summary.txt
tail -3 $OUT/split_probe.err >> $OUT/summary.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/cfg2 -o cfg2 --output-format csv -- python $R/bench.py --backbone dinov2_vits14 --img-size 672 --batch 16 --persons 8 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/cfg2.json 2> $OUT/cfg2.err
echo "== cfg2 trace rc $?" >> $OUT/summary.txt
cd $R
f=$(find $OUT/cfg2 -name "cfg2_kernel_stats.csv" | head -1)
cp "$f" $OUT/cfg2_kernel_stats.csv
find $OUT/cfg2 -name "*kernel_trace.csv" -delete
python - >> $OUT/summary.txt <<PY
import json, csv
d = json.load(open("$OUT/cfg2.json")); print("cfg2 under rocprof:", d["value"], d["ms_per_step"], d["mfma_utilisation_whole_forward"], "blocks", d["backbone_image_blocks"])
rows = list(csv.DictReader(open("$OUT/cfg2_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    print("   %-62s calls %5s avg %8.1f us %5.1f %%" % (r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:62], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
cat $OUT/summary.txt
