#!/bin/bash
# Round-5 session L: kernel trace of the f16x3 forward (ViT-L 672^2 x 8, hostile weights -> precision "auto" packs f16x3): what the mode's
# 4.4x is made of, and the fp32 attention kernel's rate against the fp32 MFMA peak.  No library change.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05l
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/x3 -o x3 --output-format csv -- python $R/tools/x3_bench.py 8 x3only > $OUT/x3.json 2> $OUT/x3.err
echo "== x3 trace rc $?" > $OUT/summary.txt
cd $R
f=$(find $OUT/x3 -name "x3_kernel_stats.csv" | head -1)
cp "$f" $OUT/x3_kernel_stats.csv
find $OUT/x3 -name "*kernel_trace.csv" -delete
cat $OUT/x3.json >> $OUT/summary.txt
python - >> $OUT/summary.txt <<PY
import csv
rows = list(csv.DictReader(open("$OUT/x3_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    n = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:70]
    print("   %-70s calls %5s avg %9.1f us %5.1f %%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
    if "attn_f32" in n:
        fl = 4.0 * 8 * 16 * 2305 * 2305 * 64
        print("      -> %.1f TFLOP/s algorithmic of the 157.3 dense fp32-MFMA peak = %.3f" % (fl / float(r["AverageNs"]) / 1e3, fl / float(r["AverageNs"]) / 1e3 / 157.3))
PY
cat $OUT/summary.txt
