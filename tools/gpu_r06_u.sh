#!/bin/bash
# Round-6 session U: config 2's kernel trace on the final build (profiles/r06_cfg2_kernel_stats.csv was taken on the mid-round build).
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r06u}; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
S=$OUT/summary.txt
echo "== library: $(python -c 'from multi_hmr_amd import _lib; print(_lib.built_source_hash())')" > $S
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/cfg2 -o cfg2 --output-format csv -- python $R/bench.py --backbone dinov2_vits14 --img-size 672 --batch 16 --persons 8 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/cfg2.json 2> $OUT/cfg2.err)
f=$(find $OUT/cfg2 -name "cfg2_kernel_stats.csv" | head -1); cp "$f" $OUT/cfg2_kernel_stats.csv 2>/dev/null
find $OUT/cfg2 -name "*kernel_trace.csv" -delete
python - >> $S <<PY
import json, csv
d = json.load(open("$OUT/cfg2.json")); print("cfg2 under rocprof:", d["value"], d["ms_per_step"], d["mfma_utilisation_whole_forward"], "blocks", d["backbone_image_blocks"])
rows = list(csv.DictReader(open("$OUT/cfg2_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    print("%-100s %6s %8.1f us %6.2f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
cat $S
