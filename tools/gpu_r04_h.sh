#!/bin/bash
# Round-4 session H: rocprofv3 kernel traces of the other BASELINE configurations on the final build (the r02 traces predate the row map,
# the LayerNorm fold and the image blocks), and a second sample of the default bench line on another box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in "cfg2 --backbone dinov2_vits14 --img-size 672 --batch 16 --persons 8" "cfg3 --backbone dinov2_vitl14 --img-size 672 --batch 32 --persons 8" "cfg5 --backbone dinov2_vitl14 --img-size 1288 --batch 8 --persons 20"; do
  set -- $c; name=$1; shift
  rocprofv3 --kernel-trace --stats -d $OUT/$name -o $name --output-format csv -- python $R/bench.py "$@" --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/$name.json 2> $OUT/$name.err
  f=$(find $OUT/$name -name "${name}_kernel_stats.csv" | head -1)
  cp "$f" $OUT/${name}_kernel_stats.csv
  find $OUT/$name -name "*kernel_trace.csv" -delete
done
cd $R
timeout 900 python bench.py > $OUT/bench_second_sample.json 2> $OUT/bench.err
python - <<PY
import json, csv
for n in ("cfg2", "cfg3", "cfg5"):
    d = json.load(open("$OUT/%s.json" % n)); print(n, "under rocprof:", d["value"], d["ms_per_step"], d["mfma_utilisation_whole_forward"], "blocks", d["backbone_image_blocks"])
    rows = list(csv.DictReader(open("$OUT/%s_kernel_stats.csv" % n)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows[:9]:
        print("   %-62s calls %5s avg %8.1f us %5.1f %%" % (r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:62], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
d = json.load(open("$OUT/bench_second_sample.json"))
print("second sample:", d["value"], d["ms_per_step"], d["mfma_utilisation_whole_forward"], d["source_hash"], "traffic", d["roofline"]["traffic"], "inference", d["inference_mode"]["host_side_share_ms_per_step"], d["inference_mode"]["host_side_share_ms_median_step"], "lbs", d["lbs"]["layer_ms"])
PY
