#!/bin/bash
# Round-6 session D: the whole GPU suite on the new defaults (row statistics inside the class-row launches, small-grid attention fallback,
# any-order launches, per-key max-norm gates, the ladder gate); headline A/B of MHMR_CLS_STATS and of the attention form with early copies
# (MHMR_ATTN_VARIANT=7); the pose kernel's timeline; batch-1 latency; one default bench run.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06d}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
S=$OUT/summary.txt
echo "== the whole -m gpu suite" > $S
rm -f gpurun_out/parity_fullsize.json
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log >> $S
grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest_gpu.log | head -30 >> $S
cp gpurun_out/parity_fullsize.json $OUT/ 2>/dev/null
echo "== headline A/B, 20 steps: CLS_STATS 1/0, ATTN_VARIANT 6/7, two alternations" >> $S
for i in 1 2; do for V in "1 6" "0 6" "1 7"; do set -- $V
  MHMR_CLS_STATS=$1 MHMR_ATTN_VARIANT=$2 timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/head.json 2> $OUT/head.err
  python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/head.json"))
print("CLS_STATS=$1 ATTN_VARIANT=$2 run $i:", d["value"], d["ms_per_step"])
PY
done; done
echo "== pose kernel timeline (debug build)" >> $S
for P in 160 1; do MHMR_LIB=tools/dbg/libmhmr_stamps.so timeout 120 python tools/lbs_pose_timeline.py $P 2>&1 | grep -v amdgpu.ids | tail -16 >> $S; done
echo "== latency_b1" >> $S
timeout 300 python bench.py --only-latency > $OUT/lat.json 2> $OUT/lat.err
python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/lat.json"))
print({k: (v["ms"], v["gpu_ms"], v.get("graph_ms")) for k, v in d.items() if isinstance(v, dict)})
PY
echo "== default bench" >> $S
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python - >> $S <<'PY'
import json, os
try:
    d = json.load(open("gpurun_out/%s/bench.json" % os.environ.get("TAG", "r06d")))
    for k in ("value", "ms_per_step", "mfma_utilisation_whole_forward", "source_hash", "precision_resolved", "roofline", "roofline_attention", "lbs", "ms_per_person_lbs", "inference_mode", "parity", "cpu_baseline", "other_precision", "configs", "latency_b1"):
        print(k, json.dumps(d.get(k))[:1200])
except Exception as e:
    print("bench parse failed", e)
PY
tail -3 $OUT/bench.err >> $S
cat $S
