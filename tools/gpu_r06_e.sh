#!/bin/bash
# Round-6 session E: after the any-order race fix (stats-carrying class-row launches are ordinary launches) and the pose kernel's register-resident
# level tasks: the whole GPU suite, the SMPL-X layer A/B + timeline, the headline twice with and without any-order launches, two-stream soak.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06e}
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
echo "== SMPL-X layer A/B (MHMR_LBS_POSE1: 1 = the one-wave pose kernel)" >> $S
for i in 1 2; do for V in 0 1; do for P in 160 20 1; do
  echo -n "POSE1=$V " >> $S; MHMR_LBS_POSE1=$V timeout 120 python tools/lbs_bench.py $P 2>/dev/null | tail -1 >> $S
done; done; done
echo "== pose kernel timeline (debug build)" >> $S
for P in 160 1; do MHMR_LIB=tools/dbg/libmhmr_stamps.so timeout 120 python tools/lbs_pose_timeline.py $P 2>&1 | grep -v amdgpu.ids | tail -13 >> $S; done
echo "== headline, 20 steps, ANYORDER 1 / 0, two alternations" >> $S
for i in 1 2; do for V in 1 0; do
  MHMR_ANYORDER=$V timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/head.json 2> $OUT/head.err
  python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/head.json"))
print("ANYORDER=$V run $i:", d["value"], d["ms_per_step"])
PY
done; done
echo "== determinism of the headline forward: 30 forwards, bit-equal features and outputs" >> $S
timeout 400 python tools/forward_determinism.py 30 >> $S 2>&1; timeout 300 python tools/forward_determinism.py 30 8 672 >> $S 2>&1
cat $S
