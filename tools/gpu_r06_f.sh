#!/bin/bash
# Round-6 session F (short): the pose kernel with the host's level schedule -- LBS tests, layer A/B, timeline -- and the class-row statistics test.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06f}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
S=$OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_checkpoint_ingest.py -q -p no:cacheprovider -k "lbs or row_statistics or training_mode or inference_mode or ingest" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log > $S
grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest.log | head -20 >> $S
echo "== SMPL-X layer A/B (MHMR_LBS_POSE1: 1 = the one-wave pose kernel)" >> $S
for i in 1 2; do for V in 0 1; do for P in 160 20 1; do
  echo -n "POSE1=$V " >> $S; MHMR_LBS_POSE1=$V timeout 120 python tools/lbs_bench.py $P 2>/dev/null | tail -1 >> $S
done; done; done
for P in 160 1; do MHMR_LIB=tools/dbg/libmhmr_stamps.so timeout 120 python tools/lbs_pose_timeline.py $P 2>&1 | grep -v amdgpu.ids | tail -13 >> $S; done
cat $S
