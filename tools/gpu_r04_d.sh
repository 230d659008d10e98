#!/bin/bash
# Round-4 session D: SMPL-X pose kernel with wide stores, person dicts before the heads, the hostile parity bound, LBS layer times,
# two-stream soak with a split backbone.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04d}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== tests" > $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_parity_fullsize.py -q -m gpu -k "lbs or model or hostile or person or capacity or hook or sharded" -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log >> $OUT/summary.txt
echo "== inference host share" >> $OUT/summary.txt
timeout 600 python tools/infer_host_trace.py >> $OUT/summary.txt 2> $OUT/infer.err
echo "== LBS layer" >> $OUT/summary.txt
for p in 160 20 1 256; do timeout 120 python tools/lbs_bench.py $p >> $OUT/summary.txt 2>> $OUT/lbs.err; done
echo "== two-stream soak (ViT-L 24 blocks, 224^2, 8 images, 2 image blocks per model, 2 host threads)" >> $OUT/summary.txt
MHMR_SOAK_BATCH=8 MHMR_SPLIT=2 REPS=200 timeout 600 python tools/two_stream_check.py 24 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tail -4 >> $OUT/summary.txt
MHMR_SOAK_BATCH=2 REPS=300 timeout 600 python tools/two_stream_check.py 24 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tail -2 >> $OUT/summary.txt
cat $OUT/summary.txt
