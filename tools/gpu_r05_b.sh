#!/bin/bash
# Round-5 session B: the fused SMPL-X launch (bit-equality with the two launches, self-cleaning workspace), the layer's timing fused / unfused
# on one box, the x3 unit tests with their final bounds, then the LBS + model tests that ride on the new entry.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05b}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -k "lbs" > $OUT/pytest_lbs.log 2>&1
echo "== pytest lbs: rc $?" > $OUT/summary.txt
tail -4 $OUT/pytest_lbs.log >> $OUT/summary.txt
grep -E "^(FAILED|ERROR)|Error|assert " $OUT/pytest_lbs.log | head -20 >> $OUT/summary.txt
for P in 160 20 1; do
  for F in 1 0; do
    echo "== lbs P=$P fused=$F" >> $OUT/summary.txt
    MHMR_LBS_FUSED=$F timeout 120 python tools/lbs_bench.py $P >> $OUT/summary.txt 2>&1
  done
done
timeout 300 python -m pytest tests/test_gpu_x3.py -q -p no:cacheprovider > $OUT/pytest_x3.log 2>&1
echo "== pytest x3: rc $?" >> $OUT/summary.txt
tail -3 $OUT/pytest_x3.log >> $OUT/summary.txt
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -q -p no:cacheprovider > $OUT/pytest_model.log 2>&1
echo "== pytest model/fullsize: rc $?" >> $OUT/summary.txt
tail -3 $OUT/pytest_model.log >> $OUT/summary.txt
grep -E "^(FAILED|ERROR)" $OUT/pytest_model.log | head >> $OUT/summary.txt
cat $OUT/summary.txt
