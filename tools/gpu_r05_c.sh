#!/bin/bash
# Round-5 session C: the fused SMPL-X launch with ONE release / acquire per workgroup: correctness again, then fused / unfused timings; the
# SLP reproducer (tools/ubench/slp_repro).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05c}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -k "lbs" > $OUT/pytest_lbs.log 2>&1
echo "== pytest lbs: rc $?" > $OUT/summary.txt
tail -4 $OUT/pytest_lbs.log >> $OUT/summary.txt
grep -E "^(FAILED|ERROR)|Error|assert " $OUT/pytest_lbs.log | head -20 >> $OUT/summary.txt
for P in 160 20 1 48; do
  for F in 1 0; do
    MHMR_LBS_FUSED=$F timeout 120 python tools/lbs_bench.py $P 2>/dev/null | grep "P=" >> $OUT/summary.txt
  done
done
echo "== slp repro" >> $OUT/summary.txt
timeout 300 ./tools/ubench/slp_repro 400 > $OUT/slp_repro.txt 2>&1
cat $OUT/slp_repro.txt >> $OUT/summary.txt
cat $OUT/summary.txt
