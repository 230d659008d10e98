#!/bin/bash
# Round-5 session G: the SMPL-X vertex kernel with all seven high-half eighths requested up front (LBS_LEAD 7, was 5): correctness, then
# the layer's time against the previous build (build_ab/libmhmr_lead5.so) interleaved on one box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05g
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -k "lbs" > $OUT/pytest_lbs.log 2>&1
echo "== pytest lbs: rc $?" > $OUT/summary.txt
tail -3 $OUT/pytest_lbs.log >> $OUT/summary.txt
for i in 1 2; do
  for P in 160 20 1; do
    echo -n "LEAD=7 " >> $OUT/summary.txt; timeout 120 python tools/lbs_bench.py $P 2>/dev/null | grep "P=" >> $OUT/summary.txt
    echo -n "LEAD=5 " >> $OUT/summary.txt; MHMR_LIB=$R/build_ab/libmhmr_lead5.so timeout 120 python tools/lbs_bench.py $P 2>/dev/null | grep "P=" >> $OUT/summary.txt
  done
done
cat $OUT/summary.txt
