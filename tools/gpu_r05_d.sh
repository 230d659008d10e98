#!/bin/bash
# Round-5 session D: fp8 low-half ranges (V and output projection of blocks 0-11 on v_mfma_scale_f32_16x16x128_f8f6f4): kernel tests, the
# full-size goldens with them on, and the headline A/B against the 16-bit low halves (MHMR_LO8=0), interleaved; the SLP reproducer with tuples.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05d}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 200 ./tools/ubench/slp_repro 200 > $OUT/slp_repro.txt 2>&1
echo "== slp repro" > $OUT/summary.txt; head -40 $OUT/slp_repro.txt | cut -c1-400 >> $OUT/summary.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -x > $OUT/pytest_kernels.log 2>&1
echo "== pytest kernels: rc $?" >> $OUT/summary.txt
tail -4 $OUT/pytest_kernels.log >> $OUT/summary.txt
grep -E "^(FAILED|ERROR)|Error|^E  " $OUT/pytest_kernels.log | head -30 >> $OUT/summary.txt
rm -f gpurun_out/parity_fullsize.json
timeout 600 python -m pytest tests/test_gpu_parity_fullsize.py -q -p no:cacheprovider -s -k "f16 or auto" > $OUT/pytest_parity.log 2>&1
echo "== pytest parity fullsize (f16 / auto): rc $?" >> $OUT/summary.txt
grep -E "^\[parity|passed|failed|^FAILED|^ERROR" $OUT/pytest_parity.log | cut -c1-330 >> $OUT/summary.txt
cp gpurun_out/parity_fullsize.json $OUT/ 2>/dev/null
for i in 1 2; do
  for V in 1 0; do
    MHMR_LO8=$V timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_lo8_${V}_$i.json 2> $OUT/bench_lo8_${V}_$i.err
    python - >> $OUT/summary.txt 2>&1 <<PY
import json
d = json.load(open("$OUT/bench_lo8_${V}_$i.json"))
print("LO8=$V run $i: %.2f img/s %.2f ms/step gemm %.1f TF/s executed %.1f avg %.4f ms attn %.1f" % (d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["executed"], d["roofline"]["avg_launch_ms"], d["roofline_attention"]["achieved"]))
PY
  done
done
cat $OUT/summary.txt
