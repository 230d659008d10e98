#!/bin/bash
# Round-6 session A: (1) does hipExtAnyOrderLaunch overlap on gfx950 (tools/ubench/anyorder.hip); (2) split-k residual linears: kernel tests,
# batch-1 latency A/B (MHMR_SPLITK), full-size goldens; (3) any-order launches of V / class-row linears: headline A/B (MHMR_ANYORDER);
# (4) the minimal low-half set on the current kernels (tools/wlo_study_gpu.py).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06a}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
S=$OUT/summary.txt
echo "== anyorder ubench" > $S
(cd tools/ubench && timeout 60 ./anyorder) >> $S 2>&1
echo "== pytest splitk kernels" >> $S
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -k "splitk or low_half or layernorm_fold" > $OUT/pytest_k.log 2>&1; tail -3 $OUT/pytest_k.log >> $S
grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest_k.log | head -20 >> $S
echo "== latency_b1 A/B (MHMR_SPLITK)" >> $S
for i in 1 2; do
  for V in 1 0; do
    MHMR_SPLITK=$V timeout 300 python bench.py --only-latency > $OUT/lat_${V}_$i.json 2> $OUT/lat_${V}_$i.err
    python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/lat_${V}_$i.json"))
print("SPLITK=$V run $i:", {k: (v["ms"], v["gpu_ms"]) for k, v in d.items() if isinstance(v, dict)})
PY
  done
done
echo "== headline A/B (MHMR_ANYORDER), 20 steps" >> $S
for i in 1 2; do
  for V in 0 1; do
    MHMR_ANYORDER=$V timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/head_${V}_$i.json 2> $OUT/head_${V}_$i.err
    python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/head_${V}_$i.json"))
print("ANYORDER=$V run $i:", d["value"], d["ms_per_step"])
PY
  done
done
echo "== parity fullsize (f16 / auto)" >> $S
rm -f gpurun_out/parity_fullsize.json
timeout 900 python -m pytest tests/test_gpu_parity_fullsize.py tests/test_gpu_fullsize.py -q -p no:cacheprovider -s -k "f16 or auto or not bf16" > $OUT/pytest_parity.log 2>&1
echo "rc $?" >> $S
grep -E "^\[parity|passed|failed|^FAILED|^ERROR" $OUT/pytest_parity.log | cut -c1-300 >> $S
cp gpurun_out/parity_fullsize.json $OUT/ 2>/dev/null
echo "== wlo study" >> $S
timeout 900 python tools/wlo_study_gpu.py --specs "none,proj@0-11,v@0-11,v+proj@0-5,v+proj@0-7,v+proj@0-9,v+proj@0-11" --cases vitl_672_full,vitl_896_full,vitl_1288_full,vitb_672_full,vitl_672_hostile_m > $OUT/wlo.json 2> $OUT/wlo.txt
cat $OUT/wlo.txt >> $S
cat $S
