#!/bin/bash
# Round-6 session R: (1) the K / V^T copies of attn16_kernel addressed as a scalar 64-bit base + a 32-bit lane offset (global_load_lds with an
# SGPR base: no 64-bit vector add per copy, 122 instead of 128 registers); (2) on top, the last workgroup of T = 128 n + 1 ... 128 n + 16 with ONE
# 16-query block per wave (NQB = 1: half the per-tile work of the "ghost").  build_ab/r06_pairsum = the validated library of sources ba23b17f,
# build_ab/r06_saddr = (1), in-tree = (1) + (2).
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r06r}; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
S=$OUT/summary.txt
echo "== library: $(python -c 'from multi_hmr_amd import _lib; print(_lib.built_source_hash())')" > $S
echo "== pytest (in-tree library): every attention test, two full-size goldens, batch invariance" >> $S
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_fullsize.py tests/test_gpu_fullsize.py -q -p no:cacheprovider -k "attention or (vitl_896_full and f16) or (vitl_672_full and f16) or vits_672_full or invariance" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log >> $S
grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest.log | head -10 >> $S
for i in 1 2; do for V in build_ab/r06_pairsum build_ab/r06_saddr default; do
  echo "== kbench attention 896^2 / 672^2, library $V (run $i)" >> $S
  timeout 300 python tools/run_with_lib.py $V tools/kbench.py --dtype f16 --only attn --variants 6 --iters 20 2>/dev/null >> $S
  timeout 300 python tools/run_with_lib.py $V tools/kbench.py --dtype f16 --only attn --variants 6 --iters 20 --img 672 2>/dev/null >> $S
done; done
echo "== headline A/B, 20 steps" >> $S
for i in 1 2 3; do for V in build_ab/r06_pairsum build_ab/r06_saddr default; do
  timeout 300 python tools/run_with_lib.py $V bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/head.json 2> $OUT/head.err
  python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/head.json"))
print("library $V run $i:", d["value"], d["ms_per_step"], d.get("source_hash"))
PY
done; done
echo "== config 3 A/B, 20 steps" >> $S
for i in 1 2; do for V in build_ab/r06_pairsum build_ab/r06_saddr default; do
  timeout 300 python tools/run_with_lib.py $V bench.py --img-size 672 --batch 32 --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/head.json 2> $OUT/head.err
  python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/head.json"))
print("cfg3 library $V run $i:", d["value"], d["ms_per_step"])
PY
done; done
cat $S
