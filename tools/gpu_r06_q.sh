#!/bin/bash
# Round-6 session Q: Op<DT>::pair_sum (the attention forms' row sums) as __builtin_amdgcn_fdot2 instead of inline assembly, so that the
# compiler's hazard recogniser sees the dot products.  build_ab/r06_final = the validated library of sources dbe963fd (asm), in-tree = builtins.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r06q}; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
S=$OUT/summary.txt
echo "== library: $(python -c 'from multi_hmr_amd import _lib; print(_lib.built_source_hash())')" > $S
echo "== pytest: every attention test, the x3 tests, two full-size goldens" >> $S
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_x3.py tests/test_gpu_parity_fullsize.py -q -p no:cacheprovider -k "attention or x3 or (vitl_896_full and f16) or (vitl_672_full and f16) or vits_672_full" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log >> $S
grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest.log | head -10 >> $S
for i in 1 2; do for V in build_ab/r06_final default; do
  echo "== kbench attention f16 + bf16, library $V (run $i)" >> $S
  timeout 300 python tools/run_with_lib.py $V tools/kbench.py --dtype f16 --only attn --variants 6 --iters 20 2>/dev/null >> $S
  timeout 300 python tools/run_with_lib.py $V tools/kbench.py --dtype bf16 --only attn --variants 6 --iters 20 2>/dev/null >> $S
done; done
echo "== headline A/B, 20 steps" >> $S
for i in 1 2 3; do for V in build_ab/r06_final default; do
  timeout 300 python tools/run_with_lib.py $V bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/head.json 2> $OUT/head.err
  python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/head.json"))
print("library $V run $i:", d["value"], d["ms_per_step"], d.get("source_hash"))
PY
done; done
cat $S
