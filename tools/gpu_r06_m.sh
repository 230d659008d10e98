#!/bin/bash
# Round-6 session M: the GELU of the fc1 epilogue as max(x, 0) - |x| exp2(P5(|x|)) (8 VALU, one transcendental; rounds 4-5: 11 with two).
# build_ab/r06_base/libmhmr.so = the library of sources c34442c0 (the round's validated build), the in-tree library = the new form.
# Also: what the lone class-token query block (T = 64 n + 1: a 33rd workgroup per (image, head) with ONE real query) costs the attention launch.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r06m}; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
S=$OUT/summary.txt
echo "== library: $(python -c 'from multi_hmr_amd import _lib; print(_lib.built_source_hash())')" > $S
echo "== pytest (new library): GELU sweep, epilogues, fold, class rows, two full-size goldens" >> $S
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_fullsize.py -q -p no:cacheprovider -k "gelu or gemm_epilogues or layernorm_fold or cls or token_row_map or (vitl_896_full and f16) or (vitl_672_full and f16) or vits_672_full" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log >> $S
grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest.log | head -10 >> $S
for i in 1 2; do for V in build_ab/r06_base default; do
  echo "== kbench fc1 shapes, library $V (run $i)" >> $S
  timeout 300 python tools/run_with_lib.py $V tools/kbench.py --dtype f16 --only gemm --rows map --iters 20 2>/dev/null | grep -E "fc1" >> $S
done; done
echo "== headline A/B, 20 steps" >> $S
for i in 1 2 3; do for V in build_ab/r06_base default; do
  timeout 300 python tools/run_with_lib.py $V bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/head.json 2> $OUT/head.err
  python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/head.json"))
print("library $V run $i:", d["value"], d["ms_per_step"], d.get("source_hash"))
PY
done; done
echo "== attention: T = 4097 (shipped shape) against T = 4096 (no lone query block, no lone key), same buffers" >> $S
for i in 1 2; do for T in 4097 4096; do
  timeout 300 python tools/kbench.py --dtype f16 --only attn --variants 6 --iters 20 --tokens $T 2>/dev/null >> $S
done; done
echo "== cfg2 / cfg3 / cfg5 through bench.py extras are not run here" >> $S
cat $S
