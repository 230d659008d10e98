#!/bin/bash
# Round-6 session S: the operand copies of gemm256_kernel addressed with 32-bit BYTE offsets (global_load_lds with an SGPR base: the element offsets
# of rounds 2-6 were shifted after their zero-extension, which cost a 64-bit vector shift-and-add per copy: 16 per pair of k tiles).
# build_ab/r06_attn2 = session R's in-tree library (attention changes only), in-tree = that + this.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r06s}; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
S=$OUT/summary.txt
echo "== library: $(python -c 'from multi_hmr_amd import _lib; print(_lib.built_source_hash())')" > $S
echo "== pytest (in-tree library): kernel tests except attention, x3, three full-size goldens, batch invariance" >> $S
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_x3.py tests/test_gpu_parity_fullsize.py tests/test_gpu_fullsize.py -q -p no:cacheprovider -k "not attention and (not fullsize or f16 or invariance)" --deselect tests/test_gpu_parity_fullsize.py > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log >> $S
grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest.log | head -10 >> $S
timeout 900 python -m pytest tests/test_gpu_parity_fullsize.py -q -p no:cacheprovider -k "(vitl_896_full and f16) or (vitl_672_full and f16) or vits_672_full or vitb_672_full" > $OUT/pytest2.log 2>&1; tail -2 $OUT/pytest2.log >> $S
grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest2.log | head -10 >> $S
for i in 1 2; do for V in build_ab/r06_attn2 default; do
  echo "== kbench GEMMs under the token-row map, library $V (run $i)" >> $S
  timeout 300 python tools/run_with_lib.py $V tools/kbench.py --dtype f16 --only gemm --rows map --iters 20 2>/dev/null | grep -v "no GELU\|ReLU\|plain" >> $S
done; done
echo "== headline A/B, 20 steps" >> $S
for i in 1 2 3; do for V in build_ab/r06_attn2 default; do
  timeout 300 python tools/run_with_lib.py $V bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/head.json 2> $OUT/head.err
  python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/head.json"))
print("library $V run $i:", d["value"], d["ms_per_step"], d.get("source_hash"))
PY
done; done
echo "== batch-1 latency" >> $S
for V in build_ab/r06_attn2 default; do
  timeout 300 python tools/run_with_lib.py $V bench.py --only-latency > $OUT/lat.json 2> $OUT/lat.err
  python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/lat.json"))
print("library $V:", "  ".join("%s %.3f (min %.3f)" % (k, v["ms"], v["min_ms"]) for k, v in d.items() if isinstance(v, dict)))
PY
done
cat $S
