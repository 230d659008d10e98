#!/bin/bash
# Round-6 session T: the copies of the 128x128 kernel (gemm.hip: ViT-S, the patch embedding, the heads) with an SGPR base + 32-bit lane byte offset.
# build_ab/r06_gemm2 = session S's in-tree library, in-tree = that + this.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r06t}; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
S=$OUT/summary.txt
echo "== library: $(python -c 'from multi_hmr_amd import _lib; print(_lib.built_source_hash())')" > $S
echo "== pytest (in-tree library): kernel tests except attention, model tests, the ViT-S / ViT-B goldens" >> $S
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -p no:cacheprovider -k "not attention" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log >> $S
grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest.log | head -10 >> $S
timeout 900 python -m pytest tests/test_gpu_parity_fullsize.py -q -p no:cacheprovider -k "vits_672_full or (vitb_672_full and f16)" > $OUT/pytest2.log 2>&1; tail -2 $OUT/pytest2.log >> $S
grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest2.log | head -10 >> $S
echo "== config 2 (ViT-S 672^2 x 16) A/B, 20 steps" >> $S
for i in 1 2 3; do for V in build_ab/r06_gemm2 default; do
  timeout 300 python tools/run_with_lib.py $V bench.py --backbone dinov2_vits14 --img-size 672 --batch 16 --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/head.json 2> $OUT/head.err
  python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/head.json"))
print("cfg2 library $V run $i:", d["value"], d["ms_per_step"])
PY
done; done
echo "== headline, 20 steps" >> $S
for V in build_ab/r06_gemm2 default; do
  timeout 300 python tools/run_with_lib.py $V bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/head.json 2> $OUT/head.err
  python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/head.json"))
print("library $V:", d["value"], d["ms_per_step"], d.get("source_hash"))
PY
done
cat $S
