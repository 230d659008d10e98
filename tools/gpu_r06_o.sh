#!/bin/bash
# Round-6 session O: attention variant 10 = variant 6 with the class query of T = 128 n + 1 on workgroups of its own (attn_cls_role: one wave per
# (image, head), vector ALU, exact online softmax) instead of a 128-query "ghost" workgroup.  One library, MHMR_ATTN_VARIANT = 6 | 10 at run time.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r06o}; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
S=$OUT/summary.txt
echo "== library: $(python -c 'from multi_hmr_amd import _lib; print(_lib.built_source_hash())')" > $S
echo "== pytest: every attention test" >> $S
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -k "attention" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log >> $S
grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest.log | head -10 >> $S
echo "== pytest with MHMR_ATTN_VARIANT=10: full-size goldens (f16), batch invariance, model tests" >> $S
MHMR_ATTN_VARIANT=10 timeout 1500 python -m pytest tests/test_gpu_parity_fullsize.py tests/test_gpu_fullsize.py -q -p no:cacheprovider -k "(vitl_896_full and f16) or (vitl_672_full and f16) or vits_672_full or invariance" > $OUT/pytest10.log 2>&1; tail -3 $OUT/pytest10.log >> $S
grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest10.log | head -10 >> $S
echo "== kbench attention variants 6 / 10, 896^2 x 32 then 672^2 x 32" >> $S
timeout 300 python tools/kbench.py --dtype f16 --only attn --variants 6,10 --iters 20 2>/dev/null >> $S
timeout 300 python tools/kbench.py --dtype f16 --only attn --variants 6,10 --iters 20 --img 672 2>/dev/null >> $S
echo "== headline A/B, 20 steps" >> $S
for i in 1 2 3; do for V in 6 10; do
  MHMR_ATTN_VARIANT=$V timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/head.json 2> $OUT/head.err
  python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/head.json"))
print("ATTN_VARIANT=$V run $i:", d["value"], d["ms_per_step"], d.get("source_hash"))
PY
done; done
echo "== batch-1 latency" >> $S
for i in 1 2; do for V in 6 10; do
  MHMR_ATTN_VARIANT=$V timeout 300 python bench.py --only-latency > $OUT/lat.json 2> $OUT/lat.err
  python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/lat.json"))
print("ATTN_VARIANT=$V run $i:", "  ".join("%s %.3f (min %.3f)" % (k, v["ms"], v["min_ms"]) for k, v in d.items() if isinstance(v, dict)))
PY
done; done
echo "== config 3 (672^2 ViT-L x 32) A/B, 20 steps" >> $S
for i in 1 2; do for V in 6 10; do
  MHMR_ATTN_VARIANT=$V timeout 300 python bench.py --img-size 672 --batch 32 --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/head.json 2> $OUT/head.err
  python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/head.json"))
print("cfg3 ATTN_VARIANT=$V run $i:", d["value"], d["ms_per_step"])
PY
done; done
cat $S
