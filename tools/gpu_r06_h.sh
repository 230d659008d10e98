#!/bin/bash
# Round-6 session H (needs tools/ubench/resid_prefetch_experiment.patch applied to csrc/: the experiment measured SLOWER and its code was not kept):
# the residual tile prefetched under the last k pair (MHMR_RESID_PREFETCH=1, gemm256.hip RPF): kernel tests, the residual
# GEMMs alone (kbench), the headline A/B.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r06h}; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
S=$OUT/summary.txt
echo "== pytest with MHMR_RESID_PREFETCH=1" > $S
MHMR_RESID_PREFETCH=1 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_fullsize.py -q -p no:cacheprovider -k "gemm_epilogues or layernorm_fold or low_half or token_row_map or rows_beyond or (vitl_896_full and f16) or (vitl_672_full and f16)" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log >> $S
grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest.log | head -10 >> $S
for V in 0 1 0 1; do
  echo "== kbench residual GEMMs, MHMR_RESID_PREFETCH=$V" >> $S
  MHMR_RESID_PREFETCH=$V timeout 300 python tools/kbench.py --dtype f16 --only gemm --rows map --iters 20 2>/dev/null | grep -E "proj|fc2" >> $S
done
echo "== headline A/B, 20 steps" >> $S
for i in 1 2 3; do for V in 0 1; do
  MHMR_RESID_PREFETCH=$V timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/head.json 2> $OUT/head.err
  python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/head.json"))
print("RESID_PREFETCH=$V run $i:", d["value"], d["ms_per_step"])
PY
done; done
cat $S
