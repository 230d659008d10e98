#!/bin/bash
# Round-3 A/B session on the GPU box (one gpurun call): kernel tests, kernel micro-benchmarks, whole-forward A/B of the round's switches,
# the low-half weight pass parity table.  Everything lands in gpurun_out/$TAG.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03a}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== kernels: gemm / cls" > $OUT/summary.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "gemm or cls or gelu" -p no:cacheprovider > $OUT/pytest_gemm.log 2>&1; tail -3 $OUT/pytest_gemm.log >> $OUT/summary.txt
echo "== kernels: attention" >> $OUT/summary.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "attention" -p no:cacheprovider > $OUT/pytest_attn.log 2>&1; tail -3 $OUT/pytest_attn.log >> $OUT/summary.txt
echo "== everything else" >> $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q -k "not (gemm or cls or gelu or attention)" -p no:cacheprovider > $OUT/pytest_rest.log 2>&1; tail -3 $OUT/pytest_rest.log >> $OUT/summary.txt
cp gpurun_out/parity_fullsize.json $OUT/ 2>/dev/null
echo "== kbench" >> $OUT/summary.txt
timeout 300 python tools/kbench.py --dtype f16 --only gemm --iters 10 > $OUT/kb_gemm.txt 2>&1
MHMR_COLGROUP=0 timeout 300 python tools/kbench.py --dtype f16 --only gemm --iters 10 --rows map > $OUT/kb_gemm_nocolgroup.txt 2>&1
timeout 300 python tools/kbench.py --dtype f16 --only attn --variants 0,4,5 --iters 10 > $OUT/kb_attn.txt 2>&1
timeout 300 python tools/kbench.py --dtype bf16 --only attn --variants 0,4 --iters 10 > $OUT/kb_attn_bf16.txt 2>&1
cat $OUT/kb_gemm.txt $OUT/kb_gemm_nocolgroup.txt $OUT/kb_attn.txt $OUT/kb_attn_bf16.txt >> $OUT/summary.txt
echo "== bench A/B (10 steps, f16)" >> $OUT/summary.txt
i=0
for cfg in "X=1" "MHMR_ROWMAP=0" "MHMR_WLO=" "MHMR_ATTN_VARIANT=4" "MHMR_ATTN_VARIANT=5" "MHMR_COLGROUP=0"; do
  i=$((i+1))
  env $cfg timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_$i.json 2> $OUT/bench_$i.err
  echo "$cfg: $(python -c "import json,sys; d=json.load(open('$OUT/bench_$i.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_attention']['achieved'])" 2>&1 | tail -1)" >> $OUT/summary.txt
done
echo "== low-half weight pass parity table" >> $OUT/summary.txt
timeout 900 python tools/wlo_study_gpu.py > $OUT/wlo_study.json 2> $OUT/wlo_study.txt
cat $OUT/wlo_study.txt >> $OUT/summary.txt
cat $OUT/summary.txt
