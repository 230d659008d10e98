#!/bin/bash
# third session: LayerNorm fold -- kernel tests, model goldens, full-size parity, A/B against MHMR_LNFOLD=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03c}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "gemm or cls or fold" -p no:cacheprovider > $OUT/pytest_gemm.log 2>&1; tail -3 $OUT/pytest_gemm.log > $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_fullsize.py tests/test_anny_model.py tests/test_gpu_fullsize.py -q -p no:cacheprovider > $OUT/pytest_model.log 2>&1; tail -3 $OUT/pytest_model.log >> $OUT/summary.txt
grep -E "^(FAILED|ERROR)" $OUT/pytest_gemm.log $OUT/pytest_model.log >> $OUT/summary.txt
cp gpurun_out/parity_fullsize.json $OUT/ 2>/dev/null
python - >> $OUT/summary.txt <<'PY'
import json
d = json.load(open("gpurun_out/parity_fullsize.json"))
for k, v in sorted(d.items()):
    if k.endswith("/f16"):
        print(k, "backbone %.2e" % v["backbone_rel_l2"], " ".join(f"{a}={b:.2e}" for a, b in v["rel_l2"].items() if a in ("scores", "offset", "shape", "expression", "rotmat", "transl", "v3d")))
PY
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace -o headline --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --only-headline-kernels > $OUT/trace_headline.json 2> $OUT/trace_err.txt)
find $OUT/trace -name "*kernel_trace.csv" -delete
head -14 $(find $OUT/trace -name "*kernel_stats.csv" | head -1) | cut -c1-160 >> $OUT/summary.txt
i=0
for cfg in "X=1" "MHMR_LNFOLD=0" "X=2" "MHMR_LNFOLD=0"; do
  i=$((i+1))
  env $cfg timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_$i.json 2> $OUT/bench_$i.err
  echo "$cfg: $(python -c "import json,sys; d=json.load(open('$OUT/bench_$i.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_attention']['achieved'])" 2>&1 | tail -1)" >> $OUT/summary.txt
done
cat $OUT/summary.txt
