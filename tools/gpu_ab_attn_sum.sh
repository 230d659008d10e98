#!/bin/bash
# row sums of the attention softmax by v_dot2 on the packed P values (shipped) against fp32 adds (build_ab/sum32): tests, kernel alone, forward
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03q; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "attention" -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/pytest_attn.txt
for i in 1 2; do for lib in default build_ab/sum32; do
  echo "== $lib" | tee -a $OUT/ab.txt
  python tools/run_with_lib.py $lib tools/kbench.py --dtype f16 --only attn --variants 0 --iters 10 2>/dev/null | grep "attention" | tail -1 | tee -a $OUT/ab.txt
  python tools/run_with_lib.py $lib bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_attention']['achieved'])" | tee -a $OUT/ab.txt
done; done
python tools/run_with_lib.py default tools/kbench.py --dtype bf16 --only attn --variants 0 --iters 10 2>/dev/null | grep attention | tail -1 | tee -a $OUT/ab.txt
python tools/run_with_lib.py build_ab/sum32 tools/kbench.py --dtype bf16 --only attn --variants 0 --iters 10 2>/dev/null | grep attention | tail -1 | tee -a $OUT/ab.txt
