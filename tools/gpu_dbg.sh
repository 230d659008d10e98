#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03l; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "lbs" -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/pytest_lbs.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $OUT/pytest_lbs.txt
for P in 160 20 1 300; do timeout 120 python tools/lbs_bench.py $P 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $OUT/lbs_bench.txt; done
timeout 600 python -m pytest tests/test_gpu_model.py -q -k "golden" -p no:cacheprovider 2>&1 | tail -3 | tee -a $OUT/pytest_lbs.txt
