#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03u; mkdir -p $OUT; cd $R
for b in gemm4w_v4; do echo "== $b" | tee -a $OUT/gemm4w_d.txt; timeout 120 ./tools/ubench/$b 2>&1 | tee -a $OUT/gemm4w_d.txt; done
