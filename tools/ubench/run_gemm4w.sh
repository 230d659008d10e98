#!/bin/bash
# builds (if missing) and runs the forms of tools/ubench/gemm4w.hip quoted in its header; output -> gpurun_out/r03u/
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03u; mkdir -p $OUT; cd $R/tools/ubench
build() { [ -x $1 ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DMHMR_NO_SLP $2 -I ../../multi_hmr_amd/csrc -o $1 gemm4w.hip; }
build gemm4w_v3 "-DASM_MFMA -DV3"
build gemm4w_v4 "-DASM_MFMA -DV3 -DEARLY_DMA"
for b in gemm4w_v3 gemm4w_v4; do echo "== $b" | tee -a $OUT/gemm4w_run.txt; timeout 120 ./$b 2>&1 | tee -a $OUT/gemm4w_run.txt; done
