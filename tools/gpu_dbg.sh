#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03j; mkdir -p $OUT; cd $R
timeout 300 python tools/gemm_timeline.py f16 2>&1 | grep -v amdgpu.ids > $OUT/gemm_timeline_b.txt
grep "==\|finish\|tile period" $OUT/gemm_timeline_b.txt | cut -c1-260
