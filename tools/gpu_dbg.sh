#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03n; mkdir -p $OUT; cd $R
for P in 1 20 160; do MHMR_LIB=tools/dbg/libmhmr_stamps.so timeout 120 python tools/lbs_timeline.py $P 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lbs_timeline.txt; done
