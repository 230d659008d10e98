#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03p; mkdir -p $OUT; cd $R
REPS=1000 timeout 600 python tools/two_stream_check.py 24 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tail -5 | tee $OUT/two_stream_soak.txt
