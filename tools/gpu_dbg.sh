#!/bin/bash
# which instruction pattern of the SLP build of vit_cls.hip misbehaves: the same device assembly re-assembled unchanged (cls_slp),
# with a wait state in front of every op_sel:[0,1,0] v_pk_fma_f32 (cls_varA), with wait states between the (mean, rstd) load and its use (cls_varB)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03g; mkdir -p $OUT; cd $R
{ for lib in build_ab/cls_slp build_ab/cls_varA build_ab/cls_varB build_ab/cls_slp; do
    echo "=== lib '$lib'"
    MHMR_LIBDIR=$lib REPS=400 python tools/two_stream_check.py 1 | cut -c1-200 | tail -3
    MHMR_LIBDIR=$lib REPS=100 python tools/two_stream_check.py 4 | cut -c1-160 | tail -2
  done; } > $OUT/dbg11.txt 2>&1
grep -v amdgpu.ids $OUT/dbg11.txt | grep "===\|done"
