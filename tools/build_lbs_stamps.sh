#!/bin/bash
# Debug build for tools/lbs_timeline.py: lbs.hip with -DMHMR_LBS_STAMPS linked against the product's other objects
# (run `python -c "import __graft_entry__ as g; g.build()"` first).  Output: tools/dbg/libmhmr_stamps.so (git-ignored).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/multi_hmr_amd/csrc
mkdir -p $R/tools/dbg
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DMHMR_NO_SLP -DMHMR_LBS_STAMPS -DMHMR_SOURCE_HASH=\"stamps\" -c $C/lbs.hip -o /tmp/lbs_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/dbg/libmhmr_stamps.so /tmp/lbs_stamps.o $(ls $C/build/*.o | grep -v "/lbs.o")
echo built $R/tools/dbg/libmhmr_stamps.so
