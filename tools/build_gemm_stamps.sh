#!/bin/bash
# Debug build for tools/gemm_timeline.py: gemm256.hip with -DMHMR_GEMM_STAMPS linked against the product's other objects
# (run `python -c "import __graft_entry__ as g; g.build()"` first).  Output: tools/dbg/libmhmr_gemm_stamps.so (git-ignored).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/multi_hmr_amd/csrc
mkdir -p $R/tools/dbg
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DMHMR_NO_SLP -DMHMR_GEMM_STAMPS -c $C/gemm256.hip -o /tmp/gemm256_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/dbg/libmhmr_gemm_stamps.so /tmp/gemm256_stamps.o $(ls $C/build/*.o | grep -v "/gemm256.o")
echo built $R/tools/dbg/libmhmr_gemm_stamps.so
