#!/bin/bash
# wave-time split counters of the headline forward's kernels and of the SMPL-X layer (one --pmc pass each), round 3
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03w; mkdir -p $OUT/fw $OUT/lbs
cd /tmp && export TMPDIR=/tmp
SET="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"
rocprofv3 --kernel-trace --pmc $SET -d $OUT/fw -o W --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --only-headline-kernels > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc $SET -d $OUT/lbs -o W --output-format csv -- python $R/tools/lbs_bench.py 160 > /dev/null 2>&1
cd $R
python tools/pmc_wait.py $(find $OUT/fw -name "W_counter_collection.csv" | head -1) > $OUT/forward_wait.json
python tools/pmc_wait.py $(find $OUT/lbs -name "W_counter_collection.csv" | head -1) > $OUT/lbs_wait.json
find $OUT -name "*_counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/forward_wait.json | head -80
