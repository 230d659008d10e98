#!/bin/bash
# Runs on the GPU box (gpurun): the round's rocprofv3 evidence for the CURRENT build of libmhmr.so.
#   tools/collect_profiles.sh rNN   ->  gpurun_out/rNN/{kernel_stats.txt, pmc.json, ...}; copy the summaries into profiles/.
# Counters are collected in their own runs (one --pmc set per run, --kernel-trace only beside them).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT/pmc/lbs $OUT/trace
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --only-headline-kernels"
LBS="python $R/tools/lbs_bench.py 160"
# (a) the headline forward alone: per-kernel averages comparable with the bench line's avg_launch_ms; (b) the whole default command
if [ "$2" != "pmc" ]; then
rocprofv3 --kernel-trace --stats -d $OUT/trace -o headline --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --only-headline-kernels > $OUT/trace/headline.json 2> $OUT/trace/err_headline.txt
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/trace/bench.json 2> $OUT/trace/err.txt
fi
# counter passes (every kernel of the forward runs on the caller's stream: one kernel at a time)
SQSET="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc -o $c --output-format csv -- $BENCH > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc/lbs -o $c --output-format csv -- $LBS > /dev/null 2>&1
done
rocprofv3 --kernel-trace --pmc $SQSET -d $OUT/pmc -o SQ --output-format csv -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc $SQSET -d $OUT/pmc/lbs -o SQ --output-format csv -- $LBS > /dev/null 2>&1
cd $R
python tools/pmc_traffic.py $OUT/pmc > $OUT/pmc.json 2> $OUT/pmc_err.txt
# keep the merge small: the raw per-dispatch csv files are large
find $OUT -name "*_counter_collection.csv" -size +8M -delete
find $OUT -name "*kernel_trace.csv" -delete
ls -la $OUT $OUT/pmc | head -40
head -c 1500 $OUT/pmc.json
