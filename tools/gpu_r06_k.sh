#!/bin/bash
# Round-6 session K (review item 6): fabric reads, shader clock and time of the fc1 / Q|K shapes as the tile order changes -- MHMR_COLGROUP =
# default (four weight panels per XCD), 0 (plain order: every XCD walks all panels), 8,8 and 4,16 -- from rocprofv3 --pmc passes over
# tools/kbench.py (token-row map, f16); one FETCH_SIZE, one WRITE_SIZE and one SQ / GRBM pass per setting (tools/pmc_traffic.py).
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r06k}; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary.txt
SQSET="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
echo "kernel (kbench: <1,1> = fc1 shape with GELU, <1,0> = fc1 / Q|K shapes plain, <1,7> = Q|K, <1,2> = fc1 shape ReLU) | read GB | written GB | pipe busy | clock GHz | ms under the counters" > $S
for CG in default 0 8,8 4,16; do
  D=$OUT/cg_$(echo $CG | tr , _); mkdir -p $D
  if [ "$CG" = default ]; then unset MHMR_COLGROUP; else export MHMR_COLGROUP=$CG; fi
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d $D -o $c --output-format csv -- python $R/tools/kbench.py --dtype f16 --only gemm --rows map --iters 3 > /dev/null 2>&1
  done
  timeout 300 rocprofv3 --kernel-trace --pmc $SQSET -d $D -o SQ --output-format csv -- python $R/tools/kbench.py --dtype f16 --only gemm --rows map --iters 3 > /dev/null 2>&1
  cd $R
  python tools/pmc_traffic.py $D > $D/pmc.json 2> $D/err.txt
  echo "== MHMR_COLGROUP=$CG" >> $S
  python - >> $S <<PY
import json
d = json.load(open("$D/pmc.json"))
for k, v in d.items():
    if isinstance(v, dict) and k.startswith("gemm256") and ("<1, 1," in k or "<1, 0," in k or "<1, 7," in k or "<1, 2," in k):
        print("   %-58s %6.3f %6.3f  %s  %s  %s" % (k, v.get("read_bytes_per_launch", 0) / 1e9, v.get("write_bytes_per_launch", 0) / 1e9, v.get("mfma_busy_frac"), v.get("shader_clock_ghz"), v.get("avg_duration_ms_under_pmc")))
PY
  find $D -name "*_counter_collection.csv" -delete; find $D -name "*kernel_trace.csv" -delete
done
cat $S
