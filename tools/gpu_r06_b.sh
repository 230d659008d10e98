#!/bin/bash
# Round-6 session B: (1) new kernels' tests: masked-width GEMM (ViT-S on the 256x256 kernel), four-wave pose kernel, split-k; the model tests
# that ride on the ViT-S path; (2) SMPL-X layer A/B (MHMR_LBS_POSE1); (3) cfg2 A/B (MHMR_VITS_256); (4) where a batch of one spends its time
# now (kernel trace, multiHMR_896_L / 672_L); (5) what MHMR_ANYORDER changes in the headline trace; (6) the low-half set: proj-only variants,
# and the bench's own parity leg under proj@0-11.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06b}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
S=$OUT/summary.txt
echo "== pytest kernels (masked / splitk / lbs)" > $S
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -k "masked or splitk or lbs or layernorm_fold or gemm_epilogues" > $OUT/pytest_k.log 2>&1; tail -3 $OUT/pytest_k.log >> $S
grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest_k.log | head -20 >> $S
echo "== pytest model / x3 / fullsize / demo / anny / graph" >> $S
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_x3.py tests/test_gpu_fullsize.py tests/test_demo_config1.py tests/test_anny_model.py tests/test_anny_hph.py tests/test_gpu_graph.py -q -p no:cacheprovider -x > $OUT/pytest_m.log 2>&1; tail -3 $OUT/pytest_m.log >> $S
grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest_m.log | head -20 >> $S
echo "== parity fullsize vits" >> $S
timeout 600 python -m pytest tests/test_gpu_parity_fullsize.py -q -p no:cacheprovider -s -k "vits" > $OUT/pytest_p.log 2>&1
grep -E "^\[parity|passed|failed|^FAILED|^ERROR" $OUT/pytest_p.log | cut -c1-330 >> $S
echo "== SMPL-X layer A/B (MHMR_LBS_POSE1: 1 = the one-wave pose kernel)" >> $S
for i in 1 2; do for V in 0 1; do for P in 160 20 1; do
  echo -n "POSE1=$V " >> $S; MHMR_LBS_POSE1=$V timeout 120 python tools/lbs_bench.py $P 2>/dev/null | tail -1 >> $S
done; done; done
echo "== cfg2 A/B (MHMR_VITS_256: 0 = ViT-S C-wide linears on the 128x128 kernel, no fold)" >> $S
for i in 1 2; do for V in 1 0; do
  echo -n "VITS_256=$V " >> $S; MHMR_VITS_256=$V timeout 300 python tools/split_probe.py cfg2 2>/dev/null | tail -1 >> $S
done; done
echo "== batch-1 kernel trace" >> $S
for M in multiHMR_896_L multiHMR_672_L; do
  rm -rf /tmp/tr_$M; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$M -- python $R/tools/latency_trace.py run $M 50 > $OUT/b1_$M.run 2>&1)
  echo "-- $M" >> $S; tail -1 $OUT/b1_$M.run >> $S; python tools/latency_trace.py parse /tmp/tr_$M 50 >> $S 2>&1
done
echo "== headline trace, MHMR_ANYORDER=0 / 1" >> $S
for V in 0 1; do
  rm -rf /tmp/tr_ao$V; (cd /tmp && MHMR_ANYORDER=$V timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_ao$V -- python $R/bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > $OUT/ao_$V.json 2> $OUT/ao_$V.err)
  echo "-- ANYORDER=$V" >> $S; python tools/trace_gaps.py /tmp/tr_ao$V 3 >> $S 2>&1
done
echo "== wlo study 2" >> $S
timeout 900 python tools/wlo_study_gpu.py --specs "proj@0-7,proj@0-11,proj@0-15,proj,v@0-3|proj@0-11" --cases vitl_672_full,vitl_896_full,vitl_1288_full,vitb_672_full,vitl_672_hostile_m > $OUT/wlo.json 2> $OUT/wlo.txt
grep -v amdgpu.ids $OUT/wlo.txt >> $S
echo "== bench parity leg under MHMR_WLO=proj@0-11 vs default" >> $S
for W in "proj@0-11" "v+proj@0-11"; do
  MHMR_WLO="$W" timeout 600 python bench.py --steps 10 --warmup 3 --no-extras > $OUT/bp.json 2> $OUT/bp.err
  python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/bp.json"))
print("WLO=$W:", d["value"], d["ms_per_step"], json.dumps(d.get("parity"))[:700])
PY
done
cat $S
