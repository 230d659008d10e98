#!/bin/bash
# FIFTH final validation (sessions R-T: copies with an SGPR base in all three MFMA kernels, the half-work ghost) -- FOURTH (pair_sum as a builtin) -- THIRD final validation of round 6 (after sessions M-P: the GELU as exp2 of a polynomial, the lone key by DMA, the class-query role: library changes, so the whole evidence set again;
# the SMPL-X A/B, the pose timeline and config 2's trace of the first final run stand: those kernels did not change).
# Final validation of round 6 on the GPU box, all on ONE build (the library says which sources it was made from): the whole -m gpu suite,
# smoke(), the default bench command, the bench under an RCCL group of one, the forward's bit-reproducibility, the SMPL-X layer A/B + the
# pose kernel's timeline, config 2's kernel trace, then the round's rocprofv3 evidence (tools/collect_profiles.sh: kernel traces + PMC).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06x}; export TAG
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
S=$OUT/summary.txt
rm -f gpurun_out/parity_fullsize.json
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log > $S
grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest_gpu.log | head -20 >> $S
cp gpurun_out/parity_fullsize.json $OUT/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $S 2>&1
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python - >> $S <<'PY'
import json, os
try:
    d = json.load(open("gpurun_out/%s/bench.json" % os.environ.get("TAG", "r06x")))
    for k in ("value", "ms_per_step", "mfma_utilisation_whole_forward", "source_hash", "precision_resolved", "backbone_image_blocks", "roofline", "roofline_attention", "lbs", "ms_per_person_lbs", "inference_mode", "parity", "cpu_baseline", "other_precision", "configs", "latency_b1"):
        print(k, json.dumps(d.get(k))[:1500])
except Exception as e:
    print("bench parse failed", e)
PY
tail -3 $OUT/bench.err >> $S
echo "== bench under torch.distributed.run, RCCL group of one" >> $S
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400 >> $S
echo "== bit-reproducibility of the headline forward" >> $S
timeout 400 python tools/forward_determinism.py 30 2>&1 | grep -v amdgpu.ids >> $S
echo "== two-stream soak" >> $S
MHMR_SOAK_BATCH=8 MHMR_SPLIT=2 REPS=200 timeout 500 python tools/two_stream_check.py 24 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tail -3 >> $S
echo "== x3 cost" >> $S
timeout 300 python tools/x3_bench.py 8 >> $S 2>/dev/null
echo "== profiles" >> $S
timeout 1500 bash tools/collect_profiles.sh r06 > $OUT/collect.log 2>&1
tail -12 $OUT/collect.log >> $S
cat $S
