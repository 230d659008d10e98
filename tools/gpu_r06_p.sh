#!/bin/bash
# Round-6 session P: does a batch of one want another attention form?  One 896^2 image is 528 workgroups on 1 024 slots: every workgroup pays the
# full copy latency per key tile (96 us per launch = 1.5 us per tile).  Forms 7 / 8 / 9 (copies in front of the score MFMAs, a three-slot ring,
# both) were measured at the headline only (slower there).  latency_b1 per MHMR_ATTN_VARIANT, interleaved twice.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r06p}; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
S=$OUT/summary.txt
echo "== library: $(python -c 'from multi_hmr_amd import _lib; print(_lib.built_source_hash())')" > $S
for i in 1 2; do for V in 6 7 8 9 10; do
  MHMR_ATTN_VARIANT=$V timeout 300 python bench.py --only-latency > $OUT/lat.json 2> $OUT/lat.err
  python - >> $S 2>&1 <<PY
import json
d = json.load(open("$OUT/lat.json"))
print("ATTN_VARIANT=$V run $i:", "  ".join("%s %.3f (min %.3f)" % (k, v["ms"], v["min_ms"]) for k, v in d.items() if isinstance(v, dict)))
PY
done; done
cat $S
