#!/bin/bash
# Round-5 session E: WHY the fp8 low-half build is slower (session D: 135.8 vs 133.6 ms): rocprofv3 kernel traces of the headline forward
# with MHMR_LO8=1 and =0 (per-kernel averages side by side); the SLP reproducer with offending tuples.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05e}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 $R/tools/ubench/slp_repro 100 > $OUT/slp_repro.txt 2>&1
for V in 1 0; do
  MHMR_LO8=$V timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace$V -o headline --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --only-headline-kernels > $OUT/headline_$V.json 2> $OUT/err_$V.txt
  find $OUT/trace$V -name "*kernel_trace.csv" -delete
  cp $(find $OUT/trace$V -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_lo8_$V.csv 2>/dev/null
done
cd $R
python - > $OUT/summary.txt 2>&1 <<PY
import csv
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        n = r["Name"]
        for a, b in (("void (anonymous namespace)::", ""), ("(anonymous namespace)::", ""), ("(GemmArgs)", "")):
            n = n.replace(a, b)
        d[n[:60]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6)
    return d
a, b = load("$OUT/kernel_stats_lo8_1.csv"), load("$OUT/kernel_stats_lo8_0.csv")
print("%-62s %6s %10s %10s | %6s %10s %10s" % ("kernel", "calls", "avg us", "total ms", "calls", "avg us", "total ms"))
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0, 0))[2] + b.get(k, (0, 0, 0))[2]))[:28]:
    x, y = a.get(k, (0, 0, 0)), b.get(k, (0, 0, 0))
    print("%-62s %6d %10.1f %10.2f | %6d %10.1f %10.2f" % (k, x[0], x[1], x[2], y[0], y[1], y[2]))
print("LO8=1 total %.1f ms, LO8=0 total %.1f ms" % (sum(v[2] for v in a.values()), sum(v[2] for v in b.values())))
PY
echo "== slp repro" >> $OUT/summary.txt; head -30 $OUT/slp_repro.txt | cut -c1-420 >> $OUT/summary.txt
cat $OUT/summary.txt
