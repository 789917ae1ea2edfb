#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace/stats + separate PMC passes of bench.py.
# usage: tools/profile_bench.sh <variant> <steps> <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
V=${1:-mfma}; K=${2:-10}; TAG=${3:-r01_$V}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --variant $V --steps $K --warmup 2 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --variant $V --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --variant $V --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
find $OUT -name "*.csv" | head -30
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f; done
for f in $(find $OUT/pmc_fetch $OUT/pmc_write -name "*counter_collection.csv"); do echo "== $f"; head -3 $f; python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    agg[(r.get("Kernel_Name", "?")[:60], r.get("Counter_Name"))].append(float(r.get("Counter_Value", 0)))
for k, v in agg.items():
    print(k, "n=%d mean=%.4g max=%.4g" % (len(v), sum(v) / len(v), max(v)))
PY
done
