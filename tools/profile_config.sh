#!/bin/bash
# rocprofv3 kernel-trace/stats + separate FETCH_SIZE / WRITE_SIZE passes of one configuration (run on the GPU box).
# usage: tools/profile_config.sh <tag> <config> <B> <variant>    -> gpurun_out/prof_<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/run_one.py "$@" > $OUT/run_trace.txt 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/run_one.py "$@" > $OUT/run_fetch.txt 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/tools/run_one.py "$@" > $OUT/run_write.txt 2> $OUT/pmc_write.err
cat $OUT/run_trace.txt
python3 $R/tools/pmc_summary.py $OUT | cut -c1-200 | grep -v "at::native\|rocclr"
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do head -4 $f | cut -c1-220; done
