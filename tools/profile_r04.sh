#!/bin/bash
# Round-4 / round-5 profile recipe (tools/profile_r04.sh r05 ...) (run on the GPU box via gpurun): the kernel statistics come from the bench command ITSELF - back-to-back
# launches on a warm clock, >= 30 dispatches per kernel, the first 5 dropped (tools/warm_kernel_stats.py) - so that the committed
# average reproduces the line's kernel time; FETCH_SIZE / WRITE_SIZE in separate passes as MI355X_MICROARCH.md prescribes.
# usage: tools/profile_r04.sh <round tag, e.g. r04> [config ...]      configs: C2 C3 C4 C4_nd C2_N30 C5_soc C5_hmpc C2_tv C2_tv_fista C4_tv C4_tv_fista (default: all but the last two)
R=${GRAFT_REPO_ROOT:-/root/repo}; RD=${1:-r04}; shift
CFGS=${*:-C2 C3 C4 C4_nd C2_N30 C5_soc C5_hmpc C2_tv C2_tv_fista}
OUT=$R/gpurun_out/profiles_$RD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A TAG=([C2]=C2_mfma4 [C3]=C3_mfma4r [C4]=C4_mfma4r [C5_soc]=C5soc_bsp [C5_hmpc]=C5hmpc_fused [C2_tv]=C2tv_mfma4r [C2_tv_fista]=C2tvfista_mfma4r [C4_nd]=C4nd_mfma4r [C2_N30]=C2N30_mfma4r [C2_N30_gen]=C2N30gen_mfma4r [C4_tv]=C4tv_mfma4r [C4_tv_fista]=C4tvfista_mfma4r)  # <config>_<default variant>: the names bench.py looks for
for C in $CFGS; do
  T=${TAG[$C]:-$C}
  D=$R/gpurun_out/prof_${RD}_$C; mkdir -p $D
  if [ "$C" = "C2" ]; then ARGS="--steps 60 --warmup 5 --no-configs --no-cpu-baseline --no-pcie"
  else ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-pcie --configs $C --config-steps 36"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $R/bench.py $ARGS > $D/bench_trace.json 2> $D/trace.err
  python3 $R/tools/warm_kernel_stats.py $D/trace --drop 5 --min-calls 20 > $OUT/${RD}_${T}_kernel_stats.csv
  if [ "$C" = "C2" ]; then PARGS="--steps 4 --warmup 2 --no-configs --no-cpu-baseline --no-pcie"
  else PARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-pcie --configs $C --config-steps 4"; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/pmc_fetch -- python3 $R/bench.py $PARGS > $D/bench_fetch.json 2> $D/pmc_fetch.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/pmc_write -- python3 $R/bench.py $PARGS > $D/bench_write.json 2> $D/pmc_write.err
  python3 $R/tools/pmc_summary.py $D | cut -c1-260 | grep -v "at::native\|rocclr\|elementwise\|eng_\|fill" > $OUT/${RD}_${T}_pmc_summary.txt
  tail -1 $D/bench_trace.json | cut -c1-600 > $OUT/${RD}_${T}_bench_under_rocprof.json
  echo "== $C"; cat $OUT/${RD}_${T}_kernel_stats.csv | cut -c1-240; cat $OUT/${RD}_${T}_pmc_summary.txt | cut -c1-220
done
