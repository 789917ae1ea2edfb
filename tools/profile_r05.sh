#!/bin/bash
# Round-5 SQ-counter passes of the streamed-block kernels (run on the GPU box via gpurun):
#   tools/profile_r05.sh [tag ...]        tags: C3 C4 C4nd C2N30 (default: all)
# Five rocprofv3 --pmc passes per kernel (tools/pmc_cmd.sh: counters in groups the hardware can collect together, kernel trace only -
# never together with a runtime trace), each over a cold and RUN_ONE_REPS warm launches of tools/run_one.py; the summary is the mean per
# launch.  Output: gpurun_out/profiles_r05/r05_<tag>_mfma4r_sq_counters.txt (copy into profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_r05; mkdir -p $OUT
export RUN_ONE_REPS=${RUN_ONE_REPS:-4}
declare -A CFG=([C3]="C3 262144" [C4]="C4 131072" [C4nd]="C4_nd 131072" [C2N30]="C2_lax_N30 65536" [C2N30gen]="C2_lax_N30_gen 65536" [C2tv]="C2_lax:tv 65536" [C2tvfista]="C2_lax_FISTA:tv 65536" [C5soc]="C5_soc 65536")
declare -A KRN=([C3]=fista_r_kernel [C4]=eadmm_r_kernel [C4nd]=eadmm_r_kernel [C2N30]=admm_r_kernel [C2N30gen]=admm_r_kernel [C2tv]=admm_tvr_kernel [C2tvfista]=fista_tvr_kernel [C5soc]=soc_bsp)
for T in ${*:-C3 C4 C4nd C2N30}; do
  V=mfma4r; [ "$T" = "C5soc" ] && V=bsp
  bash $R/tools/pmc_cmd.sh r05_$T ${KRN[$T]} ${CFG[$T]} $V > $OUT/r05_${T}_${V}${SUFFIX}_sq_counters.txt 2> $OUT/r05_${T}.err
  echo "== $T"; cat $OUT/r05_${T}_${V}${SUFFIX}_sq_counters.txt
done
