#!/bin/bash
# Round profile: kernel-trace stats + FETCH/WRITE passes (+ SQ counters) for the BASELINE configurations' default variants.
# usage (GPU box): tools/profile_round.sh r02 [tag ...]   -> gpurun_out/prof_<round>_<tag>/, summaries in gpurun_out/profiles_<round>/
# (with tags: only those configurations)
R=${GRAFT_REPO_ROOT:-/root/repo}; RD=${1:-r03}; shift; ONLY=" $* "
OUT=$R/gpurun_out/profiles_$RD; mkdir -p $OUT
prof() {  # tag config B variant kernel-substring
  if [ "$ONLY" != "  " ] && [[ "$ONLY" != *" $1 "* ]]; then return; fi
  tools/profile_config.sh ${RD}_$1 $2 $3 $4 > $OUT/${RD}_$1_profile.log 2>&1
  python3 tools/pmc_summary.py $R/gpurun_out/prof_${RD}_$1 | cut -c1-260 | grep -v "at::native\|rocclr" > $OUT/${RD}_$1_pmc_summary.txt
  for f in $(find $R/gpurun_out/prof_${RD}_$1/trace -name "*kernel_stats.csv"); do cp $f $OUT/${RD}_$1_kernel_stats.csv; done
  tools/pmc_cmd.sh ${RD}_$1 $5 $2 $3 $4 > $OUT/${RD}_$1_sq_counters.txt 2>&1
  echo "== $1"; head -3 $OUT/${RD}_$1_kernel_stats.csv | cut -c1-200; cat $OUT/${RD}_$1_pmc_summary.txt | cut -c1-200
}
prof C2_mfma4 C2 65536 mfma4 admm_mfma4
prof C3_mfma4r C3 262144 mfma4r fista_r
prof C4_mfma4r C4 131072 mfma4r eadmm_r
prof C4_mfma4g C4 131072 mfma4g eadmm_g
prof C5soc_bsp C5_soc 65536 bsp bsp
prof C5hmpc_fused C5_HMPC_SADMM 65536 fused hmpc_fused
prof C5hmpc_nosplit_fused C5_HMPC_SADMM_nosplit 65536 fused hmpc_fused
prof C2cs_fused C2_cs 65536 fused cs_fused
