#!/bin/bash
# Round-5 timing experiments on the streamed-block kernels (GPU box; WRONG RESULTS by construction, durations only): what the stage-end
# synchronisation of eadmm_r / fista_r costs, one piece at a time.  Each line re-specialises the kernel with hiprtc (-D switch through
# SPCIES_ER_RTC_FLAGS / SPCIES_FR_RTC_FLAGS) and times RUN_ONE_REPS warm launches (tools/run_one.py, host timer around the launch).
R=${GRAFT_REPO_ROOT:-/root/repo}
export RUN_ONE_REPS=${RUN_ONE_REPS:-5}
run() { echo -n "$1 | "; env $2 python3 $R/tools/run_one.py $3 $4 mfma4r 2>&1 | tail -1; }
run "C4 baseline (hiprtc build, no switch)" "SPCIES_ER_RTC_FLAGS=-DSPCIES_ER_DBG_NONE" C4 131072
run "C4 no counted wait (barrier kept)    " "SPCIES_ER_RTC_FLAGS=-DSPCIES_ER_DBG_NOVMW" C4 131072
run "C4 no barrier (counted wait kept)    " "SPCIES_ER_RTC_FLAGS=-DSPCIES_ER_DBG_NOSB" C4 131072
run "C4 neither                           " "SPCIES_ER_RTC_FLAGS=-DSPCIES_ER_DBG_NOBAR" C4 131072
run "C4 no y traffic                      " "SPCIES_ER_RTC_FLAGS=-DSPCIES_ER_DBG_NOY" C4 131072
run "C3 baseline (hiprtc build, no switch)" "SPCIES_FR_PD=7 SPCIES_FR_RTC_FLAGS=-DSPCIES_FR_DBG_NONE" C3 262144
run "C3 no counted wait (barrier kept)    " "SPCIES_FR_PD=7 SPCIES_FR_RTC_FLAGS=-DSPCIES_FR_DBG_NOVMW" C3 262144
run "C3 no barrier (counted wait kept)    " "SPCIES_FR_PD=7 SPCIES_FR_RTC_FLAGS=-DSPCIES_FR_DBG_NOSB" C3 262144
run "C3 neither                           " "SPCIES_FR_PD=7 SPCIES_FR_RTC_FLAGS=-DSPCIES_FR_DBG_NOBAR" C3 262144
