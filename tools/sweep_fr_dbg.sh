#!/bin/bash
# Timing experiments on fista_r_kernel (wrong results on purpose): what the time depends on.  usage: tools/sweep_fr_dbg.sh [config] [B]
C=${1:-C3}; B=${2:-262144}
run() { echo "== $*"; env SPCIES_FR_PD=7 "SPCIES_FR_RTC_FLAGS=$*" python tools/bench_one.py $C auto $B 4 2>&1 | tail -1 | cut -c1-170; }
run -DSPCIES_FR_NOP=1
run -DSPCIES_FR_DBG_NOA=1
run -DSPCIES_FR_DBG_NOA=1 -DSPCIES_FR_DBG_NOBAR=1
run -DSPCIES_FR_DBG_NOBAR=1
