#!/bin/bash
# MFMA4R (FISTA, fista_r_kernel.inc) tuning sweep on the GPU box (hiprtc re-specialisation).  usage: tools/sweep_fr.sh [config] [B]
C=${1:-C3}; B=${2:-262144}
run() { echo "== $*"; env "$@" python tools/bench_one.py $C auto $B 3 2>&1 | tail -1 | cut -c1-140; }
run A=0
run SPCIES_FR_RTC_FLAGS="-DSPCIES_FR_PF=4"
run SPCIES_FR_RTC_FLAGS="-DSPCIES_FR_PF=6"
run SPCIES_FR_RTC_FLAGS="-DSPCIES_FR_PF=3"
run SPCIES_FR_PD=2
run SPCIES_FR_PD=4
run SPCIES_FR_PD=6
run SPCIES_FR_NW=3
