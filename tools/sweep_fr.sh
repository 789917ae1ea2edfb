#!/bin/bash
# MFMA4R (FISTA, fista_r_kernel.inc) tuning sweep on the GPU box (hiprtc re-specialisation).  usage: tools/sweep_fr.sh [config] [B]
C=${1:-C3}; B=${2:-262144}
run() { echo "== $*"; env "$@" python tools/bench_one.py $C auto $B 3 2>&1 | tail -1 | cut -c1-140; }
run A=0
for v in 120 135 160 170; do run SPCIES_FR_MAX_REG_VECS=$v; done
