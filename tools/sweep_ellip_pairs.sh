#!/bin/bash
# banded BSP programs (ellipMPC ADMM, lax / equ with vector rho): single blocks against pairs, ring depth (GPU box)
run() { cfg=$1; shift; echo "== $cfg $*"; env "$@" python tools/bench_one.py $cfg bsp 65536 7 2>&1 | tail -1 | cut -c1-200; }
for cfg in C2_ellip C2_lax_gen C2_equ_gen; do
  run $cfg SPCIES_BSP_PAIRS=0
  run $cfg A=1
  run $cfg SPCIES_BSP_PF=8
  run $cfg SPCIES_BSP_PF=16
done
