#!/bin/bash
# BSP ellipMPC-ADMM / laxMPC-gen programs on the GPU box: prefetch ring depth, compile report
run() { echo "== $CFG $*"; env SPCIES_BSP_VERBOSE=1 "$@" python tools/bench_one.py ${CFG:-C2_ellip} auto 65536 4 2>&1 | grep "spcies bsp\|kernel_ms" | cut -c1-150; }
for c in C2_ellip C2_lax_gen; do export CFG=$c; run A=0; for p in 8 12 16 20 24; do run SPCIES_BSP_PF=$p; done; done
