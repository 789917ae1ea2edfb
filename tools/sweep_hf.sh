#!/bin/bash
# FUSED kernels on the GPU box: the two HMPC solvers at the C5 shape, MPCT-cs at the C2 shape
run() { echo "== $*"; env "$@" python tools/bench_one.py ${CFG:-C5_HMPC_SADMM} auto 65536 3 2>&1 | tail -1; }
run A=0
CFG=C5_HMPC_SADMM_nosplit run A=0
CFG=C2_cs run A=0
