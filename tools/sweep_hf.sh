#!/bin/bash
# FUSED HMPC kernel on the GPU box: the two solvers at the C5 shape
run() { echo "== $*"; env "$@" python tools/bench_one.py ${CFG:-C5_HMPC_SADMM} auto 65536 3 2>&1 | tail -1; }
run A=0
CFG=C5_HMPC_SADMM_nosplit run A=0
