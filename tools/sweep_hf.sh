#!/bin/bash
# FUSED HMPC kernel on the GPU box: the two solvers at the C5 shape, then chunk sizes (re-specialised with hiprtc)
run() { echo "== $*"; env "$@" python tools/bench_one.py ${CFG:-C5_HMPC_SADMM} auto 65536 3 2>&1 | tail -1; }
run A=0
CFG=C5_HMPC_SADMM_nosplit run A=0
run SPCIES_HFUSED_RTC=1 SPCIES_HFUSED_CHUNK=77824 SPCIES_HFUSED_FLAGS="-DSPCIES_HFUSED_CHUNK=77824"
run SPCIES_HFUSED_RTC=1 SPCIES_HFUSED_CHUNK=20480 SPCIES_HFUSED_FLAGS="-DSPCIES_HFUSED_CHUNK=20480"
run SPCIES_HFUSED_RTC=1 SPCIES_HFUSED_CHUNK=58368 SPCIES_HFUSED_FLAGS="-DSPCIES_HFUSED_CHUNK=58368"
