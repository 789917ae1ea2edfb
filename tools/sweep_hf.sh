#!/bin/bash
# FUSED HMPC kernel experiments on the GPU box (re-specialised with hiprtc): chunk size, instance groups per wavefront
run() { echo "== $*"; env "$@" python tools/bench_one.py C5_HMPC_SADMM auto 65536 3 2>&1 | tail -1; }
run A=0
run SPCIES_HFUSED_RTC=1 SPCIES_HFUSED_CHUNK=77824 SPCIES_HFUSED_FLAGS="-DSPCIES_HFUSED_CHUNK=77824"
run SPCIES_HFUSED_RTC=1 SPCIES_HFUSED_CHUNK=20480 SPCIES_HFUSED_FLAGS="-DSPCIES_HFUSED_CHUNK=20480"
