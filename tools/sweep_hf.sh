#!/bin/bash
# FUSED HMPC kernel on the GPU box: k-slabs per LDS chunk (three buffers; re-specialised with hiprtc)
run() { echo "== $CFG $*"; env "$@" python tools/bench_one.py ${CFG:-C5_HMPC_SADMM} auto 65536 3 2>&1 | tail -1 | cut -c1-140; }
for c in C5_HMPC_SADMM C5_HMPC_SADMM_nosplit; do export CFG=$c
run A=0
for k in 20480 30720 49152; do run SPCIES_HFUSED_RTC=1 SPCIES_HFUSED_CHUNK=$k SPCIES_HFUSED_FLAGS="-DSPCIES_HFUSED_CHUNK=$k"; done
done
