#!/bin/bash
# round-4 experiment batch 2 (GPU box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== C3 mfma4r (wave-laundered DMA)"; python tools/bench_one.py C3 mfma4r 262144 3
echo "== C4 PF=6"; SPCIES_ER_RTC_FLAGS="-DSPCIES_ER_PF=6" python tools/bench_one.py C4 mfma4r 131072 3
echo "== C4 PF=2"; SPCIES_ER_RTC_FLAGS="-DSPCIES_ER_PF=2" python tools/bench_one.py C4 mfma4r 131072 3
echo "== C3 PF=6"; SPCIES_FR_RTC_FLAGS="-DSPCIES_FR_PF=6" python tools/bench_one.py C3 mfma4r 262144 3
echo "== C3 PF=2"; SPCIES_FR_RTC_FLAGS="-DSPCIES_FR_PF=2" python tools/bench_one.py C3 mfma4r 262144 3
