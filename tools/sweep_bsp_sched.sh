#!/bin/bash
# BSP block programs ordered by bsp_sched.hpp: window, machine model and compiler switches (timing experiments on the GPU box)
cfg=${1:-C5_soc}
run() { echo "== $*"; env "$@" python tools/bench_one.py $cfg bsp 65536 9 2>&1 | tail -1 | cut -c1-200; }
run SPCIES_BSP_WINDOW=24
for k in 8 12 16; do run SPCIES_BSP_WINDOW=24 SPCIES_BSP_KREG=$k; done
for w in 16 20 28 40; do run SPCIES_BSP_WINDOW=$w SPCIES_BSP_KREG=8; done
run SPCIES_BSP_WINDOW=24 SPCIES_BSP_KREG=8 SPCIES_BSP_PF=4
run SPCIES_BSP_WINDOW=24 SPCIES_BSP_KREG=8 SPCIES_BSP_PF=8
run SPCIES_BSP_WINDOW=24 SPCIES_BSP_KREG=8 SPCIES_BSP_MODEL=48,6,10,10,3,2,2
run SPCIES_BSP_WINDOW=24 SPCIES_BSP_KREG=16 SPCIES_BSP_PF=4
