#!/bin/bash
# BSP block programs ordered by bsp_sched.hpp: window, machine model and compiler switches (timing experiments on the GPU box)
cfg=${1:-C5_soc}
run() { echo "== $*"; env "$@" python tools/bench_one.py $cfg bsp 65536 9 2>&1 | tail -1 | cut -c1-200; }
run A=1
for w in 16 24 40; do
  run SPCIES_BSP_WINDOW=$w SPCIES_BSP_MODEL=32,4,6,7,2,1,1,2
  run SPCIES_BSP_WINDOW=$w SPCIES_BSP_MODEL=32,4,6,7,2,1,1,4
done
run SPCIES_BSP_WINDOW=64 SPCIES_BSP_MODEL=32,4,6,7,2,1,1,2
