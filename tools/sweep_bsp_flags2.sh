#!/bin/bash
# BSP block programs: LLVM scheduler switches (timing experiments on the GPU box)
cfg=${1:-C5_soc}
run() { echo "== $*"; env "$@" python tools/bench_one.py $cfg bsp 65536 5 2>&1 | tail -1; }
run A=0
run SPCIES_BSP_FLAGS="-mllvm -amdgpu-sched-strategy=max-ilp"
run SPCIES_BSP_FLAGS="-mllvm -amdgpu-sched-strategy=iterative-ilp"
run SPCIES_BSP_FLAGS="-mllvm -enable-misched=false"
run SPCIES_BSP_FLAGS="-mllvm -enable-misched=false -mllvm -enable-post-misched=false"
run SPCIES_BSP_FLAGS="-mllvm -enable-post-misched=false"
run SPCIES_BSP_FLAGS="-mllvm -misched-topdown"
run SPCIES_BSP_FLAGS="-mllvm -misched-bottomup"
run SPCIES_BSP_SEG=1
run SPCIES_BSP_SEG=1000
