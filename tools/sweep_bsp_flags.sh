#!/bin/bash
# BSP block programs (ellipMPC soc / ADMM, laxMPC switches): compiler flag and ring-depth experiments on the GPU box
run() { cfg=$1; var=$2; shift 2; echo "== $cfg $var $*"; env "$@" python tools/bench_one.py $cfg $var 65536 3 2>&1 | tail -1; }
for cfg in C5_soc C2_ellip; do
  run $cfg bsp A=0
  run $cfg bsp SPCIES_BSP_FLAGS="-mllvm -amdgpu-mfma-vgpr-form"
  run $cfg bsp SPCIES_BSP_PF=16 SPCIES_BSP_FLAGS="-mllvm -amdgpu-mfma-vgpr-form"
  run $cfg bsp SPCIES_BSP_PF=24 SPCIES_BSP_FLAGS="-mllvm -amdgpu-mfma-vgpr-form"
done
