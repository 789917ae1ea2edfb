#!/bin/bash
# round-4 experiment batch 1 (GPU box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== C4 mfma4r"; python tools/bench_one.py C4 mfma4r 131072 3
echo "== C5_soc bsp default"; python tools/bench_one.py C5_soc bsp 65536 3
echo "== C5_soc bsp vgpr-form"; SPCIES_BSP_FLAGS="-mllvm -amdgpu-mfma-vgpr-form" python tools/bench_one.py C5_soc bsp 65536 3
echo "== C2_ellip bsp default"; python tools/bench_one.py C2_ellip bsp 65536 3
echo "== C2_ellip bsp vgpr-form"; SPCIES_BSP_FLAGS="-mllvm -amdgpu-mfma-vgpr-form" python tools/bench_one.py C2_ellip bsp 65536 3
echo "== headline RC_REGS"; python tools/sweep_mfma4_flags.py "" "-DSPCIES_MFMA4U_RC_REGS=1" "-DSPCIES_MFMA4U_RC_REGS=1 -DSPCIES_MFMA4U_PF=6" "-DSPCIES_MFMA4U_SEED_ACC=1"
