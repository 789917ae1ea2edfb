#!/bin/bash
# usage: tools/kstat.sh <kernel-name-regex>   (after `make -C spcies_amd/csrc asm`)
# prints register usage and an instruction histogram for one kernel of the gfx950 assembly.
B=/root/repo/spcies_amd/csrc/build
S=$B/spcies_hip-hip-amdgcn-amd-amdhsa-gfx950.s
NAME=$(grep -o "^_ZN6spcies[A-Za-z0-9_]*:" $S | tr -d ':' | grep -E "$1" | head -1)
[ -z "$NAME" ] && { echo "no kernel matches $1"; exit 1; }
echo "kernel: $NAME"
grep -A10 "Function Name: $NAME " $B/resource_usage.txt | grep -E "GPRs|Scratch|Occup|LDS" | sed 's/.*:0: *//; s/ \[-R.*//'
L=$(grep -n "^$NAME:" $S | cut -d: -f1)
awk -v s=$L 'NR>=s' $S | awk '/s_endpgm/{print; exit} {print}' > $B/kernel.s
echo "lines: $(wc -l < $B/kernel.s)"
for p in v_mfma ds_read ds_write v_accvgpr_read v_accvgpr_write s_nop v_fma_f64 v_mul_f64 v_add_f64 v_cndmask v_max_f64 v_min_f64 v_cmp scratch_ s_waitcnt s_load global_load global_store v_readlane v_writelane; do
  c=$(grep -c "$p" $B/kernel.s); [ "$c" != "0" ] && echo "  $p: $c"; done
