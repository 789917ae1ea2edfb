#!/bin/bash
# SQ counter passes for tools/run_one.py (run on the GPU box).  usage: tools/pmc_cmd.sh <tag> <kernel-substring> <config> <B> <variant>
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; KN=$2; shift 2
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"
P3="GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM"
P4="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_WRITE_REQ_sum"
P5="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/run_one.py "$@" > $OUT/p$i.out 2> $OUT/p$i.err
done
python3 - $OUT "$KN" <<'PY'
import csv, sys, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]; print(f"{k:32s} {sum(v)/len(v):.6g}")
PY
