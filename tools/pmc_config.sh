#!/bin/bash
# SQ counter passes for the kernel of one configuration / variant (run on the GPU box; counters in their own runs, kernel-trace only).
# usage: tools/pmc_config.sh <tag> <kernel-name-substring> <config> <B> <variant>   -> gpurun_out/pmc_<tag>/summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; KN=$2; shift; shift
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_SALU"
P3="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS SQ_WAVES"
P4="SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_FLAT TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/run_one.py "$@" > /dev/null 2> $OUT/p$i.err
done
python3 - $OUT "$KN" > $OUT/summary.txt <<'PY'
import csv, sys, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# kernel name contains:", sys.argv[2], " dispatches per counter:", {k: len(v) for k, v in list(agg.items())[:1]})
for k in sorted(agg):
    v = agg[k]; print(f"{k:32s} {sum(v)/len(v):.6g}")
PY
cat $OUT/summary.txt
