#!/bin/bash
# SQ counter passes for the dominant kernel of bench.py (run on the GPU box).  usage: tools/pmc_kernel.sh <variant> <tag> [lib.so]
R=${GRAFT_REPO_ROOT:-/root/repo}; V=${1:-mfma}; TAG=${2:-x}; LIB=${3:-libspcies_hip.so}
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
export SPCIES_HIP_LIB=$R/spcies_amd/$LIB
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC"
P3="GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS"
i=0
for P in "$P1" "$P2" "$P3"; do i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/bench.py --variant $V --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/p$i.err
done
python3 - $OUT <<'PY'
import csv, sys, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "admm_" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]; print(f"{k:32s} {sum(v)/len(v):.6g}")
PY
