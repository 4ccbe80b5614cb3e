#!/bin/bash
# SQ counter passes over the eager step probe (tools/probe_step.py) of a workload, one counter group per pass; prints the kernels
# whose name contains a pattern.  Usage (via gpurun): bash tools/debug/sq_step.sh OUTDIR WORKLOAD PATTERN [PATTERN...]
O=$PWD/$1; W=$2; shift 2; R=$PWD; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for g in "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  n=$(echo $g | cut -d" " -f1)
  timeout -k 5 200 rocprofv3 --pmc $g --output-format csv -d $O/pmc_$n -- python3 $R/tools/probe_step.py $W 8 > $O/pmc_$n.log 2>&1 || echo "pmc $n failed: $(tail -2 $O/pmc_$n.log)"
done
python3 - $O "$@" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:70]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    if any(p in k for p in sys.argv[2:]):
        print(k, {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())})
PY
