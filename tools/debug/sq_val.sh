#!/bin/bash
# SQ counter passes over the stand-alone value gradient + forward probe at the 224x224 shape (one group per pass).
# Usage (via gpurun): bash tools/debug/sq_val.sh OUTDIR
O=$PWD/$1; R=$PWD; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for g in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_COEXEC_CYCLES"; do
  n=$(echo $g | cut -d" " -f1)
  timeout -k 5 200 rocprofv3 --pmc $g --output-format csv -d $O/pmc_$n -- python3 $R/tools/probe_val.py 6 fwd > $O/pmc_$n.log 2>&1 || echo "pmc $n failed: $(tail -2 $O/pmc_$n.log)"
done
python3 - $O <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    if "ftm_" in k:
        print(k, {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())})
PY
