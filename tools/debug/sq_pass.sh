#!/bin/bash
# SQ counter passes (one group per pass) over the eager step probe: MFMA busy, wave residency / waiting, LDS conflicts.
# Usage (via gpurun): bash tools/debug/sq_pass.sh OUTDIR workload
O=$PWD/$1; W=${2:-c2}; R=$PWD; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for g in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVES SQ_INSTS_VALU"; do
  n=$(echo $g | cut -d" " -f1)
  timeout -k 5 240 rocprofv3 --pmc $g --output-format csv -d $O/pmc_$n -- python3 $R/tools/probe_step.py $W 6 > $O/pmc_$n.log 2>&1 || echo "pmc $n failed"
done
python3 - $O <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    if any(x in k for x in ("ftm_", "l1_", "tail_", "ste_", "conv_", "sgd_", "sqnorm")):
        print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
