#!/bin/bash
# Static wave priority in the LDS-DMA value gradient (NNUE_FTM_VAL_PRIO): which pairing breaks the lockstep of the two
# workgroups of a CU.  Usage (via gpurun): bash tools/debug/val_prio.sh OUTDIR
O=$PWD/$1; R=$PWD; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for m in 0 1 2 3; do
  ( export NNUE_FTM_VAL_PRIO=$m
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$m -- python3 $R/tools/probe_val.py 20 > $O/p$m.log 2>&1 || echo "p$m failed: $(tail -3 $O/p$m.log)" )
  python3 - $O/p$m $m <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ftv_values" in r["Name"]:
            print("prio mode", sys.argv[2], "avg %.1f min %.1f us" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
