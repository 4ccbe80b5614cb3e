#!/bin/bash
# Value gradient at the 224x224 shape: the LDS-DMA kernel (ftv_kernels.hip) vs the six-plane tile kernel, + timing-only ablations.
# (The NNUE_*_ABL* knobs exist only in a library built with NNUE_BUILD_ABLATIONS=1 python nnue-vision_amd/csrc/build.py --force.)
# Usage (via gpurun): bash tools/debug/val_v2.sh OUTDIR
O=$PWD/$1; R=$PWD; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { # name, env...
  n=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -- python3 $R/tools/probe_val.py 20 > $O/$n.log 2>&1 || echo "$n failed: $(tail -3 $O/$n.log)" )
  python3 - $O/$n $n <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ftv_" in r["Name"] or "bf6" in r["Name"] or "split_planes" in r["Name"]:
            print(sys.argv[2], r["Name"].replace("(anonymous namespace)::", "")[:60], r["Calls"], "avg %.1f min %.1f us" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
}
run dma
run old NNUE_FTM_VAL_DMA=0
run abl1 NNUE_FTM_VAL_ABL=1
run abl2 NNUE_FTM_VAL_ABL=2

