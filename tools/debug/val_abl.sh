#!/bin/bash
# Timing-only ablations of the six-plane value-gradient kernel at the 224x224 shape (NNUE_FTM_BF6_ABL, wrong results):
# what each phase of a K tile costs.  Usage (via gpurun): bash tools/debug/val_abl.sh OUTDIR "0 1 2 3 4 5" [ENVVAR=value ...]
O=$PWD/$1; R=$PWD; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
LIST=${2:-"0 1 2 3 4 5"}; shift 2
for kv in "$@"; do export "$kv"; done
for a in $LIST; do
  export NNUE_FTM_BF6_ABL=$a
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/abl_$a -- python3 $R/tools/probe_val.py 30 > $O/abl_$a.log 2>&1 || echo "abl $a failed"
  f=$(find $O/abl_$a -name "*kernel_stats.csv" | head -1)
  echo "ABL=$a $(grep -E 'bf6|ValEpi' $f | awk -F, '{print $1, "calls", $2, "avg_ns", $4}' | head -2)"
done
