#!/bin/bash
# usage: quick_stats.sh OUTDIR workload...   -- rocprofv3 kernel stats of bench.py per workload (top kernels printed)
O=$PWD/$1; shift; R=$PWD; mkdir -p $O
B="--steps 200 --warmup 16 --no-cpu-baseline --no-gather-compare"
cd /tmp; export TMPDIR=/tmp
for w in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -- python3 $R/bench.py --workload $w $B > $O/$w.json 2>$O/err_$w || exit 1
  f=$(find $O/kt_$w -name "*kernel_stats.csv"); cp $f $O/ks_$w.csv; rm -rf $O/kt_$w
  echo "== $w $(python3 -c "import json;d=json.load(open('$O/$w.json'));print(d['ms_per_step'], d['value'])")"
  python3 - $O/ks_$w.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f"   {float(r['AverageNs'])/1e3:8.1f} us x{r['Calls']:>5}  {r['Name'].replace('(anonymous namespace)::','')[:90]}")
PY
done
