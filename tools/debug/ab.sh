#!/bin/bash
# usage: ab.sh OUTDIR ENVVAR PATTERN workload  -- A/B of a developer knob (0/1 twice) with rocprof kernel averages for PATTERN
O=$PWD/$1; V=$2; PAT=$3; W=${4:-c4}; R=$PWD; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for x in 0 1 0 1; do
  export $V=$x
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$x -- python3 $R/bench.py --workload $W --steps 100 --warmup 10 --no-cpu-baseline --no-gather-compare > $O/${W}_$x.json 2>$O/err_$x || exit 1
  f=$(find $O/kt_$x -name "*kernel_stats.csv"); echo "== $V=$x $(python3 -c "import json;d=json.load(open('$O/${W}_$x.json'));print(d['ms_per_step'])")"
  grep -E "$PAT" $f | awk -F, '{print "   ", $(NF-4)}'
  rm -rf $O/kt_$x
done
