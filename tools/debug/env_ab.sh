#!/bin/bash
# usage: env_ab.sh OUTDIR workload PATTERN "ENV=a" "ENV=b" ...  -- rocprof kernel averages for PATTERN under each env setting
O=$PWD/$1; W=$2; PAT=$3; shift 3; R=$PWD; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for e in "$@" "$@"; do
  i=$((i+1))
  env $e python3 -c "pass" || exit 1
  ( export $e; rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$i -- python3 $R/bench.py --workload $W --steps 150 --warmup 15 --no-cpu-baseline --no-gather-compare > $O/${W}_$i.json 2>$O/err_$i ) || exit 1
  f=$(find $O/kt_$i -name "*kernel_stats.csv"); echo "== $e $(python3 -c "import json;d=json.load(open('$O/${W}_$i.json'));print(d['ms_per_step'])")"
  grep -E "$PAT" $f | awk -F, '{print "   ", $(NF-4)}'
  rm -rf $O/kt_$i
done
