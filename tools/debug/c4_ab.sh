#!/bin/bash
# A/B of the 224x224 step under rocprofv3 --kernel-trace --stats: each argument is "name ENV=VALUE ..." ("name" alone = defaults).
# (The NNUE_*_ABL* knobs exist only in a library built with NNUE_BUILD_ABLATIONS=1 python nnue-vision_amd/csrc/build.py --force.)
# Usage (via gpurun): bash tools/debug/c4_ab.sh OUTDIR [workload] -- "new" "old NNUE_FTM_VAL_DMA=0"
O=$PWD/$1; W=${2:-c4}; shift 2; [ "$1" = "--" ] && shift
R=$PWD; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for spec in "$@"; do
  set -- $spec; n=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -- python3 $R/bench.py --workload $W --steps 50 --warmup 10 --no-gather-compare --no-cpu-baseline > $O/$n.json 2> $O/$n.err || echo "$n failed: $(tail -3 $O/$n.err)" )
  python3 - $O/$n $n $O/$n.json <<'PY'
import csv, glob, json, sys
try:
    d = json.load(open(sys.argv[3])); print(sys.argv[2], "ms/step", d["ms_per_step"], "images/s", d["value"])
except Exception as e:
    print(sys.argv[2], "no bench line", e)
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    print("   %-70s calls %5s avg %8.1f us" % (r["Name"].replace("(anonymous namespace)::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
