#!/bin/bash
# Collects one round's judged artifacts on the GPU box into gpurun_out/<tag>/: GPU tests, bench lines, rocprofv3 kernel
# stats of the bench command and the three PMC passes (one counter group each) of the eager step probe.
# Usage (via gpurun): bash tools/collect_round.sh r02x [notests]
R=$PWD; TAG=${1:-r02x}; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
if [ "$2" != "notests" ]; then timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log; fi
cd /tmp
for w in c2 c3 c4; do
  S=100; [ $w = c4 ] && S=50
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -- python3 $R/bench.py --workload $w --steps $S --warmup 10 --no-gather-compare --no-cpu-baseline > $O/kt_$w.json 2> $O/kt_$w.err
  find $O/kt_$w -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$w.csv \;
done
for w in c2 c4; do
  for g in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $g | cut -d" " -f1)
    timeout -k 5 240 rocprofv3 --pmc $g --output-format csv -d $O/pmc_${w}_$n -- python3 $R/tools/probe_step.py $w 8 > $O/pmc_${w}_$n.log 2>&1 || echo "pmc $w $n failed"
  done
done
cd $R
python tools/pmc_summary.py c2 $O/pmc_c2_FETCH_SIZE $O/pmc_c2_WRITE_SIZE $O/pmc_c2_TCC_HIT_sum > $O/pmc_tmp.json && python tools/pmc_summary.py c4 $O/pmc_c4_FETCH_SIZE $O/pmc_c4_WRITE_SIZE $O/pmc_c4_TCC_HIT_sum $O/pmc_tmp.json > $O/pmc_traffic.json
# bench lines last: they read the kernel stats / traffic summaries of THIS run when profiles/ holds them (copied below)
mkdir -p $R/profiles; for w in c2 c3 c4; do cp $O/kernel_stats_$w.csv $R/profiles/${TAG}_kernel_stats_$w.csv 2>/dev/null; done; cp $O/pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json 2>/dev/null
for w in c1 c2 c3 c3k1; do python bench.py --workload $w --steps 300 --warmup 30 > $O/bench_$w.json 2> $O/bench_$w.err || echo "bench $w failed"; done
python bench.py --workload c4 --steps 100 --warmup 10 > $O/bench_c4.json 2> $O/bench_c4.err || echo "bench c4 failed"
python bench.py --workload c3 --no-spread --steps 300 --warmup 30 --no-cpu-baseline --no-gather-compare > $O/bench_c3_nospread.json 2> $O/bench_c3_nospread.err
# one graph launch per step (round-1 form) and the driver's own invocation
python bench.py --workload c2 --steps 300 --warmup 30 --steps-per-graph 1 --no-cpu-baseline --no-gather-compare > $O/bench_c2_spg1.json 2> $O/bench_c2_spg1.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver_form.json 2> $O/bench_c2_driver_form.err
for f in bench_c2_spg1 bench_c2_driver_form; do cut -c1-200 $O/$f.json; echo; done
for w in c1 c2 c3 c3k1 c4; do cut -c1-220 $O/bench_$w.json; echo; done
# density sweep: thresholds set over all slots and held (lr 0)
for w in c2 c4; do for d in 0.01 0.05 0.25 0.9; do S=200; [ $w = c4 ] && S=50; python bench.py --workload $w --density $d --steps $S --warmup 10 --no-cpu-baseline > $O/dens_${w}_$d.json 2> $O/dens_${w}_$d.err || echo "density $w $d failed"; done; done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/dens_*.json")):
    try:
        d = json.loads(open(f).read()); print(f.split("/")[-1], d["value"], d["config"]["active_density"], (d.get("gather_path") or {}).get("images_per_sec"))
    except Exception as e:
        print(f, "ERR", e)
PY
