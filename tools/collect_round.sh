#!/bin/bash
# Collects one round's judged artifacts on the GPU box into gpurun_out/<tag>/: GPU tests, bench lines c1..c4,
# rocprofv3 kernel stats (c2, c4) and the three PMC passes of tools/probe_fwd_l1.py.  Usage (via gpurun): bash tools/collect_round.sh r01s
R=$PWD; TAG=${1:-r01x}; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
for w in c1 c2 c3; do python bench.py --workload $w --steps 300 --warmup 30 > $O/bench_$w.json 2> $O/bench_$w.err || echo "bench $w failed"; done
python bench.py --workload c4 --steps 100 --warmup 10 > $O/bench_c4.json 2> $O/bench_c4.err || echo "bench c4 failed"
cd /tmp
for w in c2 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -- python3 $R/bench.py --workload $w --steps 100 --warmup 10 --no-gather-compare > $O/kt_$w.json 2> $O/kt_$w.err
  find $O/kt_$w -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$w.csv \;
done
for g in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $g | cut -d" " -f1)
  timeout -k 5 90 rocprofv3 --pmc $g --output-format csv -d $O/pmc_$n -- python3 $R/tools/probe_fwd_l1.py c2 > $O/pmc_$n.log 2>&1 || echo "pmc $n failed"
done
cd $R; for w in c1 c2 c3 c4; do cut -c1-200 $O/bench_$w.json; done
