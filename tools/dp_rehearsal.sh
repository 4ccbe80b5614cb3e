#!/bin/bash
# One-GPU rehearsal of the data-parallel step over RCCL (one rank, collectives forced on): what the captured collective,
# the eager collective and the sharded update cost against the plain single-rank step.  Usage (via gpurun): bash tools/dp_rehearsal.sh r02f
TAG=${1:-r02x}; O=gpurun_out/$TAG; mkdir -p $O
B="--steps 300 --warmup 30 --no-cpu-baseline --no-gather-compare"
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
python bench.py --workload c2 $B > $O/c2_single.json 2>$O/err1
NNUE_DP_FORCE_COLLECTIVES=1 $TR --master-port 29511 bench.py --gpus 1 --workload c2 $B > $O/c2_dp1_captured.json 2>$O/err2
NNUE_DP_FORCE_COLLECTIVES=1 $TR --master-port 29515 bench.py --gpus 1 --workload c2 $B --steps-per-graph 1 > $O/c2_dp1_captured_spg1.json 2>$O/err6
NNUE_DP_FORCE_COLLECTIVES=1 NNUE_DP_CAPTURE=0 $TR --master-port 29512 bench.py --gpus 1 --workload c2 $B > $O/c2_dp1_eager.json 2>$O/err3
C4="--workload c4 --steps 100 --warmup 10 --no-cpu-baseline --no-gather-compare"
python bench.py $C4 > $O/c4_single.json 2>$O/err7
# default under collectives at this shape: the gradient's FACTORS are all-gathered, fused update on the global batch
NNUE_DP_FORCE_COLLECTIVES=1 $TR --master-port 29516 bench.py --gpus 1 $C4 > $O/c4_dp1_factors.json 2>$O/err8
NNUE_DP_FORCE_COLLECTIVES=1 NNUE_DP_FACTOR_EXCHANGE=0 $TR --master-port 29513 bench.py --gpus 1 $C4 > $O/c4_dp1_sharded.json 2>$O/err4
NNUE_DP_FORCE_COLLECTIVES=1 NNUE_DP_FACTOR_EXCHANGE=0 NNUE_DP_SHARDED_UPDATE=0 $TR --master-port 29514 bench.py --gpus 1 $C4 > $O/c4_dp1_allreduce.json 2>$O/err5
for f in c2_single c2_dp1_captured c2_dp1_captured_spg1 c2_dp1_eager c4_single c4_dp1_factors c4_dp1_sharded c4_dp1_allreduce; do echo $f; cut -c1-260 $O/$f.json; echo; done
for e in $O/err2 $O/err8 $O/err4; do tail -n 2 $e; done
