#!/bin/bash
# The parity suites of the FeatureTransformer products and the trainer under each developer knob's non-default setting
# (the alternate code paths must stay green).  Usage (via gpurun): bash tools/debug/knob_matrix.sh OUTDIR
# KNOBS="A=1 B=0": only these.
O=$1; mkdir -p $O
ALL="NNUE_FTM_BF16=0 NNUE_FTM_BWD_W64=0 NNUE_FTM_BF_KT64=0 NNUE_FTM_VAL_BF6=0 NNUE_FTM_BWD_BF6=0 NNUE_FTM_BF6_KT=64 NNUE_SGD_SCALAR=1 NNUE_FUSE_TABLE_UPDATE=0 NNUE_FTM_XCD_REMAP=0 NNUE_DEFER_STE=0 NNUE_NORM_PARTIALS=0 NNUE_FTM_VAL_DMA=1 NNUE_FTM_VAL_PLANES=1 NNUE_FTM_BF6_BN=128 NNUE_FUSE_NEXT_FORWARD=0 NNUE_CONV_PATCHES=0 NNUE_CONV_PATCHES=1 NNUE_CONV_REFORM=1"
for e in ${KNOBS:-$ALL}; do
  n=$(echo $e | tr '=' '_')
  ( export $e; timeout -k 10 400 python -m pytest tests/test_gpu_ftm.py tests/test_gpu_step_shapes.py tests/test_gpu_trainer.py tests/test_gpu_update_forward.py tests/test_gpu_patches.py -m gpu -q -x > $O/$n.log 2>&1 )
  echo "$e: $(tail -1 $O/$n.log)"
done
