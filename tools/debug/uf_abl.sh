#!/bin/bash
# Timing-only ablations of the fused table update + next forward (update_forward.h) at the 224x224 shape; needs a library built
# with NNUE_BUILD_ABLATIONS=1.  Usage (via gpurun): bash tools/debug/uf_abl.sh OUTDIR
NNUE_BUILD_ABLATIONS=1 python nnue-vision_amd/csrc/build.py --force > /dev/null 2>&1 || exit 1
bash tools/debug/c4_ab.sh $1 c4 -- "base" "noload NNUE_FTM_UF_ABL=1" "nostore NNUE_FTM_UF_ABL=2" "nomem NNUE_FTM_UF_ABL=3" "nomem_nofwd NNUE_FTM_UF_ABL=7" 2>&1 | grep "ms/step\|update_forward"
