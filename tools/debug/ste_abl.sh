#!/bin/bash
# Timing-only ablations of the STE / conv-weight backward at the 224x224 shape (library built with NNUE_BUILD_ABLATIONS=1).
# Usage (via gpurun): bash tools/debug/ste_abl.sh OUTDIR
NNUE_BUILD_ABLATIONS=1 python nnue-vision_amd/csrc/build.py --force > /dev/null 2>&1 || exit 1
bash tools/debug/c4_ab.sh $1 c4 -- "base" "nopartials NNUE_STE_ABL=1" "nodload NNUE_STE_ABL=2" "nopatchload NNUE_STE_ABL=4" "noloads NNUE_STE_ABL=6" "nothing NNUE_STE_ABL=7" 2>&1 | grep "ms/step\|ste_conv"
