#!/bin/bash
# Builds libwm_enctiles.so: the product sources + tests/microbench/r05_enc_few_clip_tiles.patch (few-clip encoder: taller balanced tiles, residual
# GEMMs with the K loop split over two blocks and the partial folded into the next LayerNorm, side-stream weight prefetch).  Loaded through WM_LIB
# by tests/microbench/r05_enc_balance.py (profiles/r05_enc_few_clip_tiles.md); never loaded by the product.  Knobs of the variant (read per
# launch / pass): WM_ENC_BALANCE=0/1, WM_ENC_XSPLIT=0/1/2, WM_ENC_PREFETCH=0/1/2, WM_ENC_SAME_LAYER=0/1.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d)
mkdir -p $T/whisper-medusa_amd $T/include
cp -r $R/whisper-medusa_amd/csrc $T/whisper-medusa_amd/csrc
cp $R/include/wm.h $T/include/
(cd $T && patch -p1 -s < $R/tests/microbench/r05_enc_few_clip_tiles.patch)
rm -f $T/whisper-medusa_amd/csrc/*.o
WM_CSRC=$T/whisper-medusa_amd/csrc python $R/whisper-medusa_amd/build.py --variant enctiles
rm -rf $T
