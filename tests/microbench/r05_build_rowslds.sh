#!/bin/bash
# Builds libwm_rowslds.so: the product sources + tests/microbench/r05_rows_lds_ln_tail.patch (k_rows_lds: LDS-shared token tiles, register-direct
# weights; LayerNorm in the producing GEMM's tail with a write-through hand-off).  Loaded through WM_LIB by the A/B runs of round 5
# (profiles/r05_rows_lds.md); never loaded by the product.  Knobs of the variant (read per launch): WM_ROWS_LDS=0/1, WM_LN_TAIL=0/1,
# WM_RL_NW, WM_RL_FT.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d)
mkdir -p $T/whisper-medusa_amd $T/include
cp -r $R/whisper-medusa_amd/csrc $T/whisper-medusa_amd/csrc
cp $R/include/wm.h $T/include/
(cd $T && patch -p1 -s < $R/tests/microbench/r05_rows_lds_ln_tail.patch)
rm -f $T/whisper-medusa_amd/csrc/*.o
WM_CSRC=$T/whisper-medusa_amd/csrc python $R/whisper-medusa_amd/build.py --variant rowslds
rm -rf $T
