#!/bin/bash
# Round 6: builds the three upper-bound arms VERDICT r05 item 1 asks for (product sources + tests/microbench/r06_upper_bounds.patch):
#   libwm_nolo.so     -DWM_EXP_NOLO: no bf16 lo plane in k_rows_gemm / k_skinny2_gemm (loads + MFMAs) and k_ln_tiles (store)
#   libwm_noln.so     -DWM_EXP_NOLN: the 96 k_ln_tiles launches of a batched pass dropped (the GEMMs read a stale operand)
#   libwm_nolonoln.so both
# Numerically meaningless; only the time counts.  Loaded through WM_LIB; never by the product.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)     # run from a checkout of the commit tests/microbench/PATCHES.json names for the patch
T=$(mktemp -d)
mkdir -p $T/whisper-medusa_amd $T/include
cp -r $R/whisper-medusa_amd/csrc $T/whisper-medusa_amd/csrc
cp $R/include/wm.h $T/include/
(cd $T && patch -p1 -s < $R/tests/microbench/r06_upper_bounds.patch)
rm -f $T/whisper-medusa_amd/csrc/*.o
export WM_CSRC=$T/whisper-medusa_amd/csrc
python $R/whisper-medusa_amd/build.py --variant nolo -DWM_EXP_NOLO &
python $R/whisper-medusa_amd/build.py --variant noln -DWM_EXP_NOLN &
wait
python $R/whisper-medusa_amd/build.py --variant nolonoln -DWM_EXP_NOLO -DWM_EXP_NOLN
rm -rf $T
