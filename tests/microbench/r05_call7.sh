#!/bin/bash
# round 5, GPU call 7: encoder at 1-4 clips — balanced taller tiles (WM_ENC_BALANCE) and residual GEMMs with the K loop split over two blocks,
# partial folded into the next LayerNorm (WM_ENC_XSPLIT): times + output differences, kernel trace of both dispatches at one clip, encoder parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05c7; mkdir -p $O
export WM_LIB=$R/whisper-medusa_amd/whisper_medusa/libwm_enctiles.so   # tests/microbench/r05_build_enc_tiles.sh (at the time of the call these knobs were in the product library)
echo "== timing"
timeout 400 python tests/microbench/r05_enc_balance.py --out $O/r05_enc_balance.json 2>&1 | grep -v "^$" | tail -30
echo "== kernel trace, one clip"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/tests/microbench/r05_enc_balance.py --clips 1 --profile --out $O/prof.json > $O/kt.log 2>&1; echo rc $?
DB=$(find /tmp/kt -name "*.db" | head -1); python $R/tests/prof_summary.py $DB $O/r05_kernel_trace_encoder_b1.md | tail -1
grep -E "k_gemm_tiled|k_flash|k_enc_ln|k_gemm_256" $O/r05_kernel_trace_encoder_b1.md | head -30
cd $R
echo "== parity (encoder-facing tests, default dispatch = both on, FC2 only)"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -q -p no:cacheprovider -x -k "encoder or logmel or end_to_end or pinned or consistency or audio" 2>&1 | tail -8
