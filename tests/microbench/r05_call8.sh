#!/bin/bash
# round 5, GPU call 8: is the few-clip encoder bound by the latency of cold weight reads?  Same-layer weights (cache-resident ceiling) and a
# side-stream prefetch one layer ahead, at 1 and 2 clips; kernel trace of round-4 dispatch / same-layer / prefetch at one clip
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05c8; mkdir -p $O
export WM_LIB=$R/whisper-medusa_amd/whisper_medusa/libwm_enctiles.so   # tests/microbench/r05_build_enc_tiles.sh (at the time of the call these knobs were in the product library)
echo "== timing"
timeout 400 python tests/microbench/r05_enc_balance.py --prefetch --clips 1 2 --out $O/r05_enc_prefetch.json 2>&1 | grep "^encoder"
echo "== kernel trace, one clip"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/tests/microbench/r05_enc_balance.py --prefetch --clips 1 --profile --out $O/prof.json > $O/kt.log 2>&1; echo rc $?
DB=$(find /tmp/kt -name "*.db" | head -1); python $R/tests/prof_summary.py $DB $O/r05_kernel_trace_encoder_b1_prefetch.md | tail -1
grep -E "k_gemm_tiled|k_flash|k_enc_ln|k_gemm_256|k_touch" $O/r05_kernel_trace_encoder_b1_prefetch.md | head -30
