#!/bin/bash
# round 6, GPU call 19: the flash softmax maxima without inline asm (wm_encoder.hip built with -fno-honor-nans): encoder determinism on the small shapes
# (sha over 6 passes), the large-v2 32-clip sha and time against the round-5 encoder kernels (libwm_fl0.so), the parity file three times, then the whole suite
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c19; mkdir -p $O
P=$GRAFT_REPO_ROOT/whisper-medusa_amd/whisper_medusa
for shape in micro micro10block tiny; do
timeout 200 python - <<PY 2>&1 | grep -v amdgpu.ids | tee -a $O/enc_det.log
import sys, hashlib
sys.path.insert(0, "$GRAFT_REPO_ROOT/tests")
import torch
import test_gpu_parity as t
r = t.Rig("$shape", torch.device("cuda", 0))
shas = []
for i in range(6):
    r.encode(); torch.cuda.synchronize()
    shas.append(hashlib.sha256(r.eng.encoder_output(2).float().cpu().numpy().tobytes()).hexdigest()[:12])
print("$shape", shas, "equal to the first encode:", bool(torch.equal(r.eng.encoder_output(2), r.enc)))
PY
done
for rep in 1 2; do
  for v in f16 fl0; do
    WM_LIB_F16=$P/libwm_$v.so timeout 300 python tests/microbench/r06_enc_time.py 2>&1 | grep "^lib=" | tee -a $O/enc_time.log
  done
done
for rep in 1 2 3; do
  timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider 2>&1 | grep -h "^FAILED\|passed\|failed" | cut -c1-200 | tee -a $O/repeat.log
done
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1 ) 2>&1 | grep real; echo pytest rc $?
grep -h "^FAILED\|^ERROR\|passed\|failed" $O/pytest_gpu.log | cut -c1-300 | tail -8
grep -h "parity ties" $O/pytest_gpu.log | cut -c1-200
