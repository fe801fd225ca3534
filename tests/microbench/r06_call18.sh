#!/bin/bash
# round 6, GPU call 18: is the micro10block decode failure of call 17 deterministic, and does it follow the encoder kernels (product library against one built with
# the round-5 flash loop and GEMM schedule)?  The parity file's micro10block cases, five times per library.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c18; mkdir -p $O
P=$GRAFT_REPO_ROOT/whisper-medusa_amd/whisper_medusa
for lib in f16 fl0; do
  for rep in 1 2 3 4 5; do
    echo "== lib $lib rep $rep"
    WM_LIB_F16=$P/libwm_$lib.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "micro10block" 2>&1 | grep -h "^FAILED\|passed\|failed" | cut -c1-200 | tee -a $O/repeat_$lib.log
  done
done
echo "== encoder determinism (micro10block shape, two ragged clips): sha of the encoder output over 6 passes per library"
for lib in f16 fl0; do
WM_LIB_F16=$P/libwm_$lib.so timeout 200 python - <<PY 2>&1 | grep -v amdgpu.ids | tee -a $O/enc_det_$lib.log
import sys, hashlib
sys.path.insert(0, "$GRAFT_REPO_ROOT/tests")
import torch
import test_gpu_parity as t
r = t.Rig("micro10block", torch.device("cuda", 0))
shas = []
for i in range(6):
    r.encode(); torch.cuda.synchronize()
    shas.append(hashlib.sha256(r.eng.encoder_output(2).float().cpu().numpy().tobytes()).hexdigest()[:12])
print("$lib", shas, "equal to the first encode:", [bool(torch.equal(r.eng.encoder_output(2), r.enc))])
PY
done
