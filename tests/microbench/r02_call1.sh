#!/bin/bash
# round-2 GPU call 1: smoke, A/B bench old vs new skinny GEMM, in-kernel timelines, then the GPU test-suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02c1; mkdir -p $O
echo "== smoke"; timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -3 $O/smoke.log
echo "== bench new"; timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_new.json 2> $O/bench_new.err; echo rc $?; tail -c 1500 $O/bench_new.json
echo "== bench v1"; WM_ABI_ANY=1 WM_LIB=tests/microbench/libwm_v1.so timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_v1.json 2> $O/bench_v1.err; echo rc $?; tail -c 600 $O/bench_v1.json
echo "== timeline new"; WM_LIB=whisper-medusa_amd/whisper_medusa/libwm_tl.so timeout 600 python tests/microbench/timeline.py --out $O/timeline_new > $O/timeline_new.log 2>&1; echo rc $?; tail -25 $O/timeline_new.log
echo "== timeline v1"; WM_ABI_ANY=1 WM_LIB=tests/microbench/libwm_tl_v1.so timeout 600 python tests/microbench/timeline.py --out $O/timeline_v1 > $O/timeline_v1.log 2>&1; echo rc $?; tail -25 $O/timeline_v1.log
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo rc $?; tail -40 $O/pytest.log
