"""Build libwm.so (the gfx950 HIP engine) in-tree with hipcc.  No cmake / JIT cache: the built
library lives next to the Python package so it travels with the repo snapshot to the GPU box.

    python whisper-medusa_amd/build.py [--force]
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.environ.get("WM_CSRC", os.path.join(HERE, "csrc"))      # WM_CSRC: a patched copy of csrc/ (A/B variants of tests/microbench)
OUT_DIR = os.path.join(HERE, "whisper_medusa")
LIB = os.path.join(OUT_DIR, "libwm.so")
SOURCES = ["wm_engine.hip", "wm_decoder.hip", "wm_encoder.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# gfx950 can hand the first kernel arguments to a wave in SGPRs at launch (kernarg preload): the decode chain's kernels then
# compute their addresses and issue their first loads without waiting for an s_load round trip.  The code object carries a
# compatibility prologue for firmware without the feature.  WM_NO_KERNARG_PRELOAD=1 builds without it (A/B runs).
if not os.environ.get("WM_NO_KERNARG_PRELOAD"):
    FLAGS += ["-mllvm", "-amdgpu-kernarg-preload-count=16"]
# MFMA results in architectural VGPRs.  By default the register allocator parks MFMA destinations in AGPRs; every kernel whose VALU
# code consumes them (attention: scores -> softmax, lazily rescaled outputs) then pays v_accvgpr_read / _write / _mov copies — 45 % of
# the VALU instructions of the encoder flash-attention loop, 16 % of the decode self-attention kernel (ISA counts).
# The register-bound kernel is the pipelined 256 x 256 encoder GEMM (k_gemm_256p: 128 accumulator + 96 fragment registers): every default
# instance sits at 256 VGPRs with 20-56 bytes per lane of scratch — in its prologue / epilogue blocks, none in the 32-MFMA K-loop blocks (ISA,
# -Rpass-analysis=kernel-resource-usage) —, the fp8 16x16x128 GEMMs use 186-216, flash attention 180.  WM_MFMA_AGPR=1 builds the old form.
if not os.environ.get("WM_MFMA_AGPR"):
    FLAGS += ["-mllvm", "-amdgpu-mfma-vgpr-form"]
# experiments prepared for the next round (csrc/: off = the measured code, byte for byte)
if os.environ.get("WM_EP_WAIT_ONCE"):
    FLAGS += ["-DWM_EP_WAIT_ONCE"]


def _newest_src():
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))] + [os.path.join(HERE, "..", "include", "wm.h")]
    return max(os.path.getmtime(f) for f in files)


# Per-source flags.  wm_encoder.hip: -fno-honor-nans — its softmax / LayerNorm / GELU arithmetic never produces or tests a NaN (masked scores are -inf), and with
# `nnan` on the fmaxf calls the backend drops the operand canonicalisation (a v_max x, x per operand under IEEE mode) in front of every maximum over MFMA results:
# 120 -> 62 max instructions per 64-key step and wave of the encoder flash attention, same values (csrc/wm_encoder.hip: vmax2; profiles/r06_encoder_prefill.md).
SRC_FLAGS = {"wm_encoder.hip": ["-fno-honor-nans"]}


def _compile(src, extra=(), suffix=""):
    obj = os.path.join(CSRC, src.replace(".hip", suffix + ".o"))
    cmd = [HIPCC] + FLAGS + SRC_FLAGS.get(src, []) + list(extra) + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj


LIB_F16 = os.path.join(OUT_DIR, "libwm_f16.so")      # the same sources for the fp16 single-plane decode contract (csrc/wm_common.h, wm_config.act_fp16)


def build(force=False, verbose=True):
    """Both product libraries: libwm.so (bf16 hi / lo decode operands) and libwm_f16.so (-DWM_ACT_F16)."""
    newest = _newest_src()
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        build_variant(LIB, verbose=verbose)
    if force or not os.path.exists(LIB_F16) or os.path.getmtime(LIB_F16) < newest:
        build_variant(LIB_F16, ("-DWM_ACT_F16",), "_f16", verbose=verbose)
    return LIB


def build_variant(lib, extra=(), suffix="", verbose=True):
    """Compile the three sources (objects get `suffix`) and link them as `lib`.  The product build is build_variant(LIB);
    A/B arms of tests/microbench call it with another path (`--variant NAME [-DFLAG ...]` -> libwm_NAME.so, loaded through WM_LIB)."""
    with cf.ThreadPoolExecutor(len(SOURCES)) as ex:
        objs = list(ex.map(lambda s: _compile(s, extra, suffix), SOURCES))
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {lib} ({os.path.getsize(lib) / 1e6:.1f} MB)")
    return lib


def build_timeline(verbose=True):
    """Debug variant with in-kernel timeline probes (-DWM_TIMELINE, csrc/wm_skinny_gemm.h): libwm_tl.so next to the product
    library, loaded by tests/microbench/timeline.py through engine.load_library(path).  Never loaded by the product."""
    lib = os.path.join(OUT_DIR, "libwm_tl.so")
    with cf.ThreadPoolExecutor(len(SOURCES)) as ex:
        objs = list(ex.map(lambda s: _compile(s, ("-DWM_TIMELINE",), "_tl"), SOURCES))
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {lib}")
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        name = sys.argv[i + 1]
        build_variant(os.path.join(OUT_DIR, f"libwm_{name}.so"), tuple(a for a in sys.argv[i + 2:] if a.startswith("-D")), "_" + name)
        sys.exit(0)
    build(force="--force" in sys.argv)
    if "--timeline" in sys.argv:
        build_timeline()
