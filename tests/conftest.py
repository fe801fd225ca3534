import os
import sys

# never leave bytecode behind — above all not in /root/reference, which oracle/make_golden.py and the KAT tests import from
# (the environment variable below only reaches child processes: the flag is what counts for this interpreter)
sys.dont_write_bytecode = True

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "whisper-medusa_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """libwm.so built in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    sys.path.insert(0, PKG)
    import build as wm_build
    return wm_build.build(verbose=False)


@pytest.fixture(scope="session")
def gpu(built_lib):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda", 0)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """One line of parity evidence per session + tests/parity_report.json (ties followed must be counted,
    agreement numbers asserted AND visible in the driver's log)."""
    try:
        import json
        import helpers
        P = helpers.PARITY
    except Exception:  # noqa: BLE001
        return
    if not P["runs"] and not P["tables"]:
        return
    line = (f"parity ties: {len(P['ties'])} over {P['runs']} oracle-compared runs ({P['strict_equal']} strictly identical); "
            + "; ".join(f"{k}: {v}" for k, v in P["tables"].items()))
    terminalreporter.write_line(line)
    rep = dict(P, summary=line)
    for d in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "parity_report.json"), "w") as f:
                    json.dump(rep, f, indent=1)
            except OSError:
                pass
