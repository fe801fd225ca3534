"""bench.py's algorithmic byte / FLOP figures (what roofline.achieved is priced on) against SURVEY.md §8d."""
import importlib.util
import os

import pytest

from helpers import ROOT, MedusaConfig


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_decode_iteration_bytes_match_the_survey_figures():
    b = _bench()
    lin = MedusaConfig.large_v2("base_head", K=10)
    blk = MedusaConfig.large_v2("medusa_block", K=10)
    # W_a = 1.638 GB, W_v = 1.605 GB, X = 245.76 MB / stream / pass, S_kv = 0.328 MB x len (both passes)
    one = b.decode_iter_bytes(lin, 1, 64)
    assert abs(one - (1.638e9 + 1.605e9 + 2 * 245.76e6 + 0.328e6 * 64)) / one < 5e-3
    assert abs(one - 3.75e9) / 3.75e9 < 0.01                       # "B=1 Linear: ~3.76 GB/iter"
    b32 = b.decode_iter_bytes(lin, 32, 64)
    assert abs(b32 - 19.6e9) / 19.6e9 < 0.03                       # "B=32 Linear: ~19.6 GB/iter"
    assert b.decode_iter_bytes(blk, 1, 64) > one                    # the extra layer and its cross-KV
    # fp8 decoder-layer weights: the layer matrices shrink to one byte per parameter, everything else stays
    f8 = b.decode_iter_bytes(lin, 1, 64, True)
    layer_params = 32 * (6 * 1280 * 1280 + 2 * 1280 * 5120)
    assert abs((one - f8) - 2 * (layer_params - 4 * 32 * (7 * 1280 + 5120))) / one < 1e-3


def test_executed_bytes_price_what_the_carry_really_runs():
    """roofline.frac_executed: with both passes every iteration it IS SURVEY §8d's figure; a carried iteration at
    one stream is the verify pass alone; at several streams the base pass's weights always stream, only the attention reads drop."""
    b = _bench()
    lin = MedusaConfig.large_v2("base_head", K=10)
    full = b.decode_iter_bytes(lin, 1, 64)
    assert abs(b.executed_bytes(lin, 1, 64, False, 1.0, 1.0) - full) < 1.0
    p = b.decode_iter_bytes(lin, 1, 64, parts=True)
    kv = p["cross_kv_per_stream_pass"] + p["self_kv_per_stream_pass"]
    assert abs(b.executed_bytes(lin, 1, 64, False, 0.0, 0.0) - (p["w_verify"] + kv)) < 1.0
    assert abs(b.executed_bytes(lin, 32, 64, False, 1.0, 0.5) - (p["w_verify"] + p["w_base"] + 32 * 1.5 * kv)) < 64.0
    assert p["w_vanilla"] == p["w_verify"]                       # Linear: the verify pass and a vanilla step both run one head


def test_merged_step_schedule_is_priced_per_step():
    """Several streams on the merged-step schedule (wm_stats.schedule_steps > 0): every step is ONE pass — the weights once, each stream's
    K/V once — and bench.py prices an iteration as steps-per-iteration of those.  With a share p0 of iterations that accept nothing, an
    iteration is 1 + p0 steps: the K/V reads equal the lock-step schedule's (whose base pass skips the attention of carried streams), the
    weights stream 1 + p0 times instead of twice — the schedule's gain is the base pass's launch chain, not bytes."""
    b = _bench()
    lin = MedusaConfig.large_v2("base_head", K=10)
    p = b.decode_iter_bytes(lin, 32, 64, parts=True)
    kv = p["cross_kv_per_stream_pass"] + p["self_kv_per_stream_pass"]
    one_step = b.executed_bytes(lin, 32, 64, False, 0.0, 0.0)
    assert abs(one_step - (p["w_verify"] + 32 * kv)) < 64.0
    lock_step = b.executed_bytes(lin, 32, 64, False, 1.0, 0.45)
    assert abs((lock_step - 1.45 * one_step) - (p["w_base"] - 0.45 * p["w_verify"])) < 4096.0


def test_newest_committed_bench_line_reports_the_merged_step_legs():
    """The newest profiles/rNN_bench_default.json (a measurement artefact: absent or renamed, this check is skipped — the accounting itself
    is tested above on synthetic numbers) carries the 32-stream legs with steps-per-iteration between 1 and 2 and a parity check."""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json")))
    if not files:
        pytest.skip("no committed bench line")
    line = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    legs = [c for c in line.get("configs", []) if str(c.get("config", "")).startswith("configs[1] shape at 32 streams")]
    if not legs:
        pytest.skip("the newest bench line has no 32-stream leg")
    leg = legs[0]
    assert leg["merged_steps_per_iteration"] and 1.0 < leg["merged_steps_per_iteration"] < 2.0 and leg["parity_checked"] is True


def test_prefill_flops_match_the_survey_figure():
    b = _bench()
    lin = MedusaConfig.large_v2("base_head", K=10)
    assert abs(b.prefill_flops(lin) - 2.59e12) / 2.59e12 < 0.01    # encoder 2.273 + cross-KV projection 0.315 TFLOP


def test_committed_traffic_counters_were_taken_on_the_committed_decode_sources():
    """bench.py only reports `roofline.traffic` from a PMC pass stamped with the hash of the decode-path sources it was measured on
    (profiles/rNN_pmc_traffic.json, written by tests/pmc_summary.py on the GPU).  The newest committed file must match the committed
    sources — an edit to csrc/ after the last GPU pass would silently turn the figure into null in the next bench line."""
    import glob
    import json
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    assert files
    tj = json.load(open(files[-1]))
    assert tj["kernels_sha"] == bench.kernels_sha(), f"{os.path.basename(files[-1])} is stale: re-run the FETCH_SIZE pass or undo the source edit"
    assert tj["medusa_iteration_bytes"] > 3e9
