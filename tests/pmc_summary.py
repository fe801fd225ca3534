"""HBM traffic per decode iteration from a `rocprofv3 --pmc FETCH_SIZE --kernel-trace` run of bench.py.
FETCH_SIZE is reported in KB; on gfx950 it counts 64 B per 128-B request for wide coalesced streams, i.e.
HALF of the bytes (MI355X_MICROARCH.md §HBM) -> multiply by 2.  python tests/pmc_summary.py <db> [out.md [out.json [noprefetch.json]]]
Run bench.py with --no-vanilla so that every decode-path byte belongs to a Medusa iteration.  `noprefetch.json`: the json this script
wrote for the same command under WM_PREFETCH=0 — its per-iteration bytes are recorded next to the default run's as
`without_prefetch_blocks` (the in-launch prefetch blocks fill every weight / cross-K/V byte once more, one launch early)."""
import json
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(counter_value), sum(end-start)/1e3 from pmc_events where counter_name='FETCH_SIZE' "
                 "group by name order by 3 desc").fetchall()
ours = [(re.sub(r"\(.*", "", n)[:90], k, v, t) for n, k, v, t in rows
        if re.search(r"k_skinny|k_attn|k_select|k_embed|k_rows_norm|k_accept|k_set_cand|k_rows_gemm|k_ln_tiles|k_tile_gemm", n)]
n_iter = sum(k for n, k, v, t in ours if n.startswith("k_accept") and "vanilla" not in n)
n_van = sum(k for n, k, v, t in ours if "k_accept_vanilla" in n)
lines = [f"Medusa iterations profiled: {n_iter}; vanilla steps: {n_van}", "",
         "| kernel | calls | FETCH_SIZE sum (MB, raw) | x2-corrected MB | corrected MB / call |", "|---|---|---|---|---|"]
tot = 0.0
for n, k, v, t in ours:
    mb = v / 1024.0
    tot += mb
    lines.append(f"| `{n}` | {k} | {mb:.1f} | {2 * mb:.1f} | {2 * mb / k:.3f} |")
lines += ["", f"decode-path total: raw {tot:.1f} MB, corrected {2 * tot:.1f} MB over {n_iter} Medusa iterations + {n_van} vanilla steps"]
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
if len(sys.argv) > 3 and n_iter and not n_van:
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bench import kernels_sha
    cmd = "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs (MI355X)"
    json.dump({"source": cmd, "command": cmd, "kernels_sha": kernels_sha(),
               "correction": "FETCH_SIZE x 2 (gfx950 counts 64 B per 128-B request on wide coalesced streams, MI355X_MICROARCH.md HBM section)",
               "medusa_iterations": n_iter, "medusa_iteration_bytes": round(2 * tot * 1024 * 1024 / n_iter),
               "note": "host-driven hidden-state carry: an iteration is a verify pass plus a base pass only when the previous accept length was 0",
               "without_prefetch_blocks": (json.load(open(sys.argv[4]))["medusa_iteration_bytes"] if len(sys.argv) > 4 and os.path.exists(sys.argv[4]) else None),
               "counts": {"kernels": len(ours), "dispatches": int(sum(k for _, k, _, _ in ours))},
               "config": "whisper-large-v2 + medusa-linear K=10, batch 1"}, open(sys.argv[3], "w"), indent=1)
