"""Summarise a rocprofv3 rocpd database (kernel-trace) per kernel: python tests/prof_summary.py <db> [out.md] [--hbm-large-v2-b1]
--hbm-large-v2-b1: the run was bench.py's default (whisper-large-v2 + Medusa-Linear K=10, ONE stream): a second table prices every
decode-path launch on the bytes it must read from HBM (its weight matrix; the stream's cross K/V for the cross-attention) against 8 TB/s."""
import re
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
lines = [f"total kernel time {tot / 1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches", "",
         "| % | calls | avg us | min us | max us | kernel |", "|---|---|---|---|---|---|"]
for r in rows[:45]:
    name = re.sub(r"\(.*", "", r[0])[:100]
    lines.append(f"| {r[2] / tot * 100:.1f} | {r[1]} | {r[3]:.2f} | {r[4]:.2f} | {r[5]:.2f} | `{name}` |")
# reconciliation with bench.py's hipEvent timing: decode-path kernel time per Medusa iteration (run bench.py --no-vanilla)
dec = c.execute("select sum(end-start)/1e3, count(*) from kernels where name like '%k_skinny%' or name like '%k_attn%' or name like '%k_rows_gemm%' "
                "or name like '%k_ln_tiles%' or name like '%k_tile_gemm%' or name like '%k_select%' or name like '%k_embed%' or name like '%k_rows_norm%' "
                "or name like '%k_accept%' or name like '%k_set_cand%'").fetchone()
n_iter = c.execute("select count(*) from kernels where name like '%k_accept%' and name not like '%vanilla%'").fetchone()[0]
n_van = c.execute("select count(*) from kernels where name like '%k_accept_vanilla%'").fetchone()[0]
if n_iter and not n_van:
    lines += ["", f"decode path: {dec[1]} kernel launches, {dec[0] / 1e3:.3f} ms of kernel time over {n_iter} Medusa iterations = "
                  f"{dec[0] / n_iter:.1f} us busy and {dec[1] / n_iter:.0f} launches per iteration "
                  "(bench.py's ms_per_launch additionally contains the launch boundaries between them)"]
if "--hbm-large-v2-b1" in sys.argv:
    d, f, V, K1 = 1280, 5120, 51865, 11
    vpad = (V + 127) // 128 * 128
    table = [                                       # (regex on the kernel name, what it is, algorithmic HBM bytes per launch)
        (r"k_skinny_gemm<8, \d, false, LdNormT<true>, EpQKVDecT", "LN1 + QKV (3 d x d)", 3 * d * d * 2),
        (r"k_skinny_gemm<8, \d, false, LdPacked, EpResidual>", "out-proj / cross-out (d x d)", d * d * 2),
        (r"k_skinny_gemm<8, \d, false, LdNormT<true>, EpF32T<true>", "LN2 + cross-q (d x d)", d * d * 2),
        (r"k_attn_mfma<true", "cross-attention (K + V of 1500 frames, 20 heads)", 2 * 1500 * d * 2),
        (r"k_skinny_gemm<8, \d, false, LdNormT<true>, EpPackedAct<1>", "LN3 + FC1 (d x 4d)", d * f * 2),
        (r"k_skinny_gemm<16, \d, false, LdPacked, EpResidual>", "FC2 (4d x d)", d * f * 2),
        (r"k_skinny_gemm<8, 1, false, LdPacked, EpF32T<false>", "vocabulary projection (V x d)", vpad * d * 2),
        (r"k_skinny_gemm<8, 2, false, LdNormT<false>, EpHead>", "Medusa heads, base pass (11 x d x d)", K1 * d * d * 2),
        (r"k_skinny_gemm<8, 1, false, LdNormT<false>, EpHead>", "base head only, verify pass (d x d)", d * d * 2),
    ]
    lines += ["", "Per-launch HBM fraction (algorithmic bytes of the launch / average duration / 8 TB/s; large-v2 + Medusa-Linear K = 10, one stream):", "",
              "| launch | calls | avg us | MB per launch | TB/s | fraction of 8 TB/s |", "|---|---|---|---|---|---|"]
    for rx, what, nbytes in table:
        hit = [r for r in rows if re.search(rx, r[0])]
        if not hit:
            continue
        calls = sum(r[1] for r in hit); us = sum(r[2] for r in hit) / calls
        lines.append(f"| {what} | {calls} | {us:.2f} | {nbytes / 1e6:.2f} | {nbytes / us / 1e6:.2f} | {nbytes / us / 1e6 / 8.0:.3f} |")
out = "\n".join(lines)
print(out)
outs = [a for a in sys.argv[2:] if not a.startswith("--")]
if outs:
    open(outs[0], "w").write(out + "\n")
