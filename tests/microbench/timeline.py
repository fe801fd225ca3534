"""In-kernel timeline of the decode chain (debug build libwm_tl.so, -DWM_TIMELINE): one stream by default, `--batch 32` = the merged-step
schedule's 352-row launches (k_rows_gemm records: cycles to "MFMAs issued", "K-slice partials exchanged", exit).

    python whisper-medusa_amd/build.py --timeline
    WM_LIB=whisper-medusa_amd/whisper_medusa/libwm_tl.so python tests/microbench/timeline.py --out gpurun_out/timeline_new

Thread 0 of every block of the weight-streaming GEMMs and attention kernels records {realtime at entry / exit (100 MHz,
chip-wide), shader cycles entry -> operands ready -> products done -> exit}.  The script decodes one clip (large-v2 +
Medusa-Linear K=10, hipGraph replays), groups the records into launches and prints, per kernel of a decoder layer and per
pass type, the launch duration (first block entry -> last block exit), the boundary before it (previous launch's last exit ->
first entry), the entry spread and the in-block phase times.
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "whisper-medusa_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

REC = np.dtype([("tag", "<u4"), ("block", "<u4"), ("rt0", "<u8"), ("rt1", "<u8"), ("c_prep", "<u4"), ("c_mid", "<u4"),
                ("c_end", "<u4"), ("pad", "<u4")])
STEP = {1: "LN1+QKV", 2: "self-attn", 3: "out-proj", 4: "LN2+cq", 5: "cross-attn", 6: "cross-out", 7: "LN3+FC1", 8: "FC2"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/timeline")
    ap.add_argument("--max-new", type=int, default=48)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--cap", type=int, default=3_000_000)
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    import torch
    from whisper_medusa import MedusaConfig, WhisperMedusaModel, synth, ACCEPT_TYPICAL
    from whisper_medusa import engine as eng_mod
    lib = eng_mod.load_library()
    if not hasattr(lib, "wm_debug_timeline"):
        raise SystemExit("this library has no timeline probes: build with --timeline and set WM_LIB")
    dev = torch.device("cuda", 0)
    cfg = MedusaConfig.large_v2("base_head", K=10)
    cfg.decoder_layers = args.layers
    sd = synth.synth_state_dict(cfg, seed=0, device=str(dev), logit_std=4.5)
    model = WhisperMedusaModel(cfg, sd, device=dev, max_batch=args.batch)
    eng = model.engine
    buf = torch.zeros(args.cap * REC.itemsize, dtype=torch.uint8, device=dev)
    idx = torch.zeros(4, dtype=torch.int32, device=dev)
    lib.wm_debug_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    lib.wm_debug_timeline.restype = C.c_int32
    wav = np.stack([synth.synth_clip(j, cfg.n_mel_frames * 160) for j in range(args.batch)])
    feats = model.extract_features(wav)
    gp = synth.bench_gen_params(cfg, max_new_tokens=args.max_new, accept_mode=ACCEPT_TYPICAL)
    eng.encode(feats)
    eng.decode(gp, args.batch)                               # warm: graphs captured
    assert lib.wm_debug_timeline(eng.h, C.c_void_p(buf.data_ptr()), C.c_void_p(idx.data_ptr()), args.cap) == 0
    eng.decode(gp, args.batch)
    st = eng.stats()
    torch.cuda.synchronize()
    n = int(idx[0].item())
    rec = np.frombuffer(buf.cpu().numpy().tobytes(), dtype=REC)[: min(n, args.cap)]
    print(f"{n} records, {st['iterations']} iterations, {st['ms_decode'] / max(st['iterations'], 1):.3f} ms / iteration (probed build)")
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)

    # ---- group into launches: sort by entry time, split where the tag changes ----
    o = np.argsort(rec["rt0"], kind="stable")
    rec = rec[o]
    cut = np.flatnonzero(np.diff(rec["tag"].astype(np.int64)) != 0) + 1
    starts = np.concatenate([[0], cut]); ends = np.concatenate([cut, [len(rec)]])
    launches = []
    for a, b in zip(starts, ends):
        r = rec[a:b]
        launches.append(dict(tag=int(r["tag"][0]), n=int(b - a), t0=int(r["rt0"].min()), t0_last=int(r["rt0"].max()),
                             t1_first=int(r["rt1"].min()), t1=int(r["rt1"].max()),
                             c_prep=float(r["c_prep"].mean()), c_mid=float(r["c_mid"].mean()), c_end=float(r["c_end"].mean()),
                             c_end_max=float(r["c_end"].max()), blk=float(np.median((r["rt1"] - r["rt0"]).astype(np.float64))) * 0.01))
    TICK = 0.01                                               # 100 MHz realtime counter -> microseconds
    rows = {}
    for i, L in enumerate(launches):
        tag = L["tag"]
        if tag >= 1000 and tag < 8192:
            key = ("heads" if tag == 1000 else "vocab", 0)
        else:
            key = (STEP.get(tag % 16, str(tag % 16)), tag >> 13)
        gap = (L["t0"] - launches[i - 1]["t1"]) * TICK if i > 0 else 0.0
        d = rows.setdefault(key, dict(n=0, dur=[], gap=[], spread=[], prep=[], mid=[], end=[], blocks=[], endmax=[], blk=[]))
        d["n"] += 1
        d["dur"].append((L["t1"] - L["t0"]) * TICK); d["gap"].append(gap); d["spread"].append((L["t0_last"] - L["t0"]) * TICK)
        d["prep"].append(L["c_prep"]); d["mid"].append(L["c_mid"]); d["end"].append(L["c_end"]); d["blocks"].append(L["n"])
        d["endmax"].append(L["c_end_max"]); d["blk"].append(L["blk"])
    lines = [f"{n} block records over {st['iterations']} Medusa iterations; {st['ms_decode'] / max(st['iterations'], 1):.3f} ms / iteration with probes",
             "", "times in microseconds (100 MHz realtime counter, 10 ns resolution), phases in shader cycles (mean over blocks)", "",
             "| kernel | rows/stream | launches | blocks | gap before (median) | entry spread | first entry -> last exit | block entry -> exit (median) | cycles: operands ready | products done | exit (mean) | exit (max) |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    tot = {}
    for key in sorted(rows, key=lambda k: (k[1], list(STEP.values()).index(k[0]) if k[0] in STEP.values() else 99)):
        d = rows[key]
        med = lambda v: float(np.median(v))
        lines.append(f"| {key[0]} | {key[1]} | {d['n']} | {int(med(d['blocks']))} | {med(d['gap']):.2f} | {med(d['spread']):.2f} | {med(d['dur']):.2f} | {med(d['blk']):.2f} | "
                     f"{med(d['prep']):.0f} | {med(d['mid']):.0f} | {med(d['end']):.0f} | {med(d['endmax']):.0f} |")
        tot.setdefault(key[1], [0.0, 0.0])
        if key[0] in STEP.values():
            tot[key[1]][0] += med(d["dur"]); tot[key[1]][1] += med(d["gap"])
    lines.append("")
    for m, (dur, gap) in sorted(tot.items()):
        lines.append(f"rows/stream {m}: one decoder layer = {dur:.1f} us inside launches + {gap:.1f} us between them = {dur + gap:.1f} us")
    txt = "\n".join(lines)
    print(txt)
    with open(args.out + ".md", "w") as f:
        f.write(txt + "\n")
    np.savez_compressed(args.out + ".npz", rec=rec[: 400000])
    with open(args.out + ".json", "w") as f:
        json.dump({f"{k[0]}|{k[1]}": {kk: (float(np.median(v)) if isinstance(v, list) else v) for kk, v in d.items()} for k, d in rows.items()}, f)


if __name__ == "__main__":
    main()
