"""Component-by-component numeric report of the HIP engine against the oracle (run on the GPU box).
Writes gpurun_out/diag.json; every section is independent so one failure does not hide the rest."""
import json
import os
import sys
import time
import traceback

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import ROOT, MedusaConfig, synth, golden_gen_params, clip_for, ACCEPT_TYPICAL, ACCEPT_GREEDY  # noqa: E402
from oracle.whisper_medusa_oracle import Oracle, log_mel  # noqa: E402
from whisper_medusa import WhisperMedusaModel  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
rep = {}


def section(name):
    def deco(fn):
        t = time.time()
        try:
            rep[name] = fn()
        except Exception as e:  # noqa: BLE001
            rep[name] = {"error": repr(e), "trace": traceback.format_exc()[-1500:]}
        rep[name + "_sec"] = round(time.time() - t, 2)
        print(name, json.dumps(rep[name])[:600], flush=True)
        return fn
    return deco


def stats(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    d = np.abs(a - b)
    return dict(max_abs=float(d.max()), mean_abs=float(d.mean()), ref_max=float(np.abs(b).max()),
                nan=int(np.isnan(a).sum()))


dev = torch.device("cuda", 0)
print(torch.cuda.get_device_name(0), flush=True)

for tag, cfg, seed in (("micro", MedusaConfig.micro(K=4), 11), ("micro10", MedusaConfig.micro(K=10), 12),
                       ("microblock", MedusaConfig.micro(K=4, heads_type="medusa_block"), 13),
                       ("tiny", MedusaConfig.tiny_en(K=4), 0),
                       ("wide", MedusaConfig(d_model=1280, encoder_layers=2, decoder_layers=2, encoder_attention_heads=20,
                                             decoder_attention_heads=20, encoder_ffn_dim=5120, decoder_ffn_dim=5120, vocab_size=4099,
                                             medusa_num_heads=10, medusa_hidden_size=1280, medusa_choices=[1] * 11,
                                             eos_token_id=4096, pad_token_id=4096, decoder_start_token_id=4097,
                                             is_multilingual=False, lang_to_id={}, task_to_id={}, no_timestamps_token_id=4098,
                                             begin_suppress_tokens=[7, 4096]), 5)):
    sd = synth.synth_state_dict(cfg, seed=seed)
    orc = Oracle(cfg, sd, sim="bf16")
    wavs = [clip_for(cfg, 0), clip_for(cfg, 1)[: cfg.n_mel_frames * 160 // 3]]
    feats_ref = np.stack([log_mel(w, cfg.num_mel_bins, cfg.n_mel_frames * 160) for w in wavs])
    model = None

    @section(f"{tag}/create")
    def _():
        global model
        model = WhisperMedusaModel(cfg, sd, device=dev, max_batch=2)
        return {"blob_mb": model._blob.numel() / 1e6}

    if model is None:
        continue
    eng = model.engine

    @section(f"{tag}/logmel")
    def _():
        f = model.extract_features(wavs).cpu().numpy()
        return stats(f, feats_ref)

    enc_eng = None

    @section(f"{tag}/encoder")
    def _():
        global enc_eng
        eng.encode(torch.from_numpy(feats_ref).to(dev))
        enc_eng = eng.encoder_output(2)
        ref = torch.stack([orc.encode(torch.from_numpy(feats_ref[i])) for i in range(2)])
        return stats(enc_eng.numpy(), ref.numpy())

    if enc_eng is None:
        continue

    @section(f"{tag}/cross_kv")
    def _():
        ckv = orc.cross_kv(enc_eng[1])
        out = {}
        for kvl in (0, cfg.n_kv_layers - 1):
            k, v = eng.cross_kv(kvl, 1, 1)
            out[f"k{kvl}"] = stats(k.numpy(), ckv[kvl][0][1].numpy())
            out[f"v{kvl}"] = stats(v.numpy(), ckv[kvl][1][1].numpy())
        return out

    @section(f"{tag}/forward_prompt_medusa")
    def _():
        prompt = synth.default_prompt(cfg) + [11, 12]
        z = eng.forward_logits([prompt, prompt[::-1]], 0, False)          # [K+1, 2, T, V]
        out = {}
        for b, toks in enumerate((prompt, prompt[::-1])):
            st = orc.new_state(enc_eng[b])
            ref = orc.decoder_pass(st, toks, 0, disable_medusa=False)       # [K+1, T, V]
            out[f"s{b}"] = stats(z[:, b].numpy(), ref.numpy())
            out[f"s{b}_argmax_equal"] = bool((z[:, b].argmax(-1) == ref.argmax(-1)).all())
        return out

    @section(f"{tag}/forward_verify_rows")
    def _():
        prompt = synth.default_prompt(cfg)
        eng.forward_logits([prompt, prompt], 0, True)
        cands = [[5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15][: cfg.medusa_num_heads + 1]] * 2
        z = eng.forward_logits(cands, len(prompt), True)                    # [1, 2, K+1, V]
        st = orc.new_state(enc_eng[0])
        orc.decoder_pass(st, prompt, 0, True)
        st["kv_len"] = len(prompt)
        ref = orc.decoder_pass(st, cands[0], len(prompt), True)
        return stats(z[0, 0].numpy(), ref[0].numpy())

    for mode, mname in ((ACCEPT_TYPICAL, "typical"), (ACCEPT_GREEDY, "greedy")):
        for eos_free in (True, False):
            @section(f"{tag}/decode_{mname}_{'noeos' if eos_free else 'eos'}")
            def _():
                gp = golden_gen_params(cfg, mode, 40, suppress_eos=eos_free)
                eng.encode(torch.from_numpy(feats_ref).to(dev))
                seqs = eng.decode(gp, 2)
                st = eng.stats()
                out = {"stats": {k: (v if not isinstance(v, float) else round(v, 3)) for k, v in st.items()}}
                for b in range(2):
                    r = orc.decode(enc_eng[b], gp)
                    n = 0
                    while n < min(len(seqs[b]), len(r.ids)) and seqs[b][n] == r.ids[n]:
                        n += 1
                    out[f"s{b}"] = dict(equal=seqs[b] == r.ids, first_diff=n, n_eng=len(seqs[b]), n_ref=len(r.ids),
                                        ref_accepts=r.accept_lengths)
                return out

    @section(f"{tag}/vanilla")
    def _():
        gp = golden_gen_params(cfg, ACCEPT_GREEDY, 30)
        gp.vanilla = True
        eng.encode(torch.from_numpy(feats_ref).to(dev))
        seqs = eng.decode(gp, 2)
        r = orc.decode(enc_eng[0], gp)
        return dict(equal=seqs[0] == r.ids, n=len(seqs[0]), eng=seqs[0][:12], ref=r.ids[:12])

    model.engine.close()

json.dump(rep, open(os.path.join(OUT, "diag.json"), "w"), indent=1)
print("DIAG DONE")
