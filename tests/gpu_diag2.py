"""Depth / vocabulary sweep at d=1280: engine forward() vs the oracle fed with the engine's encoder output."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import MedusaConfig, synth
from oracle.whisper_medusa_oracle import Oracle
from whisper_medusa import WhisperMedusaModel

dev = torch.device("cuda", 0)
for (dl, vocab) in ((4, 4099), (12, 4099), (32, 4099), (2, 51865), (32, 51865)):
    cfg = MedusaConfig(d_model=1280, encoder_layers=2, decoder_layers=dl, encoder_attention_heads=20, decoder_attention_heads=20,
                       encoder_ffn_dim=5120, decoder_ffn_dim=5120, vocab_size=vocab, medusa_num_heads=10, medusa_hidden_size=1280,
                       medusa_choices=[1] * 11, eos_token_id=vocab - 3, pad_token_id=vocab - 3, decoder_start_token_id=vocab - 2,
                       is_multilingual=False, lang_to_id={}, task_to_id={}, no_timestamps_token_id=vocab - 1, begin_suppress_tokens=[7])
    sd = synth.synth_state_dict(cfg, seed=3, device="cuda:0")
    model = WhisperMedusaModel(cfg, sd, device=dev, max_batch=1)
    eng = model.engine
    feats = model.extract_features(synth.synth_clip(0))
    eng.encode(feats)
    enc = eng.encoder_output(1)[0]
    prompt = synth.default_prompt(cfg) + [11, 12]
    z = eng.forward_logits([prompt], 0, False)[:, 0]
    orc = Oracle(cfg, {k: v.float().cpu() for k, v in sd.items()}, sim="bf16")
    ref = orc.decoder_pass(orc.new_state(enc), prompt, 0, disable_medusa=False)
    d = (z - ref).abs()
    print(f"dec_layers={dl} vocab={vocab}: max|d|={float(d.max()):.3e} mean|d|={float(d.mean()):.3e} per-head max={[round(float(x), 4) for x in d.amax(dim=(1, 2))]}"
          f" per-row max={[round(float(x), 4) for x in d.amax(dim=(0, 2))]} refmax={float(ref.abs().max()):.2f}", flush=True)
    eng.close()
    del model, sd
    torch.cuda.empty_cache()
