"""Sweep the skinny-GEMM launch plan knobs on the layer-GEMM microbench (one process per setting)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, os
sys.path.insert(0, os.path.join(%r, "whisper-medusa_amd")); sys.path.insert(0, %r)
import torch
from whisper_medusa import MedusaConfig, WhisperMedusaModel, synth
cfg = MedusaConfig.large_v2("base_head", K=10)
cfg.encoder_layers = 1; cfg.decoder_layers = 2
sd = synth.synth_state_dict(cfg, seed=0, device="cuda:0")
m = WhisperMedusaModel(cfg, sd, device="cuda:0")
for rows in (1, 11):
    ms, nb = m.engine.profile_layer_gemms(rows=rows, reps=200)
    print("rows", rows, "us_per_layer_gemms", round(ms * 1e3, 2))
''' % (ROOT, ROOT)
for env in ({}, {"WM_PLAN_WAVE_CAP": "5"}, {"WM_PLAN_WAVE_CAP": "8"}, {"WM_PLAN_TARGET_WAVES": "512"}, {"WM_PLAN_TARGET_WAVES": "2048"},
            {"WM_PLAN_NK_MAX": "8"}, {"WM_PLAN_RT2": "0"}, {"WM_PLAN_WAVE_CAP": "4", "WM_PLAN_NK_MAX": "40"}):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
    print(env, " | ".join(l for l in r.stdout.splitlines() if l.startswith("rows")), r.stderr[-200:] if r.returncode else "", flush=True)
