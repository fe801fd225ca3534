"""Round 5: how many host threads should the oracle (test infrastructure) use on the GPU box?  Its decoder passes are chains of small torch ops
(1-11 rows x d) next to a few weight-streaming GEMVs; on a many-core host the default thread pool makes the small ops slower (round 4: the
fp32-pinned 8-clip table cost 708 s on the GPU box against ~180 s on an 8-core container).  Times 6 Medusa iterations of a large-v2 decode and
one fp32 encoder pass of a 6 s window at several torch thread counts.
    python tests/microbench/oracle_threads.py"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "whisper-medusa_amd"), ROOT):
    sys.path.insert(0, p)
import torch  # noqa: E402
from whisper_medusa.config import MedusaConfig, ACCEPT_TYPICAL  # noqa: E402
from whisper_medusa import synth  # noqa: E402
from oracle.whisper_medusa_oracle import Oracle  # noqa: E402

print("cpu_count", os.cpu_count(), "torch default threads", torch.get_num_threads(), flush=True)
cfg = MedusaConfig.large_v2("base_head", K=10)
t = time.time()
sd = synth.synth_state_dict(cfg, seed=0, device="cpu", logit_std=4.5)
print(f"checkpoint from the CPU generator: {time.time() - t:.1f} s", flush=True)
orc = Oracle(cfg, sd, sim="bf16")
g = torch.Generator().manual_seed(1)
enc = torch.randn(1500, cfg.d_model, generator=g)
gp = synth.bench_gen_params(cfg, max_new_tokens=48, accept_mode=ACCEPT_TYPICAL)
for n in [int(a) for a in sys.argv[1:]] or [64, 32, 16, 8, 4]:
    if n > (os.cpu_count() or n):
        continue
    torch.set_num_threads(n)
    t = time.time(); orc.decode(enc, gp, max_iters=6); t_dec = time.time() - t
    t = time.time(); orc.cross_kv(enc); t_kv = time.time() - t
    print(f"threads {n:3d}: 6 iterations {t_dec:6.2f} s (of which new_state ~{t_kv:.2f} s)", flush=True)
