"""BASELINE configs[0] on the GPU (BEAT n_poses=34, batch 1, fp32 path), `steps` ancestral steps (default 100 of the 1000):
used under rocprofv3 --kernel-trace to see where a batch-1 fp32 evaluation spends its time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsheg_amd.config import get_config
from diffsheg_amd.model import UniDiffuser
from diffsheg_amd.synthetic import make_inputs
from diffsheg_amd.trainer import DDPMTrainer, sampler_namespace
from diffsheg_amd.weights import make_synthetic_state_dict
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
cfg = get_config("beat")
model = UniDiffuser(cfg, make_synthetic_state_dict(cfg, 1234), device="cuda:0", precision=prec)
tr = DDPMTrainer(sampler_namespace(cfg, ddim=False, diffusion_steps=steps), model)
inp = make_inputs(cfg, 1, seed=3)
dev = "cuda:0"
a, h, p = inp["audio_emb"].to(dev), {"pretrain_aud_feat": inp["pretrain_aud_feat"].to(dev)}, inp["person_id"].to(dev)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.generate_batch(a, p, cfg.net_dim_pose, h, {}, seed=7 + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{prec} BEAT B=1, {steps} ddpm steps: {1e3 * dt:.1f} ms = {1e3 * dt / steps:.3f} ms per step", flush=True)
