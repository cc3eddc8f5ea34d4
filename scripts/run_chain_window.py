"""One chained (out-painting) window of the arbitrary-length chain at batch G (default 1): the launch-bound regime of
BASELINE config 4.  Used under rocprofv3 --kernel-trace (scripts/prof_chain.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsheg_amd.config import get_config
from diffsheg_amd.model import UniDiffuser
from diffsheg_amd.synthetic import make_inputs
from diffsheg_amd.trainer import DDPMTrainer, sampler_namespace
from diffsheg_amd.weights import make_synthetic_state_dict
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = get_config("show")
model = UniDiffuser(cfg, make_synthetic_state_dict(cfg, 1234), device="cuda:0", precision="bf16")
tr = DDPMTrainer(sampler_namespace(cfg), model)
inp = make_inputs(cfg, G, seed=3)
dev = "cuda:0"
a, h, p = inp["audio_emb"].to(dev), {"pretrain_aud_feat": inp["pretrain_aud_feat"].to(dev)}, inp["person_id"].to(dev)
T, C, L = cfg.n_poses, cfg.net_dim_pose, cfg.overlap_len
y = {"gt": torch.randn(G, T, C, device=dev), "outpainting_mask": torch.zeros(G, T, C, dtype=torch.bool, device=dev)}
y["outpainting_mask"][:, :L] = True
for i in range(reps):
    model._cond_key = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.generate_batch(a, p, C, h, y, seed=7 + i)
    torch.cuda.synchronize()
    print(f"chained window, {G} chain(s): {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
