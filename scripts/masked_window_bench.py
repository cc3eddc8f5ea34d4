"""Wall time of one out-painting window (jump schedule (3,5): 63 evaluations + 48 undo steps) at batch B, bf16 SHOW.
usage: python scripts/masked_window_bench.py B   (env: DSH_DUAL, DSH_LEVEL_CACHE)
Round 3 (MI355X): B = 200: 366 ms (two sub-batch streams, per-stream timestep cache) / 391 ms (cache off) / 502 ms (one stream);
B = 950: 1497 / 1564 / 1743 ms (cache slots: 1.75 GiB per 317-clip sub-batch)."""
import sys, time, torch, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from diffsheg_amd.config import get_config
from diffsheg_amd.model import UniDiffuser
from diffsheg_amd.synthetic import make_inputs
from diffsheg_amd.trainer import DDPMTrainer, sampler_namespace
from diffsheg_amd.weights import make_synthetic_state_dict
cfg = get_config("show"); sd = make_synthetic_state_dict(cfg, 1234)
model = UniDiffuser(cfg, sd, device="cuda:0", precision="bf16")
tr = DDPMTrainer(sampler_namespace(cfg), model)
B = int(sys.argv[1]); T, Cc, L = cfg.n_poses, cfg.net_dim_pose, cfg.overlap_len
small = make_inputs(cfg, 16, seed=3)
rep = (B + 15) // 16
a = small["audio_emb"].repeat(rep, 1, 1)[:B].cuda(); h = small["pretrain_aud_feat"].repeat(rep, 1, 1)[:B].cuda()
pid = torch.zeros(B, cfg.style_dim, device="cuda"); pid[:, 0] = 1
y = {"gt": torch.randn(B, T, Cc, device="cuda"), "outpainting_mask": torch.zeros(B, T, Cc, dtype=torch.bool, device="cuda")}
y["outpainting_mask"][:, :L] = True
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model._cond_key = None
    tr.generate_batch(a, pid, Cc, {"pretrain_aud_feat": h}, y, seed=5 + i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"B={B} masked window (63 evals + 48 undo): {dt*1e3:.1f} ms  LEVEL_CACHE={os.environ.get('DSH_LEVEL_CACHE','1')} DUAL={os.environ.get('DSH_DUAL','3')}")
