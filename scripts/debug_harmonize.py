import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch, numpy as np
from diffsheg_amd.config import get_config
from diffsheg_amd.synthetic import SeededNoise, make_inputs
from diffsheg_amd.trainer import DDPMTrainer, sampler_namespace
from oracle import denoiser_ref as D, sampler_ref as S
from util import gpu_model, synthetic_sd
cfg = get_config('show'); sd = synthetic_sd('show')
model = gpu_model('show', 'fp32')
tr = DDPMTrainer(sampler_namespace(cfg), model)
B, L = 2, 10
inp = make_inputs(cfg, B, seed=5)
g = torch.Generator().manual_seed(17); gt = torch.zeros(B, 88, 232); gt[:, :L] = torch.randn(B, L, 232, generator=g)
mask = torch.zeros_like(gt, dtype=torch.bool); mask[:, :L] = True
kw = {"audio_emb": inp["audio_emb"], "length": None, "person_id": inp["person_id"], "add_cond": {"pretrain_aud_feat": inp["pretrain_aud_feat"]}, "y": {"gt": gt, "outpainting_mask": mask}, "pe_type": "pe_sinu"}
x, trace = tr.diffusion_ddim_val.ddim_sample_loop(model, (B, 88, 232), clip_denoised=False, model_kwargs=kw, noise_source=SeededNoise(101), return_trace=True)
def eps_fn(xc, t, c1, c2):
    with torch.no_grad(): return D.unidiffuser(sd, cfg, xc, torch.full((B,), t), c1, c2, inp['audio_emb'], inp['person_id'], inp['pretrain_aud_feat'])
otr = []
xr = S.ddim_sample_loop(eps_fn, (B, 88, 232), {"gt": gt, "outpainting_mask": mask}, S.NoiseSource(seed=101), trace=otr)
tr_c = trace.cpu()
for i, (kind, k, xs, _) in enumerate(otr):
    d = (tr_c[i] - xs).abs()
    print(i, kind, k, 'max|x|=%.3g' % xs.abs().max().item(), 'maxerr=%.3g' % d.max().item(), 'rel=%.3g' % (d.max()/xs.abs().max()).item(),
          'err_masked=%.3g err_free=%.3g' % (d[:, :L].max().item(), d[:, L:].max().item()))
# sensitivity of the oracle itself: perturb x_T by 1e-6 relative
class Pert(S.NoiseSource):
    def randn(self, shape):
        o = super().randn(shape)
        if self.i == 1: o = o * (1 + 1e-6)
        return o
xp = S.ddim_sample_loop(eps_fn, (B, 88, 232), {"gt": gt, "outpainting_mask": mask}, Pert(seed=101))
print('oracle self-sensitivity to 1e-6 rel perturbation of x_T: rel diff final = %.3g' % ((xp - xr).abs().max() / xr.abs().max()).item())
