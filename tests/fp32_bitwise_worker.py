"""Worker of tests/test_gpu_sharded.py::test_fp32_chain_equality_is_bitwise_without_the_few_row_gemm: run with DSH_GEMM_KSPLIT=0 in
the environment (the switch is read once per process).  Samples a five-chain stream through sample_arbitrary_len_sharded on the fp32
path and every chain alone, and requires the two to agree BIT FOR BIT."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from diffsheg_amd.config import get_config  # noqa: E402
from diffsheg_amd.synthetic import make_inputs  # noqa: E402
from diffsheg_amd.trainer import DDPMTrainer, sampler_namespace, split_segments  # noqa: E402
from util import gpu_model  # noqa: E402

assert os.environ.get("DSH_GEMM_KSPLIT") == "0"
cfg = get_config("show")
model = gpu_model("show", "fp32")
tr = DDPMTrainer(sampler_namespace(cfg), model)
N, n_seg = 3 * 78 + 10 + 37, 3
inp = make_inputs(cfg, 1, frames=N, seed=15)
audio, cond, pid = inp["audio_emb"], {"pretrain_aud_feat": inp["pretrain_aud_feat"]}, inp["person_id"]
out = tr.sample_arbitrary_len_sharded(audio, pid, cond, n_seg, seed=31)
segs = split_segments(N, n_seg, cfg.n_poses, cfg.overlap_len)
exact = 0
for i, sg in enumerate(segs):
    solo = tr.sample_arbitrary_len(audio[:, sg.start:sg.stop].cuda(), pid, {k: v[:, sg.start:sg.stop].cuda() for k, v in cond.items()},
                                   seed=31, row_keys=[i])
    exact += int(torch.equal(solo[0], out[0, sg.start:sg.stop]))
print(f"[fp32, DSH_GEMM_KSPLIT=0] {exact} of {len(segs)} chains bit-identical to the chain sampled alone")
assert exact == len(segs)
print("FP32_BITWISE_OK")
