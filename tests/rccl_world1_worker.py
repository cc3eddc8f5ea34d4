"""Worker of tests/test_gpu_sharded.py::test_rccl_collectives_run_at_world_size_1: ONE rank, backend nccl (= RCCL on ROCm).
DSH_FORCE_COLLECTIVES=1 keeps the per-rank code paths of the long-audio mode from short-circuiting at world size 1, so that the
device-side broadcast of the feature stream and the device-side gather of the outputs execute on a single-GPU box; the result must
equal the plain single-process call bit for bit.  Reference behaviour replaced: mp.spawn + NCCL in runner.py:80-122, the per-rank
sharding of ddpm_show_trainer.py:743-750,924-931."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from diffsheg_amd.config import get_config  # noqa: E402
from diffsheg_amd.synthetic import make_inputs  # noqa: E402
from diffsheg_amd.trainer import DDPMTrainer, sampler_namespace  # noqa: E402
from util import gpu_model  # noqa: E402


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    t = torch.tensor([3.5], device="cuda:0", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    assert float(t) == 3.5
    cfg = get_config("show")
    model = gpu_model("show", "fp32")
    tr = DDPMTrainer(sampler_namespace(cfg), model)
    N = 400
    inp = make_inputs(cfg, 1, frames=N, seed=9)
    audio, hub, pid = inp["audio_emb"].cuda(), inp["pretrain_aud_feat"].cuda(), inp["person_id"].cuda()
    os.environ["DSH_FORCE_COLLECTIVES"] = "0"
    ref = tr.sample_arbitrary_len_sharded(audio, pid, {"pretrain_aud_feat": hub}, 3, seed=11)
    os.environ["DSH_FORCE_COLLECTIVES"] = "1"
    out = tr.sample_arbitrary_len_sharded(audio, pid, {"pretrain_aud_feat": hub}, 3, seed=11, inputs_on_rank0_only=True)
    torch.cuda.synchronize()
    assert out is not None and tuple(out.shape) == (1, N, cfg.net_dim_pose)
    assert torch.equal(out, ref), float((out - ref).abs().max())
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_WORLD1_OK")


if __name__ == "__main__":
    main()
