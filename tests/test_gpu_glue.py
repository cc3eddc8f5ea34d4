"""§8f rows on a real MI355X: HuBERT-feature interpolation and inv_standardize vs the exact torch ops the
reference calls (datasets/show.py:98,157-162), and the checkpoint loader round trip."""
import os
import tempfile

import pytest
import torch
import torch.nn.functional as F

from diffsheg_amd.config import get_config
from diffsheg_amd.glue import interpolate_features, inv_standardize, load_checkpoint, split_motion
from util import synthetic_sd


@pytest.mark.gpu
@pytest.mark.parametrize("B,Tin,Tout,C", [(1, 499, 300, 1024), (2, 50, 88, 1024), (1, 7, 1, 16), (1, 1, 5, 8), (3, 88, 88, 128)])
def test_interpolate_matches_torch_align_corners(B, Tin, Tout, C):
    g = torch.Generator().manual_seed(Tin)
    x = torch.randn(B, Tin, C, generator=g)
    ref = F.interpolate(x.swapaxes(-1, -2), size=Tout, mode="linear", align_corners=True).swapaxes(-1, -2)
    got = interpolate_features(x.cuda(), Tout).cpu()
    assert got.shape == ref.shape
    # source positions are computed in fp32 on both sides (t * (Tin-1)/(Tout-1)): ~1e-5 relative slack on the weights
    assert (got - ref).abs().max().item() < 1e-4
    if B == 1:
        assert torch.equal(interpolate_features(x[0].cuda(), Tout).cpu(), got[0])


@pytest.mark.gpu
def test_inv_standardize_and_split():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 166, 232, generator=g)
    mean, std = torch.randn(232, generator=g), torch.rand(232, generator=g) + 0.5
    got = inv_standardize(x.cuda(), mean, std).cpu()
    assert torch.allclose(got, x * std + mean, rtol=1e-6, atol=1e-6)
    ges, exp = split_motion(got, 129)
    assert ges.shape[-1] == 129 and exp.shape[-1] == 103


def test_checkpoint_loader_roundtrip():
    """Reference save format: dict with the weights under 'encoder' (possibly DDP-prefixed)."""
    cfg = get_config("beat")
    sd = synthetic_sd("beat")
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "latest.tar")
        torch.save({"encoder": {"module." + k: v for k, v in list(sd.items())[:40]}, "ep": 7, "total_it": 123, "best_fgd": 1.5}, p)
        got, meta = load_checkpoint(p)
        assert meta == {"ep": 7, "total_it": 123, "best_fgd": 1.5}
        for k in list(sd)[:40]:
            assert torch.equal(got[k], sd[k])
        torch.save({"ep": 1}, p)
        with pytest.raises(KeyError):
            load_checkpoint(p)
