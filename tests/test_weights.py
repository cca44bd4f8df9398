"""CPU: the parameter-shape tables used for random-weight benchmarking (diffuman4d_amd/host/weights.py) list exactly
the keys and shapes of the oracle modules' state_dict (= the diffusers checkpoint key skeleton),
for the plain, temporal-embedding and pose-encoder configurations."""
from dataclasses import asdict

import pytest

from diffuman4d_amd.host.unet import UNetConfig as HostUNetConfig
from diffuman4d_amd.host.weights import unet_param_shapes, vae_param_shapes
from oracle.unet import UNetConfig, UNetMultiviewConditionModel


@pytest.mark.parametrize("kw", [dict(), dict(enable_tem_embeds=True), dict(enable_pose_encoder=True, in_channels=11),
                                dict(use_linear_projection=False)])
def test_unet_shapes_match_oracle_state_dict(kw):
    cfg = UNetConfig.tiny(**kw)
    sd = UNetMultiviewConditionModel(cfg).state_dict()
    shapes = unet_param_shapes(HostUNetConfig.from_dict(asdict(cfg)))
    assert set(shapes) == set(sd), sorted(set(shapes) ^ set(sd))[:10]
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k


def test_vae_shapes_match_oracle_state_dict():
    from diffuman4d_amd.host.vae import VAEConfig as HostVAEConfig
    from oracle.vae import AutoencoderKL, VAEConfig
    cfg = VAEConfig.tiny()
    sd = AutoencoderKL(cfg).state_dict()
    shapes = vae_param_shapes(HostVAEConfig.from_dict(asdict(cfg)))
    assert set(shapes) == set(sd)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
