"""CPU: the HOST side of the three precisions -- weight layouts (K-duplicated matrices, fused QKV, padded channels), operand planes, which
tensor feeds which launch, the planned scheduler rows, the pose encoder -- driven through the REAL host classes (host/unet.py, vae.py,
pipeline.py, scheduler.py) with `tests/cpu_standin_ops.py` standing in for the kernel wrappers (a torch restatement of every wrapper's
documented semantics and rounding points; the kernels themselves are checked on the GPU by tests/opcheck.py).  Compared with the fixtures
the REFERENCE's own pipeline code produced (tests/golden/pipeline_*.pt, pose_encoder.pt): a wiring defect of the parity precision is orders
of magnitude above its 1e-4, and the fast precision has to stay inside its bf16-oracle yardstick.  Nothing here runs product arithmetic
on the CPU: the stand-in is test infrastructure, installed for these tests only and removed afterwards."""
import sys
from dataclasses import asdict
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture()
def cpu_standin(monkeypatch):
    import cpu_standin_ops as fake_ops
    import modelcheck as mc
    import diffuman4d_amd.host.pipeline as hp
    from diffuman4d_amd.host.unet import UNetConfig, UNetMultiviewConditionModel
    from diffuman4d_amd.host.vae import AutoencoderKL, VAEConfig
    fake_ops.install()
    monkeypatch.setattr(mc, "hip_unet", lambda cfg, om, precision="fast": UNetMultiviewConditionModel(UNetConfig.from_dict(asdict(cfg)), om.state_dict(), "cpu", precision))
    monkeypatch.setattr(mc, "hip_vae", lambda cfg, om, precision="fast": AutoencoderKL(VAEConfig.from_dict(asdict(cfg)), om.state_dict(), "cpu", precision))
    real = hp.Diffuman4DPipeline
    monkeypatch.setattr(hp, "Diffuman4DPipeline", lambda v, u, s, dev: real(v, u, s, "cpu"))
    try:
        yield mc
    finally:
        fake_ops.uninstall()


FIXTURES = ["spatial", "temporal_v", "round2_shift", "pose_encoder", "dpm_temporal_v_heun_round2", "unipc_temporal_v_bh1_round2", "deis3_spatial_bidir", "pndm_spatial_bidir",
            "pndm_temporal_v_round2"]


@pytest.mark.parametrize("name", FIXTURES)
def test_parity_precision_host_wiring_reproduces_the_reference_fixture(cpu_standin, name):
    mc = cpu_standin
    err, _ = mc.case_golden_pipeline(name, precision="parity")
    assert "bookkeeping" not in err, "timestep bookkeeping differs from the reference pipeline's"
    assert err["latents"] <= 1e-4 and err["images"] <= 5e-4, err  # images: the fixtures store RGB in fp16 (floor 1.7e-4)


@pytest.mark.parametrize("name", FIXTURES)
def test_fp16_precision_host_wiring_meets_its_fixed_bounds(cpu_standin, name):
    """precision "fp16" (fp32 tensors, one fp16 plane per MFMA operand, fp16 weights) through the same host classes: decoded RGB within
    north_star's 1e-3 of the reference pipeline's fp32 output, latents within 2e-3 (modelcheck.FP16_BOUNDS)."""
    mc = cpu_standin
    err, _ = mc.case_golden_pipeline(name, precision="fp16")
    assert "bookkeeping" not in err, "timestep bookkeeping differs from the reference pipeline's"
    for q in err:
        assert err[q] <= mc.FP16_BOUNDS[q], (q, err[q])


@pytest.mark.parametrize("name", ["spatial", "dpm_temporal_v_heun_round2", "unipc_temporal_v_bh1_round2"])
def test_fast_precision_host_wiring_stays_inside_the_bf16_yardstick(cpu_standin, name):
    mc = cpu_standin
    err, yard = mc.case_golden_pipeline(name)
    assert "bookkeeping" not in err
    for q in err:
        assert err[q] <= mc.YARD_FACTOR * yard[q], (q, err[q], yard[q])


@pytest.mark.parametrize("name,precision,kw", [("spatial", "parity", {}), ("pose_encoder", "parity", dict(copies=2)),
                                               ("dpm_temporal_v_heun_round2", "fp16", dict(copies=2)), ("spatial", "parity", dict(global_rng=True))])
def test_task_stack_host_wiring_returns_what_each_task_returns_alone(cpu_standin, name, precision, kw):
    """runner.task_batch, host side (stacked tensors, copies of the plan tables, frames per attention group, per-task decode, the order
    of the random draws): `sliding_iterative_denoise_stack` against one `sliding_iterative_denoise` per task.  On the GPU the equality is
    bitwise (modelcheck task_stack_*); the CPU stand-in's fp32 matmuls block differently for a taller matrix, hence a bound here."""
    worst, _ = cpu_standin.case_task_stack(name, precision=precision, **kw)
    assert worst <= (2e-5 if precision == "parity" else 2e-3), worst


def test_the_standin_is_removed_again():
    """After the fixture the real wrappers are back: they refuse CPU tensors (no CPU compute path in the product)."""
    import torch
    from diffuman4d_amd.host import lib, ops
    with pytest.raises(lib.Dm4dError):
        ops.silu(torch.zeros(4, 8, dtype=torch.bfloat16))


def test_wide_input_checkpoint_and_config_field_messages(cpu_standin):
    """unet/config.json generality (unet_multiview_condition.py:149-212): a checkpoint with up to 64 input channels loads and runs (conv_in
    padded to two 32-wide K slabs); what this path does not build is refused AT LOAD with a message naming the config.json field."""
    import torch
    from diffuman4d_amd.host import ops
    from diffuman4d_amd.host.unet import UNetConfig, UNetMultiviewConditionModel
    mc = cpu_standin
    cfg, om = mc.make_unet(3, in_channels=40)
    hm = UNetMultiviewConditionModel(UNetConfig.from_dict(asdict(cfg)), om.state_dict(), "cpu", "parity")
    assert hm.IN_PAD == 64
    g = torch.Generator().manual_seed(4)
    x = torch.randn(4, 40, 16, 8, generator=g).to(torch.bfloat16)
    t = torch.randint(0, 1000, (4,), generator=g)
    with torch.no_grad():
        ref = om(x.float(), t, domains=["spatial"] * 2, num_frames=2)
    out = ops.nhwc_to_nchw(hm(ops.split(x.float().permute(0, 2, 3, 1).contiguous(), cpad=hm.IN_PAD), t.float(), domains=["spatial"] * 2, num_frames=2))
    assert float((out - ref).norm() / ref.norm()) < 1e-4
    base = asdict(mc.make_unet(0)[0])
    sd = mc.make_unet(0)[1].state_dict()
    with pytest.raises(NotImplementedError, match="in_channels = 80"):
        UNetMultiviewConditionModel(UNetConfig.from_dict(dict(base, in_channels=80)), sd, "cpu")
    with pytest.raises(NotImplementedError, match="cross_attention_dim = 1024"):
        UNetMultiviewConditionModel(UNetConfig.from_dict(dict(base, cross_attention_dim=1024)), sd, "cpu")
    heads = tuple(2 * h for h in base["attention_head_dim"])  # twice the heads over the same channels: head dimension 32
    with pytest.raises(NotImplementedError, match="attention_head_dim give a head dimension of 32"):
        UNetMultiviewConditionModel(UNetConfig.from_dict(dict(base, attention_head_dim=heads)), sd, "cpu")
    with pytest.raises(ValueError, match="in_channels = 40, but this pipeline assembles 15"):
        import diffuman4d_amd.host.pipeline as hp
        from diffuman4d_amd.host.scheduler import DDIMScheduler
        hp.Diffuman4DPipeline(None, hm, DDIMScheduler(), "cpu")
