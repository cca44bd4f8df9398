"""CPU: the oracle and the host-side bookkeeping against golden vectors produced by the REFERENCE's own
modules (tests/golden/make_golden.py, run in the build container where /root/reference is mounted).
Nothing here reads /root/reference."""
from pathlib import Path

import pytest
import torch

import modelcheck as mc
from diffuman4d_amd.host.dataset import SyntheticSpaTemDataset
from diffuman4d_amd.host.sampler import SlidingIterativeSampler
from diffuman4d_amd.host.schedule import plan_sweep
from oracle.ddim import DDIMConfig, DDIMScheduler
from oracle.pipeline import OraclePipeline
from stubs import StubPipeline

G = Path(__file__).resolve().parent / "golden"


def rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


@pytest.mark.parametrize("name", ["spatial", "temporal_temb", "conv_proj", "2d_only"])
def test_oracle_unet_matches_reference_unet(name):
    g = torch.load(G / "unet_forward.pt")[name]
    cfg, om = mc.make_unet(g["seed"], **g["cfg_kw"])
    if g["cfg_kw"].get("enable_tem_embeds"):
        gen = torch.Generator().manual_seed(8)
        with torch.no_grad():
            for p in om.temporal_pos_embed.parameters():
                p.copy_(torch.randn(p.shape, generator=gen) * 0.05)
    with torch.no_grad():
        y = om(g["x"], g["t"], domains=[g["domain"]] * 2, num_frames=g["num_frames"])
    assert rel(y, g["y"]) <= 1e-6


def test_oracle_pose_encoder_unet_matches_reference():
    g = torch.load(G / "pose_encoder.pt")["unet"]
    cfg, om = mc.make_unet(g["seed"], **g["cfg_kw"])
    nf = g["num_frames"]
    gen = torch.Generator().manual_seed(g["data_seed"])
    x = torch.randn(2 * nf, cfg.in_channels, 16, 8, generator=gen)
    t = torch.randint(0, 1000, (2 * nf,), generator=gen)
    sk = torch.rand(2 * nf, 3, 128, 64, generator=gen) * 2 - 1
    with torch.no_grad():
        y = om(x, t, skeletons=sk, domains=["spatial"] * 2, num_frames=nf)
    assert rel(y, g["y"]) <= 1e-6


def test_oracle_pose_encoder_pipeline_matches_reference():
    g = torch.load(G / "pose_encoder.pt")["pipeline"]
    c, seeds = g["case"], g["seeds"]
    _, ou = mc.make_unet(seeds["unet"], **g["cfg_kw"])
    _, ov = mc.make_vae(seeds["vae"])
    pv, pl, sk, cm = mc.synthetic_task(c["n"], 64, 64, c["inputs"], seeds["task"])
    op = OraclePipeline(ov, ou, DDIMScheduler(DDIMConfig(prediction_type=c["pred"])), torch.float32)
    res = op.sliding_iterative_denoise(pv, pl, sk, cm, None, c["domain"], g["timestep_indices_in"], g["noise"], **c["kw"])
    assert torch.equal(res["timestep_indices"], g["timestep_indices"])
    assert torch.equal(res["fully_denoised"], g["fully_denoised"])
    assert rel(res["latents"], g["latents"]) <= 1e-5
    assert rel(res["images"], g["images"]) <= 1e-3


@pytest.mark.parametrize("name", ["spatial", "temporal_v", "bidir_nocfg", "round2_shift"])
def test_oracle_pipeline_matches_reference_pipeline(name):
    g = torch.load(G / "pipeline_sliding.pt")[name]
    c, seeds = g["case"], g["seeds"]
    _, ou = mc.make_unet(seeds["unet"])
    _, ov = mc.make_vae(seeds["vae"])
    pv, pl, sk, cm = mc.synthetic_task(c["n"], 64, 64, c["inputs"], seeds["task"])
    op = OraclePipeline(ov, ou, DDIMScheduler(DDIMConfig(prediction_type=c["pred"])), torch.float32)
    res = op.sliding_iterative_denoise(pv, pl, sk, cm, g["latents_in"], c["domain"], g["timestep_indices_in"], g["noise"],
                                       **c["kw"])
    assert torch.equal(res["timestep_indices"], g["timestep_indices"])  # bit-exact bookkeeping
    assert torch.equal(res["fully_denoised"], g["fully_denoised"])
    assert rel(res["latents"], g["latents"]) <= 1e-5
    assert rel(res["images"], g["images"]) <= 1e-3  # fixture images are stored in fp16


@pytest.mark.parametrize("name", ["spatial", "temporal_v", "bidir_nocfg", "round2_shift"])
def test_host_planner_matches_reference_bookkeeping(name):
    g = torch.load(G / "pipeline_sliding.pt")[name]
    c = g["case"]
    cond = [i in c["inputs"] for i in range(c["n"])]
    k = c["kw"]
    plan = plan_sweep(cond, g["timestep_indices_in"].tolist(), c["domain"], k["window_size"], k["sliding_stride"],
                      k["sliding_shift"], k["bidirectional"], k["num_denoising_steps"], k["alternation_rounds"])
    assert plan.final_timestep_indices.tolist() == g["timestep_indices"].tolist()
    assert (torch.from_numpy(plan.final_timestep_indices) == plan.num_inference_steps).tolist() == g["fully_denoised"].tolist()


def _product_sampler(kw):
    ds = SyntheticSpaTemDataset(height=16, width=16, num_cameras=48)
    pipe = StubPipeline()
    return SlidingIterativeSampler(ds, [pipe], "/tmp/unused", result_writer=lambda *a, **k: None, **kw), pipe


@pytest.mark.parametrize("name", ["tiny", "demo_4d_tiny"])
def test_sampler_task_lists_match_reference_sampler(name):
    g = torch.load(G / "sampler_bookkeeping.pt")[name]
    s, _ = _product_sampler(g["kw"])
    assert s.all_tasks == g["all_tasks"]
    assert (s.spa_labels, s.tem_labels, s.target_spa_labels) == (g["spa_labels"], g["tem_labels"], g["target_spa_labels"])


def test_sampler_grid_bookkeeping_matches_reference_sampler():
    g = torch.load(G / "sampler_bookkeeping.pt")["tiny"]
    s, pipe = _product_sampler(g["kw"])
    for tasks in s.all_tasks:
        for t in tasks:
            s.execute_one_task(t)
    assert pipe.calls == g["calls"]  # same task order, same cond rows, same latents-None decisions
    assert {c: dict(v) for c, v in s.timestep_indices.items()} == g["final_idx"]
    assert {c: {f: float(l.flatten()[0]) for f, l in v.items()} for c, v in s.latents.items()} == g["final_lat0"]
