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


# ---- stateful scheduler (DPM-Solver++): fixture = the REFERENCE's pipeline run with one deep copy of the scheduler per latent ----
DPM_CASES = ["dpm_spatial_bidir", "dpm_temporal_v_heun_round2"]


def _dpm_task(name):
    g = torch.load(G / "pipeline_dpm.pt")[name]
    c, seeds = g["case"], g["seeds"]
    _, ou = mc.make_unet(seeds["unet"])
    _, ov = mc.make_vae(seeds["vae"])
    pv, pl, sk, cm = mc.synthetic_task(c["n"], 64, 64, c["inputs"], seeds["task"])
    return g, c, ou, ov, (pv, pl, sk, cm)


@pytest.mark.parametrize("name", DPM_CASES)
def test_oracle_pipeline_with_stateful_scheduler_matches_reference_pipeline(name):
    from oracle.dpmsolver import DPMSolverConfig, DPMSolverMultistepScheduler
    g, c, ou, ov, (pv, pl, sk, cm) = _dpm_task(name)
    op = OraclePipeline(ov, ou, DPMSolverMultistepScheduler(DPMSolverConfig(**c["sched"])), torch.float32)
    res = op.sliding_iterative_denoise(pv, pl, sk, cm, g["latents_in"], c["domain"], g["timestep_indices_in"], g["noise"], **c["kw"])
    assert torch.equal(res["timestep_indices"], g["timestep_indices"])
    assert rel(res["latents"], g["latents"]) <= 1e-5
    assert rel(res["images"], g["images"]) <= 1e-3


class _PlannedRows:
    """What the PRODUCT does with a multistep scheduler, on the CPU: no scheduler object per latent, only the host's coefficient
    rows (host/scheduler.py::step_rows) and one stored x0 prediction per latent.  Deep-copied per latent by the oracle pipeline
    exactly like a real scheduler; records whether each step had a previous prediction."""
    init_noise_sigma = 1.0

    def __init__(self, host_sched, log):
        self.h, self.log, self.p, self.timesteps = host_sched, log, None, None

    def __deepcopy__(self, memo):
        c = _PlannedRows(self.h, self.log)  # the row tables and the log are shared, the state is per latent
        c.timesteps = self.timesteps
        return c

    def set_timesteps(self, n):
        self.timesteps = torch.from_numpy(self.h.set_timesteps(n))
        return self.timesteps

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, t, sample):
        i = int((self.timesteps == int(t)).nonzero()[0])
        has_prev = self.p is not None
        self.log.append(has_prev)
        a, b, cc, d, e = (float(v) for v in self.h.step_rows([i], [has_prev])[0, :5])
        p = self.p if has_prev else torch.zeros_like(sample)
        self.p = d * sample + e * model_output
        return a * sample + b * model_output + cc * p


@pytest.mark.parametrize("name", DPM_CASES)
def test_planned_rows_reproduce_the_reference_pipeline_with_a_stateful_scheduler(name):
    """The reference pipeline's output with one stateful scheduler object per latent == a pipeline that only has the product's
    planned rows + one stored prediction per latent; and the plan's history flags are the ones observed."""
    from diffuman4d_amd.host.schedule import history_flags
    from diffuman4d_amd.host.scheduler import DPMSolverConfig as HC, DPMSolverMultistepScheduler as HS
    g, c, ou, ov, (pv, pl, sk, cm) = _dpm_task(name)
    log = []
    op = OraclePipeline(ov, ou, _PlannedRows(HS(HC(**c["sched"])), log), torch.float32)
    res = op.sliding_iterative_denoise(pv, pl, sk, cm, g["latents_in"], c["domain"], g["timestep_indices_in"], g["noise"], decode=False, **c["kw"])
    assert rel(res["latents"], g["latents"]) <= 2e-5
    k = c["kw"]
    cond = [i in c["inputs"] for i in range(c["n"])]
    plan = plan_sweep(cond, g["timestep_indices_in"].tolist(), c["domain"], k["window_size"], k["sliding_stride"],
                      k["sliding_shift"], k["bidirectional"], k["num_denoising_steps"], k["alternation_rounds"])
    flags = history_flags(plan.windows, plan.is_cond)
    planned = [bool(f) for fl, ic in zip(flags, plan.is_cond) for f, is_c in zip(fl, ic) if not is_c]
    assert planned == log and any(log) and not all(log)


# ---- UniPC / DEIS: fixture = the REFERENCE's pipeline run with one stateful scheduler object per latent (make_golden.py multistep) ----
MULTISTEP_CASES = ["unipc_spatial_bidir", "unipc_temporal_v_bh1_round2", "deis3_spatial_bidir", "deis2_temporal_v_round2",
                   "pndm_spatial_bidir", "pndm_temporal_v_round2"]  # PNDM with skip_prk_steps (PLMS): round 6


def _multistep_task(name):
    g = torch.load(G / "pipeline_multistep.pt")[name]
    c, seeds = g["case"], g["seeds"]
    _, ou = mc.make_unet(seeds["unet"])
    _, ov = mc.make_vae(seeds["vae"])
    return g, c, ou, ov, mc.synthetic_task(c["n"], 64, 64, c["inputs"], seeds["task"])


def _oracle_multistep(c):
    from oracle import multistep as ms
    if c["kind"] == "pndm":
        return ms.PNDMScheduler(ms.PNDMConfig(**c["sched"]))
    return ms.UniPCMultistepScheduler(ms.UniPCConfig(**c["sched"])) if c["kind"] == "unipc" else ms.DEISMultistepScheduler(ms.DEISConfig(**c["sched"]))


def _host_multistep(c):
    from diffuman4d_amd.host import scheduler as hs
    if c["kind"] == "pndm":
        return hs.PNDMScheduler(hs.PNDMConfig.from_dict(c["sched"]))
    return (hs.UniPCMultistepScheduler(hs.UniPCConfig.from_dict(c["sched"])) if c["kind"] == "unipc"
            else hs.DEISMultistepScheduler(hs.DEISConfig.from_dict(c["sched"])))


@pytest.mark.parametrize("name", MULTISTEP_CASES)
def test_oracle_pipeline_with_stateful_unipc_deis_matches_reference_pipeline(name):
    g, c, ou, ov, (pv, pl, sk, cm) = _multistep_task(name)
    op = OraclePipeline(ov, ou, _oracle_multistep(c), torch.float32)
    res = op.sliding_iterative_denoise(pv, pl, sk, cm, g["latents_in"], c["domain"], g["timestep_indices_in"], g["noise"], **c["kw"])
    assert torch.equal(res["timestep_indices"], g["timestep_indices"])
    assert rel(res["latents"], g["latents"]) <= 1e-5
    assert rel(res["images"], g["images"]) <= 1e-3


class _PlannedRows16:
    """What the PRODUCT does with UniPC / DEIS, on the CPU: no scheduler object per latent, only the host's 16-float rows
    (host/scheduler.py::step_rows) and up to three stored tensors per latent; records how many steps each latent had taken."""
    init_noise_sigma = 1.0

    def __init__(self, host_sched, log):
        self.h, self.log, self.s, self.k, self.timesteps = host_sched, log, None, 0, None

    def __deepcopy__(self, memo):
        c = _PlannedRows16(self.h, self.log)
        c.timesteps = self.timesteps
        return c

    def set_timesteps(self, n):
        self.timesteps = torch.from_numpy(self.h.set_timesteps(n))
        return self.timesteps

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, t, sample):
        i = int((self.timesteps == int(t)).nonzero()[0])
        self.log.append(self.k)
        r = [float(v) for v in self.h.step_rows([i], [self.k])[0]]
        s1, s2, s3 = self.s if self.s is not None else (torch.zeros_like(sample),) * 3
        conv = r[0] * sample + r[1] * model_output
        xc = r[2] * sample + r[3] * s3 + r[4] * s1 + r[5] * s2 + r[6] * conv
        out = r[7] * xc + r[8] * conv + r[9] * s1 + r[10] * s2 + r[11] * s3
        self.s = {0: (conv, s1, xc), 1: (s1, s2, s3), 2: (conv, s1, s2)}[int(r[12])]  # k12: as planned / kept / shifted (PLMS)
        self.k += 1
        return out


@pytest.mark.parametrize("name", MULTISTEP_CASES)
def test_planned_16_float_rows_reproduce_the_reference_pipeline_with_stateful_unipc_deis(name):
    from diffuman4d_amd.host.schedule import history_counts
    g, c, ou, ov, (pv, pl, sk, cm) = _multistep_task(name)
    log = []
    op = OraclePipeline(ov, ou, _PlannedRows16(_host_multistep(c), log), torch.float32)
    res = op.sliding_iterative_denoise(pv, pl, sk, cm, g["latents_in"], c["domain"], g["timestep_indices_in"], g["noise"], decode=False, **c["kw"])
    assert rel(res["latents"], g["latents"]) <= 5e-5
    k = c["kw"]
    cond = [i in c["inputs"] for i in range(c["n"])]
    plan = plan_sweep(cond, g["timestep_indices_in"].tolist(), c["domain"], k["window_size"], k["sliding_stride"],
                      k["sliding_shift"], k["bidirectional"], k["num_denoising_steps"], k["alternation_rounds"])
    counts = history_counts(plan.windows, plan.is_cond)
    planned = [int(n) for cn, ic in zip(counts, plan.is_cond) for n, is_c in zip(cn, ic) if not is_c]
    assert planned == log and max(log) >= 1
