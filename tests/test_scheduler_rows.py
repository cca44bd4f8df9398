"""CPU: the DPM-Solver++ coefficient rows of host/scheduler.py against the stateful oracle scheduler (one object per latent, as
the reference keeps them: pipeline_diffuman4d.py:265-271, 420), the scheduler factory, and the history flags of a planned sweep."""
import copy
import itertools
import json

import numpy as np
import pytest
import torch

from diffuman4d_amd.host.schedule import plan_sweep
from diffuman4d_amd.host.scheduler import (DDIMScheduler, DPMSolverConfig, DPMSolverMultistepScheduler, load_scheduler)
from oracle.dpmsolver import DPMSolverConfig as OC, DPMSolverMultistepScheduler as OS

GRID = list(itertools.product((1, 2), ("epsilon", "v_prediction"), ("midpoint", "heun"), ("zero", "sigma_min"),
                              ("linspace", "leading", "trailing")))


@pytest.mark.parametrize("order,pred,solver,final,spacing", GRID)
def test_rows_reproduce_the_stateful_scheduler(order, pred, solver, final, spacing):
    kw = dict(solver_order=order, prediction_type=pred, solver_type=solver, final_sigmas_type=final, timestep_spacing=spacing)
    torch.manual_seed(0)
    for n in (6, 18, 36):
        h, o = DPMSolverMultistepScheduler(DPMSolverConfig(**kw)), OS(OC(**kw))
        ts, to = h.set_timesteps(n), o.set_timesteps(n)
        assert (ts == to.numpy()).all()
        for start in (0, n // 3, n - 2, n - 1):  # a latent enters a call at any step index with a FRESH scheduler copy
            oc = copy.deepcopy(o)
            x = torch.randn(3, 5, dtype=torch.float64)
            p = torch.zeros_like(x)
            xo = x.clone().float()
            for k, i in enumerate(range(start, n)):
                m = torch.randn(3, 5)
                a, b, c, d, e = h.step_rows(np.array([i]), np.array([k > 0]))[0, :5].astype(np.float64)
                x, p = a * x + b * m.double() + c * p, d * x + e * m.double()
                xo = oc.step(m, int(to[i]), xo)
                assert (x - xo.double()).abs().max().item() <= 2e-5 * max(1.0, xo.abs().max().item()), (n, start, i)
    if final == "zero":  # the last step lands exactly on the x0 prediction
        a, b, c, d, e = h.step_rows(np.array([n - 1]), np.array([True]))[0, :5]
        assert abs(a - d) < 1e-6 and abs(b - e) < 1e-6 and c == 0


def _run_general_rows(host, stateful, n_list=(6, 18, 36)):
    """The product's 16-float rows + three stored tensors against one stateful scheduler object, for latents that enter a call at any step."""
    worst = 0.0
    for n in n_list:
        ts, to = host.set_timesteps(n), stateful.set_timesteps(n)
        assert (ts == to.numpy()).all()
        for start in (0, n // 3, n - 2, n - 1):
            oc = copy.deepcopy(stateful)
            x = torch.randn(3, 5, dtype=torch.float64)
            s1, s2, s3 = torch.zeros_like(x), torch.zeros_like(x), torch.zeros_like(x)
            xo = x.clone().float()
            for k, i in enumerate(range(start, n)):
                m = torch.randn(3, 5)
                r = host.step_rows(np.array([i]), np.array([k]))[0].astype(np.float64)
                conv = r[0] * x + r[1] * m.double()
                xc = r[2] * x + r[3] * s3 + r[4] * s1 + r[5] * s2 + r[6] * conv
                xn = r[7] * xc + r[8] * conv + r[9] * s1 + r[10] * s2 + r[11] * s3
                s1, s2, s3 = {0: (conv, s1, xc), 1: (s1, s2, s3), 2: (conv, s1, s2)}[int(r[12])]  # k12: as planned / kept / shifted (PLMS)
                x = xn
                xo = oc.step(m, int(to[i]), xo)
                err = (x - xo.double()).abs().max().item() / max(1.0, xo.abs().max().item())
                assert err <= 5e-5, (n, start, i, err)
                worst = max(worst, err)
    return worst


SD_BETAS = dict(beta_schedule="scaled_linear", beta_start=0.00085, beta_end=0.012)


@pytest.mark.parametrize("order,pred,solver,final,spacing,off", list(itertools.product(
    (1, 2), ("epsilon", "v_prediction"), ("bh1", "bh2"), ("zero", "sigma_min"), ("linspace", "leading", "trailing"), ((), (0, 3)))))
def test_unipc_rows_reproduce_the_stateful_scheduler(order, pred, solver, final, spacing, off):
    from diffuman4d_amd.host.scheduler import UniPCConfig, UniPCMultistepScheduler
    from oracle.multistep import UniPCConfig as OC_, UniPCMultistepScheduler as OS_
    kw = dict(solver_order=order, prediction_type=pred, solver_type=solver, final_sigmas_type=final, timestep_spacing=spacing, **SD_BETAS)
    torch.manual_seed(0)
    _run_general_rows(UniPCMultistepScheduler(UniPCConfig(disable_corrector=off, **kw)), OS_(OC_(disable_corrector=list(off), **kw)))


@pytest.mark.parametrize("order,pred,spacing,lof", list(itertools.product((1, 2, 3), ("epsilon", "v_prediction"),
                                                                           ("linspace", "leading", "trailing"), (True, False))))
def test_deis_rows_reproduce_the_stateful_scheduler(order, pred, spacing, lof):
    from diffuman4d_amd.host.scheduler import DEISConfig, DEISMultistepScheduler
    from oracle.multistep import DEISConfig as OC_, DEISMultistepScheduler as OS_
    kw = dict(solver_order=order, prediction_type=pred, timestep_spacing=spacing, lower_order_final=lof, **SD_BETAS)
    torch.manual_seed(0)
    _run_general_rows(DEISMultistepScheduler(DEISConfig(**kw)), OS_(OC_(**kw)))


@pytest.mark.parametrize("pred,spacing,off,alpha_one", list(itertools.product(("epsilon", "v_prediction"), ("leading", "linspace", "trailing"), (0, 1),
                                                                              (False, True))))
def test_pndm_rows_reproduce_the_stateful_scheduler(pred, spacing, off, alpha_one):
    """PLMS as planned rows: the first step, the repeated second step on the stored sample (wherever in the table the latent stands when a
    call starts), Adams-Bashforth orders 2-4 on a shifted history -- against the stateful object the reference deep-copies per latent."""
    from diffuman4d_amd.host.scheduler import PNDMConfig, PNDMScheduler
    from oracle.multistep import PNDMConfig as OC_, PNDMScheduler as OS_
    kw = dict(skip_prk_steps=True, prediction_type=pred, timestep_spacing=spacing, steps_offset=off, set_alpha_to_one=alpha_one, **SD_BETAS)
    h = PNDMScheduler(PNDMConfig(**kw))
    assert len(h.set_timesteps(18)) == 19 and h.timesteps[1] == h.timesteps[2]  # the repeated second step of the table
    _run_general_rows(h, OS_(OC_(**kw)))


def test_pndm_without_skip_prk_steps_is_refused():
    from diffuman4d_amd.host.scheduler import PNDMConfig, PNDMScheduler
    with pytest.raises(NotImplementedError, match="skip_prk_steps"):
        PNDMScheduler(PNDMConfig(skip_prk_steps=False))


def test_oracle_multistep_schedulers_converge_on_a_solvable_ode():
    """The stateful oracle schedulers (restated from the papers / diffusers 0.33.1, unpinned) integrate the probability-flow ODE of a
    Gaussian data distribution, whose solution is known in closed form: every one converges, and order 2 beats order 1 by a wide margin
    (a wrong second-order coefficient would leave a first-order error)."""
    from oracle.multistep import DEISConfig, DEISMultistepScheduler, UniPCConfig, UniPCMultistepScheduler, _alpha_sigma
    s2 = 0.49

    def run(sch, n):
        ts = sch.set_timesteps(n)
        sig = sch.sigmas.double()
        a0, s0 = _alpha_sigma(sig[0])
        z = torch.tensor([1.3, -0.7, 0.2], dtype=torch.float64)
        x = (z * float((a0 ** 2 * s2 + s0 ** 2) ** 0.5)).float()
        for i, t in enumerate(ts):
            a, s = _alpha_sigma(sig[i])
            x = sch.step((s * x.double() / (a ** 2 * s2 + s ** 2)).float(), int(t), x)  # exact noise prediction for N(0, s2) data
        aL, sL = _alpha_sigma(sig[-1])
        return float((x.double() - z * float((aL ** 2 * s2 + sL ** 2) ** 0.5)).abs().max())
    e = {name: [run(mk(), n) for n in (20, 40)] for name, mk in (
        ("unipc1", lambda: UniPCMultistepScheduler(UniPCConfig(solver_order=1, final_sigmas_type="sigma_min"))),
        ("unipc2", lambda: UniPCMultistepScheduler(UniPCConfig(solver_order=2, final_sigmas_type="sigma_min"))),
        ("deis1", lambda: DEISMultistepScheduler(DEISConfig(solver_order=1))),
        ("deis2", lambda: DEISMultistepScheduler(DEISConfig(solver_order=2))),
        ("deis3", lambda: DEISMultistepScheduler(DEISConfig(solver_order=3))))}
    assert all(v[1] < v[0] * 1.05 for v in e.values()), e                      # more steps never hurt
    assert e["unipc2"][0] < 0.3 * e["unipc1"][0] and e["deis2"][0] < 0.4 * e["deis1"][0] and e["deis3"][0] < 0.5 * e["deis2"][0], e
    assert 1.7 < e["deis1"][0] / e["deis1"][1] < 2.3, e                         # first order: the error halves with the step


def test_history_counts_follow_the_plan():
    from diffuman4d_amd.host.schedule import history_counts, history_flags
    cond = np.array([True, False, False, True, False, False, False, False])
    plan = plan_sweep(cond, np.zeros(8, dtype=np.int64), "spatial", 4, 2, 0, True, 2, 1)
    counts, flags = history_counts(plan.windows, plan.is_cond), history_flags(plan.windows, plan.is_cond)
    seen = {}
    for w, c, n, f in zip(plan.windows, plan.is_cond, counts, flags):
        for k, (i, ic) in enumerate(zip(w, c)):
            assert n[k] == (0 if ic else seen.get(int(i), 0)) and bool(f[k]) == (n[k] > 0)
        for i, ic in zip(w, c):
            if not ic:
                seen[int(i)] = seen.get(int(i), 0) + 1
    assert max(int(n.max()) for n in counts) == 7  # 8 steps per latent in this call


def test_factory_dispatch(tmp_path):
    (tmp_path / "scheduler_config.json").write_text(json.dumps({"_class_name": "DPMSolverMultistepScheduler", "solver_order": 2,
                                                                "prediction_type": "v_prediction", "_diffusers_version": "0.33.1"}))
    s = load_scheduler(tmp_path)
    assert isinstance(s, DPMSolverMultistepScheduler) and s.is_multistep and s.config.prediction_type == "v_prediction"
    assert isinstance(DDIMScheduler.from_pretrained(tmp_path), DPMSolverMultistepScheduler)  # the old entry point dispatches too
    (tmp_path / "scheduler_config.json").write_text(json.dumps({"_class_name": "DDIMScheduler"}))
    assert isinstance(load_scheduler(tmp_path), DDIMScheduler) and not load_scheduler(tmp_path).is_multistep
    from diffuman4d_amd.host.scheduler import DEISMultistepScheduler, UniPCMultistepScheduler
    (tmp_path / "scheduler_config.json").write_text(json.dumps({"_class_name": "UniPCMultistepScheduler", "solver_order": 2, "disable_corrector": [0],
                                                                "_diffusers_version": "0.33.1"}))
    u = load_scheduler(tmp_path)
    assert isinstance(u, UniPCMultistepScheduler) and u.general_rows and u.state_slots == 3 and u.config.disable_corrector == (0,)
    (tmp_path / "scheduler_config.json").write_text(json.dumps({"_class_name": "DEISMultistepScheduler", "solver_order": 3}))
    assert isinstance(load_scheduler(tmp_path), DEISMultistepScheduler)
    from diffuman4d_amd.host.scheduler import PNDMScheduler
    # the Stable Diffusion family's stock scheduler/scheduler_config.json
    (tmp_path / "scheduler_config.json").write_text(json.dumps({"_class_name": "PNDMScheduler", "_diffusers_version": "0.8.0", "beta_end": 0.012,
                                                                "beta_schedule": "scaled_linear", "beta_start": 0.00085, "num_train_timesteps": 1000,
                                                                "set_alpha_to_one": False, "skip_prk_steps": True, "steps_offset": 1,
                                                                "trained_betas": None, "clip_sample": False}))
    pn = load_scheduler(tmp_path)
    assert isinstance(pn, PNDMScheduler) and pn.general_rows and pn.state_slots == 3 and pn.config.steps_offset == 1
    for bad in ({"_class_name": "UniPCMultistepScheduler", "solver_order": 3}, {"_class_name": "UniPCMultistepScheduler", "predict_x0": False},
                {"_class_name": "DEISMultistepScheduler", "use_karras_sigmas": True}):
        (tmp_path / "scheduler_config.json").write_text(json.dumps(bad))
        with pytest.raises(NotImplementedError):
            load_scheduler(tmp_path)
    for name in ("EulerDiscreteScheduler", "PNDMScheduler", "DDPMScheduler"):
        (tmp_path / "scheduler_config.json").write_text(json.dumps({"_class_name": name}))
        with pytest.raises(NotImplementedError, match=name):
            load_scheduler(tmp_path)
    (tmp_path / "scheduler_config.json").write_text(json.dumps({"_class_name": "DPMSolverMultistepScheduler", "use_karras_sigmas": True}))
    with pytest.raises(NotImplementedError, match="use_karras_sigmas"):
        load_scheduler(tmp_path)


def test_history_flags_follow_the_plan():
    """A latent has a previous prediction from its second step of a call on, whichever windows those steps fall into."""
    from diffuman4d_amd.host.schedule import history_flags
    cond = np.array([True, False, False, True, False, False, False, False])
    plan = plan_sweep(cond, np.zeros(8, dtype=np.int64), "spatial", 4, 2, 0, True, 2, 1)
    flags = history_flags(plan.windows, plan.is_cond)
    seen = set()
    for w, c, f in zip(plan.windows, plan.is_cond, flags):
        for k, (i, ic) in enumerate(zip(w, c)):
            assert f[k] == ((not ic) and int(i) in seen)
        seen.update(int(i) for i, ic in zip(w, c) if not ic)
    assert flags[0].sum() == 0 and any(f.any() for f in flags)


def test_dpm_config_keys_that_change_the_sigma_table_are_not_dropped():
    """scheduler_config.json keys that alter the sigma table must raise, not be ignored (host/scheduler.py policy)."""
    import pytest
    from diffuman4d_amd.host.scheduler import DPMSolverConfig, DPMSolverMultistepScheduler
    DPMSolverMultistepScheduler(DPMSolverConfig.from_dict({"rescale_betas_zero_snr": False, "trained_betas": None, "lambda_min_clipped": -float("inf")}))
    for bad in ({"rescale_betas_zero_snr": True}, {"trained_betas": [0.1, 0.2]}, {"lambda_min_clipped": -5.1}):
        with pytest.raises(NotImplementedError):
            DPMSolverMultistepScheduler(DPMSolverConfig.from_dict(bad))
