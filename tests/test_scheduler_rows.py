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


def test_factory_dispatch(tmp_path):
    (tmp_path / "scheduler_config.json").write_text(json.dumps({"_class_name": "DPMSolverMultistepScheduler", "solver_order": 2,
                                                                "prediction_type": "v_prediction", "_diffusers_version": "0.33.1"}))
    s = load_scheduler(tmp_path)
    assert isinstance(s, DPMSolverMultistepScheduler) and s.is_multistep and s.config.prediction_type == "v_prediction"
    assert isinstance(DDIMScheduler.from_pretrained(tmp_path), DPMSolverMultistepScheduler)  # the old entry point dispatches too
    (tmp_path / "scheduler_config.json").write_text(json.dumps({"_class_name": "DDIMScheduler"}))
    assert isinstance(load_scheduler(tmp_path), DDIMScheduler) and not load_scheduler(tmp_path).is_multistep
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
