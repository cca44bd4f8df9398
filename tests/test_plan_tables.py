"""CPU: the device tables a planned sweep is turned into (Diffuman4DPipeline.upload_plan) -- plain, stacked for task batching,
and with a multistep scheduler.  No kernel is called: the tables are built on the host and `uploaded` to a CPU device."""
import numpy as np
import pytest
import torch

from diffuman4d_amd.host.pipeline import Diffuman4DPipeline
from diffuman4d_amd.host.schedule import history_flags, plan_sweep
from diffuman4d_amd.host.scheduler import DDIMScheduler, DPMSolverMultistepScheduler


def _plan():
    cond = [i in (1, 5) for i in range(8)]
    return cond, plan_sweep(cond, [0] * 8, "spatial", 4, 2, 0, True, 2, 1)


def test_plain_tables():
    cond, plan = _plan()
    pipe = Diffuman4DPipeline(None, None, DDIMScheduler(), "cpu")
    tb = pipe.upload_plan(plan, 2.0)
    F = len(plan.windows[0])
    assert tb["calls"] == len(plan.windows) and tb["cfg"] == 2 and tb["frames_per_group"] == F and tb["copies"] == 1
    assert tb["win"].shape == (tb["calls"], F) and tb["t"].shape == (tb["calls"], 2 * F) and tb["coef"].shape == (tb["calls"], F, 4)
    ts = pipe.scheduler.timesteps
    for i in range(tb["calls"]):
        assert tb["win"][i].tolist() == plan.windows[i].tolist()
        want_t = np.where(plan.is_cond[i], 0, ts[plan.timestep_index[i]])  # conditioning rows run at t = 0 (:277)
        assert tb["t"][i, :F].tolist() == want_t.tolist() == tb["t"][i, F:].tolist()
        consumed = np.nonzero(~plan.is_cond[i])[0]
        assert tb["keep"][i].tolist() == consumed.tolist() + (consumed + F).tolist()
    assert pipe.upload_plan(plan, 1.0)["cfg"] == 1


def test_stacked_tables_repeat_the_plan_with_row_offsets():
    cond, plan = _plan()
    pipe = Diffuman4DPipeline(None, None, DDIMScheduler(), "cpu")
    one, three = pipe.upload_plan(plan, 2.0), pipe.upload_plan(plan, 2.0, copies=3, rows_per_task=8)
    F = one["win"].shape[1]
    assert three["frames_per_group"] == F and three["copies"] == 3 and three["win"].shape == (one["calls"], 3 * F)
    for k in range(3):
        assert torch.equal(three["win"][:, k * F:(k + 1) * F], one["win"] + 8 * k)
        assert torch.equal(three["cond"][:, k * F:(k + 1) * F], one["cond"])
        assert torch.equal(three["coef"][:, k * F:(k + 1) * F], one["coef"])
    # the CFG batch is [negative: task 0, 1, 2 | positive: task 0, 1, 2], F frames per attention group
    assert torch.equal(three["t"][:, :3 * F], three["t"][:, 3 * F:]) and torch.equal(three["t"][:, :F], one["t"][:, :F])
    for i in range(one["calls"]):
        consumed = np.nonzero(three["cond"][i].numpy() == 0)[0]
        assert three["keep"][i].tolist() == consumed.tolist() + (consumed + 3 * F).tolist()
    with pytest.raises(ValueError):
        pipe.upload_plan(plan, 2.0, copies=2)  # rows_per_task missing


def test_multistep_tables_carry_history_rows():
    cond, plan = _plan()
    sched = DPMSolverMultistepScheduler()
    pipe = Diffuman4DPipeline(None, None, sched, "cpu")
    tb = pipe.upload_plan(plan, 2.0)
    assert tb["coef"].shape[-1] == 8
    flags = history_flags(plan.windows, plan.is_cond)
    for i in range(tb["calls"]):
        for f in range(tb["win"].shape[1]):
            if plan.is_cond[i][f]:
                continue
            row = sched.step_rows(np.array([plan.timestep_index[i][f]]), np.array([flags[i][f]]))[0]
            assert np.allclose(tb["coef"][i, f].numpy(), row)
            assert (row[2] != 0) == (bool(flags[i][f]) and plan.timestep_index[i][f] != plan.num_inference_steps - 1)
