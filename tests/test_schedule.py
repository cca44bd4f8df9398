"""CPU: bit-exact integer bookkeeping of the sliding iterative denoiser.

Known-answer vectors are the ones derived in SURVEY.md section 4 by replaying
``pipeline_diffuman4d.py:503-518,542`` plus the reference's own runtime self-checks (:463-467,
:480-487, :546-551; sliding_iterative_sampler.py:71-88).  The host planner (product) and the oracle's
window builder are both checked against them and against each other.
"""
import numpy as np
import pytest
import torch

from diffuman4d_amd.host.schedule import build_windows, plan_sweep, steps_per_alternation
from diffuman4d_amd.host.scheduler import DDIMConfig, DDIMScheduler
from oracle import pipeline as opipe
from oracle.ddim import DDIMScheduler as OracleDDIM

INPUTS = [1, 13, 25, 37]
TARGETS = [i for i in range(48) if i not in INPUTS]


def test_spatial_windows_kat():
    tws, iws = build_windows(TARGETS, INPUTS, "spatial", 12, 2, 0, False)
    assert len(tws) == 22
    assert tws[0].tolist() == [0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]
    assert tws[1].tolist() == [3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 15]
    assert tws[2].tolist() == [5, 6, 7, 8, 9, 10, 11, 12, 14, 15, 16, 17]
    assert tws[20].tolist() == [44, 45, 46, 47, 0, 2, 3, 4, 5, 6, 7, 8]
    assert tws[21].tolist() == [46, 47, 0, 2, 3, 4, 5, 6, 7, 8, 9, 10]
    assert all(iw.tolist() == INPUTS for iw in iws)


def test_spatial_timestep_trajectory_kat():
    cond = [i in INPUTS for i in range(48)]
    plan = plan_sweep(cond, [0] * 48, "spatial", 12, 2, 0, False, 1, 3)
    assert plan.num_inference_steps == 18
    assert len(plan.windows) == 22
    assert plan.windows[0].tolist() == INPUTS + [0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]  # inputs first, then targets
    assert plan.timestep_index[0].tolist() == [0] * 16
    assert plan.timestep_index[1].tolist() == [0, 0, 0, 0] + [1] * 10 + [0, 0]
    assert plan.timestep_index[2].tolist() == [0, 0, 0, 0] + [2] * 8 + [1, 1, 0, 0]
    assert plan.timestep_index[20].tolist() == [0, 0, 0, 0, 5, 5] + [4] * 10
    assert plan.timestep_index[21].tolist() == [0, 0, 0, 0] + [5] * 12
    assert plan.final_timestep_indices[TARGETS].tolist() == [6] * 44
    assert plan.final_timestep_indices[INPUTS].tolist() == [0] * 4
    assert max(len(set(t.tolist())) for t in plan.timestep_index) <= 6  # a call mixes several noise levels


@pytest.mark.parametrize("T,nwin,last_in,last_tg", [
    (16, 8, [14, 15, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9], [30, 31, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25]),
    (150, 75, None, [298, 299] + list(range(150, 160))),
])
def test_temporal_windows_kat(T, nwin, last_in, last_tg):
    tws, iws = build_windows(list(range(T, 2 * T)), list(range(T)), "temporal", 12, 2, 0, False)
    assert len(tws) == nwin
    assert iws[0].tolist() == list(range(12)) and tws[0].tolist() == list(range(T, T + 12))
    assert tws[-1].tolist() == last_tg
    if last_in is not None:
        assert iws[-1].tolist() == last_in


def test_ddim_timesteps_kat():
    for sched in (DDIMScheduler(DDIMConfig()), OracleDDIM()):
        assert list(map(int, sched.set_timesteps(18))) == [936, 881, 826, 771, 716, 661, 606, 551, 496, 441, 386, 331,
                                                           276, 221, 166, 111, 56, 1]
        assert list(map(int, sched.set_timesteps(12))) == [914, 831, 748, 665, 582, 499, 416, 333, 250, 167, 84, 1]
        ts36 = list(map(int, sched.set_timesteps(36)))
        assert ts36[:2] == [946, 919] and ts36[-2:] == [28, 1] and len(ts36) == 36


def test_ddim_tables_match_oracle_bitwise():
    h, o = DDIMScheduler(DDIMConfig()), OracleDDIM()
    assert np.array_equal(h.alphas_cumprod, o.alphas_cumprod.numpy())
    h.set_timesteps(18), o.set_timesteps(18)
    for t in (936, 496, 56, 1):
        a_t, a_p = o.coefficients(t)
        c = h.step_coefficients(np.array([t]))[0]
        exp = np.array([float(a_t) ** 0.5, (1 - float(a_t)) ** 0.5, float(a_p) ** 0.5, (1 - float(a_p)) ** 0.5], dtype=np.float32)
        assert np.allclose(c, exp, rtol=1e-6, atol=0)
    # t=1 steps to prev_t < 0 => final_alpha_cumprod = alphas_cumprod[0] (set_alpha_to_one=False)
    assert h.step_coefficients(np.array([1]))[0][2] == np.sqrt(h.alphas_cumprod[0])


def test_ddim_linspace_and_zero_terminal_snr_match_oracle():
    """The other DDIM configurations a checkpoint may name: "linspace" spacing and zero-terminal-SNR betas (v-prediction
    fine-tunes), host tables against the oracle's; the last training step then has alpha_cumprod = 0 exactly."""
    from oracle.ddim import DDIMConfig as OC
    kw = dict(timestep_spacing="linspace", rescale_betas_zero_snr=True, prediction_type="v_prediction")
    h, o = DDIMScheduler(DDIMConfig(**kw)), OracleDDIM(OC(**kw))
    assert np.array_equal(h.alphas_cumprod, o.alphas_cumprod.numpy()) and h.alphas_cumprod[-1] == 0.0
    ts = h.set_timesteps(12)
    assert list(map(int, ts)) == list(map(int, o.set_timesteps(12))) and int(ts[0]) == 999 and int(ts[-1]) == 0
    c = h.step_coefficients(np.array([999]))[0]
    assert c[0] == 0.0 and c[1] == 1.0  # pure noise at the first step: x0 = -v (v-prediction), no division by sqrt(alpha)
    x, v = torch.randn(3, 4), torch.randn(3, 4)
    sa, sb, sap, sbp = (float(k) for k in c)
    mine = sap * (sa * x - sb * v) + sbp * (sa * v + sb * x)
    assert torch.allclose(mine, o.step(v, 999, x), atol=1e-6)
    import json, tempfile, pathlib
    from diffuman4d_amd.host.scheduler import load_scheduler
    with tempfile.TemporaryDirectory() as d:
        (pathlib.Path(d) / "scheduler_config.json").write_text(json.dumps({"_class_name": "DDIMScheduler", "clip_sample": True}))
        with pytest.raises(NotImplementedError, match="clip_sample"):
            load_scheduler(d)


@pytest.mark.parametrize("domain,n_t,n_i", [("spatial", 44, 4), ("temporal", 16, 16), ("temporal", 150, 150)])
@pytest.mark.parametrize("window,stride,shift,bidir,steps", [(12, 1, 0, False, 1), (12, 2, 0, False, 1), (4, 1, 0, False, 1),
                                                             (12, 2, 0, True, 1), (12, 3, 1, False, 2), (8, 4, 2, True, 2)])
def test_planner_matches_oracle_windows(domain, n_t, n_i, window, stride, shift, bidir, steps):
    if domain == "spatial":
        inputs = [1, 13, 25, 37]
        targets = [i for i in range(48) if i not in inputs]
    else:
        inputs, targets = list(range(n_i)), list(range(n_i, 2 * n_i))
    if n_t % stride or (window * steps) % stride:
        pytest.skip("combination rejected by the sampler/pipeline validation")
    h_t, h_i = build_windows(targets, inputs, domain, window, stride, shift, bidir)
    o_t, o_i = opipe.build_windows(torch.tensor(targets), torch.tensor(inputs), domain, window, stride, shift, bidir)
    assert [w.tolist() for w in h_t] == [w.tolist() for w in o_t]
    assert [w.tolist() for w in h_i] == [w.tolist() for w in o_i]
    # the reference's exit check (:546-551): every target advanced by exactly steps_per_alternation
    n = len(inputs) + len(targets)
    cond = [i in set(inputs) for i in range(n)]
    plan = plan_sweep(cond, [0] * n, domain, window, stride, shift, bidir, steps, 3)
    per = steps_per_alternation(window, stride, bidir, steps)
    assert set(plan.final_timestep_indices[targets].tolist()) == {per}
    assert set(plan.final_timestep_indices[inputs].tolist()) == {0}
    assert plan.num_inference_steps == 3 * per
    visits = np.zeros(n, dtype=int)
    for w, c in zip(plan.windows, plan.is_cond):
        visits[w[~c]] += 1
    assert set(visits[targets].tolist()) == {per}


def test_validation_errors_match_reference():
    with pytest.raises(ValueError, match="divisible by the sliding stride"):
        steps_per_alternation(12, 5, False, 1)
    cond = [True] * 4 + [False] * 8
    with pytest.raises(ValueError, match="same for all target samples"):
        plan_sweep(cond, [0, 0, 0, 0, 1, 1, 1, 2, 1, 1, 1, 1], "spatial", 4, 1, 0, False, 1, 3)
    with pytest.raises(ValueError, match="should be 0 for all input samples"):
        plan_sweep(cond, [1, 0, 0, 0] + [0] * 8, "spatial", 4, 1, 0, False, 1, 3)
    with pytest.raises(ValueError, match="mismatch the config"):  # stride not dividing #targets leaves uneven visits
        plan_sweep([True] + [False] * 7, [0] * 8, "spatial", 4, 2, 0, False, 1, 3)


def test_second_round_starts_from_previous_indices():
    cond = [i in INPUTS for i in range(48)]
    start = [0 if c else 6 for c in cond]
    plan = plan_sweep(cond, start, "spatial", 12, 2, 0, False, 1, 3)
    assert plan.timestep_index[0].tolist() == [0, 0, 0, 0] + [6] * 12
    assert set(plan.final_timestep_indices[TARGETS].tolist()) == {12}
