"""CPU: sampler task lists, label formatting, validation errors and grid hand-over between rounds
(sliding_iterative_sampler.py:16-100,192-199) with a stub pipeline, plus the config composer."""
import re

import pytest
import torch

from diffuman4d_amd.host import config as cfglib
from diffuman4d_amd.host.dataset import SyntheticSpaTemDataset
from diffuman4d_amd.host.sampler import SlidingIterativeSampler
from stubs import StubPipeline


def make(**kw):
    ds = SyntheticSpaTemDataset(height=16, width=16, num_cameras=48)
    args = dict(window_size=12, sliding_stride=2, bidirectional=False, alternation_rounds=3,
                spa_label_range=[0, 48, 1], tem_label_range=[0, 16, 1], result_writer=lambda *a, **k: None)
    args.update(kw)
    return SlidingIterativeSampler(ds, [StubPipeline()], "/tmp/unused", **args)


def test_task_lists_demo_4d_tiny():
    s = make()
    assert [len(t) for t in s.all_tasks] == [16, 44, 16]
    assert [t[0]["domain"] for t in s.all_tasks] == ["spatial", "temporal", "spatial"]
    assert s.all_tasks[0][3] == {"alt": 1, "domain": "spatial", "domain_label": "000003"}
    assert s.all_tasks[1][0] == {"alt": 2, "domain": "temporal", "domain_label": "00"}
    assert s.all_tasks[1][1]["domain_label"] == "02"  # camera 01 is an input camera
    assert s.input_spa_labels == ["01", "13", "25", "37"] and len(s.target_spa_labels) == 44


def test_task_counts_demo_4d_and_3d():
    assert [len(t) for t in make(tem_label_range=[0, 150, 1]).all_tasks] == [150, 44, 150]
    assert [len(t) for t in make(tem_label_range=[0, 1, 1], alternation_rounds=1, sliding_stride=1).all_tasks] == [1]


def test_validation_errors():
    """Conditions AND texts of sliding_iterative_sampler.py:55,63,71-88."""
    with pytest.raises(ValueError, match=re.escape("window_size(=45) must be <= len(target_spa_labels)(=44)")):
        make(window_size=45)
    with pytest.raises(ValueError, match=re.escape("len(target_spa_labels)(=44) % sliding_stride(=5) must be 0")):
        make(sliding_stride=5)
    # README's "15 frames" is rejected with stride 2
    with pytest.raises(ValueError, match=re.escape("len(tem_labels)(=15) % sliding_stride(=2) must be 0")):
        make(tem_label_range=[0, 15, 1])
    with pytest.raises(ValueError, match=re.escape("window_size(=12) must be <= the number of tem_labels(=8) when "
                                                   "alternation_rounds > 1")):
        make(tem_label_range=[0, 8, 1], sliding_stride=1)
    with pytest.raises(ValueError, match="spa_labels or spa_label_range must be provided"):
        make(spa_label_range=None)
    with pytest.raises(ValueError, match="tem_labels or tem_label_range must be provided"):
        make(tem_label_range=None)


def test_three_rounds_visit_every_latent_18_times_and_hand_the_grid_over():
    s = make(tem_label_range=[0, 12, 1], spa_label_range=[0, 20, 1], input_spa_labels=[1, 9], window_size=6,
             sliding_stride=2)
    s.execute_tasks_no_check = None
    for tasks in s.all_tasks:
        for t in tasks:
            s.execute_one_task(t)
    per_alt = 6 // 2
    for c in s.target_spa_labels:
        for f in s.tem_labels:
            assert s.timestep_indices[c][f] == 3 * per_alt
            assert float(s.latents[c][f].flatten()[0]) == 3 * per_alt  # the stub adds 1 per denoising step
    for c in s.input_spa_labels:
        for f in s.tem_labels:
            assert s.timestep_indices[c][f] == 0


def test_latents_none_only_in_first_round():
    s = make(tem_label_range=[0, 12, 1])
    pipe = s.pipelines[0]
    s.execute_one_task(s.all_tasks[0][0])
    assert pipe.calls[-1]["latents_was_none"] is True
    for t in s.all_tasks[0][1:]:
        s.execute_one_task(t)
    s.execute_one_task(s.all_tasks[1][0])
    assert pipe.calls[-1]["latents_was_none"] is False and pipe.calls[-1]["domain"] == "temporal"
    assert pipe.calls[-1]["n"] == 24 and pipe.calls[-1]["cond_rows"] == list(range(12))


def test_config_composer_matches_reference_presets():
    c = cfglib.compose(["exp=demo_4d", "data.scene_label=0023_06", "data.data_dir=./d"])
    assert c["sampler"]["sliding_stride"] == 2 and c["sampler"]["alternation_rounds"] == 3
    assert c["sampler"]["output_dir"] == "./output/results/demo_4d/0023_06"
    assert c["model"]["_target_"].endswith("load_pipelines") and c["data"]["scene_label"] == "0023_06"
    assert cfglib.compose(["exp=demo_3d"])["sampler"]["alternation_rounds"] == 1
    assert cfglib.compose(["exp=demo_3d"])["sampler"]["sliding_stride"] == 1
    assert cfglib.compose(["exp=demo_4d_tiny"])["sampler"]["tem_label_range"] == [0, 16, 1]
    low = cfglib.compose(["exp=demo_4d", "sampler=sliding_low_mem"])["sampler"]
    assert (low["window_size"], low["guidance_scale"], low["sliding_stride"]) == (4, 1.0, 1)
    assert cfglib.compose(["exp=demo_4d", "sampler=sliding_premium"])["sampler"]["alternation_rounds"] == 5
    with pytest.raises(ValueError, match="exp"):
        cfglib.compose([])
    assert cfglib.locate("src.samplers.sliding_iterative_sampler.SlidingIterativeSampler") is SlidingIterativeSampler


def test_load_pipelines_rejects_unknown_dtype():
    from diffuman4d_amd.host.loader import load_pipelines
    with pytest.raises(ValueError, match="Unsupported torch_dtype"):
        load_pipelines(torch_dtype="fp32", gpu_ids=[])


def test_task_noise_seeds_do_not_collide():
    """Every task of a large job gets its own seed (the frame-shard groups draw their noise from it): the weighted-sum form of round 4
    mapped (round 1, frame 100003 + x) onto (round 2, frame x) and spatial frame 50021 + c onto temporal camera c."""
    s = make()
    assert s.task_noise_seed(1, "spatial", "100003") != s.task_noise_seed(2, "spatial", "000000")
    assert s.task_noise_seed(1, "spatial", "050021") != s.task_noise_seed(1, "temporal", "00")
    seeds = {s.task_noise_seed(alt, dom, lab) for alt in (1, 2, 3, 4, 5) for dom, labs in
             (("spatial", [f"{f:06d}" for f in range(0, 3000, 7)]), ("temporal", [f"{c:02d}" for c in range(48)])) for lab in labs}
    assert len(seeds) == 5 * (len(range(0, 3000, 7)) + 48)
    assert all(0 <= v < 2 ** 63 for v in seeds) and any(v >= 2 ** 31 for v in seeds) and s.task_noise_seed(1, "spatial", "000003") == s.task_noise_seed(1, "spatial", "000003")
    load_pipelines_precisions = ("auto", "fast", "parity", "fp16")
    from diffuman4d_amd.host.loader import load_pipelines
    with pytest.raises(ValueError, match="Unsupported precision"):
        load_pipelines(gpu_ids=[], precision="fp8")
    for prec in load_pipelines_precisions:  # no GPU ids: nothing is loaded, the argument checks run
        assert load_pipelines(model_dir="/nonexistent-but-unused", gpu_ids=[], precision=prec) == []


def test_only_tasks_with_one_window_plan_are_stacked():
    """sampler.stackable / denoise_stack (runner.task_batch): tasks of one round share their plan and go to the pipeline as ONE stack; tasks
    whose plans differ (another domain, targets at another timestep index), a frame-sharded sampler, or a pipeline without the stack entry
    run one by one -- and every sample ends up exactly as `denoise` leaves it."""
    from stubs import StackStubPipeline
    s = make(window_size=6, spa_label_range=[0, 20, 1], tem_label_range=[0, 12, 1], input_spa_labels=[1, 9])
    s.pipelines[0] = pipe = StackStubPipeline()
    a, b = (s.load_sample(**t) for t in s.all_tasks[0][:2])
    assert s.stackable([a, b]) and not s.stackable([a])
    out = s.denoise_stack([a, b])
    assert pipe.stacks == [2] and [o["domain_label"] for o in out] == ["000000", "000001"]
    assert all(int(o["timestep_indices"][o["target_indices"][0]]) == 3 for o in out)  # window 6, stride 2: 3 steps per round
    # a spatial task of round 1 that has not run next to one that has: targets at different timestep indices
    c = s.load_sample(**s.all_tasks[0][2])
    a2 = s.load_sample(**s.all_tasks[0][0])  # frame 0 again: its cells now sit at index 3 and carry latents
    assert not s.stackable([c, a2])
    # another domain (a round never mixes them; the check is by the samples, not by where they came from)
    assert not s.stackable([c, dict(c, domain="temporal")])
    # fall-back: one plain call per sample, in order
    before = len(pipe.calls)
    s.denoise_stack([c, a2])
    assert pipe.stacks == [2] and len(pipe.calls) == before + 2
    # a frame-sharded sampler never stacks (collectives are issued task by task); neither does a pipeline without the entry
    d, e = (s.load_sample(**t_) for t_ in s.all_tasks[0][3:5])
    s.frame_shard = object()
    assert not s.stackable([d, e])
    s.frame_shard = None
    assert s.stackable([d, e])
    s.pipelines[0] = StubPipeline()
    assert not s.stackable([d, e])
