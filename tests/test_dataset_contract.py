"""CPU: the reference dataset's runtime self-checks (`check_output`, spatem_dataset.py:191-228 -- the only executable statement
of what a sample must look like) applied to the synthetic stand-in and to the samples the sampler builds."""
import pytest
import torch

from diffuman4d_amd.host.dataset import SyntheticSpaTemDataset
from diffuman4d_amd.host.sampler import SlidingIterativeSampler
from stubs import StubPipeline


def check_output(sample, domain):
    """The reference's checks, restated: label consistency per domain and value ranges within [-1 - 1e-6, 1 + 1e-6]."""
    labels = sample["labels"]
    if domain == "temporal":
        half = len(labels) // 2
        spa = [l[1] for l in labels]
        assert all(s == spa[0] for s in spa[:half]) and all(s == spa[-1] for s in spa[half:])
        assert [l[2] for l in labels[:half]] == [l[2] for l in labels[half:]]  # the same frames for both cameras
    else:
        assert all(l[2] == labels[0][2] for l in labels)
    lo, hi = -1.0 - 1e-6, 1.0 + 1e-6
    for key in ("pixel_values", "skeletons", "plucker_embeds", "cond_masks"):
        t = sample[key]
        if t is not None:
            assert lo <= float(t.min()) and float(t.max()) <= hi, key


@pytest.mark.parametrize("mode", ["host", "cameras"])
def test_synthetic_samples_pass_the_reference_checks(mode):
    ds = SyntheticSpaTemDataset(height=32, width=16, num_cameras=48, plucker=mode)
    spa = [f"{c:02d}" for c in range(0, 48, 4)]
    s = ds.get_item("synthetic", spa, ["000003"], ["04", "28"])
    check_output(s, "spatial")
    n = len(spa)
    assert s["pixel_values"].shape == (n, 3, 32, 16) and s["cond_masks"].shape == (n, 1, 32, 16) and len(s["crops"]) == n
    assert s["Ks"].shape == (n, 3, 3) and s["poses"].shape == (n, 4, 4)
    assert torch.allclose(s["poses"][0], torch.eye(4), atol=1e-6)  # poses are relative to the sample's first camera (:169-173)
    assert (s["plucker_embeds"] is None) == (mode == "cameras")
    t = ds.get_item("synthetic", ["12"], [f"{f:06d}" for f in range(5)], ["04", "28"])
    check_output(t, "temporal")
    assert len(t["labels"]) == 10 and t["labels"][0][1] in ("04", "28") and t["labels"][-1][1] == "12"
    if mode == "host":  # unit ray directions; moments bounded by the scene scale
        d = t["plucker_embeds"][:, :3]
        assert torch.allclose(d.norm(dim=1), torch.ones_like(d[:, 0]), atol=1e-4)


def test_sampler_samples_pass_the_reference_checks():
    ds = SyntheticSpaTemDataset(height=16, width=16, num_cameras=48)
    s = SlidingIterativeSampler(ds, [StubPipeline()], "/tmp/unused", spa_label_range=[0, 20, 1], tem_label_range=[0, 6, 1],
                                input_spa_labels=[1, 9], window_size=6, sliding_stride=2, alternation_rounds=2, bidirectional=False)
    for tasks in s.all_tasks[:2]:
        sample = s.load_sample(**tasks[0])
        check_output(sample, tasks[0]["domain"])
        masks = sample["cond_masks"][:, 0, 0, 0]
        assert set(masks.tolist()) <= {0.0, 1.0} and 0 < int((masks == 0).sum()) < len(masks)  # inputs 0, targets 1 (:134-139)
