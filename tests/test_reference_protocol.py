"""CPU, build container only (needs /root/reference): the reference's OWN SlidingIterativeSampler + SamplingRunner
(imported through oracle/refshim.py) against (a) oracle/sampler.py, the restatement the GPU test uses where the reference
checkout is absent, and (b) this repo's product sampler + runner -- all three driving the same recording pipeline.

What must be identical: the sequence of ``sliding_iterative_denoise`` calls (keyword names, tensor shapes / dtypes /
devices, None-ness of ``latents``, scalar arguments) and the final grid (latents and timestep indices of every cell)."""
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
HAVE_REF = Path("/root/reference/src/samplers/sliding_iterative_sampler.py").exists()
pytestmark = pytest.mark.skipif(not HAVE_REF, reason="the reference checkout exists only in the build container")

from diffuman4d_amd.host.dataset import SyntheticSpaTemDataset  # noqa: E402
from stubs import StubPipeline  # noqa: E402

KW = dict(spa_label_range=[0, 20, 1], tem_label_range=[0, 12, 1], input_spa_labels=[1, 9], window_size=6,
          sliding_stride=2, sliding_shift=0, bidirectional=False, num_denoising_steps=1, alternation_rounds=3,
          guidance_scale=2.0)
PROTOCOL_KEYS = ["pixel_values", "plucker_embeds", "skeletons", "cond_masks", "latents", "domain", "timestep_indices",
                 "window_size", "sliding_stride", "sliding_shift", "bidirectional", "num_denoising_steps",
                 "alternation_rounds", "guidance_scale", "tqdm"]  # sliding_iterative_sampler.py:161-178


class RecordingPipeline(StubPipeline):
    """StubPipeline that also records HOW it was called."""

    def __init__(self):
        super().__init__()
        self.trace = []

    def sliding_iterative_denoise(self, **kw):
        def sig(v):
            if torch.is_tensor(v):
                return ("tensor", tuple(v.shape), str(v.dtype), v.device.type)
            if callable(v):
                return "callable"
            return v
        self.trace.append({k: sig(v) for k, v in kw.items()})
        return super().sliding_iterative_denoise(**kw)


def _grid(s):
    return ({c: {f: (None if l is None else l.clone()) for f, l in d.items()} for c, d in s.latents.items()},
            {c: dict(d) for c, d in s.timestep_indices.items()})


def _run(kind, n_pipes=1):
    ds = SyntheticSpaTemDataset(height=16, width=16, num_cameras=48)
    pipes = [RecordingPipeline() for _ in range(n_pipes)]
    if kind == "reference":
        from oracle import refshim
        refshim.install()
        import src.samplers.sliding_iterative_sampler as ref_mod
        from src.samplers.sampling_runner import SamplingRunner
        import src.samplers.sampling_runner as run_mod
        ref_mod.save_sampling_results = lambda *a, **k: None  # the writer is tested separately (tests/test_results.py)
        ref_mod.check_sampling_results = lambda *a, **k: True
        run_mod.check_sampling_results = lambda *a, **k: True
        s = ref_mod.SlidingIterativeSampler(ds, pipes, "/tmp/dm4d_unused", **KW)
        SamplingRunner(s).inference()
    elif kind == "oracle":
        from oracle.sampler import OracleRunner, OracleSampler
        s = OracleSampler(ds, pipes, "/tmp/dm4d_unused", **KW)
        OracleRunner(s).inference()
    else:
        from diffuman4d_amd.host.runner import SamplingRunner
        from diffuman4d_amd.host.sampler import SlidingIterativeSampler
        s = SlidingIterativeSampler(ds, pipes, "/tmp/dm4d_unused", **KW)
        s.result_writer = None
        SamplingRunner(s, prefetch_depth=0, writers=1, gpu_streams=1).inference()
    return s, pipes


def _same_grid(a, b):
    (la, ia), (lb, ib) = _grid(a), _grid(b)
    assert ia == ib
    for c in la:
        for f in la[c]:
            assert (la[c][f] is None) == (lb[c][f] is None)
            if la[c][f] is not None:
                assert torch.equal(la[c][f].float().cpu(), lb[c][f].float().cpu())


def test_restatement_is_pinned_to_the_reference_sampler():
    ref, (rp,) = _run("reference")
    ora, (op,) = _run("oracle")
    assert list(rp.trace[0]) == PROTOCOL_KEYS  # what the reference really passes, in its order
    assert rp.trace == op.trace and rp.calls == op.calls
    assert ref.all_tasks == ora.all_tasks and ref.spa_labels == ora.spa_labels and ref.tem_labels == ora.tem_labels
    _same_grid(ref, ora)


def test_product_sampler_speaks_the_reference_protocol():
    ref, (rp,) = _run("reference")
    pro, (pp,) = _run("product")
    assert len(rp.trace) == len(pp.trace) == sum(len(t) for t in ref.all_tasks)
    for a, b in zip(rp.trace, pp.trace):
        assert set(b) == set(a)  # no extension keyword unless one is switched on
        assert a == b
    assert rp.calls == pp.calls
    _same_grid(ref, pro)


def test_reference_runner_threads_over_two_pipelines_cover_every_task_once():
    ref, pipes = _run("reference", n_pipes=2)
    ora, opipes = _run("oracle", n_pipes=2)
    total = sum(len(t) for t in ref.all_tasks)
    assert sum(len(p.calls) for p in pipes) == total == sum(len(p.calls) for p in opipes)
    _same_grid(ref, ora)  # tasks of a round touch disjoint target cells: the result does not depend on the thread interleaving
