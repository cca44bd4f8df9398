"""GPU: the reference's sampler + runner logic (oracle/sampler.py -- the restatement that tests/test_reference_protocol.py
pins to the reference's real classes in the build container; /root/reference does not exist on the GPU box) drives the HIP
``Diffuman4DPipeline`` exactly as the reference would -- CPU fp32 images in, ``latents=None`` in round 1 and CPU bf16
stacks afterwards, ``.cpu()`` / ``.item()`` on what comes back -- and the grid it ends with equals, bit for bit, the grid
this repo's own sampler + runner produce with the same pipeline and the same random draws."""
import pytest
import torch

pytestmark = pytest.mark.gpu

KW = dict(spa_label_range=[0, 8, 1], tem_label_range=[0, 4, 1], input_spa_labels=[1, 5], window_size=4, sliding_stride=2,
          sliding_shift=0, bidirectional=False, num_denoising_steps=1, alternation_rounds=3, guidance_scale=2.0)


def _pipeline(tmp_path):
    from diffuman4d_amd.host.loader import load_pipelines
    from diffuman4d_amd.host.unet import UNetConfig
    from diffuman4d_amd.host.vae import VAEConfig
    from diffuman4d_amd.host.weights import write_synthetic_checkpoint
    ckpt = write_synthetic_checkpoint(tmp_path / "ckpt", UNetConfig(block_out_channels=(64, 128, 128, 128), attention_head_dim=(1, 2, 2, 2)),
                                      VAEConfig(block_out_channels=(32, 32, 64, 64), norm_num_groups=8), seed=3)
    return load_pipelines(model_dir=ckpt, torch_dtype="bf16", gpu_ids=[0])  # the Hydra factory seam, as inference.py:23 uses it


def _grid(s):
    return {(c, f): (s.timestep_indices[c][f], None if s.latents[c][f] is None else s.latents[c][f].float().cpu())
            for c in s.spa_labels for f in s.tem_labels}


def test_reference_sampler_logic_drives_the_hip_pipeline(tmp_path, hip_device):
    from diffuman4d_amd.host.dataset import SyntheticSpaTemDataset
    from diffuman4d_amd.host.runner import SamplingRunner
    from diffuman4d_amd.host.sampler import SlidingIterativeSampler
    from oracle.sampler import OracleRunner, OracleSampler
    pipes = _pipeline(tmp_path)
    saved = []

    def fresh_dataset():
        return SyntheticSpaTemDataset(height=64, width=64, num_cameras=8)

    # (a) the reference's control flow
    torch.manual_seed(1234)
    ref = OracleSampler(fresh_dataset(), pipes, str(tmp_path / "ref"), save=lambda sample, output_dir: saved.append(sample), **KW)
    OracleRunner(ref).inference()
    # (b) this repo's sampler + runner, one task at a time so that the random draws come in the same order
    torch.manual_seed(1234)
    own = SlidingIterativeSampler(fresh_dataset(), pipes, str(tmp_path / "own"), **KW)
    own.result_writer = None
    SamplingRunner(own, prefetch_depth=0, writers=1, gpu_streams=1).inference()

    steps = KW["window_size"] // KW["sliding_stride"] * KW["alternation_rounds"]
    ga, gb = _grid(ref), _grid(own)
    assert set(ga) == set(gb)
    for cell, (idx, lat) in ga.items():
        assert idx == gb[cell][0] and torch.equal(lat, gb[cell][1]), cell
    assert all(ga[(c, f)][0] == steps for c in ref.target_spa_labels for f in ref.tem_labels)
    assert all(ga[(c, f)][0] == 0 for c in ref.input_spa_labels for f in ref.tem_labels)
    # the cells the reference logic stores are CPU tensors in the model dtype, as its load_sample expects to stack them
    some = ref.latents[ref.target_spa_labels[0]][ref.tem_labels[0]]
    assert some.device.type == "cpu" and some.dtype == torch.bfloat16 and tuple(some.shape) == (4, 8, 8)
    # what it hands to the writer: fp32 CPU images in [0, 1], CPU bookkeeping tensors (sampling_utils.py:64-67,103)
    last = saved[-1]
    assert last["images"].dtype == torch.float32 and last["images"].device.type == "cpu"
    assert float(last["images"].min()) >= 0.0 and float(last["images"].max()) <= 1.0 and bool(last["fully_denoised"][last["target_indices"]].all())
