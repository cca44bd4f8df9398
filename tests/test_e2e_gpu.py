"""GPU: the whole CLI path on a tiny synthetic checkpoint -- config composition, `load_pipelines`
(`Diffuman4DPipeline.from_pretrained` on the diffusers directory layout, sampling_utils.py:17-51), sampler, the
pipelined runner, VAE, JPEG writer -- in the strict mode and with the VAE extensions; both must fully denoise the grid
and write one image per cell, and a task run through the loaded pipeline must equal the same task run through a
pipeline built directly from the same weights."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tiny_cfgs():
    from diffuman4d_amd.host.unet import UNetConfig
    from diffuman4d_amd.host.vae import VAEConfig
    return (UNetConfig(block_out_channels=(64, 128, 128, 128), attention_head_dim=(1, 2, 2, 2)),
            VAEConfig(block_out_channels=(32, 32, 64, 64), norm_num_groups=8))


@pytest.mark.parametrize("fast", [False, True])
def test_cli_path_on_synthetic_checkpoint(tmp_path, fast):
    from diffuman4d_amd.host import config as cfglib
    from diffuman4d_amd.host.results import check_sampling_results
    from diffuman4d_amd.host.runner import SamplingRunner
    from diffuman4d_amd.host.weights import write_synthetic_checkpoint
    ucfg, vcfg = _tiny_cfgs()
    ckpt = write_synthetic_checkpoint(tmp_path / "ckpt", ucfg, vcfg, seed=3)
    ov = ["exp=demo_4d_tiny", "model=diffuman4d_mi355x", "data=synthetic", f"model.model_dir={ckpt}", "model.gpu_ids=[0]",
          "data.height=64", "data.width=64", "data.num_cameras=8", f"result_dir={tmp_path / 'results'}",
          "sampler.spa_label_range=[0,8,1]", "sampler.tem_label_range=[0,4,1]", "sampler.input_spa_labels=[1,5]",
          "sampler.window_size=4", "sampler.sliding_stride=2"]
    if fast:  # every opt-in extension at once: VAE moment cache, decode-on-demand, Pluecker maps from the cameras on the device,
        # result arithmetic on the device + file encoding in writer processes
        ov += ["sampler.vae_cache=true", "sampler.decode_policy=denoised", "sampler.plucker_on_device=true", "data.plucker=cameras",
               "sampler.device_results=true"]
    cfg = cfglib.compose(ov)
    pipelines = cfglib.instantiate(cfg["model"])
    assert len(pipelines) == 1 and pipelines[0].device.type == "cuda"
    sampler = cfglib.instantiate(cfg["sampler"], dataset=cfglib.instantiate(cfg["data"]), pipelines=pipelines)
    SamplingRunner(sampler, prefetch_depth=2, writers=2, writer_processes=2 if fast else 0).inference()
    steps = 4 // 2 * 3
    assert all(sampler.timestep_indices[c][f] == steps for c in sampler.target_spa_labels for f in sampler.tem_labels)
    assert check_sampling_results(sampler.spa_labels, sampler.tem_labels, sampler.output_dir)
    lat = torch.stack([sampler.latents[c][f].float() for c in sampler.target_spa_labels for f in sampler.tem_labels])
    assert bool(torch.isfinite(lat).all())
    if fast:
        assert len(pipelines[0]._vae_cache["pixel"]) == 8 * 4
    import glob
    assert len(glob.glob(f"{sampler.output_dir}/grids/*.webp")) == 4 + 6 + 4  # one snapshot mosaic per task


def test_cli_path_in_the_wide_precisions(tmp_path):
    """`model.precision=parity` and `model.precision=fp16` through the same CLI path (config key -> load_pipelines -> from_pretrained -> sampler -> runner -> writer), with
    the VAE cache, decode-on-demand, device-side Pluecker maps and device-side result packing switched on: the job completes, the pipeline
    really is the fp32-tensor one, and its grid differs from the fast precision's (same seed, one task at a time) by bf16 rounding
    noise -- not by zero (the key was ignored) and not by more (a wiring defect of the mode); the two wide precisions sit an order of
    magnitude closer to each other than either does to the fast one."""
    from diffuman4d_amd.host import config as cfglib
    from diffuman4d_amd.host.results import check_sampling_results
    from diffuman4d_amd.host.runner import SamplingRunner
    from diffuman4d_amd.host.weights import write_synthetic_checkpoint
    ucfg, vcfg = _tiny_cfgs()
    ckpt = write_synthetic_checkpoint(tmp_path / "ckpt", ucfg, vcfg, seed=3)
    grids = {}
    for prec in ("fast", "parity", "fp16"):
        ov = ["exp=demo_4d_tiny", "model=diffuman4d_mi355x", "data=synthetic", f"model.model_dir={ckpt}", "model.gpu_ids=[0]",
              f"model.precision={prec}", "data.height=64", "data.width=64", "data.num_cameras=8", f"result_dir={tmp_path / prec}",
              "sampler.spa_label_range=[0,8,1]", "sampler.tem_label_range=[0,4,1]", "sampler.input_spa_labels=[1,5]",
              "sampler.window_size=4", "sampler.sliding_stride=2", "sampler.vae_cache=true", "sampler.decode_policy=denoised",
              "sampler.plucker_on_device=true", "data.plucker=cameras", "sampler.device_results=true"]
        cfg = cfglib.compose(ov)
        pipelines = cfglib.instantiate(cfg["model"])
        assert pipelines[0].precision == prec and pipelines[0].dtype == (torch.bfloat16 if prec == "fast" else torch.float32)
        sampler = cfglib.instantiate(cfg["sampler"], dataset=cfglib.instantiate(cfg["data"]), pipelines=pipelines)
        torch.manual_seed(1234)
        SamplingRunner(sampler, prefetch_depth=0, writers=1, gpu_streams=1).inference()
        assert all(sampler.timestep_indices[c][f] == 6 for c in sampler.target_spa_labels for f in sampler.tem_labels)
        assert check_sampling_results(sampler.spa_labels, sampler.tem_labels, sampler.output_dir)
        grids[prec] = torch.stack([sampler.latents[c][f].float().cpu() for c in sampler.target_spa_labels for f in sampler.tem_labels])
    assert bool(torch.isfinite(grids["parity"]).all())
    d = float((grids["fast"] - grids["parity"]).norm() / grids["parity"].norm())
    assert 1e-4 < d < 6e-2, d
    dh = float((grids["fp16"] - grids["parity"]).norm() / grids["parity"].norm())
    assert 1e-6 < dh < 0.2 * d, (dh, d)


@pytest.mark.parametrize("fast", [False, True])
def test_task_stacks_through_the_runner_equal_the_task_by_task_job(tmp_path, fast):
    """runner.task_batch through the CLI path (config -> pipelines -> sampler -> SamplingRunner): a job whose rounds run as stacks of
    three tasks sharing their window calls (one stream, so that the random draws come in task order in both jobs) leaves BITWISE the grid
    of the job run task by task and writes byte-identical files -- with the reference's protocol (strict) and with the extensions on."""
    import filecmp
    import os
    from glob import glob
    from diffuman4d_amd.host import config as cfglib
    from diffuman4d_amd.host.runner import SamplingRunner
    from diffuman4d_amd.host.weights import write_synthetic_checkpoint
    ucfg, vcfg = _tiny_cfgs()
    ckpt = write_synthetic_checkpoint(tmp_path / "ckpt", ucfg, vcfg, seed=3)
    grids, dirs, stacks = {}, {}, []
    for batch in (1, 3):
        ov = ["exp=demo_4d_tiny", "model=diffuman4d_mi355x", "data=synthetic", f"model.model_dir={ckpt}", "model.gpu_ids=[0]",
              "data.height=64", "data.width=64", "data.num_cameras=8", f"result_dir={tmp_path / f'b{batch}'}",
              "sampler.spa_label_range=[0,8,1]", "sampler.tem_label_range=[0,4,1]", "sampler.input_spa_labels=[1,5]",
              "sampler.window_size=4", "sampler.sliding_stride=2"]
        if fast:
            ov += ["sampler.vae_cache=true", "sampler.decode_policy=denoised", "sampler.plucker_on_device=true", "data.plucker=cameras",
                   "sampler.device_results=true"]
        cfg = cfglib.compose(ov)
        pipelines = cfglib.instantiate(cfg["model"])
        sampler = cfglib.instantiate(cfg["sampler"], dataset=cfglib.instantiate(cfg["data"]), pipelines=pipelines)
        if batch > 1:
            real = pipelines[0].sliding_iterative_denoise_stack

            def counted(tasks, **kw):
                stacks.append(len(tasks))
                return real(tasks, **kw)
            pipelines[0].sliding_iterative_denoise_stack = counted
        torch.manual_seed(1234)
        SamplingRunner(sampler, prefetch_depth=1, writers=1, gpu_streams=1, task_batch=batch).inference()
        grids[batch] = torch.stack([sampler.latents[c][f].float().cpu() for c in sampler.target_spa_labels for f in sampler.tem_labels])
        dirs[batch] = sampler.output_dir
    assert stacks == [2, 2, 3, 3, 2, 2]  # 4 frames, 6 target cameras, 4 frames: stacks of at most three, as even as the round allows
    assert bool(torch.isfinite(grids[3]).all()) and torch.equal(grids[1], grids[3])
    fa = sorted(os.path.relpath(f, dirs[1]) for f in glob(dirs[1] + "/**/*.*", recursive=True))
    fb = sorted(os.path.relpath(f, dirs[3]) for f in glob(dirs[3] + "/**/*.*", recursive=True))
    assert fa == fb and len(fa) > 8 * 4
    for f in fa:
        assert filecmp.cmp(os.path.join(dirs[1], f), os.path.join(dirs[3], f), shallow=False), f


@pytest.mark.parametrize("domain,n", [("spatial", 8), ("temporal", 12)])
def test_device_results_match_the_host_writer(tmp_path, domain, n):
    """results.pack_results_on_device on the GPU + imgwrite.write_package vs save_sampling_results on the host copy of the
    same decoded images: identical file sets, byte-identical JPEGs (same uint8 pixels into the same encoder); the WebP
    mosaics agree up to the device's antialiased down-scale (a few uint8 levels on isolated pixels)."""
    import filecmp
    import os
    from glob import glob
    import numpy as np
    from PIL import Image
    from diffuman4d_amd.host.results import pack_results_on_device, save_sampling_results, write_package
    from test_results import _pack_sample
    s = _pack_sample(n, 192, 96, domain)
    img_dev = s["images"].to(torch.bfloat16).cuda()  # what the VAE decode returns
    s["images"] = img_dev.float().cpu()              # what sampler.denoise hands the host writer
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    save_sampling_results(s, output_dir=a, save_crop_param=True)
    write_package(pack_results_on_device(s, img_dev, output_dir=b, save_crop_param=True))
    fa = sorted(os.path.relpath(f, a) for f in glob(a + "/**/*.*", recursive=True))
    fb = sorted(os.path.relpath(f, b) for f in glob(b + "/**/*.*", recursive=True))
    assert fa == fb
    for f in fa:
        if f.endswith(".webp"):
            x, y = (np.asarray(Image.open(os.path.join(d, f)).convert("RGB"), dtype=np.int32) for d in (a, b))
            assert x.shape == y.shape and float(np.abs(x - y).mean()) < 1.0
        else:
            assert filecmp.cmp(os.path.join(a, f), os.path.join(b, f), shallow=False), f


def test_from_pretrained_equals_direct_construction(tmp_path):
    from safetensors.torch import load_file
    from diffuman4d_amd.host.pipeline import Diffuman4DPipeline
    from diffuman4d_amd.host.scheduler import DDIMScheduler
    from diffuman4d_amd.host.unet import UNetMultiviewConditionModel
    from diffuman4d_amd.host.vae import AutoencoderKL
    from diffuman4d_amd.host.weights import write_synthetic_checkpoint
    from modelcheck import synthetic_task
    ucfg, vcfg = _tiny_cfgs()
    ckpt = write_synthetic_checkpoint(tmp_path / "ckpt", ucfg, vcfg, seed=5)
    loaded = Diffuman4DPipeline.from_pretrained(ckpt, device="cuda:0")
    usd = load_file(f"{ckpt}/unet/diffusion_pytorch_model.safetensors")
    vsd = load_file(f"{ckpt}/vae/diffusion_pytorch_model.safetensors")
    direct = Diffuman4DPipeline(AutoencoderKL(vcfg, vsd, "cuda"), UNetMultiviewConditionModel(ucfg, usd, "cuda"),
                                DDIMScheduler(), "cuda")
    n = 8
    pv, pl, sk, cm = synthetic_task(n, 64, 64, [1, 5], 9)
    g = torch.Generator().manual_seed(10)
    noise = {k: torch.randn(n, 4, 8, 8, generator=g).to(torch.bfloat16) for k in ("pixel", "skeleton", "latents")}
    kw = dict(pixel_values=pv, plucker_embeds=pl, skeletons=sk, cond_masks=cm, latents=None, domain="spatial",
              timestep_indices=torch.zeros(n, dtype=torch.int64), window_size=4, sliding_stride=2, sliding_shift=0,
              bidirectional=False, num_denoising_steps=1, alternation_rounds=1, guidance_scale=2.0, noise=noise)
    a, b = loaded.sliding_iterative_denoise(**kw), direct.sliding_iterative_denoise(**kw)
    assert torch.equal(a["latents"], b["latents"]) and torch.equal(a["images"], b["images"])


def test_tasks_on_concurrent_streams_equal_serial(tmp_path):
    """runner.run_round_pipelined(gpu_streams=2): two tasks denoised at the same time, each on its own HIP stream and
    worker thread but through ONE pipeline object, must return bitwise what they return one after the other."""
    import threading
    from safetensors.torch import load_file
    from diffuman4d_amd.host.pipeline import Diffuman4DPipeline
    from diffuman4d_amd.host.scheduler import DDIMScheduler
    from diffuman4d_amd.host.unet import UNetMultiviewConditionModel
    from diffuman4d_amd.host.vae import AutoencoderKL
    from diffuman4d_amd.host.weights import write_synthetic_checkpoint
    from modelcheck import synthetic_task
    ucfg, vcfg = _tiny_cfgs()
    ckpt = write_synthetic_checkpoint(tmp_path / "ckpt", ucfg, vcfg, seed=7)
    usd = load_file(f"{ckpt}/unet/diffusion_pytorch_model.safetensors")
    vsd = load_file(f"{ckpt}/vae/diffusion_pytorch_model.safetensors")
    pipe = Diffuman4DPipeline(AutoencoderKL(vcfg, vsd, "cuda"), UNetMultiviewConditionModel(ucfg, usd, "cuda"),
                              DDIMScheduler(), "cuda")
    n = 8
    kws = []
    for seed, domain in ((11, "spatial"), (12, "temporal"), (13, "spatial")):
        # temporal tasks: rows 0..T-1 are the input camera's frames, rows T..2T-1 the target camera's (load_sample :112-118)
        pv, pl, sk, cm = synthetic_task(n, 64, 64, [1, 5] if domain == "spatial" else [0, 1, 2, 3], seed)
        g = torch.Generator().manual_seed(seed + 100)
        noise = {k: torch.randn(n, 4, 8, 8, generator=g).to(torch.bfloat16) for k in ("pixel", "skeleton", "latents")}
        kws.append(dict(pixel_values=pv, plucker_embeds=pl, skeletons=sk, cond_masks=cm, latents=None, domain=domain,
                        timestep_indices=torch.zeros(n, dtype=torch.int64), window_size=4, sliding_stride=2,
                        sliding_shift=0, bidirectional=False, num_denoising_steps=1, alternation_rounds=1,
                        guidance_scale=2.0, noise=noise))
    serial = [pipe.sliding_iterative_denoise(**kw) for kw in kws]
    out, errs = [None] * len(kws), []

    def work(i):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                out[i] = pipe.sliding_iterative_denoise(**kws[i])
                out[i] = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in out[i].items()}
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    for _ in range(3):  # a few rounds: interleavings differ from run to run
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(kws))]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs
        for a, b in zip(serial, out):
            assert torch.equal(a["latents"].cpu(), b["latents"]) and torch.equal(a["images"].cpu(), b["images"])


def test_fp16_checkpoint_files_through_the_factory(tmp_path):
    """`torch_dtype: fp16` (sampling_utils.py:28-30) selects the *.fp16.safetensors files.  With `precision: "fast"` the arithmetic stays bf16
    MFMA: both file sets hold the same numbers here (weights below fp16's normal range are zeroed so that every bf16 value is exactly an fp16
    value), so that load must reproduce the bf16 load BIT FOR BIT.  With the default `precision: "auto"` an fp16 pipeline computes in the fp16
    precision (fp16 MFMA operands over fp32 tensors, the faithful form of the reference's fp16 pipelines): same weights, so its result is the
    bf16 pipeline's up to the bf16 path's own rounding noise, and the same files loaded as a bf16 pipeline in `precision: "fp16"` give it bit
    for bit.  The plain files are removed before the fp16 loads to prove which ones were read; `.to()` of a loaded pipeline is the identity
    on its own device."""
    import os
    from safetensors.torch import load_file, save_file
    from diffuman4d_amd.host.loader import load_pipelines
    from diffuman4d_amd.host.weights import write_synthetic_checkpoint
    from modelcheck import synthetic_task
    ucfg, vcfg = _tiny_cfgs()
    ckpt = write_synthetic_checkpoint(tmp_path / "ckpt", ucfg, vcfg, seed=5)
    for sub in ("unet", "vae"):
        f = f"{ckpt}/{sub}/diffusion_pytorch_model.safetensors"
        sd = {k: torch.where(v.float().abs() < 2.0 ** -14, torch.zeros_like(v), v) for k, v in load_file(f).items()}
        assert all(torch.equal(v.to(torch.float16).to(torch.bfloat16), v) for v in sd.values())
        save_file(sd, f)
        save_file({k: v.to(torch.float16) for k, v in sd.items()}, f"{ckpt}/{sub}/diffusion_pytorch_model.fp16.safetensors")
    ref_pipe = load_pipelines(model_dir=ckpt, torch_dtype="bf16", gpu_ids=[0])[0]
    h16_ref = load_pipelines(model_dir=ckpt, torch_dtype="bf16", gpu_ids=[0], precision="fp16")[0]
    for sub in ("unet", "vae"):
        os.remove(f"{ckpt}/{sub}/diffusion_pytorch_model.safetensors")
    pipe = load_pipelines(model_dir=ckpt, torch_dtype="fp16", gpu_ids=[0], precision="fast")[0]
    auto = load_pipelines(model_dir=ckpt, torch_dtype="fp16", gpu_ids=[0])[0]
    assert pipe.checkpoint_variant == "fp16" and ref_pipe.checkpoint_variant is None and pipe.to("cuda:0") is pipe
    assert pipe.precision == "fast" and auto.precision == "fp16" and auto.dtype == torch.float32 and ref_pipe.precision == "fast"
    n = 8
    pv, pl, sk, cm = synthetic_task(n, 64, 64, [1, 5], 9)
    g = torch.Generator().manual_seed(10)
    noise = {k: torch.randn(n, 4, 8, 8, generator=g).to(torch.bfloat16) for k in ("pixel", "skeleton", "latents")}
    kw = dict(pixel_values=pv, plucker_embeds=pl, skeletons=sk, cond_masks=cm, latents=None, domain="spatial",
              timestep_indices=torch.zeros(n, dtype=torch.int64), window_size=4, sliding_stride=2, sliding_shift=0,
              bidirectional=False, num_denoising_steps=1, alternation_rounds=1, guidance_scale=2.0, noise=noise)
    a, b = pipe.sliding_iterative_denoise(**kw), ref_pipe.sliding_iterative_denoise(**kw)
    assert torch.equal(a["timestep_indices"], b["timestep_indices"])
    assert torch.equal(a["latents"], b["latents"]) and torch.equal(a["images"], b["images"])
    c, d = auto.sliding_iterative_denoise(**kw), h16_ref.sliding_iterative_denoise(**kw)
    assert torch.equal(c["latents"], d["latents"]) and torch.equal(c["images"], d["images"]) and c["latents"].dtype == torch.float32
    rel = float((c["images"].float() - b["images"].float()).norm() / b["images"].float().norm())
    assert 1e-4 < rel < 2e-2, rel  # the fp16 precision against the bf16 one: the bf16 path's rounding noise, not a different model
