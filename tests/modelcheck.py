"""Model-level parity checks on a GPU box: HIP UNet / VAE / pipeline (through the C ABI) vs the CPU
oracle on identical seeded weights, inputs and injected noise.

`python tests/modelcheck.py` prints one line per case; tests/test_model_gpu.py wraps the same cases.

Tolerances (bf16 activations end to end, fp32 accumulation; the fp32 oracle is the ground truth):
  * the yardstick of a case is the error of the ORACLE run in bf16 -- the reference's own arithmetic
    (configs/model/diffuman4d.yaml: bf16) -- against the same fp32 truth, on the same weights, inputs and noise;
  * a model-level case passes when  rel-L2(HIP, oracle-fp32) <= YARD_FACTOR x yardstick  (per compared quantity:
    UNet output, latents, decoded RGB), i.e. when the HIP path is as close to fp32 truth as the reference's bf16 path;
  * BASELINE.json's north_star asks for 1e-3 on decoded RGB.  No bf16 pipeline meets that against an fp32 oracle --
    neither this one nor the reference's (yardsticks are 0.6-1.4e-2) -- so 1e-3 is reported as NOT met (NORTH_STAR);
    it would take fp32 activations;
  * cases with their own fixed bound (bitwise equalities, extension-vs-strict comparisons, the golden fixtures whose
    generator did not record a yardstick) keep a TOL entry;
  * integer bookkeeping (timestep indices, fully_denoised) must be bit-exact;
  * `par_*` cases run the same models with precision="parity" and must stay within PARITY_TOL (= north_star's 1e-3) of the fp32
    oracle on every compared quantity, decoded RGB included;
  * `fp16_*` cases run them with precision="fp16" (single-term fp16 MFMA operands over fp32 tensors) under FIXED bounds per compared
    quantity (FP16_BOUNDS): decoded RGB within north_star's 1e-3, latents and a single UNet call within 2e-3 (the operand rounding alone
    measures 8.5e-4 on the judged UNet call, tools/error_budget.py).
"""
from __future__ import annotations

import math
import sys
import time
import traceback
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

BF = torch.bfloat16
YARD_FACTOR = 1.15   # HIP error may exceed the bf16-oracle yardstick by at most 15 %
NORTH_STAR = 1e-3    # BASELINE.json: decoded RGB within 1e-3 rel-L2 -- not met by any bf16 path (see above); met by precision="parity"
# precision="parity" (fp32 tensors between kernels, two-term bf16 MFMA operands; include/dm4d.h "Parity precision"): every `par_*`
# case is judged against a FIXED bound on rel-L2 vs the fp32 oracle -- PARITY_TOL unless PARITY_TOLS names a tighter one.
PARITY_TOL = 1e-3
# HIP fast precision vs the rounding-matched oracle (oracle/matched.py).  Two bf16 networks decorrelate within a few rounding layers
# (oracle/replay.py header: measured 1.29e-2 between HIP and the matched oracle, each 1.06e-2 from fp32), so the DIRECT distance cannot be
# bounded tightly; what the matched oracle predicts sharply is the SIZE of the fast path's error.  A `*_matched` case passes when
# err(HIP vs fp32) / err(matched vs fp32) lies in MATCHED_BAND (two-sided: fewer roundings than the model of the path is a finding
# too) and the direct distance stays below sqrt(2) x 1.1 of the larger of the two.
MATCHED_BAND = (0.85, 1.10)
# precision="fp16": fixed bounds per compared quantity.  `images` IS north_star's bar; a single UNet call and the latents of a task sit
# above the decoded RGB (the VAE decoder averages the latents' error down by ~0.55) and get twice that.
FP16_BOUNDS = {"images": 1e-3, "latents": 2e-3, "unet_out": 2e-3, "bookkeeping": 0.0}


def matched_verdict(err_hip, err_matched, direct):
    ratio = err_hip / err_matched
    excess = max(0.0, ratio - MATCHED_BAND[1], MATCHED_BAND[0] - ratio) + max(0.0, direct - 1.1 * math.sqrt(2.0) * max(err_hip, err_matched))
    return ratio, excess
# Fast precision, every launch of a model call recomputed in fp64 from the tensors the device was given (oracle/replay.py): what is
# left is summation order and the hardware's exp2 / rcp, i.e. the bf16 roundings those flip.  Fixed bound per launch.
REPLAY_TOL = 5e-4
GOLDEN = Path(__file__).resolve().parent / "golden"


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def bf16_weights(model):
    """Round parameters to bf16-representable fp32 so that oracle and HIP see identical weights."""
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.to(BF).float())
    return model


def make_unet(seed=0, **cfg_kw):
    from oracle.unet import UNetConfig, UNetMultiviewConditionModel, init_unet_weights
    cfg = UNetConfig.tiny(**cfg_kw)
    m = UNetMultiviewConditionModel(cfg).eval()
    init_unet_weights(m, seed)
    return cfg, bf16_weights(m)


def make_vae(seed=1):
    from oracle.unet import init_unet_weights
    from oracle.vae import AutoencoderKL, VAEConfig
    cfg = VAEConfig.tiny()
    v = AutoencoderKL(cfg).eval()
    init_unet_weights(v, seed)
    return cfg, bf16_weights(v)


_SD_CACHE: dict = {}


def sd_weights(kind: str, seed: int):
    """The seeded SD-2.1 UNet / SD VAE state dict on the CPU (random_state_dict: 6 s for the UNet's 816 M parameters), built once per
    process: a dozen cases at the judged geometry use the same two dictionaries and none of them modifies a tensor."""
    key = (kind, int(seed))
    if key not in _SD_CACHE:
        from diffuman4d_amd.host.unet import UNetConfig
        from diffuman4d_amd.host.vae import VAEConfig
        from diffuman4d_amd.host.weights import random_state_dict, unet_param_shapes, vae_param_shapes
        shapes = unet_param_shapes(UNetConfig()) if kind == "unet" else vae_param_shapes(VAEConfig())
        _SD_CACHE[key] = random_state_dict(shapes, int(seed), "cpu")
    return _SD_CACHE[key]


def hip_unet(cfg, oracle_model, precision="fast"):
    from dataclasses import asdict
    from diffuman4d_amd.host.unet import UNetConfig, UNetMultiviewConditionModel
    return UNetMultiviewConditionModel(UNetConfig.from_dict(asdict(cfg)), oracle_model.state_dict(), "cuda", precision)


def hip_vae(cfg, oracle_model, precision="fast"):
    from dataclasses import asdict
    from diffuman4d_amd.host.vae import AutoencoderKL, VAEConfig
    return AutoencoderKL(VAEConfig.from_dict(asdict(cfg)), oracle_model.state_dict(), "cuda", precision)


def unet_sample(hm, x):
    """NCHW input batch (CPU) -> what the HIP UNet's forward takes: NHWC bf16 padded to 32 channels, or (precision "parity") the
    two-term operand of the fp32 sample."""
    from diffuman4d_amd.host import ops
    if hm.wide:  # parity: two-term operand [hi | lo]; fp16: one fp16 plane
        return ops.split(x.float().permute(0, 2, 3, 1).contiguous().cuda(), cpad=hm.IN_PAD, h16=hm.h16)
    return ops.nchw_to_nhwc(x.to(BF).cuda(), hm.IN_PAD)


def case_unet(num_frames=4, cfg_batch=2, h=16, w=8, tem=False, domain="spatial", seed=0, pose=False, precision="fast", matched=False):
    from diffuman4d_amd.host import ops
    cfg, om = make_unet(seed, enable_tem_embeds=tem, **(dict(enable_pose_encoder=True, in_channels=11) if pose else {}))
    if tem:  # the temporal embedding MLP is zero-initialised in training; randomise it so it is exercised
        g = torch.Generator().manual_seed(seed + 5)
        with torch.no_grad():
            for p in om.temporal_pos_embed.parameters():
                p.copy_((torch.randn(p.shape, generator=g) * 0.05).to(BF).float())
    hm = hip_unet(cfg, om, precision)
    g = torch.Generator().manual_seed(seed + 1)
    B = num_frames * cfg_batch
    x = (torch.randn(B, cfg.in_channels, h, w, generator=g)).to(BF)
    t = torch.randint(0, 1000, (B,), generator=g)
    domains = [domain] * cfg_batch
    sk = (torch.rand(B, 3, 8 * h, 8 * w, generator=g) * 2 - 1).to(BF) if pose else None  # raw skeleton images (:551)
    with torch.no_grad():
        ref = om(x.float(), t, skeletons=sk.float() if pose else None, domains=domains, num_frames=num_frames)
        om_bf = om.to(BF)
        ref_bf = om_bf(x, t, skeletons=sk, domains=domains, num_frames=num_frames).float()
        om.float()
    xd = unet_sample(hm, x)
    out = hm(xd, t.float().cuda(), skeletons=ops.nchw_to_nhwc(sk.cuda(), 4) if pose else None, domains=domains,
             num_frames=num_frames)
    out = ops.nhwc_to_nchw(out)
    if matched:  # HIP fast precision vs the rounding-matched oracle, run on the spot (MATCHED_BAND)
        from oracle import matched as mo
        mref = mo.unet_forward(om, x.float(), t, domains=domains, num_frames=num_frames)
        e_h, e_m, direct = rel_l2(out, ref), rel_l2(mref, ref), rel_l2(out, mref)
        ratio, excess = matched_verdict(e_h, e_m, direct)
        print(f"    [unet {domain} fast vs rounding-matched oracle] HIP vs fp32 {e_h:.3e}, matched vs fp32 {e_m:.3e} (ratio {ratio:.3f}, band "
              f"{MATCHED_BAND}), HIP vs matched {direct:.3e}", flush=True)
        return {"band_excess": excess}, {"band_excess": 0.0}
    return {"unet_out": rel_l2(out, ref)}, {"unet_out": rel_l2(ref_bf, ref)}


class _RecordShard:
    """world-size-1 stand-in for parallel.FrameShard that records every K|V block it is asked to gather."""
    rank, world = 0, 1

    def __init__(self):
        self.kv = []

    def local_frames(self, n):
        return slice(0, n)

    def gather_kv(self, kv_local):
        self.kv.append(kv_local.clone())
        return kv_local

    # split form used by the UNet (the Q projection runs between start and finish)
    def gather_kv_start(self, kv_local):
        return self.gather_kv(kv_local)

    def gather_kv_finish(self, handle):
        return handle


class _ReplayShard:
    """Rank r of P, single process: the all-gather is answered from the recording of the unsharded run, after
    checking that this rank's contribution is bitwise the slice the real collective would have sent."""

    def __init__(self, rank, world, recorded):
        self.rank, self.world, self.rec, self.i, self.max_dev = rank, world, recorded, 0, 0.0

    def local_frames(self, n):
        fl = n // self.world
        return slice(self.rank * fl, (self.rank + 1) * fl)

    def gather_kv(self, kv_local):
        full = self.rec[self.i]
        self.i += 1
        ls = kv_local.shape[1]
        mine = full[:, self.rank * ls:(self.rank + 1) * ls]
        self.max_dev = max(self.max_dev, float((mine.float() - kv_local.float()).abs().max()))
        return full

    def gather_kv_start(self, kv_local):
        return self.gather_kv(kv_local)

    def gather_kv_finish(self, handle):
        return handle


def case_unet_frame_shard(P=4, num_frames=8, h=16, w=8, tem=True, seed=0, precision="fast"):
    """In-window frame sharding: every rank's slice of the UNet output must equal the unsharded
    output bitwise, and what it would contribute to each K/V all-gather must equal the unsharded K/V slice.  In the wide
    precisions the gathered blocks are the operand planes of K | V (parity: hi and lo planes, fp16: one fp16 plane)."""
    cfg, om = make_unet(seed, enable_tem_embeds=tem)
    hm = hip_unet(cfg, om, precision)
    g = torch.Generator().manual_seed(seed + 1)
    B = 2 * num_frames
    x = unet_sample(hm, torch.randn(B, cfg.in_channels, h, w, generator=g).to(BF))
    t = torch.randint(0, 1000, (B,), generator=g).float().cuda()
    rec = _RecordShard()
    full = hm(x, t, domains=["temporal"] * 2, num_frames=num_frames, shard=rec)
    base = hm(x, t, domains=["temporal"] * 2, num_frames=num_frames)
    worst = float((full.float() - base.float()).abs().max())  # split Q / KV projection vs fused QKV
    fl = num_frames // P
    for r in range(P):
        rows = torch.cat([torch.arange(r * fl, (r + 1) * fl), num_frames + torch.arange(r * fl, (r + 1) * fl)]).cuda()
        sh = _ReplayShard(r, P, rec.kv)
        out = hm(x.index_select(0, rows).contiguous(), t.index_select(0, rows).contiguous(), domains=["temporal"] * 2,
                 num_frames=fl, shard=sh)
        assert sh.i == len(rec.kv)
        worst = max(worst, float((out.float() - full.index_select(0, rows).float()).abs().max()), sh.max_dev)
    return worst, 0.0


def case_pipeline_shard_world1(seed=21, precision="fast"):
    """denoise_latents with a REAL process group (RCCL, world size 1): the sharded code path (table slicing, K/V and
    latent-row all-gathers, index_copy) must reproduce the plain path bitwise."""
    import os
    import torch.distributed as dist
    from diffuman4d_amd.host.parallel import FrameShard
    from diffuman4d_amd.host.pipeline import Diffuman4DPipeline
    from diffuman4d_amd.host.schedule import plan_sweep
    from diffuman4d_amd.host.scheduler import DDIMScheduler
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        cfg, om = make_unet(seed)
        pipe = Diffuman4DPipeline(None, hip_unet(cfg, om, precision), DDIMScheduler(), "cuda")
        g = torch.Generator(device="cuda").manual_seed(seed)
        n, h, w = 8, 16, 8
        rnd = lambda c, s=1.0: (torch.randn(n, h, w, c, generator=g, device="cuda") * s).to(pipe.dtype)  # noqa: E731
        cond = [i in (1, 5) for i in range(n)]
        mask = torch.tensor([0.0 if c else 1.0 for c in cond], device="cuda").to(pipe.dtype)[:, None, None, None].expand(n, h, w, 1).contiguous()
        pv, pl, sk, lat0 = rnd(4), rnd(6, 0.5), rnd(4), rnd(4)
        plan = plan_sweep(cond, [0] * n, "spatial", 4, 2, 0, False, 1, 1)
        a = pipe.denoise_latents(pv, pl, sk, mask, lat0.clone(), plan, "spatial", 2.0)
        b = pipe.denoise_latents(pv, pl, sk, mask, lat0.clone(), plan, "spatial", 2.0, shard=FrameShard())
        return float((a.float() - b.float()).abs().max()), 0.0
    finally:
        if created:
            dist.destroy_process_group()


def case_task_batching(domain="spatial", copies=2, sched="ddim", seed=23):
    """Tasks of a round stacked into one window call (upload_plan copies): every task's latents equal, bit for bit, what the
    task gives alone -- GEMMs and convolutions are per row, GroupNorm per sample, attention per (batch, head), and every
    tile choice the larger batch may trigger is bit-identical."""
    from diffuman4d_amd.host.pipeline import Diffuman4DPipeline
    from diffuman4d_amd.host.schedule import plan_sweep
    if sched == "dpm":
        from diffuman4d_amd.host.scheduler import DPMSolverMultistepScheduler as S_
    else:
        from diffuman4d_amd.host.scheduler import DDIMScheduler as S_
    cfg, om = make_unet(seed)
    pipe = Diffuman4DPipeline(None, hip_unet(cfg, om), S_(), "cuda")
    g = torch.Generator(device="cuda").manual_seed(seed)
    h, w = 16, 8
    if domain == "spatial":
        n, cond = 8, [i in (1, 5) for i in range(8)]
    else:
        n, cond = 8, [i < 4 for i in range(8)]
    rnd = lambda c, s=1.0: (torch.randn(copies * n, h, w, c, generator=g, device="cuda") * s).to(BF)  # noqa: E731
    mask = torch.tensor([0.0 if c else 1.0 for c in cond] * copies, device="cuda").to(BF)[:, None, None, None].expand(copies * n, h, w, 1).contiguous()
    pv, pl, sk, lat0 = rnd(4), rnd(6, 0.5), rnd(4), rnd(4)
    plan = plan_sweep(cond, [0] * n, domain, 4, 2 if domain == "spatial" else 1, 0, True, 2, 1)
    alone = [pipe.denoise_latents(pv[k * n:(k + 1) * n], pl[k * n:(k + 1) * n], sk[k * n:(k + 1) * n], mask[k * n:(k + 1) * n],
                                  lat0[k * n:(k + 1) * n].clone(), plan, domain, 2.0) for k in range(copies)]
    tables = pipe.upload_plan(plan, 2.0, copies=copies, rows_per_task=n)
    stacked = pipe.denoise_latents(pv, pl, sk, mask, lat0.clone(), plan, domain, 2.0, tables=tables)
    ref = torch.cat(alone)
    assert bool(torch.isfinite(stacked.float()).all()) and not torch.equal(stacked, lat0)
    return float((stacked.float() - ref.float()).abs().max()), 0.0


def latents_nhwc(hv, z):
    """NCHW latents (CPU) -> the NHWC device tensor decode_to_images takes (bf16, or fp32 under the wide precisions)."""
    from diffuman4d_amd.host import ops
    if hv.wide:
        return z.float().permute(0, 2, 3, 1).contiguous().cuda()
    return ops.nchw_to_nhwc(z.to(BF).contiguous().cuda())


def case_vae(h=64, w=64, seed=1, precision="fast"):
    from diffuman4d_amd.host import ops
    cfg, ov = make_vae(seed)
    hv = hip_vae(cfg, ov, precision)
    g = torch.Generator().manual_seed(seed + 1)
    img = (torch.rand(3, 3, h, w, generator=g) * 2 - 1).to(BF)
    noise = torch.randn(3, 4, h // 8, w // 8, generator=g).to(BF)
    with torch.no_grad():
        z_ref = ov.sample_posterior(ov.moments(img.float()), noise.float()) * cfg.scaling_factor
        z_in = z_ref.to(BF)  # the decoder is checked in isolation on the bf16-rounded ORACLE latents
        img_ref = (ov.decode(z_in.float() / cfg.scaling_factor) / 2 + 0.5).clamp(0, 1)
        ov.to(BF)
        z_bf = (ov.sample_posterior(ov.moments(img), noise) * cfg.scaling_factor).float()
        img_bf = (ov.decode(z_in / cfg.scaling_factor) / 2 + 0.5).clamp(0, 1).float()
        ov.float()
    z = hv.encode_scaled(img, noise)  # NHWC
    out = hv.decode_to_images(latents_nhwc(hv, z_in))
    return ({"latents": rel_l2(ops.nhwc_to_nchw(z), z_ref), "images": rel_l2(out, img_ref)},
            {"latents": rel_l2(z_bf, z_ref), "images": rel_l2(img_bf, img_ref)})


def _check_fixture_inputs(what, got, want):
    if abs(got - want) > 1e-6 * abs(want):
        raise RuntimeError(f"{what}: checksum {got!r} != fixture {want!r} -- torch's CPU generator produced different numbers "
                           "than when tests/golden/sd21_72x40.pt was made; regenerate it (tests/golden/make_golden_sd21.py)")


def case_unet_sd21(name, precision="fast", fixture="sd21_72x40.pt", matched=False):
    """The JUDGED configuration: full SD-2.1 geometry (320, 640, 1280, 1280; heads 5/10/20/20), 72x40 latents, one spatial
    (F = 16, CFG batch 32, 3-D attention over 46 080 / 11 520 / 2 880 / 720 tokens) or temporal (F = 24, CFG batch 48,
    69 120 / 17 280 / 4 320 / 1 080 tokens) window call, HIP vs the fp32 CPU oracle's output recorded in
    tests/golden/sd21_72x40.pt together with the bf16-oracle yardstick (made by tests/golden/make_golden_sd21.py)."""
    from diffuman4d_amd.host import ops
    from diffuman4d_amd.host.unet import UNetConfig, UNetMultiviewConditionModel
    from diffuman4d_amd.host.weights import random_state_dict, unet_param_shapes
    sys.path.insert(0, str(GOLDEN))
    import make_golden_sd21 as mk
    g = torch.load(GOLDEN / fixture)[name]
    cfg = UNetConfig()
    sd = sd_weights("unet", mk.UNET_SEED)
    _check_fixture_inputs("UNet weights", float(sum(v.float().abs().sum() for v in sd.values())), g["weights_checksum"])
    x, t = mk.unet_inputs(g["num_frames"], g["n_cond"], g["seed"], g.get("size"))
    _check_fixture_inputs("UNet input", float(x.float().abs().sum()), g["x_checksum"])
    assert torch.equal(t, g["t"])
    hm = UNetMultiviewConditionModel(cfg, sd, "cuda", precision)
    del sd
    torch.cuda.synchronize()
    t0 = time.time()
    out = hm(unet_sample(hm, x), t.float().cuda(), domains=[g["domain"]] * 2, num_frames=g["num_frames"])
    torch.cuda.synchronize()
    secs = time.time() - t0
    out = ops.nhwc_to_nchw(out)
    if matched:
        # HIP (fast precision) and the ROUNDING-MATCHED oracle (oracle/matched.py: the fp32 oracle rounded to bf16 wherever the HIP
        # path stores a tensor): see MATCHED_BAND above for what can and cannot be asserted between two bf16 networks
        e_h, e_m, direct = rel_l2(out, g.get("out_f32", g["out"])), g["matched_vs_fp32"], rel_l2(out, g["matched_out"])
        ratio, excess = matched_verdict(e_h, e_m, direct)
        print(f"    [{name} fast vs rounding-matched oracle] HIP vs fp32 {e_h:.3e}, matched vs fp32 {e_m:.3e} (ratio {ratio:.3f}, band "
              f"{MATCHED_BAND}), HIP vs matched {direct:.3e}", flush=True)
        del hm
        torch.cuda.empty_cache()
        return {"band_excess": excess}, {"band_excess": 0.0}
    if "out_sub" in g:  # the 128 x 128 fixture keeps every sub-th pixel of the fp32 output
        err, yard = rel_l2(out[..., ::g["sub"], ::g["sub"]], g["out_sub"]), g["yard_bf16_sub"]
    else:  # fp32 copy of the oracle output where the fixture has one (round 4), else the fp16 copy (its own floor: 2.1e-4)
        err, yard = rel_l2(out, g.get("out_f32", g["out"])), g["yard_bf16"]
    print(f"    [{name} {precision}, first call incl. allocation {secs * 1e3:.0f} ms] unet_out rel_l2={err:.3e} (oracle-bf16 {yard:.3e})", flush=True)
    del hm
    torch.cuda.empty_cache()
    return {"unet_out": err}, {"unet_out": yard}


def case_vae_sd(name="vae_576x320", precision="fast"):
    """AutoencoderKL at the SD geometry (128, 256, 512, 512; mid-block attention d = 512 over 2 880 tokens) on two 576x320
    images vs the fp32 oracle's recorded posterior sample and decoded images (tests/golden/sd21_72x40.pt)."""
    from diffuman4d_amd.host import ops
    from diffuman4d_amd.host.vae import AutoencoderKL, VAEConfig
    from diffuman4d_amd.host.weights import random_state_dict, vae_param_shapes
    sys.path.insert(0, str(GOLDEN))
    import make_golden_sd21 as mk
    g = torch.load(GOLDEN / "sd21_72x40.pt")[name]
    cfg = VAEConfig.from_dict(g["config"])
    sd = sd_weights("vae", mk.VAE_SEED)
    _check_fixture_inputs("VAE weights", float(sum(v.float().abs().sum() for v in sd.values())), g["weights_checksum"])
    img, noise = mk.vae_inputs(g["n"], g["seed"])
    _check_fixture_inputs("VAE input", float(img.float().abs().sum()), g["img_checksum"])
    hv = AutoencoderKL(cfg, sd, "cuda", precision)
    z = hv.encode_scaled(img, noise)
    out = hv.decode_to_images(latents_nhwc(hv, g["z"].to(BF)))  # the fixture's decoder was fed the bf16-rounded latents
    return ({"latents": rel_l2(ops.nhwc_to_nchw(z), g["z"]), "images": rel_l2(out, g["images"])},
            {"latents": g["yard_z"], "images": g["yard_images"]})


def case_demo3d_sd21(precision="fast", matched=False):
    """BASELINE.json configs[0] END TO END on the judged geometry: `demo_3d` (configs/exp/demo_3d.yaml:3-10 + sampler/sliding_3d.yaml:
    48 cameras x 1 frame, 4 input cameras, window 12, stride 1, one alternation round => 12 steps per latent, 44 UNet calls of F = 16
    = CFG batch 32) through `sliding_iterative_denoise` (pipeline_diffuman4d.py:439-559) with the SD-2.1 UNet, the SD VAE and 576 x 320
    images: 96 VAE encodes, 44 window calls, 48 decodes.  Compared with the fp32 CPU oracle's latents (all 48 rows), its decoded RGB (six
    rows kept in the fixture as 16-bit fixed point) and -- bit for bit -- its timestep bookkeeping, all recorded in
    tests/golden/demo3d_sd21_72x40.pt by tests/golden/make_golden_demo3d.py together with the bf16-oracle yardsticks."""
    from diffuman4d_amd.host.pipeline import Diffuman4DPipeline
    from diffuman4d_amd.host.scheduler import DDIMConfig as HC, DDIMScheduler as HS
    from diffuman4d_amd.host.unet import UNetConfig, UNetMultiviewConditionModel
    from diffuman4d_amd.host.vae import AutoencoderKL, VAEConfig
    sys.path.insert(0, str(GOLDEN))
    import make_golden_demo3d as mk
    g = torch.load(GOLDEN / "demo3d_sd21_72x40.pt")
    usd, vsd = sd_weights("unet", mk.UNET_SEED), sd_weights("vae", mk.VAE_SEED)
    pv, pl, sk, cm = mk.task_inputs()
    noise = mk.task_noise()
    got, want = mk.checksums(pv, pl, sk, cm, noise, usd, vsd), g["checksums"]
    for k in ("pixel_values", "plucker", "skeletons", "cond_masks", "unet_weights", "vae_weights"):
        _check_fixture_inputs("demo3d " + k, got[k], want[k])
    for k in noise:
        _check_fixture_inputs("demo3d noise " + k, got["noise"][k], want["noise"][k])
    hp = Diffuman4DPipeline(AutoencoderKL(VAEConfig(), vsd, "cuda", precision), UNetMultiviewConditionModel(UNetConfig(), usd, "cuda", precision),
                            HS(HC()), "cuda")
    del usd, vsd
    t0 = time.time()
    out = hp.sliding_iterative_denoise(pixel_values=pv, plucker_embeds=pl, skeletons=sk, cond_masks=cm, latents=None, domain="spatial",
                                       timestep_indices=torch.zeros(mk.N_CAMS, dtype=torch.int64), noise=noise, **g["kw"])
    torch.cuda.synchronize()
    secs = time.time() - t0
    exact = torch.equal(out["timestep_indices"].cpu(), g["timestep_indices"]) and torch.equal(out["fully_denoised"].cpu(), g["fully_denoised"])
    if matched:  # the whole task against the rounding-matched oracle pipeline (MATCHED_BAND)
        mimg = g["matched_images_u16"].to(torch.int32).float() / 65535.0
        ref_img = g["images_u16"].to(torch.int32).float() / 65535.0
        res = {}
        for q, hip_q, ref_q, m_q, e_m in (("latents", out["latents"], g["latents"], g["matched_latents"], g["matched_vs_fp32_latents"]),
                                          ("images", out["images"][g["image_rows"]], ref_img, mimg, g["matched_vs_fp32_images"])):
            e_h, direct = rel_l2(hip_q, ref_q), rel_l2(hip_q, m_q)
            ratio, res[q] = matched_verdict(e_h, e_m, direct)
            print(f"    [demo_3d fast vs rounding-matched oracle, {q}] HIP vs fp32 {e_h:.3e}, matched vs fp32 {e_m:.3e} (ratio {ratio:.3f}, band "
                  f"{MATCHED_BAND}), HIP vs matched {direct:.3e} bookkeeping_exact={exact}", flush=True)
        del hp
        torch.cuda.empty_cache()
        return ({"bookkeeping": 1.0}, {"bookkeeping": 0.0}) if not exact else (res, {"latents": 0.0, "images": 0.0})
    ref_img = g["images_u16"].to(torch.int32).float() / 65535.0
    fd = g["fully_denoised"]
    e = {"latents": rel_l2(out["latents"], g["latents"]), "images": rel_l2(out["images"][g["image_rows"]], ref_img)}
    e_t = rel_l2(out["latents"].cpu()[fd], g["latents"][fd])
    # yardsticks of the bf16 pass of the generator; until that pass has run the bound falls back to the F = 16 UNet call's yardstick
    fallback = 1.22e-2
    y = {"latents": g.get("yard_latents", fallback), "images": g.get("yard_images", fallback)}
    print(f"    [demo_3d {precision}, SD-2.1 + SD VAE, 72x40, 44 calls, {secs:.1f}s] latents rel_l2={e['latents']:.3e} (targets only {e_t:.3e}; oracle-bf16 "
          f"{y['latents']:.3e}) images rel_l2={e['images']:.3e} (oracle-bf16 {y['images']:.3e}; north_star {NORTH_STAR:.0e}: "
          f"{'met' if e['images'] <= NORTH_STAR else 'NOT met'}) bookkeeping_exact={exact}", flush=True)
    del hp
    torch.cuda.empty_cache()
    if not exact:
        return {"bookkeeping": 1.0}, {"bookkeeping": 0.0}
    return e, y


def case_demo4dtiny_temporal(precision="fast"):
    """BASELINE.json configs[1], ONE WHOLE TEMPORAL TASK on the judged geometry: `demo_4d_tiny` (configs/exp/demo_4d_tiny.yaml:6-9 +
    sampler/sliding_fast.yaml: 16 frames, window 12, stride 2, 3 alternation rounds), the middle round: one target camera = 32 rows
    (sliding_iterative_sampler.py:112-118: 16 input frames + 16 target frames), targets entering at timestep index 6 with the grid's
    latents, 8 window calls of F = 24 frames = CFG batch 48 (pipeline_diffuman4d.py:504-518), 2 x 32 VAE encodes, decode of the four
    target rows the fixture keeps.  Compared with the fp32 CPU oracle's latents (all 32 rows), its decoded RGB (16-bit fixed point) and
    -- bit for bit -- its timestep bookkeeping (tests/golden/demo4dtiny_temporal_sd21_72x40.pt, made by
    tests/golden/make_golden_demo4d_tiny_temporal.py together with the bf16-oracle yardsticks).  Temporal tasks are one third of the
    judged job's window calls and F = 24 is where the fp16 precision is weakest on a single call."""
    from diffuman4d_amd.host.pipeline import Diffuman4DPipeline
    from diffuman4d_amd.host.scheduler import DDIMConfig as HC, DDIMScheduler as HS
    from diffuman4d_amd.host.unet import UNetConfig, UNetMultiviewConditionModel
    from diffuman4d_amd.host.vae import AutoencoderKL, VAEConfig
    sys.path.insert(0, str(GOLDEN))
    import make_golden_demo4d_tiny_temporal as mk
    g = torch.load(GOLDEN / "demo4dtiny_temporal_sd21_72x40.pt")
    usd, vsd = sd_weights("unet", mk.UNET_SEED), sd_weights("vae", mk.VAE_SEED)
    pv, pl, sk, cm = mk.task_inputs()
    noise, lat = mk.task_noise(), mk.grid_latents()
    got, want = mk.checksums(pv, pl, sk, cm, noise, lat, usd, vsd), g["checksums"]
    for k in ("pixel_values", "plucker", "skeletons", "cond_masks", "latents_in", "unet_weights", "vae_weights"):
        _check_fixture_inputs("demo4dtiny temporal " + k, got[k], want[k])
    for k in noise:
        _check_fixture_inputs("demo4dtiny temporal noise " + k, got["noise"][k], want["noise"][k])
    hp = Diffuman4DPipeline(AutoencoderKL(VAEConfig(), vsd, "cuda", precision), UNetMultiviewConditionModel(UNetConfig(), usd, "cuda", precision),
                            HS(HC()), "cuda")
    del usd, vsd
    t0 = time.time()
    out = hp.sliding_iterative_denoise(pixel_values=pv, plucker_embeds=pl, skeletons=sk, cond_masks=cm, latents=lat, domain="temporal",
                                       timestep_indices=mk.start_indices(), noise=noise, **g["kw"])
    torch.cuda.synchronize()
    secs = time.time() - t0
    exact = torch.equal(out["timestep_indices"].cpu(), g["timestep_indices"]) and torch.equal(out["fully_denoised"].cpu(), g["fully_denoised"])
    ref_img = g["images_u16"].to(torch.int32).float() / 65535.0
    e = {"latents": rel_l2(out["latents"], g["latents"]), "images": rel_l2(out["images"][g["image_rows"]], ref_img)}
    e_t = rel_l2(out["latents"].cpu()[mk.T:], g["latents"][mk.T:])
    y = {"latents": g["yard_latents"], "images": g["yard_images"]}
    print(f"    [demo_4d_tiny temporal task {precision}, SD-2.1 + SD VAE, 72x40, 8 calls of F = 24, {secs:.1f}s] latents rel_l2={e['latents']:.3e} (targets only "
          f"{e_t:.3e}; oracle-bf16 {y['latents']:.3e}) images rel_l2={e['images']:.3e} (oracle-bf16 {y['images']:.3e}; north_star {NORTH_STAR:.0e}: "
          f"{'met' if e['images'] <= NORTH_STAR else 'NOT met'}) bookkeeping_exact={exact}", flush=True)
    del hp
    torch.cuda.empty_cache()
    if not exact:
        return {"bookkeeping": 1.0}, {"bookkeeping": 0.0}
    return e, y


def case_vae_1024(name="vae_1024"):
    """AutoencoderKL at the SD geometry on ONE 1024 x 1024 image -- the reference's native image size (spatem_dataset.py:27-28 ->
    pipeline_diffuman4d.py:47-72, 553): mid-block attention over L = 16 384 tokens at d = 512, 1024^2 x 128-channel activations.
    Compared with the fp32 oracle's posterior sample and four bands of its decoded image (tests/golden/sd21_72x40.pt, made by
    make_golden_sd21.py vae1024).  Also asserted: the attention's fp32 logits stay query-blocked (<= 128 MB) and the whole encode +
    decode of one image peaks below 8 GB of device memory (no halo tiling is needed on a 288 GB part; the number is printed)."""
    from diffuman4d_amd.host import ops
    from diffuman4d_amd.host.vae import AutoencoderKL, VAEConfig
    from diffuman4d_amd.host.weights import random_state_dict, vae_param_shapes
    sys.path.insert(0, str(GOLDEN))
    import make_golden_sd21 as mk
    g = torch.load(GOLDEN / "sd21_72x40.pt")[name]
    cfg = VAEConfig.from_dict(g["config"])
    sd = sd_weights("vae", mk.VAE_SEED)
    _check_fixture_inputs("VAE weights", float(sum(v.float().abs().sum() for v in sd.values())), g["weights_checksum"])
    img, noise = mk.vae1024_inputs(g["seed"])
    _check_fixture_inputs("VAE 1024 input", float(img.float().abs().sum()), g["img_checksum"])
    hv = AutoencoderKL(cfg, sd, "cuda")
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    t0 = time.time()
    z = hv.encode_scaled(img, noise)
    torch.cuda.synchronize()
    t_enc = time.time() - t0
    t0 = time.time()
    out = hv.decode_to_images(ops.nchw_to_nhwc(g["z"].to(BF).contiguous().cuda()))
    torch.cuda.synchronize()
    t_dec = time.time() - t0
    peak = (torch.cuda.max_memory_allocated() - base) / 2 ** 30
    bands = torch.cat([out[..., a:b, :] for a, b in g["bands"]], dim=-2)
    ref = g["image_bands_u16"].to(torch.int32).float() / 65535.0
    print(f"    [vae 1024x1024] encode {t_enc * 1e3:.0f} ms, decode {t_dec * 1e3:.0f} ms (first call, incl. allocation), peak device memory above "
          f"the weights {peak:.2f} GiB, attention logits block {hv.mid_attention_block_bytes() / 2 ** 20:.0f} MiB", flush=True)
    assert hv.mid_attention_block_bytes() <= 128 * 2 ** 20 and peak < 8.0, (hv.mid_attention_block_bytes(), peak)
    return ({"latents": rel_l2(ops.nhwc_to_nchw(z), g["z"]), "images": rel_l2(bands, ref)},
            {"latents": g["yard_z"], "images": g["yard_images"]})


def case_resize(seed=3):
    import torch.nn.functional as F
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(2, 6, 64, 48, generator=g)
    ref_b = F.interpolate(x, size=(8, 6), mode="bilinear").to(BF)
    ref_n = F.interpolate(x, size=(8, 6), mode="nearest").to(BF)
    ob = ops.resize_to_nhwc(x.cuda(), (8, 6), "bilinear").permute(0, 3, 1, 2)
    on = ops.resize_to_nhwc(x.cuda(), (8, 6), "nearest").permute(0, 3, 1, 2)
    e1 = float((ob.float().cpu() - ref_b.float()).abs().max())
    e2 = float((on.float().cpu() - ref_n.float()).abs().max())
    # odd ratio
    x2 = torch.randn(1, 1, 50, 30, generator=g)
    r2 = F.interpolate(x2, size=(7, 5), mode="bilinear").to(BF)
    o2 = ops.resize_to_nhwc(x2.cuda(), (7, 5), "bilinear").permute(0, 3, 1, 2)
    e3 = rel_l2(o2, r2)
    r3 = F.interpolate(x2, size=(7, 5), mode="nearest").to(BF)
    o3 = ops.resize_to_nhwc(x2.cuda(), (7, 5), "nearest").permute(0, 3, 1, 2)
    e4 = float((o3.float().cpu() - r3.float()).abs().max())
    return max(e1, e2, e3, e4), 0.0


def synthetic_task(n, H, W, input_rows, seed=7):
    """Synthetic task tensors with the value ranges of spatem_dataset.py:191-228."""
    g = torch.Generator().manual_seed(seed)
    pv = torch.rand(n, 3, H, W, generator=g) * 2 - 1
    sk = -torch.ones(n, 3, H, W)
    sk[:, :, H // 4: H // 2, W // 4: W // 2] = torch.rand(n, 3, H // 4, W // 4, generator=g) * 2 - 1
    pl = torch.rand(n, 6, H, W, generator=g) * 2 - 1
    cm = torch.ones(n, 1, H, W)
    cm[input_rows] = 0.0
    return pv, pl, sk, cm


def case_pipeline(domain="spatial", n_cams=8, T=4, window=4, stride=2, rounds=1, steps=1, bidir=False, gs=2.0,
                  pred="epsilon", seed=11, pose=False, sched="ddim", sched_kw=None, precision="fast", matched=False):
    """One full task through sliding_iterative_denoise (VAE encode -> window sweep -> VAE decode).
    sched="dpm": DPM-Solver++ multistep -- the oracle keeps one stateful scheduler object per latent as the reference does
    (pipeline_diffuman4d.py:265-271, 420), the HIP path runs its planned coefficient rows (host/scheduler.py)."""
    from diffuman4d_amd.host.pipeline import Diffuman4DPipeline
    from oracle.pipeline import OraclePipeline
    if sched == "dpm":
        from diffuman4d_amd.host.scheduler import DPMSolverConfig as HC, DPMSolverMultistepScheduler as HS
        from oracle.dpmsolver import DPMSolverConfig as _OC, DPMSolverMultistepScheduler as _OS

        def HC_(**kw):
            return HC(**{**(sched_kw or {}), **kw})

        def DDIMConfig(**kw):
            return _OC(**{**(sched_kw or {}), **kw})
        DDIMScheduler, HCf = _OS, HC_
    else:
        from diffuman4d_amd.host.scheduler import DDIMConfig as HC, DDIMScheduler as HS
        from oracle.ddim import DDIMConfig, DDIMScheduler
        HCf = HC
    cfg_u, ou = make_unet(seed, **(dict(enable_pose_encoder=True, in_channels=11) if pose else {}))
    cfg_v, ov = make_vae(seed + 1)
    H, W = 64, 64
    if domain == "spatial":
        n, inputs = n_cams, [1, 5]
    else:
        n, inputs = 2 * T, list(range(T))
    pv, pl, sk, cm = synthetic_task(n, H, W, inputs, seed)
    g = torch.Generator().manual_seed(seed + 2)
    noise = {k: torch.randn(n, 4, H // 8, W // 8, generator=g).to(BF) for k in ("pixel", "skeleton", "latents")}
    tidx = torch.zeros(n, dtype=torch.int64)
    kw = dict(window_size=window, sliding_stride=stride, sliding_shift=0, bidirectional=bidir,
              num_denoising_steps=steps, alternation_rounds=rounds, guidance_scale=gs)
    op = OraclePipeline(ov, ou, DDIMScheduler(DDIMConfig(prediction_type=pred)), torch.float32)
    noise_f = {k: v.float() for k, v in noise.items()}
    ref = op.sliding_iterative_denoise(pv, pl, sk, cm, None, domain, tidx, noise_f, **kw)
    hp = Diffuman4DPipeline(hip_vae(cfg_v, ov, precision), hip_unet(cfg_u, ou, precision), HS(HCf(prediction_type=pred)), "cuda")
    out = hp.sliding_iterative_denoise(pixel_values=pv, plucker_embeds=pl, skeletons=sk, cond_masks=cm, latents=None,
                                       domain=domain, timestep_indices=tidx, noise=noise, **kw)
    exact = bool((out["timestep_indices"].cpu() == ref["timestep_indices"]).all()) and \
        bool((out["fully_denoised"].cpu() == ref["fully_denoised"]).all())
    if matched:  # HIP fast precision vs the rounding-matched oracle pipeline, run on the spot (MATCHED_BAND)
        from oracle import matched as mo
        assert sched == "ddim" and not pose
        mref = mo.MatchedPipeline(ov, ou, DDIMScheduler(DDIMConfig(prediction_type=pred))).sliding_iterative_denoise(
            pv, pl, sk, cm, None, domain, tidx, noise, **kw)
        res = {}
        for q in ("latents", "images"):
            e_h, e_m, direct = rel_l2(out[q], ref[q]), rel_l2(mref[q], ref[q]), rel_l2(out[q], mref[q])
            ratio, res[q] = matched_verdict(e_h, e_m, direct)
            print(f"    [pipeline {domain} fast vs rounding-matched oracle, {q}] HIP vs fp32 {e_h:.3e}, matched vs fp32 {e_m:.3e} (ratio "
                  f"{ratio:.3f}, band {MATCHED_BAND}), HIP vs matched {direct:.3e}", flush=True)
        return res, {"latents": 0.0, "images": 0.0}
    e_lat = rel_l2(out["latents"], ref["latents"])
    e_img = rel_l2(out["images"], ref["images"])
    # yardstick: the oracle in bf16 (what the reference computes) vs the fp32 oracle
    opb = OraclePipeline(ov, ou, DDIMScheduler(DDIMConfig(prediction_type=pred)), BF)
    refb = opb.sliding_iterative_denoise(pv, pl, sk, cm, None, domain, tidx, noise, **kw)
    ov.float(), ou.float()
    y_lat, y_img = rel_l2(refb["latents"], ref["latents"]), rel_l2(refb["images"], ref["images"])
    print(f"    [pipeline {domain} {precision}] latents rel_l2={e_lat:.3e} (oracle-bf16 {y_lat:.3e}) images rel_l2={e_img:.3e} "
          f"(oracle-bf16 {y_img:.3e}; north_star {NORTH_STAR:.0e}: {'met' if e_img <= NORTH_STAR else 'NOT met'}) "
          f"bookkeeping_exact={exact}", flush=True)
    if not exact:
        return {"bookkeeping": 1.0}, {"bookkeeping": 0.0}
    return {"latents": e_lat, "images": e_img}, {"latents": y_lat, "images": y_img}


def case_pipeline_cache_lazy(seed=31):
    """Pipeline extensions vs the strict path on the same inputs and injected noise: VAE moments served from the
    cache (second call: every image is a hit) and decode restricted to fully denoised rows must give bitwise
    identical latents, identical images on the decoded rows and zero images elsewhere."""
    from diffuman4d_amd.host.pipeline import Diffuman4DPipeline
    from diffuman4d_amd.host.scheduler import DDIMConfig as HC, DDIMScheduler as HS
    cfg_u, ou = make_unet(seed)
    cfg_v, ov = make_vae(seed + 1)
    n, inputs = 8, [1, 5]
    pv, pl, sk, cm = synthetic_task(n, 64, 64, inputs, seed)
    g = torch.Generator().manual_seed(seed + 2)
    noise = {k: torch.randn(n, 4, 8, 8, generator=g).to(BF) for k in ("pixel", "skeleton", "latents")}
    tidx = torch.zeros(n, dtype=torch.int64)
    # window 4, stride 2, ONE round of 1: two steps per latent = fully denoised after this call
    kw = dict(pixel_values=pv, plucker_embeds=pl, skeletons=sk, cond_masks=cm, latents=None, domain="spatial",
              timestep_indices=tidx, window_size=4, sliding_stride=2, sliding_shift=0, bidirectional=False,
              num_denoising_steps=1, alternation_rounds=1, guidance_scale=2.0, noise=noise)
    hp = Diffuman4DPipeline(hip_vae(cfg_v, ov), hip_unet(cfg_u, ou), HS(HC()), "cuda")
    ref = hp.sliding_iterative_denoise(**kw)
    keys = [("cam%02d" % i, "000000") for i in range(n)]
    bad = 0.0
    for rep in range(2):  # rep 0 fills the cache, rep 1 is served from it
        out = hp.sliding_iterative_denoise(cache_keys=keys, decode="denoised", **kw)
        fd = out["fully_denoised"]
        ok = torch.equal(out["latents"], ref["latents"]) and torch.equal(out["timestep_indices"], ref["timestep_indices"])
        ok = ok and torch.equal(out["images"][fd], ref["images"][fd]) and bool((out["images"][~fd] == 0).all())
        ok = ok and int(fd.sum()) == n - len(inputs) and len(hp._vae_cache["pixel"]) == n
        bad += 0.0 if ok else 1.0
    # a later round: latents handed back, nothing fully denoised -> no decode at all
    kw2 = dict(kw, alternation_rounds=3, latents=None)
    out = hp.sliding_iterative_denoise(cache_keys=keys, decode="denoised", **kw2)
    if bool(out["fully_denoised"].any()) or not bool((out["images"] == 0).all()):
        bad += 1.0
    return bad, 0.0


def case_pipeline_prune(domain="spatial", seed=41, precision="fast"):
    """prune_cond_rows extension: the UNet tail after the last 3-D attention runs only for non-conditioning rows.  The
    latents must agree with the strict path to within kernel-configuration noise (other tile shapes for the smaller
    batch) and the bookkeeping exactly."""
    from diffuman4d_amd.host.pipeline import Diffuman4DPipeline
    from diffuman4d_amd.host.scheduler import DDIMConfig as HC, DDIMScheduler as HS
    cfg_u, ou = make_unet(seed)
    cfg_v, ov = make_vae(seed + 1)
    n, inputs = (8, [1, 5]) if domain == "spatial" else (8, [0, 1, 2, 3])
    pv, pl, sk, cm = synthetic_task(n, 64, 64, inputs, seed)
    g = torch.Generator().manual_seed(seed + 2)
    noise = {k: torch.randn(n, 4, 8, 8, generator=g).to(BF) for k in ("pixel", "skeleton", "latents")}
    kw = dict(pixel_values=pv, plucker_embeds=pl, skeletons=sk, cond_masks=cm, latents=None, domain=domain,
              timestep_indices=torch.zeros(n, dtype=torch.int64), window_size=4, sliding_stride=2 if domain == "spatial" else 1,
              sliding_shift=0, bidirectional=False, num_denoising_steps=1, alternation_rounds=1, guidance_scale=2.0, noise=noise)
    hp = Diffuman4DPipeline(hip_vae(cfg_v, ov, precision), hip_unet(cfg_u, ou, precision), HS(HC()), "cuda")
    ref = hp.sliding_iterative_denoise(**kw)
    hp.prune_cond_rows = True
    out = hp.sliding_iterative_denoise(**kw)
    exact = torch.equal(out["timestep_indices"], ref["timestep_indices"]) and torch.equal(out["fully_denoised"], ref["fully_denoised"])
    err = max(rel_l2(out["latents"], ref["latents"]), rel_l2(out["images"], ref["images"]))
    return (err if exact else 1.0), 0.0


def case_pipeline_plucker_on_device(seed=51):
    """the Pluecker conditioning evaluated on the device at latent resolution from the cameras vs the
    reference route (full-resolution fp32 maps built on the host, shipped, resized on the device) -- same task, same
    noise.  The two maps differ by isolated bf16 ulps (tests/opcheck.py::plucker_*), so the results agree to noise."""
    from diffuman4d_amd.host.pipeline import Diffuman4DPipeline
    from diffuman4d_amd.host.scheduler import DDIMConfig as HC, DDIMScheduler as HS
    from oracle import plucker as op
    from test_plucker import cameras
    cfg_u, ou = make_unet(seed)
    cfg_v, ov = make_vae(seed + 1)
    n, inputs, H, W = 8, [1, 5], 64, 64
    pv, _, sk, cm = synthetic_task(n, H, W, inputs, seed)
    Ks, poses = cameras(n, H, W, seed)
    poses = op.calc_relative_poses(poses)
    pl = op.calc_plucker_embeds(H, W, Ks, poses)
    g = torch.Generator().manual_seed(seed + 2)
    noise = {k: torch.randn(n, 4, 8, 8, generator=g).to(BF) for k in ("pixel", "skeleton", "latents")}
    kw = dict(pixel_values=pv, skeletons=sk, cond_masks=cm, latents=None, domain="spatial",
              timestep_indices=torch.zeros(n, dtype=torch.int64), window_size=4, sliding_stride=2, sliding_shift=0,
              bidirectional=False, num_denoising_steps=1, alternation_rounds=1, guidance_scale=2.0, noise=noise)
    hp = Diffuman4DPipeline(hip_vae(cfg_v, ov), hip_unet(cfg_u, ou), HS(HC()), "cuda")
    ref = hp.sliding_iterative_denoise(plucker_embeds=pl, **kw)
    out = hp.sliding_iterative_denoise(plucker_embeds=None, cameras={"Ks": Ks, "poses": poses, "image_size": (H, W)}, **kw)
    exact = torch.equal(out["timestep_indices"], ref["timestep_indices"]) and torch.equal(out["fully_denoised"], ref["fully_denoised"])
    err = max(rel_l2(out["latents"], ref["latents"]), rel_l2(out["images"], ref["images"]))
    return (err if exact else 1.0), 0.0


def _golden_setup(name):
    """(fixture, host scheduler, oracle scheduler factory) of one golden-pipeline case."""
    gdir = Path(__file__).resolve().parent / "golden"
    dpm = name.startswith("dpm_")  # fixture of the reference pipeline run with one stateful DPM-Solver++ object per latent
    multistep = name.startswith(("unipc", "deis", "pndm"))  # ... with one stateful UniPC / DEIS / PNDM object per latent (make_golden.py multistep)
    if multistep:
        from diffuman4d_amd.host import scheduler as hs
        from oracle import multistep as ms
        g = torch.load(gdir / "pipeline_multistep.pt")[name]
        sc, kind = g["case"]["sched"], g["case"]["kind"]
        if kind == "pndm":
            host_sched, oracle_sched = hs.PNDMScheduler(hs.PNDMConfig.from_dict(sc)), (lambda: ms.PNDMScheduler(ms.PNDMConfig(**sc)))
        else:
            host_sched = (hs.UniPCMultistepScheduler(hs.UniPCConfig.from_dict(sc)) if kind == "unipc" else hs.DEISMultistepScheduler(hs.DEISConfig.from_dict(sc)))
            oracle_sched = (lambda: ms.UniPCMultistepScheduler(ms.UniPCConfig(**sc))) if kind == "unipc" else (lambda: ms.DEISMultistepScheduler(ms.DEISConfig(**sc)))
    elif dpm:
        from diffuman4d_amd.host.scheduler import DPMSolverConfig as HC, DPMSolverMultistepScheduler as HS
        from oracle.dpmsolver import DPMSolverConfig as OC_, DPMSolverMultistepScheduler as OS_
        g = torch.load(gdir / "pipeline_dpm.pt")[name]
        host_sched, oracle_sched = HS(HC(**g["case"]["sched"])), (lambda: OS_(OC_(**g["case"]["sched"])))
    else:
        from diffuman4d_amd.host.scheduler import DDIMConfig as HC, DDIMScheduler as HS
        from oracle.ddim import DDIMConfig, DDIMScheduler
        g = torch.load(gdir / "pose_encoder.pt")["pipeline"] if name == "pose_encoder" else torch.load(gdir / "pipeline_sliding.pt")[name]
        host_sched = HS(HC(prediction_type=g["case"]["pred"]))
        oracle_sched = lambda: DDIMScheduler(DDIMConfig(prediction_type=g["case"]["pred"]))  # noqa: E731
    return g, host_sched, oracle_sched


def case_task_stack(name="spatial", precision="fast", copies=3, global_rng=False):
    """runner.task_batch: `sliding_iterative_denoise_stack` (several tasks of a round through shared window calls, VAE encode and decode
    included) returns, task by task, what `sliding_iterative_denoise` returns for each task alone -- bit for bit on the GPU.  Task 0 is
    the golden fixture's task (so the stack is also held to the reference pipeline's output), the others differ in images, cameras and
    noise.  global_rng: no injected noise, the draws come from the device's global generator, reseeded before the serial and before the
    stacked pass (the stack prepares its tasks in list order, so the draw sequence is that of the serial order)."""
    from diffuman4d_amd.host.pipeline import Diffuman4DPipeline
    g, host_sched, _ = _golden_setup(name)
    c, seeds = g["case"], g["seeds"]
    cfg_u, ou = make_unet(seeds["unet"], **g.get("cfg_kw", {}))
    cfg_v, ov = make_vae(seeds["vae"])
    hp = Diffuman4DPipeline(hip_vae(cfg_v, ov, precision), hip_unet(cfg_u, ou, precision), host_sched, "cuda")
    dev = hp.device
    common = dict(domain=c["domain"], **c["kw"])
    tasks = []
    for k in range(copies):
        pv, pl, sk, cm = synthetic_task(c["n"], 64, 64, c["inputs"], seeds["task"] + 17 * k)
        if k == 0:
            noise = {q: v.to(hp.dtype) for q, v in g["noise"].items()}
            lat_in = g["latents_in"].to(hp.dtype) if g["latents_in"] is not None else None
        else:
            gen = torch.Generator().manual_seed(seeds["task"] + 1000 + k)
            noise = {q: torch.randn(v.shape, generator=gen).to(BF).to(hp.dtype) for q, v in g["noise"].items()}
            lat_in = None if g["latents_in"] is None else torch.randn(g["latents_in"].shape, generator=gen).to(BF).to(hp.dtype)
        t = dict(pixel_values=pv, plucker_embeds=pl, skeletons=sk, cond_masks=cm, latents=lat_in, timestep_indices=g["timestep_indices_in"])
        if not global_rng:
            t["noise"] = noise
        tasks.append(t)

    def seed():
        torch.manual_seed(4321)
        if dev.type == "cuda":
            torch.cuda.manual_seed_all(4321)

    seed()
    alone = [hp.sliding_iterative_denoise(**t, **common) for t in tasks]
    seed()
    stacked = hp.sliding_iterative_denoise_stack(tasks, **common)
    worst = 0.0
    for a, b in zip(alone, stacked):
        assert torch.equal(a["timestep_indices"], b["timestep_indices"]) and torch.equal(a["fully_denoised"], b["fully_denoised"])
        assert bool(torch.isfinite(b["latents"].float()).all()) and bool(torch.isfinite(b["images"].float()).all())
        for q in ("latents", "images"):
            worst = max(worst, float((a[q].float() - b[q].float()).abs().max()) / max(float(a[q].float().abs().max()), 1e-30))
    assert not torch.equal(stacked[0]["latents"], stacked[1]["latents"])  # the tasks are different tasks
    if not global_rng:  # task 0 of the stack against the reference pipeline's own output
        e_lat = rel_l2(stacked[0]["latents"], g["latents"])
        print(f"    [task stack {name} {precision} x{copies}] worst |stack - alone| / max = {worst:.3e}; task 0 vs the reference fixture: latents rel_l2 {e_lat:.3e}", flush=True)
        bound = {"fast": 0.1, "fp16": FP16_BOUNDS["latents"], "parity": 1e-4}[precision]
        assert e_lat <= bound, (e_lat, bound)
    return worst, 0.0


def case_golden_pipeline(name, precision="fast"):
    """HIP sliding_iterative_denoise vs the output of the REFERENCE's own pipeline code (committed
    fixture tests/golden/pipeline_sliding.pt, fp32): same seeded weights, task tensors and noise."""
    from diffuman4d_amd.host.pipeline import Diffuman4DPipeline
    g, host_sched, oracle_sched = _golden_setup(name)
    c, seeds = g["case"], g["seeds"]
    cfg_u, ou = make_unet(seeds["unet"], **g.get("cfg_kw", {}))
    cfg_v, ov = make_vae(seeds["vae"])
    pv, pl, sk, cm = synthetic_task(c["n"], 64, 64, c["inputs"], seeds["task"])
    hp = Diffuman4DPipeline(hip_vae(cfg_v, ov, precision), hip_unet(cfg_u, ou, precision), host_sched, "cuda")
    lat_in = g["latents_in"].to(hp.dtype) if g["latents_in"] is not None else None  # parity precision takes the fixture's fp32 draws as they are
    out = hp.sliding_iterative_denoise(pixel_values=pv, plucker_embeds=pl, skeletons=sk, cond_masks=cm, latents=lat_in,
                                       domain=c["domain"], timestep_indices=g["timestep_indices_in"],
                                       noise={k: v.to(hp.dtype) for k, v in g["noise"].items()}, **c["kw"])
    exact = torch.equal(out["timestep_indices"].cpu(), g["timestep_indices"]) and \
        torch.equal(out["fully_denoised"].cpu(), g["fully_denoised"])
    e_lat, e_img = rel_l2(out["latents"], g["latents"]), rel_l2(out["images"], g["images"])
    if precision in ("parity", "fp16"):  # fixed bound against the reference pipeline's own fp32 output
        print(f"    [golden {name} {precision}] latents rel_l2={e_lat:.3e} images rel_l2={e_img:.3e} (fixture images are fp16) bookkeeping_exact={exact}", flush=True)
        return ({"bookkeeping": 1.0}, {"bookkeeping": 0.0}) if not exact else ({"latents": e_lat, "images": e_img}, {"latents": 0.0, "images": 0.0})
    # yardstick: the oracle run in bf16 on the same task, measured against the same fixture (= the reference's fp32 output)
    from oracle.pipeline import OraclePipeline
    ov.to(BF), ou.to(BF)
    opb = OraclePipeline(ov, ou, oracle_sched(), BF)
    refb = opb.sliding_iterative_denoise(pv, pl, sk, cm, None if lat_in is None else lat_in.to(BF), c["domain"], g["timestep_indices_in"],
                                         {k: v.to(BF) for k, v in g["noise"].items()}, **c["kw"])
    y_lat, y_img = rel_l2(refb["latents"], g["latents"]), rel_l2(refb["images"], g["images"])
    print(f"    [golden {name}] latents rel_l2={e_lat:.3e} (oracle-bf16 {y_lat:.3e}) images rel_l2={e_img:.3e} "
          f"(oracle-bf16 {y_img:.3e}) bookkeeping_exact={exact}", flush=True)
    if not exact:
        return {"bookkeeping": 1.0}, {"bookkeeping": 0.0}
    return {"latents": e_lat, "images": e_img}, {"latents": y_lat, "images": y_img}


def _replay_report(tag, trace, secs):
    from oracle import replay
    t0 = time.time()
    stats = replay.replay_trace(trace)
    unknown = sorted(n for n in stats if n not in replay.REPLAY)
    worst = max((st["worst"] for st in stats.values()), default=0.0)
    rows = "; ".join(f"{n} {st['checked']}/{st['launches']} worst {st['worst']:.1e} mean {st['mean']:.1e}" for n, st in sorted(stats.items()))
    print(f"    [{tag}: {len(trace)} launches traced in {secs * 1e3:.0f} ms, replayed on the CPU in {time.time() - t0:.1f}s] {rows}", flush=True)
    bad = {n: st["where"] for n, st in stats.items() if st["worst"] > REPLAY_TOL}
    if bad:
        print(f"    launches above {REPLAY_TOL:.0e}: {bad}", flush=True)
    assert not unknown, f"traced operators without a replay function: {unknown}"
    return worst


def case_opreplay_unet(sd21=False, num_frames=4, tem=True, domain="temporal"):
    """EVERY launch of one HIP UNet call (fast precision) recomputed in fp64 from the tensors the device was given (oracle/replay.py):
    the tight model-level check of the fast path.  sd21: the judged call (SD-2.1 geometry, 72 x 40, F = 16, CFG batch 32: ~450 launches)."""
    from diffuman4d_amd.host import ops
    if sd21:
        from diffuman4d_amd.host.unet import UNetConfig, UNetMultiviewConditionModel
        from diffuman4d_amd.host.weights import random_state_dict, unet_param_shapes
        sys.path.insert(0, str(GOLDEN))
        import make_golden_sd21 as mk
        cfg = UNetConfig()
        hm = UNetMultiviewConditionModel(cfg, sd_weights("unet", mk.UNET_SEED), "cuda")
        x, t = mk.unet_inputs(16, 4, 101)
        num_frames, domain = 16, "spatial"
    else:
        cfg, om = make_unet(0, enable_tem_embeds=tem)
        hm = hip_unet(cfg, om)
        g = torch.Generator().manual_seed(3)
        x = torch.randn(2 * num_frames, cfg.in_channels, 16, 8, generator=g).to(BF)
        t = torch.randint(0, 1000, (2 * num_frames,), generator=g)
    xin = unet_sample(hm, x)
    hm(xin, t.float().cuda(), domains=[domain] * 2, num_frames=num_frames)  # warm: allocator, lazily prepared tables
    torch.cuda.synchronize()
    ops.TRACE = trace = []
    try:
        t0 = time.time()
        hm(xin, t.float().cuda(), domains=[domain] * 2, num_frames=num_frames)
        torch.cuda.synchronize()
        secs = time.time() - t0
    finally:
        ops.TRACE = None
    worst = _replay_report("UNet call, SD-2.1 geometry 72x40 F=16" if sd21 else "UNet call, small geometry", trace, secs)
    del hm, trace
    torch.cuda.empty_cache()
    return worst, 0.0


def case_opreplay_pipeline(domain="spatial", seed=11):
    """The same for a whole task on the small geometry: VAE encode, every window call, VAE decode."""
    from diffuman4d_amd.host import ops
    from diffuman4d_amd.host.pipeline import Diffuman4DPipeline
    from diffuman4d_amd.host.scheduler import DDIMConfig as HC, DDIMScheduler as HS
    cfg_u, ou = make_unet(seed)
    cfg_v, ov = make_vae(seed + 1)
    n, inputs = (8, [1, 5]) if domain == "spatial" else (8, [0, 1, 2, 3])
    pv, pl, sk, cm = synthetic_task(n, 64, 64, inputs, seed)
    g = torch.Generator().manual_seed(seed + 2)
    noise = {k: torch.randn(n, 4, 8, 8, generator=g).to(BF) for k in ("pixel", "skeleton", "latents")}
    hp = Diffuman4DPipeline(hip_vae(cfg_v, ov), hip_unet(cfg_u, ou), HS(HC()), "cuda")
    ops.TRACE = trace = []
    try:
        t0 = time.time()
        hp.sliding_iterative_denoise(pixel_values=pv, plucker_embeds=pl, skeletons=sk, cond_masks=cm, latents=None, domain=domain,
                                     timestep_indices=torch.zeros(n, dtype=torch.int64), noise=noise, window_size=4,
                                     sliding_stride=2 if domain == "spatial" else 1, sliding_shift=0, bidirectional=False,
                                     num_denoising_steps=1, alternation_rounds=1, guidance_scale=2.0)
        torch.cuda.synchronize()
        secs = time.time() - t0
    finally:
        ops.TRACE = None
    return _replay_report(f"whole {domain} task, small geometry (VAE + window calls)", trace, secs), 0.0


def case_opreplay_vae_sd():
    """SD VAE (128, 256, 512, 512) on one 576 x 320 image: encode + decode, every launch replayed."""
    from diffuman4d_amd.host import ops
    from diffuman4d_amd.host.vae import AutoencoderKL, VAEConfig
    from diffuman4d_amd.host.weights import random_state_dict, vae_param_shapes
    sys.path.insert(0, str(GOLDEN))
    import make_golden_sd21 as mk
    cfg = VAEConfig()
    hv = AutoencoderKL(cfg, sd_weights("vae", mk.VAE_SEED), "cuda")
    img, noise = mk.vae_inputs(1, 5)
    ops.TRACE = trace = []
    try:
        t0 = time.time()
        z = hv.encode_scaled(img, noise)
        hv.decode_to_images(z)
        torch.cuda.synchronize()
        secs = time.time() - t0
    finally:
        ops.TRACE = None
    return _replay_report("SD VAE encode + decode of one 576x320 image", trace, secs), 0.0


def case_multiround_sd21(precision="fast"):
    """A multi-round job THROUGH THE SAMPLER at the judged geometry: 8 cameras x 4 frames, window 4, stride 2, 3 alternation rounds
    (spatial -> temporal -> spatial: 14 tasks, 36 window calls, 6 steps per latent; sliding_iterative_sampler.py:192-212,
    sampling_runner.py:18-62) with the SD-2.1 UNet + SD VAE on 576 x 320 images: this repo's SlidingIterativeSampler + SamplingRunner
    drive the HIP pipeline; the final grid, the decoded RGB of three cells and -- bit for bit -- the timestep grid are compared
    with what the fp32 CPU oracle pipeline gave under the reference's control flow (tests/golden/multiround_sd21_72x40.pt, made by
    tests/golden/make_golden_multiround.py; same per-call random draws on both sides)."""
    from diffuman4d_amd.host.pipeline import Diffuman4DPipeline
    from diffuman4d_amd.host.runner import SamplingRunner
    from diffuman4d_amd.host.sampler import SlidingIterativeSampler
    from diffuman4d_amd.host.scheduler import DDIMConfig as HC, DDIMScheduler as HS
    from diffuman4d_amd.host.unet import UNetConfig, UNetMultiviewConditionModel
    from diffuman4d_amd.host.vae import AutoencoderKL, VAEConfig
    sys.path.insert(0, str(GOLDEN))
    import make_golden_demo3d as mk3
    import make_golden_multiround as mk
    g = torch.load(GOLDEN / "multiround_sd21_72x40.pt")
    usd, vsd = sd_weights("unet", mk3.UNET_SEED), sd_weights("vae", mk3.VAE_SEED)
    f = lambda t: float(t.float().abs().sum())  # noqa: E731
    _check_fixture_inputs("multiround UNet weights", float(sum(f(v) for v in usd.values())), g["checksums"]["unet_weights"])
    _check_fixture_inputs("multiround VAE weights", float(sum(f(v) for v in vsd.values())), g["checksums"]["vae_weights"])
    hp = Diffuman4DPipeline(AutoencoderKL(VAEConfig(), vsd, "cuda", precision), UNetMultiviewConditionModel(UNetConfig(), usd, "cuda", precision),
                            HS(HC()), "cuda")
    del usd, vsd
    # tasks run in the sampler's order: 4 spatial (one per frame), 6 temporal, 4 spatial; the kept cells belong to the last round,
    # whose task of frame f is call 10 + f and whose rows are the cameras in label order
    cells = [tuple(c) for c in g["image_cells"]]
    wrapped = mk.SeededNoise(hp, oracle=False, keep_images_of={10 + int(f) for _, f in cells})
    own = SlidingIterativeSampler(mk.dataset(), [wrapped], "/tmp/dm4d_multiround_unused", **g["kw"])
    first = own.dataset.get_item("synthetic", own.spa_labels, [own.tem_labels[0]], own.input_spa_labels)
    for k, key in (("pixel_values", "pixel_values"), ("plucker", "plucker_embeds"), ("skeletons", "skeletons")):
        _check_fixture_inputs("multiround dataset " + k, f(first[key]), g["checksums"][k])
    own.result_writer = None  # nothing is written (and the runner's completeness check of the files is skipped)
    t0 = time.time()
    # one stream: the draws come in the job's order; the runner's default task_batch: the rounds run as stacks of two tasks sharing their
    # window calls (4 spatial -> 2 + 2, 6 temporal -> 2 + 2 + 2), and the job is still held to the task-by-task fixture
    SamplingRunner(own, prefetch_depth=0, writers=1, gpu_streams=1).inference()
    torch.cuda.synchronize()
    secs = time.time() - t0
    lat = torch.stack([torch.stack([own.latents[c][fr].float().cpu() for fr in own.tem_labels]) for c in own.spa_labels])
    idx = torch.tensor([[own.timestep_indices[c][fr] for fr in own.tem_labels] for c in own.spa_labels])
    exact = torch.equal(idx, g["timestep_indices"])
    images = torch.stack([wrapped.images[10 + int(fr)][own.spa_labels.index(c)] for c, fr in cells])
    ref_img = g["images_u16"].to(torch.int32).float() / 65535.0
    tgt = g["timestep_indices"] > 0
    e = {"latents": rel_l2(lat[tgt], g["latents"][tgt]), "images": rel_l2(images, ref_img)}
    y = {"latents": g.get("yard_latents_targets", 1.5e-2), "images": g.get("yard_images", 1.5e-2)}
    print(f"    [multi-round job {precision}: 8 x 4 grid, 3 rounds, 14 tasks / 36 calls through the sampler, {secs:.1f}s] target latents rel_l2="
          f"{e['latents']:.3e} (oracle-bf16 {y['latents']:.3e}) images rel_l2={e['images']:.3e} (oracle-bf16 {y['images']:.3e}; north_star "
          f"{NORTH_STAR:.0e}: {'met' if e['images'] <= NORTH_STAR else 'NOT met'}) bookkeeping_exact={exact}", flush=True)
    del hp, own
    torch.cuda.empty_cache()
    if not exact:
        return {"bookkeeping": 1.0}, {"bookkeeping": 0.0}
    return e, y


CASES = {
    "unet_spatial": (case_unet, dict(num_frames=4, cfg_batch=2)),
    "unet_temporal_temb": (case_unet, dict(num_frames=4, cfg_batch=2, tem=True, domain="temporal")),
    "unet_2d_only": (case_unet, dict(num_frames=1, cfg_batch=3, h=8, w=8)),
    "unet_frame_shard_p4": (case_unet_frame_shard, dict(P=4, num_frames=8)),
    "unet_frame_shard_p8": (case_unet_frame_shard, dict(P=8, num_frames=8, tem=False)),
    "pipeline_shard_rccl_world1": (case_pipeline_shard_world1, dict()),
    "task_batching_spatial": (case_task_batching, dict(domain="spatial")),
    "task_batching_temporal_x3": (case_task_batching, dict(domain="temporal", copies=3)),
    "task_batching_dpm": (case_task_batching, dict(domain="spatial", sched="dpm")),
    # runner.task_batch: whole tasks (VAE encode, window calls, decode) through pipeline.sliding_iterative_denoise_stack
    "task_stack_spatial": (case_task_stack, dict(name="spatial")),
    "task_stack_temporal_v_x2": (case_task_stack, dict(name="temporal_v", copies=2)),
    "task_stack_round2_shift": (case_task_stack, dict(name="round2_shift")),
    "task_stack_pose_encoder": (case_task_stack, dict(name="pose_encoder", copies=2)),
    "task_stack_dpm_heun_round2": (case_task_stack, dict(name="dpm_temporal_v_heun_round2")),
    "task_stack_unipc_round2": (case_task_stack, dict(name="unipc_temporal_v_bh1_round2", copies=2)),
    "task_stack_global_rng": (case_task_stack, dict(name="spatial", global_rng=True)),
    "fp16_task_stack_spatial": (case_task_stack, dict(name="spatial", precision="fp16")),
    "par_task_stack_temporal_v": (case_task_stack, dict(name="temporal_v", precision="parity", copies=2)),
    "unet_pose_encoder": (case_unet, dict(num_frames=4, cfg_batch=2, pose=True)),
    "pipeline_pose_encoder": (case_pipeline, dict(domain="spatial", pose=True)),
    "pipeline_cache_lazy_decode": (case_pipeline_cache_lazy, dict()),
    "pipeline_prune_cond_rows": (case_pipeline_prune, dict(domain="spatial")),
    "pipeline_prune_cond_rows_temporal": (case_pipeline_prune, dict(domain="temporal")),
    "pipeline_plucker_on_device": (case_pipeline_plucker_on_device, dict()),
    "vae": (case_vae, dict()),
    # an image whose latent area (33 x 41 = 1353) is not a multiple of 32, nor of 4: padded key axis in the mid-block attention
    "vae_odd_latent_area": (case_vae, dict(h=264, w=328)),
    "resize": (case_resize, dict()),
    # stateful scheduler: DPM-Solver++ 2M (first-order first / final steps, second order in between), two denoising steps per
    # window so that latents carry history inside a window and across windows
    # (the spatial bidirectional DPM configuration: golden_dpm_spatial_bidir, against the reference pipeline's own output; its 20-second
    #  oracle-on-the-spot twin was dropped in round 6 with pipeline_bidir_nocfg -> golden_bidir_nocfg, to keep the GPU suite short)
    "pipeline_dpm_temporal_v_heun": (case_pipeline, dict(domain="temporal", T=4, window=4, stride=1, pred="v_prediction", sched="dpm",
                                                         sched_kw=dict(solver_type="heun", final_sigmas_type="sigma_min",
                                                                       timestep_spacing="leading", steps_offset=1))),
    "pipeline_spatial": (case_pipeline, dict(domain="spatial")),
    "pipeline_temporal_v": (case_pipeline, dict(domain="temporal", T=4, window=4, stride=1, pred="v_prediction")),
    "golden_spatial": (case_golden_pipeline, dict(name="spatial")),
    "golden_temporal_v": (case_golden_pipeline, dict(name="temporal_v")),
    "golden_bidir_nocfg": (case_golden_pipeline, dict(name="bidir_nocfg")),
    "golden_round2_shift": (case_golden_pipeline, dict(name="round2_shift")),
    "golden_pose_encoder": (case_golden_pipeline, dict(name="pose_encoder")),
    "golden_dpm_spatial_bidir": (case_golden_pipeline, dict(name="dpm_spatial_bidir")),
    "golden_dpm_temporal_v_heun_round2": (case_golden_pipeline, dict(name="dpm_temporal_v_heun_round2")),
    # UniPC (with its corrector) and DEIS: the reference pipeline with one stateful object per latent vs the planned 16-float rows
    "golden_unipc_spatial_bidir": (case_golden_pipeline, dict(name="unipc_spatial_bidir")),
    "golden_unipc_temporal_v_bh1_round2": (case_golden_pipeline, dict(name="unipc_temporal_v_bh1_round2")),
    "golden_deis3_spatial_bidir": (case_golden_pipeline, dict(name="deis3_spatial_bidir")),
    "golden_deis2_temporal_v_round2": (case_golden_pipeline, dict(name="deis2_temporal_v_round2")),
    "golden_pndm_spatial_bidir": (case_golden_pipeline, dict(name="pndm_spatial_bidir")),
    "golden_pndm_temporal_v_round2": (case_golden_pipeline, dict(name="pndm_temporal_v_round2")),
    # the judged configuration (BASELINE.json configs[1..2]): SD-2.1 geometry at 72x40, vs tests/golden/sd21_72x40.pt
    "unet_sd21_72x40_f16": (case_unet_sd21, dict(name="unet_f16_spatial")),
    "unet_sd21_72x40_f24": (case_unet_sd21, dict(name="unet_f24_temporal")),
    "vae_sd_576x320": (case_vae_sd, dict()),
}
# precision="parity" on the small configurations (fp32 oracle run on the spot): every layer type, both domains, CFG on / off, DPM
PAR = dict(precision="parity")
CASES.update({
    "par_unet_spatial": (case_unet, dict(num_frames=4, cfg_batch=2, **PAR)),
    "par_unet_temporal_temb": (case_unet, dict(num_frames=4, cfg_batch=2, tem=True, domain="temporal", **PAR)),
    "par_unet_2d_only": (case_unet, dict(num_frames=1, cfg_batch=3, h=8, w=8, **PAR)),
    "par_vae": (case_vae, dict(**PAR)),
    "par_vae_odd_latent_area": (case_vae, dict(h=264, w=328, **PAR)),
    "par_pipeline_spatial": (case_pipeline, dict(domain="spatial", **PAR)),
    # (temporal / bidirectional / DPM configurations of this precision: the golden_* cases below, against the reference pipeline's own output;
    #  their oracle-on-the-spot twins were dropped in round 6 to keep the GPU suite inside the driver's window)
    # parity precision against the fixtures made by the REFERENCE's own pipeline code (fp32): all three scheduler families, both domains
    "par_golden_spatial": (case_golden_pipeline, dict(name="spatial", **PAR)),
    "par_golden_temporal_v": (case_golden_pipeline, dict(name="temporal_v", **PAR)),
    "par_golden_round2_shift": (case_golden_pipeline, dict(name="round2_shift", **PAR)),
    "par_golden_pose_encoder": (case_golden_pipeline, dict(name="pose_encoder", **PAR)),
    "par_golden_dpm_temporal_v_heun_round2": (case_golden_pipeline, dict(name="dpm_temporal_v_heun_round2", **PAR)),
    "par_golden_unipc_temporal_v_bh1_round2": (case_golden_pipeline, dict(name="unipc_temporal_v_bh1_round2", **PAR)),
    "par_golden_unipc_spatial_bidir": (case_golden_pipeline, dict(name="unipc_spatial_bidir", **PAR)),
    "par_golden_deis3_spatial_bidir": (case_golden_pipeline, dict(name="deis3_spatial_bidir", **PAR)),
    "par_golden_pndm_spatial_bidir": (case_golden_pipeline, dict(name="pndm_spatial_bidir", **PAR)),
    "par_golden_pndm_temporal_v_round2": (case_golden_pipeline, dict(name="pndm_temporal_v_round2", **PAR)),
    # ... and on the judged geometry, against the committed fp32 fixtures
    "par_unet_sd21_72x40_f16": (case_unet_sd21, dict(name="unet_f16_spatial", **PAR)),
    "par_vae_sd_576x320": (case_vae_sd, dict(**PAR)),
})
# the parity precision's holes of round 4: the F = 24 temporal call at SD width, frame sharding (K | V operand planes all-gathered),
# the cond-row pruning extension
CASES.update({
    "par_unet_sd21_72x40_f24": (case_unet_sd21, dict(name="unet_f24_temporal", **PAR)),
    "par_unet_frame_shard_p4": (case_unet_frame_shard, dict(P=4, num_frames=8, **PAR)),
    "par_pipeline_shard_rccl_world1": (case_pipeline_shard_world1, dict(**PAR)),
    "par_pipeline_prune_cond_rows": (case_pipeline_prune, dict(domain="spatial", **PAR)),
})
# precision="fp16" (round 5): single-term fp16 MFMA operands over fp32 tensors -- fixed bounds per quantity (FP16_BOUNDS)
FP16 = dict(precision="fp16")
CASES.update({
    "fp16_unet_spatial": (case_unet, dict(num_frames=4, cfg_batch=2, **FP16)),
    "fp16_unet_temporal_temb": (case_unet, dict(num_frames=4, cfg_batch=2, tem=True, domain="temporal", **FP16)),
    "fp16_unet_2d_only": (case_unet, dict(num_frames=1, cfg_batch=3, h=8, w=8, **FP16)),
    "fp16_unet_pose_encoder": (case_unet, dict(num_frames=4, cfg_batch=2, pose=True, **FP16)),
    "fp16_vae": (case_vae, dict(**FP16)),
    "fp16_vae_odd_latent_area": (case_vae, dict(h=264, w=328, **FP16)),
    "fp16_pipeline_spatial": (case_pipeline, dict(domain="spatial", **FP16)),
    "fp16_golden_spatial": (case_golden_pipeline, dict(name="spatial", **FP16)),
    "fp16_golden_temporal_v": (case_golden_pipeline, dict(name="temporal_v", **FP16)),
    "fp16_golden_round2_shift": (case_golden_pipeline, dict(name="round2_shift", **FP16)),
    "fp16_golden_pose_encoder": (case_golden_pipeline, dict(name="pose_encoder", **FP16)),
    "fp16_golden_dpm_temporal_v_heun_round2": (case_golden_pipeline, dict(name="dpm_temporal_v_heun_round2", **FP16)),
    "fp16_golden_unipc_temporal_v_bh1_round2": (case_golden_pipeline, dict(name="unipc_temporal_v_bh1_round2", **FP16)),
    "fp16_golden_deis3_spatial_bidir": (case_golden_pipeline, dict(name="deis3_spatial_bidir", **FP16)),
    "fp16_golden_pndm_spatial_bidir": (case_golden_pipeline, dict(name="pndm_spatial_bidir", **FP16)),
    "fp16_golden_pndm_temporal_v_round2": (case_golden_pipeline, dict(name="pndm_temporal_v_round2", **FP16)),
    "fp16_unet_sd21_72x40_f16": (case_unet_sd21, dict(name="unet_f16_spatial", **FP16)),
    "fp16_unet_sd21_72x40_f24": (case_unet_sd21, dict(name="unet_f24_temporal", **FP16)),
    "fp16_vae_sd_576x320": (case_vae_sd, dict(**FP16)),
    "fp16_unet_frame_shard_p4": (case_unet_frame_shard, dict(P=4, num_frames=8, **FP16)),
    "fp16_pipeline_shard_rccl_world1": (case_pipeline_shard_world1, dict(**FP16)),
    "fp16_pipeline_prune_cond_rows": (case_pipeline_prune, dict(domain="spatial", **FP16)),
})
# fast precision, launch by launch against fp64 on the device's own tensors (oracle/replay.py): fixed bound REPLAY_TOL
CASES.update({
    "opreplay_unet_small": (case_opreplay_unet, dict()),
    "opreplay_pipeline_small": (case_opreplay_pipeline, dict()),
    "opreplay_unet_sd21_72x40_f16": (case_opreplay_unet, dict(sd21=True)),
    "opreplay_vae_sd_576x320": (case_opreplay_vae_sd, dict()),
})
# HIP fast precision vs the rounding-matched oracle (oracle/matched.py): small configurations on the spot ...
CASES.update({
    "unet_spatial_matched": (case_unet, dict(num_frames=4, cfg_batch=2, matched=True)),
    "unet_temporal_temb_matched": (case_unet, dict(num_frames=4, cfg_batch=2, tem=True, domain="temporal", matched=True)),
    "pipeline_spatial_matched": (case_pipeline, dict(domain="spatial", matched=True)),
    "pipeline_temporal_v_matched": (case_pipeline, dict(domain="temporal", T=4, window=4, stride=1, pred="v_prediction", matched=True)),
})
# ... and on the judged UNet calls (make_golden_sd21.py matched16 / matched24)
_sd21 = torch.load(GOLDEN / "sd21_72x40.pt")
for _n, _k in (("unet_sd21_72x40_f16_matched", "unet_f16_spatial"), ("unet_sd21_72x40_f24_matched", "unet_f24_temporal")):
    if "matched_out" in _sd21.get(_k, {}):
        CASES[_n] = (case_unet_sd21, dict(name=_k, matched=True))
del _sd21
# the reference's NATIVE latent size (spatem_dataset.py:27-28: 1024^2 images -> 128 x 128 latents; make_golden_sd21.py unet16_128)
if (GOLDEN / "sd21_128x128.pt").exists():
    CASES["unet_sd21_128x128_f16"] = (case_unet_sd21, dict(name="unet_f16_spatial_128", fixture="sd21_128x128.pt"))
    CASES["par_unet_sd21_128x128_f16"] = (case_unet_sd21, dict(name="unet_f16_spatial_128", fixture="sd21_128x128.pt", **PAR))
    CASES["fp16_unet_sd21_128x128_f16"] = (case_unet_sd21, dict(name="unet_f16_spatial_128", fixture="sd21_128x128.pt", **FP16))
if (GOLDEN / "multiround_sd21_72x40.pt").exists():  # spatial -> temporal -> spatial through the sampler (make_golden_multiround.py)
    CASES["multiround_sd21_72x40"] = (case_multiround_sd21, dict())
    CASES["par_multiround_sd21_72x40"] = (case_multiround_sd21, dict(**PAR))
    CASES["fp16_multiround_sd21_72x40"] = (case_multiround_sd21, dict(**FP16))
# BASELINE.json configs[0] end to end at the judged geometry, vs tests/golden/demo3d_sd21_72x40.pt (made by
# tests/golden/make_golden_demo3d.py: two CPU-hours; the case exists once the fixture does)
if "vae_1024" in torch.load(GOLDEN / "sd21_72x40.pt"):  # the VAE at the reference's native 1024 x 1024 (make_golden_sd21.py vae1024)
    CASES["vae_sd_1024x1024"] = (case_vae_1024, dict())
if (GOLDEN / "demo3d_sd21_72x40.pt").exists():
    CASES["demo3d_sd21_72x40"] = (case_demo3d_sd21, dict())
    CASES["par_demo3d_sd21_72x40"] = (case_demo3d_sd21, dict(**PAR))  # north_star: decoded RGB within 1e-3 of the fp32 reference path
    CASES["fp16_demo3d_sd21_72x40"] = (case_demo3d_sd21, dict(**FP16))  # the same bar at one MFMA per product
    if "matched_latents" in torch.load(GOLDEN / "demo3d_sd21_72x40.pt"):  # make_golden_demo3d.py matched
        CASES["demo3d_sd21_72x40_matched"] = (case_demo3d_sd21, dict(matched=True))
# BASELINE.json configs[1]: one whole temporal task of demo_4d_tiny (8 calls of F = 24) at the judged geometry, all three precisions
if (GOLDEN / "demo4dtiny_temporal_sd21_72x40.pt").exists():
    CASES["demo4dtiny_temporal_sd21_72x40"] = (case_demo4dtiny_temporal, dict())
    CASES["par_demo4dtiny_temporal_sd21_72x40"] = (case_demo4dtiny_temporal, dict(**PAR))    # fixed bound 1e-4
    CASES["fp16_demo4dtiny_temporal_sd21_72x40"] = (case_demo4dtiny_temporal, dict(**FP16))  # fixed bound: decoded RGB <= 1e-3
# Cases with a fixed bound of their own: bitwise equalities (0.0), extension-vs-strict comparisons, exact resampling.
# Every other case is judged against its bf16-oracle yardstick (YARD_FACTOR, see the module docstring).
TOL = {"task_batching_spatial": 0.0, "task_batching_temporal_x3": 0.0, "task_batching_dpm": 0.0, "pipeline_shard_rccl_world1": 0.0, "unet_frame_shard_p4": 0.0, "unet_frame_shard_p8": 0.0,
       "par_unet_frame_shard_p4": 0.0, "par_pipeline_shard_rccl_world1": 0.0, "fp16_unet_frame_shard_p4": 0.0, "fp16_pipeline_shard_rccl_world1": 0.0,
       "par_pipeline_prune_cond_rows": 1e-4, "fp16_pipeline_prune_cond_rows": 1e-3,
       "pipeline_cache_lazy_decode": 0.0, "pipeline_prune_cond_rows": 5e-3, "pipeline_prune_cond_rows_temporal": 5e-3,
       "pipeline_plucker_on_device": 5e-3, "resize": 4e-3}
TOL.update({n: 0.0 for n in CASES if "task_stack_" in n})  # bit for bit the tasks run alone
# Tighter fixed bounds of individual par_* cases, ~8x above what MI355X measured (every quantity below 1.6e-5 where
# the fixture is fp32 / 16-bit fixed point).  Cases whose fixture stores the decoded RGB in fp16 (floor 1.7-1.8e-4) keep 5e-4.
PARITY_TOLS: dict = {n: 1e-4 for n in CASES if n.startswith("par_") and not n.startswith(("par_golden", "par_vae_sd"))}
PARITY_TOLS.update({n: 5e-4 for n in CASES if n.startswith(("par_golden", "par_vae_sd"))})
TOL.update({n: PARITY_TOLS.get(n, PARITY_TOL) for n in CASES if n.startswith("par_") and n not in TOL})
TOL.update({n: FP16_BOUNDS for n in CASES if n.startswith("fp16_") and n not in TOL})  # a dict: one fixed bound per compared quantity
TOL.update({n: 0.0 for n in CASES if n.endswith("_matched")})  # band_excess must be zero (MATCHED_BAND)
TOL.update({n: REPLAY_TOL for n in CASES if n.startswith("opreplay_")})


def judge(name, err, yard):
    """-> (worst error, its yardstick, its bound): the compared quantity with the largest error / bound ratio."""
    errs = err if isinstance(err, dict) else {"value": err}
    yards = yard if isinstance(yard, dict) else {k: yard for k in errs}
    rows = []
    for q, e in errs.items():
        bound = (TOL[name][q] if isinstance(TOL[name], dict) else TOL[name]) if name in TOL else YARD_FACTOR * yards[q]
        rows.append((e / bound if bound > 0 else (0.0 if e == 0 else math.inf), e, yards.get(q, 0.0), bound, q))
    rows.sort(reverse=True)
    _, e, y, bound, q = rows[0]
    return e, y, bound, q, errs, yards


def run_case(name):
    fn, kw = CASES[name]
    err, yard = fn(**kw)
    e, y, bound, _, _, _ = judge(name, err, yard)
    return e, y, bound


def main():
    bad = 0
    names = [n for n in CASES if not sys.argv[1:] or any(n.startswith(a) for a in sys.argv[1:])]
    for name in names:
        t0 = time.time()
        try:
            fn, kw = CASES[name]
            e, y, bound, q, errs, yards = judge(name, *fn(**kw))
            ok = e <= bound and math.isfinite(e)
            detail = " ".join(f"{k}={v:.3e}/{yards.get(k, 0.0):.3e}" for k, v in errs.items())
            print(f"{'PASS' if ok else 'FAIL'} {name:28s} worst={q} rel_l2={e:.3e} bound={bound:.2e} "
                  f"[err/oracle-bf16 yardstick: {detail}] ({time.time() - t0:.1f}s)", flush=True)
            bad += 0 if ok else 1
        except Exception as e:
            bad += 1
            print(f"ERROR {name}: {type(e).__name__}: {e}", flush=True)
            traceback.print_exc()
    print(f"modelcheck: {len(names) - bad}/{len(names)} passed", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
