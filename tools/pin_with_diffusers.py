#!/usr/bin/env python
"""One-command pinning of the oracle's restatement of diffusers internals -- for a machine where `import diffusers` works.

The reference delegates its arithmetic to the un-vendored `diffusers==0.33.1` (/root/reference/requirements.txt:5; call sites
unet_multiview_blocks.py:26-27, attention.py:7-10, transformer_multiview.py:20, unet_multiview_condition.py:26,
pipeline_diffuman4d.py:27).  `oracle/*` restates those building blocks from the published release, but the build container has no
diffusers wheel, so that restatement is UNPINNED.  This script closes the gap wherever the package is
available:  it builds each upstream block, loads the ORACLE's seeded weights into it with `strict=True` (pins the key names),
feeds both the same seeded input and requires the outputs to agree to fp32 round-off.

    python tools/pin_with_diffusers.py            # exit 0: every block pinned | 1: a mismatch | 77: diffusers not importable (skipped)

Blocks: ResnetBlock2D (with / without shortcut, output_scale_factor), Downsample2D, Upsample2D, Timesteps + TimestepEmbedding,
Attention (AttnProcessor2_0), FeedForward(GEGLU), BasicTransformerBlock wiring (norm1 / attn1 / norm3 / ff as the reference's
MultiviewTransformerBlock inherits it), AutoencoderKL (moments, decode), DDIMScheduler, DPMSolverMultistepScheduler,
UniPCMultistepScheduler, DEISMultistepScheduler and PNDMScheduler (skip_prk_steps) steps (oracle/ddim.py, dpmsolver.py, multistep.py).

    python tools/pin_with_diffusers.py --self-test    # no diffusers needed: the scheduler comparison loop run oracle-against-oracle, so that
                                                      # the code of the check itself is exercised where the package is absent
Test infrastructure: imports `oracle/`, never imported by the product.
"""
from __future__ import annotations

import sys
import traceback
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
TOL = 2e-5  # fp32 round-off of differently ordered sums; a wrong eps / activation / order shows up at 1e-2 and above


def _gen(seed):
    return torch.Generator().manual_seed(seed)


def _rand_like_init(module, seed):
    g = _gen(seed)
    with torch.no_grad():
        for p in module.parameters():
            # matrices / kernels ~ N(0, 1/fan_in) so activations stay O(1); vectors (biases, norm gains) ~ N(0, 0.3^2)
            p.copy_(torch.randn(p.shape, generator=g) * (1.0 / max(1.0, float(p[0].numel()) ** 0.5) if p.dim() > 1 else 0.3))
    return module


def _maxdiff(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12))


def check_resnet():
    from diffusers.models.resnet import ResnetBlock2D as Up
    from oracle.unet import ResnetBlock2D as Or
    worst = 0.0
    for cin, cout, scale in ((32, 32, 1.0), (32, 64, 1.0), (64, 32, 2.0)):
        o = _rand_like_init(Or(cin, cout, 128, groups=8, eps=1e-5, output_scale_factor=scale), 1).eval()
        u = Up(in_channels=cin, out_channels=cout, temb_channels=128, groups=8, eps=1e-5, output_scale_factor=scale).eval()
        u.load_state_dict(o.state_dict(), strict=True)
        x, t = torch.randn(2, cin, 12, 10, generator=_gen(2)), torch.randn(2, 128, generator=_gen(3))
        with torch.no_grad():
            worst = max(worst, _maxdiff(u(x, t), o(x, t)))
    return worst


def check_samplers():
    from diffusers.models.downsampling import Downsample2D as UD
    from diffusers.models.upsampling import Upsample2D as UU
    from oracle.unet import Downsample2D as OD, Upsample2D as OU
    x = torch.randn(2, 16, 10, 8, generator=_gen(4))
    worst = 0.0
    for pad in (1, 0):  # UNet (padding 1) and VAE encoder (padding 0 + asymmetric F.pad)
        o = _rand_like_init(OD(16, padding=pad), 5).eval()
        u = UD(16, use_conv=True, out_channels=16, padding=pad, name="op").eval()  # name="op": the module's only attribute is `conv`
        u.load_state_dict(o.state_dict(), strict=True)
        with torch.no_grad():
            worst = max(worst, _maxdiff(u(x), o(x)))
    o = _rand_like_init(OU(16), 6).eval()
    u = UU(16, use_conv=True, out_channels=16).eval()
    u.load_state_dict(o.state_dict(), strict=True)
    with torch.no_grad():
        worst = max(worst, _maxdiff(u(x), o(x)))
    return worst


def check_time_embedding():
    from diffusers.models.embeddings import TimestepEmbedding as UT, Timesteps
    from oracle.unet import TimestepEmbedding as OT, timestep_embedding
    t = torch.tensor([0, 1, 17, 500, 999])
    worst = 0.0
    for flip, shift in ((True, 0), (False, 1)):
        worst = max(worst, _maxdiff(Timesteps(320, flip, shift)(t), timestep_embedding(t, 320, flip, shift)))
    o = _rand_like_init(OT(320, 1280), 7).eval()
    u = UT(320, 1280).eval()
    u.load_state_dict(o.state_dict(), strict=True)
    e = timestep_embedding(t, 320, True, 0)
    with torch.no_grad():
        worst = max(worst, _maxdiff(u(e), o(e)))
    return worst


def check_attention_ff_block():
    from diffusers.models.attention import BasicTransformerBlock, FeedForward as UF
    from diffusers.models.attention_processor import Attention as UA
    from oracle.unet import Attention as OA, FeedForward as OF, MultiviewTransformerBlock as OB
    x = torch.randn(3, 40, 128, generator=_gen(8))
    oa = _rand_like_init(OA(128, 2, 64, bias=False), 9).eval()
    ua = UA(query_dim=128, heads=2, dim_head=64, bias=False).eval()
    ua.load_state_dict(oa.state_dict(), strict=True)
    of = _rand_like_init(OF(128), 10).eval()
    uf = UF(128, mult=4, activation_fn="geglu").eval()
    uf.load_state_dict(of.state_dict(), strict=True)
    ob = _rand_like_init(OB(128, 2, 64), 11).eval()
    # the reference's block is BasicTransformerBlock(dim, heads, dim_head, cross_attention_dim=None) with norm2 / attn2 absent
    ub = BasicTransformerBlock(128, 2, 64, cross_attention_dim=None, activation_fn="geglu", only_cross_attention=False,
                               double_self_attention=False, norm_type="layer_norm").eval()
    missing = ub.load_state_dict(ob.state_dict(), strict=False)
    extra = [k for k in missing.missing_keys if not (k.startswith("norm2.") or k.startswith("attn2."))]
    assert not extra and not missing.unexpected_keys, (extra, missing.unexpected_keys)
    with torch.no_grad():
        w = max(_maxdiff(ua(x), oa(x)), _maxdiff(uf(x), of(x)))
        if not any(k.startswith("attn2.") for k in ub.state_dict()):  # with cross_attention_dim=None diffusers builds no attn2
            w = max(w, _maxdiff(ub(x), ob(x, num_frames=1)))
    return w


def check_vae():
    from diffusers import AutoencoderKL as UV
    from oracle.vae import AutoencoderKL as OV, VAEConfig
    cfg = VAEConfig.tiny()
    o = _rand_like_init(OV(cfg), 12).eval()
    u = UV(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4,
           block_out_channels=cfg.block_out_channels, layers_per_block=cfg.layers_per_block, latent_channels=cfg.latent_channels,
           norm_num_groups=cfg.norm_num_groups, scaling_factor=cfg.scaling_factor).eval()
    u.load_state_dict(o.state_dict(), strict=True)
    x = torch.rand(2, 3, 64, 48, generator=_gen(13)) * 2 - 1
    z = torch.randn(2, cfg.latent_channels, 8, 6, generator=_gen(14))
    with torch.no_grad():
        return max(_maxdiff(u.encode(x).latent_dist.parameters, o.moments(x)), _maxdiff(u.decode(z).sample, o.decode(z)))


def check_schedulers():
    from dataclasses import asdict
    from diffusers import DDIMScheduler as UD, DPMSolverMultistepScheduler as UP
    from oracle.ddim import DDIMConfig, DDIMScheduler as OD
    from oracle.dpmsolver import DPMSolverConfig, DPMSolverMultistepScheduler as OP
    worst = 0.0
    x0, eps = torch.randn(1, 4, 8, 8, generator=_gen(15)), [torch.randn(1, 4, 8, 8, generator=_gen(20 + i)) for i in range(12)]
    for pred in ("epsilon", "v_prediction"):
        oc = DDIMConfig(prediction_type=pred)
        o, u = OD(oc), UD(clip_sample=False, **asdict(oc))
        ts = o.set_timesteps(12)
        u.set_timesteps(12)
        assert [int(t) for t in u.timesteps] == [int(t) for t in ts]
        xo = xu = x0
        for i, t in enumerate(ts):
            xo, xu = o.step(eps[i], int(t), xo), u.step(eps[i], int(t), xu).prev_sample
            worst = max(worst, _maxdiff(xu, xo))
    for kw in (dict(), dict(solver_type="heun", final_sigmas_type="sigma_min", timestep_spacing="leading", steps_offset=1,
                           prediction_type="v_prediction")):
        oc = DPMSolverConfig(**kw)
        o, u = OP(oc), UP(**asdict(oc))
        ts = o.set_timesteps(12)
        u.set_timesteps(12)
        assert [int(t) for t in u.timesteps] == [int(t) for t in ts]
        xo = xu = x0
        for i, t in enumerate(ts):
            xo, xu = o.step(eps[i], int(t), xo), u.step(eps[i], int(t), xu).prev_sample
            worst = max(worst, _maxdiff(xu, xo))
    return worst


def _walk(make_o, make_u, n=12):
    """Both schedulers over the same n steps from the same start and the same model outputs: worst relative deviation of the sample."""
    x0, eps = torch.randn(1, 4, 8, 8, generator=_gen(15)), [torch.randn(1, 4, 8, 8, generator=_gen(20 + i)) for i in range(n + 1)]
    o, u = make_o(), make_u()
    ts = o.set_timesteps(n)  # n entries; n + 1 for PNDM (its second step is repeated)
    u.set_timesteps(n)
    assert [int(t) for t in u.timesteps] == [int(t) for t in ts], "timestep tables differ"
    xo = xu = x0
    worst = 0.0
    for i, t in enumerate(ts):
        ro, ru = o.step(eps[i], int(t), xo), u.step(eps[i], int(t), xu)
        xo, xu = ro, getattr(ru, "prev_sample", ru)
        worst = max(worst, _maxdiff(xu, xo))
    return worst


MULTISTEP_GRID = (
    ("unipc", dict()),
    ("unipc", dict(solver_type="bh1", prediction_type="v_prediction", timestep_spacing="leading", steps_offset=1, final_sigmas_type="sigma_min",
                   disable_corrector=[5, 6])),
    ("unipc", dict(solver_order=1, timestep_spacing="trailing")),
    ("deis", dict()),
    ("deis", dict(solver_order=3, prediction_type="v_prediction", timestep_spacing="trailing")),
    ("deis", dict(solver_order=1, timestep_spacing="leading", steps_offset=1)),
    # PNDM with skip_prk_steps (PLMS): the table repeats its second step, the object's second call re-does the first transfer (round 6)
    ("pndm", dict(skip_prk_steps=True, beta_schedule="scaled_linear", beta_start=0.00085, beta_end=0.012, steps_offset=1)),
    ("pndm", dict(skip_prk_steps=True, prediction_type="v_prediction", timestep_spacing="trailing", set_alpha_to_one=True)),
)


def check_multistep(upstream=True):
    """oracle/multistep.py (UniPC with its corrector, DEIS) against diffusers' classes of the same names; upstream=False (--self-test) walks
    two oracle objects against each other: it proves nothing about diffusers and everything about this loop."""
    from dataclasses import asdict
    from oracle import multistep as ms
    if upstream:
        from diffusers import DEISMultistepScheduler as UDe, PNDMScheduler as UPn, UniPCMultistepScheduler as UUn
    worst = 0.0
    cfgs = {"unipc": ms.UniPCConfig, "deis": ms.DEISConfig, "pndm": ms.PNDMConfig}
    ours = {"unipc": ms.UniPCMultistepScheduler, "deis": ms.DEISMultistepScheduler, "pndm": ms.PNDMScheduler}
    for kind, kw in MULTISTEP_GRID:
        cfg = cfgs[kind](**kw)
        make_o = (lambda c=cfg, k=kind: ours[k](c))
        make_u = (lambda c=cfg, k=kind: {"unipc": UUn, "deis": UDe, "pndm": UPn}[k](**asdict(c))) if upstream else make_o
        worst = max(worst, _walk(make_o, make_u))
    return worst


CHECKS = [("ResnetBlock2D", check_resnet), ("Downsample2D / Upsample2D", check_samplers),
          ("Timesteps + TimestepEmbedding", check_time_embedding), ("Attention / FeedForward / BasicTransformerBlock", check_attention_ff_block),
          ("AutoencoderKL", check_vae), ("DDIMScheduler / DPMSolverMultistepScheduler", check_schedulers),
          ("UniPCMultistepScheduler / DEISMultistepScheduler / PNDMScheduler", check_multistep)]


def main() -> int:
    if "--self-test" in sys.argv[1:]:
        w = check_multistep(upstream=False)
        print(f"SELF-TEST scheduler comparison loop, oracle against oracle over {len(MULTISTEP_GRID)} configurations: deviation {w:.1e} (must be 0)")
        return 0 if w == 0.0 else 1
    try:
        import diffusers
    except Exception as e:  # noqa: BLE001
        print(f"SKIP: diffusers is not importable here ({type(e).__name__}: {e}); the oracle's restatement of its blocks stays unpinned")
        return 77
    print(f"diffusers {diffusers.__version__} (the reference pins 0.33.1), torch {torch.__version__}")
    bad = 0
    for name, fn in CHECKS:
        try:
            w = fn()
            ok = w <= TOL
            print(f"{'PINNED  ' if ok else 'MISMATCH'} {name:52s} max rel. deviation {w:.2e} (tolerance {TOL:.0e})")
            bad += 0 if ok else 1
        except Exception:  # noqa: BLE001 -- report every block
            bad += 1
            print(f"ERROR    {name}")
            traceback.print_exc()
    print("all blocks pinned" if not bad else f"{bad} block(s) NOT pinned")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
