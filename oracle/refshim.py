"""Import shim that lets the REFERENCE's own Python modules run in this container.

TEST INFRASTRUCTURE ONLY -- used by ``tests/golden/make_golden.py`` (and nothing else) to produce the
golden vectors that pin the oracle.

The reference's hot-path modules (``/root/reference/src/diffusers/**``, ``src/samplers/**``) import
``diffusers==0.33.1``, ``torchvision``, ``hydra`` ... at module scope; none is installed here.  This
module registers minimal stand-ins in ``sys.modules`` so that those files import and execute
unmodified.  The stand-ins for diffusers *building blocks* (ResnetBlock2D, Attention, Transformer2DModel
base, BasicTransformerBlock base, Timesteps, DDIMScheduler, AutoencoderKL ...) are thin adapters with
diffusers' constructor signatures and attribute names over the oracle's restated primitives
(``oracle/unet.py``, ``oracle/vae.py``, ``oracle/ddim.py``).

What this pins: everything the reference itself owns -- UNet wiring and ``num_frames`` routing
(unet_multiview_condition.py, unet_multiview_blocks.py), the transformer wrapper, the 3-D attention
frame folding (attention.py), the pipeline's CFG / conditioning / aliasing / per-latent stepping and
window sweep (pipeline_diffuman4d.py), the sampler's task list and grid bookkeeping -- and, through
``load_state_dict(strict=True)``, the state_dict key names those module trees produce.
What it cannot pin: the numerics inside the diffusers building blocks themselves (restated, unpinned).
"""
from __future__ import annotations

import contextlib
import inspect
import sys
import types
from collections import deque
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import ddim as _ddim
from . import dpmsolver as _dpm
from . import unet as _ou
from . import vae as _ov

REFERENCE_ROOT = "/root/reference"

# queue of noise tensors consumed, in call order, by latent_dist.sample() and randn_tensor()
NOISE_QUEUE: deque = deque()


def _module(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


# ---------------------------------------------------------------------------------------------
# configuration_utils / modeling_utils / utils
# ---------------------------------------------------------------------------------------------
class _Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class ConfigMixin:
    def register_to_config(self, **kwargs):
        object.__setattr__(self, "_internal_dict", _Config(kwargs))

    @property
    def config(self):
        return self._internal_dict


def register_to_config(init):
    sig = inspect.signature(init)

    def wrapper(self, *args, **kwargs):
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        ConfigMixin.register_to_config(self, **cfg)
        init(self, *args, **kwargs)

    return wrapper


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


class _Empty:
    pass


class BaseOutput(dict):
    def __post_init__(self):
        pass


class _Logger:
    def warning(self, *a, **k):
        pass

    info = debug = error = warning


def _get_logger(name=None):
    return _Logger()


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    t = NOISE_QUEUE.popleft()
    assert tuple(t.shape) == tuple(shape), (t.shape, shape)
    return t.to(device=device, dtype=dtype)


def replace_example_docstring(doc):
    def deco(fn):
        return fn
    return deco


# ---------------------------------------------------------------------------------------------
# building blocks with diffusers signatures
# ---------------------------------------------------------------------------------------------
class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift, scale=1):
        super().__init__()
        self.num_channels, self.flip, self.shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(self, timesteps):
        return _ou.timestep_embedding(timesteps, self.num_channels, self.flip, self.shift)


class TimestepEmbedding(_ou.TimestepEmbedding):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None, cond_proj_dim=None):
        assert act_fn == "silu" and post_act_fn is None and cond_proj_dim is None and out_dim is None
        super().__init__(in_channels, time_embed_dim)

    def forward(self, sample, condition=None):
        return super().forward(sample)


def get_activation(name):
    assert name in ("silu", "swish")
    return nn.SiLU()


class ResnetBlock2D(_ou.ResnetBlock2D):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512, groups=32,
                 groups_out=None, pre_norm=True, eps=1e-6, non_linearity="swish", skip_time_act=False,
                 time_embedding_norm="default", kernel=None, output_scale_factor=1.0, use_in_shortcut=None, up=False,
                 down=False, conv_shortcut_bias=True, conv_2d_out_channels=None):
        assert time_embedding_norm == "default" and not up and not down and dropout == 0.0
        assert groups_out in (None, groups) and non_linearity in ("swish", "silu")
        super().__init__(in_channels, out_channels or in_channels, temb_channels, groups, eps, output_scale_factor)

    def forward(self, input_tensor, temb=None, *a, **k):
        return super().forward(input_tensor, temb)


class Downsample2D(_ou.Downsample2D):
    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv", **kw):
        assert use_conv and (out_channels in (None, channels))
        super().__init__(channels, padding=padding)

    def forward(self, hidden_states, *a, **k):
        return super().forward(hidden_states)


class Upsample2D(_ou.Upsample2D):
    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None, name="conv", **kw):
        assert use_conv and not use_conv_transpose and (out_channels in (None, channels))
        super().__init__(channels)

    def forward(self, hidden_states, output_size=None, *a, **k):
        assert output_size is None
        return super().forward(hidden_states)


class DownBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, dropout=0.0, num_layers=1, resnet_eps=1e-6,
                 resnet_time_scale_shift="default", resnet_act_fn="swish", resnet_groups=32, resnet_pre_norm=True,
                 output_scale_factor=1.0, add_downsample=True, downsample_padding=1):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels=in_channels if i == 0 else out_channels, out_channels=out_channels,
                          temb_channels=temb_channels, eps=resnet_eps, groups=resnet_groups,
                          output_scale_factor=output_scale_factor) for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, use_conv=True, out_channels=out_channels,
                                                        padding=downsample_padding, name="op")]) if add_downsample else None

    def forward(self, hidden_states, temb=None, *a, **k):
        outs = ()
        for r in self.resnets:
            hidden_states = r(hidden_states, temb)
            outs += (hidden_states,)
        if self.downsamplers is not None:
            hidden_states = self.downsamplers[0](hidden_states)
            outs += (hidden_states,)
        return hidden_states, outs


class UpBlock2D(nn.Module):
    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, resolution_idx=None, dropout=0.0,
                 num_layers=1, resnet_eps=1e-6, resnet_time_scale_shift="default", resnet_act_fn="swish",
                 resnet_groups=32, resnet_pre_norm=True, output_scale_factor=1.0, add_upsample=True):
        super().__init__()
        res = []
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            res.append(ResnetBlock2D(in_channels=rin + skip, out_channels=out_channels, temb_channels=temb_channels,
                                     eps=resnet_eps, groups=resnet_groups, output_scale_factor=output_scale_factor))
        self.resnets = nn.ModuleList(res)
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels, use_conv=True, out_channels=out_channels)]) if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, upsample_size=None, *a, **k):
        for r in self.resnets:
            res = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = r(torch.cat([hidden_states, res], dim=1), temb)
        if self.upsamplers is not None:
            hidden_states = self.upsamplers[0](hidden_states, upsample_size)
        return hidden_states


class Attention(_ou.Attention):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, out_bias=True, **kw):
        assert cross_attention_dim is None and out_bias
        super().__init__(query_dim, heads, dim_head, bias=bias)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        assert encoder_hidden_states is None and attention_mask is None
        return super().forward(hidden_states)


class BasicTransformerBlock(nn.Module):
    """Constructor + attributes of diffusers' BasicTransformerBlock (norm_type='layer_norm'); the
    reference's MultiviewTransformerBlock subclasses this and supplies its own forward."""

    def __init__(self, dim, num_attention_heads, attention_head_dim, dropout=0.0, cross_attention_dim=None,
                 activation_fn="geglu", num_embeds_ada_norm=None, attention_bias=False, only_cross_attention=False,
                 double_self_attention=False, upcast_attention=False, norm_elementwise_affine=True,
                 norm_type="layer_norm", norm_eps=1e-5, final_dropout=False, attention_type="default", **kw):
        super().__init__()
        assert norm_type == "layer_norm" and activation_fn == "geglu" and cross_attention_dim is None
        assert not double_self_attention and norm_elementwise_affine
        self.dim, self.norm_type, self.only_cross_attention = dim, norm_type, only_cross_attention
        self.num_attention_heads, self.attention_head_dim = num_attention_heads, attention_head_dim
        self.pos_embed = None
        self.norm1 = nn.LayerNorm(dim, eps=norm_eps)
        self.attn1 = Attention(query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim, dropout=dropout,
                               bias=attention_bias, cross_attention_dim=None, upcast_attention=upcast_attention)
        self.norm2, self.attn2 = None, None
        self.norm3 = nn.LayerNorm(dim, eps=norm_eps)
        self.ff = _ou.FeedForward(dim)
        self._chunk_size, self._chunk_dim = None, 0


def _chunked_feed_forward(ff, hidden_states, chunk_dim, chunk_size):
    raise NotImplementedError


class Transformer2DModel(ModelMixin, ConfigMixin):
    """Continuous-input part of diffusers' Transformer2DModel (the base of TransformerMultiviewModel)."""

    @register_to_config
    def __init__(self, num_attention_heads=16, attention_head_dim=88, in_channels=None, out_channels=None, num_layers=1,
                 dropout=0.0, norm_num_groups=32, cross_attention_dim=None, attention_bias=False, sample_size=None,
                 num_vector_embeds=None, patch_size=None, activation_fn="geglu", num_embeds_ada_norm=None,
                 use_linear_projection=False, only_cross_attention=False, double_self_attention=False,
                 upcast_attention=False, norm_type="layer_norm", norm_elementwise_affine=True, norm_eps=1e-5,
                 attention_type="default", caption_channels=None, interpolation_scale=None, use_additional_conditions=None):
        super().__init__()
        self.use_linear_projection = use_linear_projection
        self.num_attention_heads, self.attention_head_dim = num_attention_heads, attention_head_dim
        self.inner_dim = num_attention_heads * attention_head_dim
        self.in_channels = in_channels
        self.out_channels = in_channels if out_channels is None else out_channels
        self.gradient_checkpointing = False
        self.is_input_continuous, self.is_input_vectorized, self.is_input_patches = True, False, False
        self._init_continuous_input(norm_type=norm_type)

    def _operate_on_continuous_inputs(self, hidden_states):
        batch, _, height, width = hidden_states.shape
        hidden_states = self.norm(hidden_states)
        if not self.use_linear_projection:
            hidden_states = self.proj_in(hidden_states)
            inner_dim = hidden_states.shape[1]
            hidden_states = hidden_states.permute(0, 2, 3, 1).reshape(batch, height * width, inner_dim)
        else:
            inner_dim = hidden_states.shape[1]
            hidden_states = hidden_states.permute(0, 2, 3, 1).reshape(batch, height * width, inner_dim)
            hidden_states = self.proj_in(hidden_states)
        return hidden_states, inner_dim

    def _get_output_for_continuous_inputs(self, hidden_states, residual, batch_size, height, width, inner_dim):
        if not self.use_linear_projection:
            hidden_states = hidden_states.reshape(batch_size, height, width, inner_dim).permute(0, 3, 1, 2).contiguous()
            hidden_states = self.proj_out(hidden_states)
        else:
            hidden_states = self.proj_out(hidden_states)
            hidden_states = hidden_states.reshape(batch_size, height, width, inner_dim).permute(0, 3, 1, 2).contiguous()
        return hidden_states + residual


@dataclass
class Transformer2DModelOutput:
    sample: torch.Tensor = None


# ---------------------------------------------------------------------------------------------
# pipeline-level stand-ins
# ---------------------------------------------------------------------------------------------
class _LatentDist:
    def __init__(self, vae, moments):
        self.vae, self.moments = vae, moments

    def sample(self, generator=None):
        noise = NOISE_QUEUE.popleft()
        return self.vae.oracle.sample_posterior(self.moments, noise)


class AutoencoderKL(nn.Module):
    """diffusers-API adapter over the oracle VAE."""

    def __init__(self, oracle_vae: _ov.AutoencoderKL):
        super().__init__()
        self.oracle = oracle_vae
        c = oracle_vae.cfg
        self.config = _Config(scaling_factor=c.scaling_factor, block_out_channels=c.block_out_channels)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def encode(self, x):
        out = _Empty()
        out.latent_dist = _LatentDist(self, self.oracle.moments(x))
        return out

    def decode(self, z, return_dict=True, generator=None):
        return (self.oracle.decode(z),)


class DDIMSchedulerAdapter:
    """diffusers-API adapter over the oracle DDIM scheduler (deep-copyable, as the pipeline requires)."""

    def __init__(self, cfg: _ddim.DDIMConfig = _ddim.DDIMConfig()):
        self._s = _ddim.DDIMScheduler(cfg)
        self.init_noise_sigma = 1.0
        self.timesteps = None

    def set_timesteps(self, n, device=None):
        self.timesteps = self._s.set_timesteps(n)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, return_dict=True, **kw):
        return (self._s.step(model_output, int(timestep), sample),)


class DPMSolverSchedulerAdapter:
    """diffusers-API adapter over the oracle DPM-Solver++ scheduler.  STATEFUL (history of x0 predictions, step index) and
    deep-copyable: exactly what the reference's "a scheduler for each latent" copies (pipeline_diffuman4d.py:265-271)."""

    def __init__(self, cfg: "_dpm.DPMSolverConfig" = None):
        self._s = _dpm.DPMSolverMultistepScheduler(cfg or _dpm.DPMSolverConfig())
        self.init_noise_sigma = 1.0
        self.timesteps = None

    def set_timesteps(self, n, device=None):
        self.timesteps = self._s.set_timesteps(n)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, return_dict=True, **kw):
        return (self._s.step(model_output, int(timestep), sample),)


class StatefulSchedulerAdapter:
    """diffusers-API adapter over ANY stateful oracle scheduler (oracle/multistep.py: UniPC, DEIS; oracle/dpmsolver.py), deep-copyable:
    what the reference's "a scheduler for each latent" copies (pipeline_diffuman4d.py:265-271)."""

    def __init__(self, scheduler):
        self._s = scheduler
        self.init_noise_sigma = 1.0
        self.timesteps = None

    def set_timesteps(self, n, device=None):
        self.timesteps = self._s.set_timesteps(n)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, return_dict=True, **kw):
        return (self._s.step(model_output, int(timestep), sample),)


class VaeImageProcessor:
    def __init__(self, vae_scale_factor=8, **kw):
        self.vae_scale_factor = vae_scale_factor

    def postprocess(self, image, output_type="pt", do_denormalize=None):
        assert output_type == "pt" and all(do_denormalize)
        return (image / 2 + 0.5).clamp(0, 1)


class DiffusionPipeline:
    def register_modules(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def device(self):
        return next(self.unet.parameters()).device

    @property
    def _execution_device(self):
        return self.device

    @contextlib.contextmanager
    def progress_bar(self, iterable=None, total=None):
        class _Bar:
            def update(self, *a):
                pass
        yield _Bar()

    def maybe_free_model_hooks(self):
        pass

    def set_progress_bar_config(self, **kw):
        pass


def install():
    """Register the stand-ins and put the reference checkout on sys.path.  Idempotent."""
    if "diffusers" in sys.modules and getattr(sys.modules["diffusers"], "_dm4d_shim", False):
        return
    mix = lambda n: type(n, (), {})  # noqa: E731  empty mixin classes
    _module("diffusers", _dm4d_shim=True)
    _module("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    _module("diffusers.loaders", PeftAdapterMixin=mix("PeftAdapterMixin"),
            UNet2DConditionLoadersMixin=mix("UNet2DConditionLoadersMixin"), FromSingleFileMixin=mix("FromSingleFileMixin"),
            IPAdapterMixin=mix("IPAdapterMixin"), StableDiffusionLoraLoaderMixin=mix("StableDiffusionLoraLoaderMixin"),
            TextualInversionLoaderMixin=mix("TextualInversionLoaderMixin"))
    _module("diffusers.loaders.single_file_model", FromOriginalModelMixin=mix("FromOriginalModelMixin"))
    logging_mod = types.SimpleNamespace(get_logger=_get_logger)
    _module("diffusers.utils", BaseOutput=BaseOutput, logging=logging_mod, deprecate=lambda *a, **k: None,
            is_torch_version=lambda op, v: True, replace_example_docstring=replace_example_docstring)
    _module("diffusers.utils.torch_utils", apply_freeu=None, maybe_allow_in_graph=lambda c: c, randn_tensor=randn_tensor)
    _module("diffusers.models", AutoencoderKL=AutoencoderKL)
    _module("diffusers.models.activations", get_activation=get_activation)
    _module("diffusers.models.embeddings", TimestepEmbedding=TimestepEmbedding, Timesteps=Timesteps)
    _module("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    _module("diffusers.models.attention_processor", Attention=Attention, AttnAddedKVProcessor=None,
            AttnAddedKVProcessor2_0=None)
    _module("diffusers.models.normalization", AdaGroupNorm=None)
    _module("diffusers.models.resnet", Downsample2D=Downsample2D, ResnetBlock2D=ResnetBlock2D, Upsample2D=Upsample2D)
    _module("diffusers.models.unets")
    _module("diffusers.models.unets.unet_2d_blocks", DownBlock2D=DownBlock2D, UpBlock2D=UpBlock2D)
    _module("diffusers.models.transformers")
    _module("diffusers.models.transformers.transformer_2d", Transformer2DModel=Transformer2DModel)
    _module("diffusers.models.modeling_outputs", Transformer2DModelOutput=Transformer2DModelOutput)
    _module("diffusers.models.attention", _chunked_feed_forward=_chunked_feed_forward,
            BasicTransformerBlock=BasicTransformerBlock)
    _module("diffusers.image_processor", VaeImageProcessor=VaeImageProcessor)
    _module("diffusers.schedulers", KarrasDiffusionSchedulers=object)
    _module("diffusers.pipelines")
    _module("diffusers.pipelines.pipeline_utils", DiffusionPipeline=DiffusionPipeline, StableDiffusionMixin=mix("StableDiffusionMixin"))
    # sampler-side imports: file I/O, logging and config helpers the golden generator never calls.
    # Any attribute of these placeholder modules resolves to an inert object so `from x import y` works.
    class _Inert(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return type(k, (), {"__init__": lambda self, *a, **kw: None, "__call__": lambda self, *a, **kw: None})

    for name in ("torchvision", "torchvision.utils", "torchvision.transforms", "torchvision.transforms.functional",
                 "hydra", "hydra.core", "hydra.core.hydra_config", "omegaconf", "ipdb", "easyvolcap",
                 "easyvolcap.utils", "easyvolcap.utils.parallel_utils", "torchmetrics", "torchmetrics.image",
                 "torchmetrics.image.lpip", "torchmetrics.image.ssim", "torchmetrics.image.psnr", "cv2",
                 "lightning_utilities", "lightning_utilities.core", "lightning_utilities.core.rank_zero",
                 # pulled in by src/samplers/sampling_runner.py (metrics + nerfstudio export, never called here)
                 "fire", "lpips", "scripts.preprocess", "scripts.preprocess.remove_background"):
        if name not in sys.modules:
            m = _Inert(name)
            m.__path__ = []  # importable as a package (`import a.b.c` walks __path__)
            sys.modules[name] = m
            parent, _, child = name.rpartition(".")
            if parent in sys.modules:
                setattr(sys.modules[parent], child, m)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
