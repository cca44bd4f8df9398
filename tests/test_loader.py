"""Checkpoint file selection and the dtype seam of ``load_pipelines`` (sampling_utils.py:17-51).  CPU only."""
import json

import pytest
import torch
from safetensors.torch import save_file

from diffuman4d_amd.host.loader import load_pipelines
from diffuman4d_amd.host.weights import load_component_state_dict


def _write(path, value):
    save_file({"w": torch.full((2,), float(value))}, str(path))


def test_bf16_run_never_picks_the_fp16_variant(tmp_path):
    # a model_dir shared with an fp16 run of the reference holds both files; "*.fp16.*" sorts first alphabetically
    _write(tmp_path / "diffusion_pytorch_model.fp16.safetensors", 16)
    _write(tmp_path / "diffusion_pytorch_model.safetensors", 32)
    assert float(load_component_state_dict(tmp_path)["w"][0]) == 32.0
    assert float(load_component_state_dict(tmp_path, "fp16")["w"][0]) == 16.0


def test_fp16_falls_back_to_the_plain_file(tmp_path):
    _write(tmp_path / "diffusion_pytorch_model.safetensors", 32)
    assert float(load_component_state_dict(tmp_path, "fp16")["w"][0]) == 32.0


def test_sharded_checkpoint_is_merged(tmp_path):
    save_file({"a": torch.ones(1)}, str(tmp_path / "diffusion_pytorch_model-00001-of-00002.safetensors"))
    save_file({"b": torch.zeros(1)}, str(tmp_path / "diffusion_pytorch_model-00002-of-00002.safetensors"))
    (tmp_path / "diffusion_pytorch_model.safetensors.index.json").write_text(json.dumps({"weight_map": {
        "a": "diffusion_pytorch_model-00001-of-00002.safetensors", "b": "diffusion_pytorch_model-00002-of-00002.safetensors"}}))
    assert sorted(load_component_state_dict(tmp_path)) == ["a", "b"]


def test_sharded_fp16_variant_uses_the_diffusers_index_name(tmp_path):
    # diffusers names a sharded variant's index `<stem>.safetensors.index.fp16.json` and its shards `<stem>.fp16-0000x-of-0000y.safetensors`;
    # an un-suffixed (bf16) checkpoint in the same folder must not be picked by an fp16 run
    _write(tmp_path / "diffusion_pytorch_model.safetensors", 32)
    s1, s2 = "diffusion_pytorch_model.fp16-00001-of-00002.safetensors", "diffusion_pytorch_model.fp16-00002-of-00002.safetensors"
    save_file({"a": torch.full((1,), 16.0)}, str(tmp_path / s1))
    save_file({"b": torch.full((1,), 16.0)}, str(tmp_path / s2))
    (tmp_path / "diffusion_pytorch_model.safetensors.index.fp16.json").write_text(json.dumps({"weight_map": {"a": s1, "b": s2}}))
    sd = load_component_state_dict(tmp_path, "fp16")
    assert sorted(sd) == ["a", "b"] and float(sd["a"][0]) == 16.0
    assert sorted(load_component_state_dict(tmp_path)) == ["w"]  # the bf16 run still reads the plain file


def test_missing_checkpoint(tmp_path):
    with pytest.raises(FileNotFoundError):
        load_component_state_dict(tmp_path)


def test_load_pipelines_dtype_seam(tmp_path):
    # both spellings the reference accepts are accepted (no device -> no pipeline is built); anything else is its ValueError
    d = tmp_path / "ckpt"
    d.mkdir()
    (d / "model_index.json").write_text("{}")
    assert load_pipelines(model_dir=str(d), torch_dtype="bf16", gpu_ids=[]) == []
    assert load_pipelines(model_dir=str(d), torch_dtype="fp16", gpu_ids=[]) == []
    with pytest.raises(ValueError, match="Unsupported torch_dtype: fp32. Supported types are 'bf16' and 'fp16'."):
        load_pipelines(model_dir=str(d), torch_dtype="fp32", gpu_ids=[])


def test_weight_dtypes_of_the_three_precisions():
    """What the kernels of each precision are handed (host/unet.py::_Weights): the fast and parity precisions compute on the bf16 rounding of
    whatever was loaded (parity: duplicated along K); the fp16 precision holds the values of the pipeline's weight dtype in fp16 -- fp16
    checkpoint values exactly, bf16 values exactly (down to 2^-14), fp32 values rounded to the weight dtype first, as the reference's
    `from_pretrained(torch_dtype=...)` would."""
    from diffuman4d_amd.host.unet import _Weights
    w = torch.tensor([[1.0009765625, -0.333251953125, 3.0517578125e-05, 1.2345678]])  # fp16-exact, fp16-exact, 2^-15, neither
    bias = torch.tensor([0.1, -2.5])
    fast = _Weights({"b": bias}, "cpu")
    assert fast.mat(w).dtype == torch.bfloat16 and torch.equal(fast.mat(w), w.to(torch.bfloat16)) and fast.vec("b").dtype == torch.bfloat16
    par = _Weights({}, "cpu", parity=True)
    assert par.mat(w).shape == (1, 8) and torch.equal(par.mat(w)[:, :4], par.mat(w)[:, 4:])
    h_bf = _Weights({"b": bias}, "cpu", h16=True, wdtype=torch.bfloat16)   # bf16 pipeline in the fp16 precision
    assert h_bf.mat(w).dtype == torch.float16 and torch.equal(h_bf.mat(w).float(), w.to(torch.bfloat16).float())
    assert h_bf.vec("b").dtype == torch.float16 and torch.equal(h_bf.vec("b").float(), bias.to(torch.bfloat16).float())
    h_fp = _Weights({}, "cpu", h16=True, wdtype=torch.float16)             # fp16 pipeline: the checkpoint's own values
    assert torch.equal(h_fp.mat(w.to(torch.float16)), w.to(torch.float16)) and float(h_fp.mat(w)[0, 0]) == 1.0009765625
    assert h_bf.wide and h_fp.wide and par.wide and not fast.wide


def test_auto_precision_follows_the_torch_dtype(monkeypatch):
    """`precision: "auto"` (the default of load_pipelines and of configs/model/diffuman4d_mi355x.yaml): fp16 pipelines compute with fp16 MFMA
    operands, as the reference's fp16 pipelines do (sampling_utils.py:27-29); bf16 pipelines stay on the bf16 kernels."""
    import diffuman4d_amd.host.pipeline as hp
    seen = []

    def fake(model_dir, torch_dtype=None, device=None, precision=None):
        seen.append((torch_dtype, precision))
        return object()
    monkeypatch.setattr(hp.Diffuman4DPipeline, "from_pretrained", staticmethod(fake))
    load_pipelines(model_dir="/tmp", torch_dtype="fp16", gpu_ids=[0])
    load_pipelines(model_dir="/tmp", torch_dtype="bf16", gpu_ids=[0])
    load_pipelines(model_dir="/tmp", torch_dtype="fp16", gpu_ids=[0], precision="fast")
    load_pipelines(model_dir="/tmp", torch_dtype="bf16", gpu_ids=[0], precision="fp16")
    assert seen == [(torch.float16, "fp16"), (torch.bfloat16, "fast"), (torch.float16, "fast"), (torch.bfloat16, "fp16")]
