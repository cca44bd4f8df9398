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
