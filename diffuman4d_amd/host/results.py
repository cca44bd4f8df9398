"""On-disk result contract of the sampler (PIL + torch only; the reference uses torchvision).

Mirror of ``/root/reference/src/samplers/utils/sampling_utils.py:54-129``:
  * ``{output_dir}/grids/alt{r}_{spa|tem}{label}.webp`` -- the per-task snapshot mosaic (:70-93): rows = skeleton
    blend (when skeletons are given), input images, outputs (conditioning views dimmed to 20 %), |output - input|;
    one column per frame of the task; downscaled so the mosaic stays below ``max_image_size`` pixels wide;
  * ``{output_dir}/images/{cam}/{frame}.jpg`` (quality 90) for the input views and every fully denoised target
    view, with the dataset's crop undone onto a white canvas (``restore_cropped_image``,
    ``/root/reference/src/data/utils/image_utils.py:62-93``: crop parameters are the ``(ct, cl, ch, cw[, h, w])``
    tuples ``SpaTemDataset`` returns, spatem_dataset.py:58,157);
  * optional ``{output_dir}/crops/{cam}/{frame}.json`` (:112-113);
  * ``check_sampling_results`` counts the images (:117-129).
``pack_results_on_device`` + ``imgwrite.write_package`` are the same contract split for the MI355X host pipeline:
everything arithmetic (mosaic rows, |output - input|, the antialiased down-scale, the float -> uint8 conversions) runs on the
device right after the decode, one small uint8 package per task crosses PCIe, and the file encoding runs in writer processes.
``write_nerfstudio_transforms`` is the camera-file half of scripts/nerfstudio/diffuman4d_to_nerfstudio.py:14-35 (the
matting half of that script runs a third-party segmentation network and is out of scope).
"""
from __future__ import annotations

import json
import logging
import os
from copy import deepcopy
from glob import glob
from typing import Any, Dict, Optional, Sequence

import torch

log = logging.getLogger(__name__)


def _to_pil(img: torch.Tensor):
    """torchvision.transforms.functional.to_pil_image for a float CHW tensor in [0, 1]: ``mul(255).byte()``."""
    from PIL import Image
    arr = (img.detach().float().clamp(0, 1) * 255.0).to(torch.uint8).permute(1, 2, 0).cpu().numpy()
    return Image.fromarray(arr)


from .imgwrite import restore_cropped_image, write_package  # noqa: E402,F401  (PIL-only half, importable without torch)


def _resize_smaller_edge(x: torch.Tensor, size: int) -> torch.Tensor:
    """torchvision ``resize(tensor, int)``: the smaller edge becomes ``size``, aspect kept, antialiased bilinear."""
    h, w = x.shape[-2:]
    if min(h, w) == size:
        return x
    if h <= w:
        nh, nw = size, max(1, int(size * w / h))
    else:
        nh, nw = max(1, int(size * h / w)), size
    if x.is_cuda:  # on the device: the library's kernel (the same separable triangle filter), no torch operator on the result path
        from . import ops
        return ops.resize_aa(x.float().contiguous(), (nh, nw))
    return torch.nn.functional.interpolate(x, size=(nh, nw), mode="bilinear", antialias=True, align_corners=False)


def make_image_grid(images: torch.Tensor, nrow: int, padding: int = 2, pad_value: float = 0.0) -> torch.Tensor:
    """torchvision.utils.make_grid layout: images [N, C, H, W] -> [C, rows*(H+p)+p, cols*(W+p)+p], row-major."""
    n, c, h, w = images.shape
    cols = min(nrow, n)
    rows = (n + cols - 1) // cols
    grid = torch.full((c, rows * (h + padding) + padding, cols * (w + padding) + padding), float(pad_value))
    for k in range(n):
        y, x = divmod(k, cols)
        grid[:, y * (h + padding) + padding: y * (h + padding) + padding + h,
             x * (w + padding) + padding: x * (w + padding) + padding + w] = images[k]
    return grid


def make_image_grid_device(images: torch.Tensor, nrow: int, padding: int = 2) -> torch.Tensor:
    """``make_image_grid`` for a full rectangle of equally sized images (len(images) a multiple of nrow), vectorised: pad every
    image on its top / left, interleave, pad the mosaic on its bottom / right.  Same layout as torchvision's make_grid."""
    n, c, h, w = images.shape
    assert n % nrow == 0
    rows = n // nrow
    x = torch.nn.functional.pad(images, (padding, 0, padding, 0))
    x = x.view(rows, nrow, c, h + padding, w + padding).permute(2, 0, 3, 1, 4).reshape(c, rows * (h + padding), nrow * (w + padding))
    return torch.nn.functional.pad(x, (0, padding, 0, padding))


@torch.no_grad()
def pack_results_on_device(sample: Dict[str, Any], images: torch.Tensor, output_dir: str = "./results", save_image_grid: bool = True,
                           save_output_image: bool = True, save_crop_param: bool = False, image_ext: str = ".jpg",
                           image_quality: int = 90, max_image_size: int = 8192, device=None) -> Dict[str, Any]:
    """``save_sampling_results`` up to (not including) the file encoding, evaluated on the device `images` lives on.

    images: the pipeline's decoded output [N, 3, H, W] in [0, 1], still on the GPU.  ``sample`` as the sampler builds it
    (``pixel_values`` / ``skeletons`` are the host tensors: they are uploaded once more in fp32, which is what the reference's
    arithmetic reads).  Returns the package ``imgwrite.write_package`` consumes: uint8 HWC numpy arrays + paths + crop tuples."""
    # `device`: where the arithmetic runs (the sampler passes the pipeline's).  With decode_policy "denoised" a task of an early round
    # returns a zero-stride HOST placeholder for its images (nothing was decoded): the packer must not follow it onto the CPU -- 194 of
    # the 344 tasks of demo_4d then built their mosaics with host arithmetic, 130 s of a 303 s run (profiles/r04_e2e_demo_4d_fast.json)
    dev = torch.device(device) if device is not None else images.device
    if images.device != dev:
        out = torch.zeros(images.shape, dtype=torch.float32, device=dev) if images.stride(0) == 0 else images.to(dev).float()
    else:
        out = images.float()
    input_indices = sample["input_indices"].to(dev)
    target_indices = set(int(i) for i in sample["target_indices"])
    inp = sample["pixel_values"].to(dev, non_blocking=True).float() * 0.5 + 0.5  # denorm_vae_tensor
    pkg: Dict[str, Any] = {"grid": None, "images": [], "crops": [], "quality": int(image_quality)}
    staged = []  # (device uint8 tensor, consumer) pairs: one synchronisation for all D2H copies

    if save_image_grid:
        n = len(out)
        max_size = max(1, min(max_image_size // n, max(out.shape[-2:])))
        # every row of the mosaic is down-scaled on its own and only the small images are concatenated: the resize works per
        # image, so this equals resizing the concatenation (what save_sampling_results does) without its [4 N, 3, H, W] copy --
        # 2.6 GB for a 300-frame temporal task
        small = []
        if sample.get("skeletons") is not None:
            small.append(_resize_smaller_edge((sample["skeletons"].to(dev, non_blocking=True).float() * 0.5 + 0.5) * 0.8 + inp * 0.2, max_size))
        small.append(_resize_smaller_edge(inp, max_size))
        dimmed = out.clone()
        dimmed[input_indices] *= 0.2
        small.append(_resize_smaller_edge(dimmed, max_size))
        del dimmed
        small.append(_resize_smaller_edge((out - inp).abs().clamp(0, 1), max_size))
        mosaic = torch.cat(small)
        axis = "spa" if sample["domain"] == "temporal" else "tem"
        path = f'{output_dir}/grids/alt{sample["alt"]}_{axis}{sample["domain_label"]}.webp'
        grid = make_image_grid_device(mosaic, nrow=n, padding=2)
        staged.append(((grid * 255.0 + 0.5).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).contiguous(), ("grid", path)))

    crops = sample.get("crops")
    if crops is None:
        crops = [None] * len(out)
    fully = sample["fully_denoised"]
    rows_to_save, meta = [], []
    for i, (crop, (_, spa_label, tem_label)) in enumerate(zip(crops, sample["labels"])):
        if save_output_image:
            path = f"{output_dir}/images/{spa_label}/{tem_label}{image_ext}"
            if (not bool(fully[i]) and i in target_indices) or os.path.isfile(path):
                continue  # still noisy, or written by an earlier task: the reference skips the row's crop file too
            rows_to_save.append(i)
            meta.append((path, crop))
        if save_crop_param:
            pkg["crops"].append((f"{output_dir}/crops/{spa_label}/{tem_label}.json", crop))
    if rows_to_save:
        idx = torch.tensor(rows_to_save, device=dev)
        is_input = torch.zeros(len(out), dtype=torch.bool, device=dev)
        is_input[input_indices] = True
        sel = torch.where(is_input[idx][:, None, None, None], inp[idx], out[idx])  # input views are saved from the input images
        u8 = (sel.clamp(0, 1) * 255.0).to(torch.uint8).permute(0, 2, 3, 1).contiguous()  # to_pil_image: mul(255).byte()
        staged.append((u8, ("images", meta)))
    host = []
    for t, tag in staged:
        if dev.type == "cuda":
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
        else:  # CPU tensors (tests): nothing to stage
            h = t
        host.append((h, tag))
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()
    for h, (kind, info) in host:
        arr = h.numpy()
        if kind == "grid":
            pkg["grid"] = (info, arr)
        else:
            pkg["images"] = [(path, arr[k], crop) for k, (path, crop) in enumerate(info)]
    return pkg


def write_packed_results(sample: Dict[str, Any], output_dir: str = "./results", **_kw) -> None:
    """``result_writer`` of a sampler built with ``device_results=True`` when no writer-process pool is in use: the package
    ``denoise`` left in ``sample["_package"]`` is written on the calling thread."""
    pkg = sample.get("_package")
    if pkg is None:
        raise ValueError("write_packed_results: the sample carries no '_package' (was it denoised with device_results on a HIP device?)")
    write_package(pkg)


def save_sampling_results(sample: Dict[str, Any], output_dir: str = "./results", save_image_grid: bool = True,
                          save_output_image: bool = True, save_crop_param: bool = False, image_ext: str = ".jpg",
                          image_quality: int = 90, max_image_size: int = 8192) -> None:
    from PIL import Image
    output_images = sample["images"].clone().float()  # the caller keeps its tensor (the reference edits it in place)
    input_indices = sample["input_indices"]
    target_indices = set(int(i) for i in sample["target_indices"])
    input_images = sample["pixel_values"].float() * 0.5 + 0.5  # denorm_vae_tensor

    if save_image_grid:
        errors = (output_images - input_images).abs().clamp(0, 1)
        dimmed = output_images.clone()
        dimmed[input_indices] *= 0.2
        rows = [input_images, dimmed, errors]
        if sample.get("skeletons") is not None:
            rows.insert(0, (sample["skeletons"].float() * 0.5 + 0.5) * 0.8 + input_images * 0.2)
        mosaic = torch.cat(rows)
        n = len(output_images)
        max_size = min(max_image_size // n, max(mosaic.shape[-2:]))
        mosaic = _resize_smaller_edge(mosaic, max(1, max_size))
        axis = "spa" if sample["domain"] == "temporal" else "tem"
        path = f'{output_dir}/grids/alt{sample["alt"]}_{axis}{sample["domain_label"]}.webp'
        os.makedirs(os.path.dirname(path), exist_ok=True)
        grid = make_image_grid(mosaic, nrow=n, padding=2, pad_value=0.0)
        arr = (grid * 255.0 + 0.5).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).numpy()
        Image.fromarray(arr).save(path)

    output_images[input_indices] = input_images[input_indices]
    crops = sample.get("crops")
    if crops is None:
        crops = [None] * len(output_images)
    for i, (img, crop, (_, spa_label, tem_label)) in enumerate(zip(output_images, crops, sample["labels"])):
        if save_output_image:
            path = f"{output_dir}/images/{spa_label}/{tem_label}{image_ext}"
            if not bool(sample["fully_denoised"][i]) and i in target_indices:
                continue  # still noisy
            if os.path.isfile(path):
                continue  # e.g. input views written by an earlier task
            os.makedirs(os.path.dirname(path), exist_ok=True)
            restore_cropped_image(_to_pil(img), crop).save(path, quality=image_quality)
        if save_crop_param:
            cpath = f"{output_dir}/crops/{spa_label}/{tem_label}.json"
            os.makedirs(os.path.dirname(cpath), exist_ok=True)
            with open(cpath, "w") as f:
                json.dump(None if crop is None else [int(v) for v in crop], f, indent=4)


def check_sampling_results(spa_labels, tem_labels, output_dir: str) -> bool:
    found = len(glob(f"{output_dir}/images/**/*.*"))
    expected = len(spa_labels) * len(tem_labels)
    if found != expected:
        log.warning("Found incomplete sampling results: Num of saved images: %d != Num of expected images: %d in %s.",
                    found, expected, output_dir)
        return False
    log.info("Found complete results in %s.", output_dir)
    return True


def write_nerfstudio_transforms(data_dir: str, result_dir: str, input_cameras: Optional[Sequence[str]] = None) -> None:
    """Camera files for 4DGS reconstruction from the sampled views (scripts/nerfstudio/diffuman4d_to_nerfstudio.py:14-35):
    ``transforms.json`` with every frame's ``file_path`` pointed at ``images_alpha/*.png`` and ``transforms_input.json``
    holding only the input cameras."""
    with open(f"{data_dir}/transforms.json") as f:
        cameras = json.load(f)
    cameras_input = None
    if input_cameras is not None:
        cameras_input = deepcopy(cameras)
        cameras_input["frames"] = []
    for frame in cameras["frames"]:
        ext = os.path.splitext(frame["file_path"])[1]
        frame["file_path"] = frame["file_path"].replace(ext, ".png").replace("images/", "images_alpha/")
        if cameras_input is not None and frame.get("camera_label") in input_cameras:
            cameras_input["frames"].append(frame)
    os.makedirs(result_dir, exist_ok=True)
    with open(f"{result_dir}/transforms.json", "w") as f:
        json.dump(cameras, f, indent=4)
    if cameras_input is not None:
        with open(f"{result_dir}/transforms_input.json", "w") as f:
            json.dump(cameras_input, f, indent=4)
