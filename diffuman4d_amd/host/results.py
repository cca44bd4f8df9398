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


def restore_cropped_image(image, crop_param: Optional[Sequence[int]], ori_size=None, background_color: str = "white"):
    """Undo ``crop (ct, cl, ch, cw)`` + ``resize to (h, w)`` (image_utils.py:62-93): bicubic-resize the image back to
    the crop's ``(ch, cw)`` and paste it at ``(cl, ct)`` of a ``(w, h)`` canvas; parts of the crop that lay outside the
    original frame (negative ``ct`` / ``cl``, or a crop larger than the frame) are cut off, uncovered canvas is white."""
    from PIL import Image
    if crop_param is None:
        return image
    crop_param = tuple(int(v) for v in crop_param)
    if len(crop_param) == 4:
        ct, cl, ch, cw = crop_param
        w, h = image.size
    elif len(crop_param) == 6:
        ct, cl, ch, cw, h, w = crop_param
    else:
        raise ValueError(f"Invalid crop_param: {crop_param}")
    patch = image.resize((cw, ch), Image.BICUBIC)
    canvas = Image.new(image.mode, (w, h), (255, 255, 255) if background_color == "white" else (0, 0, 0))
    canvas.paste(patch, (cl, ct))  # PIL clips what falls outside the canvas
    return canvas


def _resize_smaller_edge(x: torch.Tensor, size: int) -> torch.Tensor:
    """torchvision ``resize(tensor, int)``: the smaller edge becomes ``size``, aspect kept, antialiased bilinear."""
    h, w = x.shape[-2:]
    if min(h, w) == size:
        return x
    if h <= w:
        nh, nw = size, max(1, int(size * w / h))
    else:
        nh, nw = max(1, int(size * h / w)), size
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
