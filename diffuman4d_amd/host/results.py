"""On-disk result contract of the sampler (PIL only; the reference uses torchvision).

Mirror of ``/root/reference/src/samplers/utils/sampling_utils.py:54-129``: fully denoised target
views and the input views are written to ``{output_dir}/images/{cam}/{frame}.jpg`` (quality 90,
crop restored onto a white canvas when the dataset provides one); ``check_sampling_results`` counts
them.  The debug mosaics (``grids/*.webp``, :70-93) are cosmetic and not produced.
"""
from __future__ import annotations

import os
from glob import glob
from typing import Any, Dict

import numpy as np
import torch


def _to_pil(img: torch.Tensor):
    from PIL import Image
    arr = (img.detach().float().clamp(0, 1) * 255.0).round().to(torch.uint8).permute(1, 2, 0).cpu().numpy()
    return Image.fromarray(arr)


def restore_cropped_image(image, crop: Dict[str, Any]):
    """image_utils.py:62-93 contract: paste the (resized) crop back onto a white full-size canvas."""
    from PIL import Image
    if not crop:
        return image
    x0, y0, x1, y1 = [int(v) for v in crop["bbox"]]
    full_w, full_h = [int(v) for v in crop["size"]]
    canvas = Image.new("RGB", (full_w, full_h), (255, 255, 255))
    canvas.paste(image.resize((x1 - x0, y1 - y0), Image.BICUBIC), (x0, y0))
    return canvas


def save_sampling_results(sample: Dict[str, Any], output_dir: str = "./results", image_ext: str = ".jpg",
                          image_quality: int = 90) -> None:
    output_images = sample["images"].clone()
    input_indices = sample["input_indices"]
    target_indices = set(int(i) for i in sample["target_indices"])
    input_images = (sample["pixel_values"].float() / 2 + 0.5).clamp(0, 1)  # denorm_vae_tensor
    output_images[input_indices] = input_images[input_indices]
    crops = sample.get("crops") or [None] * len(output_images)
    for i, (img, crop, (_, spa_label, tem_label)) in enumerate(zip(output_images, crops, sample["labels"])):
        path = f"{output_dir}/images/{spa_label}/{tem_label}{image_ext}"
        if not bool(sample["fully_denoised"][i]) and i in target_indices:
            continue  # still noisy
        if os.path.isfile(path):
            continue  # e.g. input views written by an earlier task
        os.makedirs(os.path.dirname(path), exist_ok=True)
        restore_cropped_image(_to_pil(img), crop).save(path, quality=image_quality)


def check_sampling_results(spa_labels, tem_labels, output_dir: str) -> bool:
    return len(glob(f"{output_dir}/images/**/*.*")) == len(spa_labels) * len(tem_labels)
