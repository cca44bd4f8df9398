"""Result writer vs the reference's on-disk contract (sampling_utils.py:54-129, image_utils.py:62-93). CPU only."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

from diffuman4d_amd.host import results


def _ref_restore(image: Image.Image, crop_param, background_color="white"):
    """Literal restatement of image_utils.restore_cropped_image (:62-93): resize back to (ch, cw), drop onto a
    2h x 2w canvas at (h/2 + ct, w/2 + cl), cut the middle h x w out."""
    if len(crop_param) == 4:
        ct, cl, ch, cw = crop_param
        w, h = image.size
    else:
        ct, cl, ch, cw, h, w = crop_param
    img = np.asarray(image.resize((cw, ch), Image.BICUBIC)).astype(np.float32) / 255.0
    canvas = np.ones((h * 2, w * 2, 3), np.float32) if background_color == "white" else np.zeros((h * 2, w * 2, 3), np.float32)
    top, left = h // 2 + ct, w // 2 + cl
    canvas[top:top + ch, left:left + cw] = img
    out = canvas[h // 2: h * 3 // 2, w // 2: w * 3 // 2]
    return Image.fromarray((out * 255.0).astype(np.uint8))


def _rand_image(w, h, seed=0):
    rng = np.random.default_rng(seed)
    return Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))


@pytest.mark.parametrize("crop", [
    (5, 4, 30, 20),                   # 4-tuple: canvas = the image's own size
    (10, 20, 40, 30, 96, 80),         # 6-tuple: explicit original (h, w)
    (-7, -5, 60, 50, 64, 48),         # negative top / left: the crop window started outside the frame
    (30, 25, 60, 40, 64, 48),         # crop runs past the bottom / right edge
    (0, 0, 64, 48, 64, 48),           # identity geometry
])
def test_restore_cropped_image_matches_reference(crop):
    img = _rand_image(32, 40, seed=sum(abs(c) for c in crop))
    got = results.restore_cropped_image(img, crop)
    ref = _ref_restore(img, crop)
    assert got.size == ref.size
    assert np.array_equal(np.asarray(got), np.asarray(ref))


def test_restore_cropped_image_rejects_other_lengths_and_passes_none():
    img = _rand_image(8, 8)
    assert results.restore_cropped_image(img, None) is img
    with pytest.raises(ValueError, match="Invalid crop_param"):
        results.restore_cropped_image(img, (1, 2, 3))


def _sample(n=4, H=24, W=16, crops=None, skeletons=True):
    g = torch.Generator().manual_seed(3)
    return {
        "images": torch.rand(n, 3, H, W, generator=g),
        "pixel_values": torch.rand(n, 3, H, W, generator=g) * 2 - 1,
        "skeletons": (torch.rand(n, 3, H, W, generator=g) * 2 - 1) if skeletons else None,
        "input_indices": torch.tensor([0]), "target_indices": torch.tensor([1, 2, 3]),
        "fully_denoised": torch.tensor([False, True, False, True]),
        "labels": [("scene", f"{c:02d}", "000007") for c in range(n)],
        "crops": crops, "alt": 2, "domain": "spatial", "domain_label": "000007",
    }


def test_save_sampling_results_layout(tmp_path):
    crops = [(2, 3, 30, 20, 40, 32)] * 4  # the tuples SpaTemDataset hands out (spatem_dataset.py:58,157)
    s = _sample(crops=crops)
    before = s["images"].clone()
    results.save_sampling_results(s, output_dir=str(tmp_path), save_crop_param=True)
    assert torch.equal(s["images"], before)  # the caller's tensor is left alone
    # input view 00 and the two fully denoised targets are written; the still-noisy target 02 is not
    names = sorted(os.path.relpath(p, tmp_path) for p in (tmp_path / "images").rglob("*.jpg"))
    assert names == ["images/00/000007.jpg", "images/01/000007.jpg", "images/03/000007.jpg"]
    assert Image.open(tmp_path / "images/01/000007.jpg").size == (32, 40)  # crop undone onto the (w, h) canvas
    # spatial task -> the fixed axis is the frame: grids/alt2_tem000007.webp; 4 rows (skeletons, inputs, outputs, errors)
    grid = Image.open(tmp_path / "grids/alt2_tem000007.webp")
    # the reference sizes by max(H, W) = 24 but torchvision's resize(int) sets the SMALLER edge: 24x16 becomes 36x24
    assert grid.size == (4 * (24 + 2) + 2, 4 * (36 + 2) + 2)
    assert json.load(open(tmp_path / "crops/01/000007.json")) == [2, 3, 30, 20, 40, 32]
    assert not (tmp_path / "crops/02/000007.json").exists()  # the reference's `continue` skips an unsaved view's crop too
    assert not results.check_sampling_results(["00", "01", "02", "03"], ["000007"], str(tmp_path))
    s["fully_denoised"] = torch.tensor([False, True, True, True])
    results.save_sampling_results(s, output_dir=str(tmp_path), save_image_grid=False)
    assert results.check_sampling_results(["00", "01", "02", "03"], ["000007"], str(tmp_path))


def test_grid_without_skeletons_and_downscale(tmp_path):
    s = _sample(skeletons=False, crops=None)
    s["domain"], s["domain_label"] = "temporal", "05"
    results.save_sampling_results(s, output_dir=str(tmp_path), max_image_size=32)  # 32 // 4 frames = 8-pixel smaller edge
    grid = Image.open(tmp_path / "grids/alt2_spa05.webp")
    assert grid.size == (4 * (8 + 2) + 2, 3 * (12 + 2) + 2)


def test_make_image_grid_matches_torchvision_layout():
    imgs = torch.arange(5 * 1 * 2 * 3, dtype=torch.float32).view(5, 1, 2, 3)
    g = results.make_image_grid(imgs, nrow=3, padding=1, pad_value=-1.0)
    assert g.shape == (1, 2 * 3 + 1, 3 * 4 + 1)
    assert torch.equal(g[0, 1:3, 1:4], imgs[0, 0]) and torch.equal(g[0, 4:6, 5:8], imgs[4, 0])
    assert float(g[0, 0, 0]) == -1.0 and float(g[0, 4, 9]) == -1.0  # padding and the empty sixth slot


def test_nerfstudio_transforms(tmp_path):
    src, dst = tmp_path / "data", tmp_path / "out"
    src.mkdir()
    frames = [{"file_path": f"images/{c}/000000.jpg", "camera_label": c} for c in ("00", "01", "13")]
    json.dump({"fl_x": 1.0, "frames": frames}, open(src / "transforms.json", "w"))
    results.write_nerfstudio_transforms(str(src), str(dst), input_cameras=["01", "13"])
    full = json.load(open(dst / "transforms.json"))
    assert [f["file_path"] for f in full["frames"]] == [f"images_alpha/{c}/000000.png" for c in ("00", "01", "13")]
    assert [f["camera_label"] for f in json.load(open(dst / "transforms_input.json"))["frames"]] == ["01", "13"]


def _pack_sample(n, H, W, domain):
    g = torch.Generator().manual_seed(0)
    if domain == "spatial":
        inp = [1, 3]
        tgt = [i for i in range(n) if i not in inp]
        labels = [(i, "%02d" % i, "000007") for i in range(n)]
    else:
        T = n // 2
        inp, tgt = list(range(T)), list(range(T, n))
        labels = [(i, "01", "%06d" % i) for i in range(T)] + [(T + i, "05", "%06d" % i) for i in range(T)]
    fd = torch.zeros(n, dtype=torch.bool)
    fd[tgt[::2]] = True
    return dict(images=torch.rand(n, 3, H, W, generator=g), pixel_values=torch.rand(n, 3, H, W, generator=g) * 2 - 1,
                skeletons=torch.rand(n, 3, H, W, generator=g) * 2 - 1, input_indices=torch.tensor(inp), target_indices=torch.tensor(tgt),
                domain=domain, alt=2, domain_label="000007" if domain == "spatial" else "05", labels=labels, fully_denoised=fd,
                crops=[(3, 5, H - 10, W - 8, H + 6, W + 4)] * n)


@pytest.mark.parametrize("domain,n", [("spatial", 8), ("temporal", 12)])
def test_packed_writer_produces_the_same_files_as_the_reference_order(tmp_path, domain, n):
    """pack_results_on_device (arithmetic, here on CPU tensors) + imgwrite.write_package (PIL only, runs in writer processes)
    write byte-for-byte the files save_sampling_results writes: same mosaic, same JPEG inputs, same crop files, same skips."""
    import filecmp
    import sys
    from glob import glob
    from diffuman4d_amd.host.results import pack_results_on_device, save_sampling_results, write_package
    s = _pack_sample(n, 64, 40, domain)
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    save_sampling_results(s, output_dir=a, save_crop_param=True)
    pkg = pack_results_on_device(s, s["images"], output_dir=b, save_crop_param=True)
    import pickle
    pkg = pickle.loads(pickle.dumps(pkg))  # what crossing into a writer process does to it
    write_package(pkg)
    fa = sorted(os.path.relpath(f, a) for f in glob(a + "/**/*.*", recursive=True))
    fb = sorted(os.path.relpath(f, b) for f in glob(b + "/**/*.*", recursive=True))
    assert fa == fb and len(fa) > 4
    assert [f for f in fa if not filecmp.cmp(os.path.join(a, f), os.path.join(b, f), shallow=False)] == []
    # a second task that shares the input views: they exist already and are skipped (and never staged)
    pkg2 = pack_results_on_device(s, s["images"], output_dir=b)
    assert pkg2["images"] == []


def test_imgwrite_imports_without_torch():
    """The writer-process half must not import torch (a spawned writer pays the import and its threads)."""
    import subprocess
    import sys
    code = "import sys; import diffuman4d_amd.host.imgwrite as m; assert 'torch' not in sys.modules, 'torch imported'; print('ok')"
    from pathlib import Path
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(Path(__file__).resolve().parent.parent))
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr


def test_writer_pool_writes_packages_in_processes_and_surfaces_errors(tmp_path):
    """imgwrite.WriterPool: plain subprocesses fed pickled uint8 packages over pipes; results and worker exceptions come back as
    futures on the submitting side."""
    from diffuman4d_amd.host.imgwrite import WriterPool
    rng = np.random.default_rng(0)
    with WriterPool(2) as pool:
        futs = []
        for k in range(5):
            pkg = {"grid": (str(tmp_path / "grids" / f"g{k}.webp"), rng.integers(0, 255, (40, 300, 3), dtype=np.uint8)),
                   "images": [(str(tmp_path / "images" / f"{k:02d}" / f"{i:06d}.jpg"), rng.integers(0, 255, (64, 40, 3), dtype=np.uint8),
                               (2, 3, 50, 30, 70, 44)) for i in range(3)],
                   "crops": [(str(tmp_path / "crops" / f"{k:02d}" / "000000.json"), (2, 3, 50, 30, 70, 44))], "quality": 90}
            futs.append(pool.submit(pkg))
        assert [f.result(timeout=120) for f in futs] == [3] * 5
        bad = pool.submit({"grid": ("/proc/definitely/not/writable.webp", np.zeros((4, 4, 3), np.uint8))})
        with pytest.raises(RuntimeError, match="writer process"):
            bad.result(timeout=120)
        again = pool.submit({"images": [(str(tmp_path / "images" / "00" / "000000.jpg"), np.zeros((8, 8, 3), np.uint8), None)]})
        assert again.result(timeout=120) == 0  # exists already: skipped, and the pool survived the failed package
    assert len(list(tmp_path.rglob("*.jpg"))) == 15 and len(list(tmp_path.rglob("*.webp"))) == 5
    assert Image.open(tmp_path / "images" / "03" / "000001.jpg").size == (44, 70)  # crop undone onto the (w, h) canvas


def test_writer_pool_survives_a_dead_worker_and_fails_loudly_when_all_are_gone(tmp_path):
    """A writer process that is killed takes at most the package it was writing with it: the remaining processes serve the queue; with
    no process left, submitted packages fail instead of waiting for ever."""
    from diffuman4d_amd.host.imgwrite import WriterPool
    pkg = lambda k: {"images": [(str(tmp_path / "images" / f"{k:02d}" / "000000.jpg"), np.full((16, 16, 3), k, np.uint8), None)]}  # noqa: E731
    pool = WriterPool(2)
    try:
        assert pool.submit(pkg(0)).result(timeout=120) == 1
        pool._procs[0].kill()
        pool._procs[0].wait(timeout=30)
        outcomes = []
        for k in range(1, 7):  # whichever feeder takes a package: the dead worker's fails once, then the live one serves the rest
            f = pool.submit(pkg(k))
            try:
                outcomes.append(f.result(timeout=120))
            except Exception:  # noqa: BLE001
                outcomes.append("failed")
        assert outcomes.count("failed") <= 1 and outcomes.count(1) >= 5
        pool._procs[1].kill()
        pool._procs[1].wait(timeout=30)
        tail = []
        for k in range(7, 10):
            f = pool.submit(pkg(k))
            with pytest.raises(Exception):
                f.result(timeout=120)
            tail.append(f)
        assert all(f.done() for f in tail)
    finally:
        pool.shutdown()
