"""Image-file half of the result writer: PIL + numpy only (NO torch import), so that it can run in writer PROCESSES.

``write_package`` is the consumer of ``results.pack_results_on_device``: by the time a package reaches it every pixel is
final (uint8, HWC); what is left is exactly the host work the reference does per image and per task --
``restore_cropped_image`` (``/root/reference/src/data/utils/image_utils.py:62-93``), ``Image.save(path, quality=90)``
(``/root/reference/src/samplers/utils/sampling_utils.py:95-114``) and the ``.webp`` snapshot mosaic (:70-93).  JPEG / WebP
encoding is CPU work of 3-5 ms per 576 x 320 image and 1-2 s per mosaic; with writer THREADS it competes with the loader
and GPU-launch threads for the interpreter lock, in processes it does not (runner.run_round_pipelined, ``writer_processes``).
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, Optional, Sequence


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


def write_package(pkg: Dict[str, Any]) -> int:
    """Write one task's files.  pkg = {"grid": (path, uint8 [H, W, 3]) | None, "images": [(path, uint8 [H, W, 3], crop | None)],
    "crops": [(path, crop | None)], "quality": int}.  Returns the number of image files written."""
    from PIL import Image
    grid = pkg.get("grid")
    if grid is not None:
        path, arr = grid
        os.makedirs(os.path.dirname(path), exist_ok=True)
        Image.fromarray(arr).save(path)
    written = 0
    for path, arr, crop in pkg.get("images", ()):
        if os.path.isfile(path):
            continue  # e.g. input views written by an earlier task (sampling_utils.py:105-106)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        restore_cropped_image(Image.fromarray(arr), crop).save(path, quality=pkg.get("quality", 90))
        written += 1
    for path, crop in pkg.get("crops", ()):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(None if crop is None else [int(v) for v in crop], f, indent=4)
    return written


# ---------------------------------------------------------------------------------------------------------------------
# writer processes: `python -m diffuman4d_amd.host.imgwrite` reads length-prefixed pickled packages on stdin and answers
# one line per package on stdout.  Plain subprocesses instead of multiprocessing: a spawned / fork-served multiprocessing child
# re-imports the parent's __main__ (inference.py: torch, the HIP runtime, ...), these import PIL and numpy only.
# ---------------------------------------------------------------------------------------------------------------------
def _worker_main() -> None:
    import pickle
    import struct
    import sys
    import traceback
    rd, wr = sys.stdin.buffer, sys.stdout.buffer
    while True:
        head = rd.read(8)
        if len(head) < 8:
            return  # parent closed the pipe
        (n,) = struct.unpack("<Q", head)
        blob = rd.read(n)
        try:
            msg = "ok %d\n" % write_package(pickle.loads(blob))
        except BaseException:  # noqa: BLE001 -- reported to the parent, which raises it on the caller's thread
            msg = "err " + traceback.format_exc().replace("\n", "\\n") + "\n"
        wr.write(msg.encode())
        wr.flush()


class WriterPool:
    """N writer processes fed through pipes.  ``submit(pkg)`` returns a concurrent.futures.Future that resolves to the number of
    image files written (or raises what the worker raised).  Packages are handed to whichever worker is free."""

    def __init__(self, processes: int):
        import queue
        import subprocess
        import sys
        import threading
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        env = dict(os.environ)
        env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        env.setdefault("OMP_NUM_THREADS", "1")
        self._q: "queue.Queue" = queue.Queue()
        self._procs, self._threads = [], []
        self._lock, self._alive = threading.Lock(), max(1, int(processes))
        for _ in range(max(1, int(processes))):
            p = subprocess.Popen([sys.executable, "-m", "diffuman4d_amd.host.imgwrite"], stdin=subprocess.PIPE,
                                 stdout=subprocess.PIPE, env=env)  # the parent's cwd: output paths may be relative to it
            t = threading.Thread(target=self._feed, args=(p,), name="dm4d-writer-feed", daemon=True)
            t.start()
            self._procs.append(p)
            self._threads.append(t)

    def _feed(self, p) -> None:
        import pickle
        import struct
        while True:
            item = self._q.get()
            if item is None:
                return
            pkg, fut = item
            try:
                blob = pickle.dumps(pkg, protocol=pickle.HIGHEST_PROTOCOL)
                p.stdin.write(struct.pack("<Q", len(blob)))
                p.stdin.write(blob)
                p.stdin.flush()
                line = p.stdout.readline().decode()
                if line.startswith("ok "):
                    fut.set_result(int(line.split()[1]))
                else:
                    fut.set_exception(RuntimeError("writer process: " + line[4:].replace("\\n", "\n") if line else
                                                   f"writer process exited with code {p.poll()}"))
            except BaseException as e:  # noqa: BLE001
                fut.set_exception(e)
            if p.poll() is not None:  # the worker is gone (killed, out of memory): stop feeding it; the others take the queue
                with self._lock:
                    self._alive -= 1
                    last = self._alive == 0
                while last:  # nobody left to write: fail what is queued instead of leaving its futures pending for ever
                    item = self._q.get()
                    if item is None:
                        return
                    item[1].set_exception(RuntimeError(f"all writer processes have exited (last code {p.poll()})"))
                return

    def submit(self, pkg: Dict[str, Any]):
        from concurrent.futures import Future
        fut: Future = Future()
        self._q.put((pkg, fut))
        return fut

    def shutdown(self) -> None:
        for _ in self._threads:
            self._q.put(None)
        for t in self._threads:
            t.join()
        for p in self._procs:
            try:
                p.stdin.close()
            except Exception:  # noqa: BLE001
                pass
            p.wait(timeout=60)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.shutdown()


if __name__ == "__main__":
    _worker_main()
