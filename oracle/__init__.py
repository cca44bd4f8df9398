"""CPU oracle for the Diffuman4D sliding iterative denoiser hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``diffuman4d_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker.

It is a plain-PyTorch (CPU, fp32 or bf16) restatement of the reference's
algorithm:

* reference-owned code (window schedule, CFG, per-latent scheduler stepping,
  UNet wiring, 3-D attention frame folding) is restated from
  ``/root/reference/src/**`` and **pinned** against golden vectors produced by
  importing the reference's own modules (see ``tests/golden/make_golden.py``);
* arithmetic that lives in the un-vendored dependency ``diffusers==0.33.1``
  (``/root/reference/requirements.txt:5``: ResnetBlock2D, Attention,
  Transformer2DModel, AutoencoderKL, DDIMScheduler ...) is restated from that
  release's published algorithm.  The package is not installed here and no
  checkpoint is available, so that part is **parity unpinned** (DESIGN.md §3).
"""
