"""Operator replay: every launch of a REAL model call recomputed on the CPU from the tensors the device was given.
TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Why.  The fast precision stores bf16 between kernels, and two bf16 networks that differ anywhere -- a summation order is enough --
decorrelate within about five rounding layers (a perturbation d of a value that is then rounded changes the rounding error by
sqrt(ulp * d) rms: 1e-6 -> 5e-5 -> 4e-4 -> 1e-3 -> the ulp itself).  Measured in round 4: the HIP UNet and the rounding-matched oracle
(oracle/matched.py) sit 1.06e-2 and 1.07e-2 from the fp32 oracle and 1.29e-2 from EACH OTHER.  No whole-network comparison of the fast
precision can therefore be tighter than its own rounding noise; the loose model-level bound (1.15 x the bf16-oracle yardstick) is the
best such a comparison can do.  What CAN be tight is a comparison that never lets the noise compound: ``diffuman4d_amd.host.ops.TRACE``
records (operator, inputs, output) for every launch of a model call, and this module recomputes each launch in fp64 from those very
inputs -- the tensors the previous HIP launch produced -- rounds once where the operator rounds, and compares.  What is left between
the two is the kernel's fp32 summation order and the hardware's exp2 / rcp, i.e. the roundings THOSE flip: 3e-5 .. 2e-4 rel-L2
(tests/modelcheck.py ``*_opreplay``: fixed bound REPLAY_TOL).  A mis-rounded epilogue, a wrong flag, a stride or tile bug on a shape only
the real model uses shows up here at the size of its effect, in the launch that has it.  (A wrong argument the HOST passes -- an eps, a
scale -- is replayed wrongly too; those are what precision="parity" against the fp32 oracle at 1e-5 catches, the host wiring being
shared by both precisions.)

Each function restates the documented semantics of one wrapper of ``diffuman4d_amd/host/ops.py`` (which cites the reference call site
it replaces in include/dm4d.h); big launches are checked on a subset of rows / samples / queries -- every operator here is independent
per row (GEMM, LayerNorm, fused feed-forward), per sample (convolution, GroupNorm) or per query (attention).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

BF = torch.bfloat16
MAX_ROWS = 2048      # rows of a GEMM / LayerNorm / feed-forward launch that are recomputed
MAX_SAMPLES = 2      # images of a convolution / GroupNorm launch
MAX_QUERIES = 128    # queries per (batch, head) of an attention launch


def _cpu(t):
    return None if t is None else t.detach().cpu()


def _take(t, idx):
    """Rows / samples `idx` of a (device) tensor, selected where the tensor lives so that only the subset crosses to the host."""
    return None if t is None else t.detach()[idx.to(t.device)].cpu()


def _r(v: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """Round the fp64 reference the way the operator's output is stored."""
    return v.to(like.dtype).double()


def rel_l2(a, b) -> float:
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _row_subset(m: int) -> torch.Tensor:
    if m <= MAX_ROWS:
        return torch.arange(m)
    blocks = MAX_ROWS // 32
    starts = torch.linspace(0, m - 32, blocks).long()
    return (starts[:, None] + torch.arange(32)[None]).reshape(-1).unique()


def _gemm_ref(a, w, a2, bias, rowbias, rows_per_rowbias, residual, geglu, silu, out_scale, rows):
    """a, a2, residual: already the rows `rows`; rowbias: whole."""
    x = a.double() if a2 is None else torch.cat([a, a2], dim=1).double()
    v = x @ w.double().t()
    if bias is not None:
        v = v + bias.double()
    if geglu:
        h, g = v.chunk(2, dim=-1)
        v = h * F.gelu(g)
    if silu:
        v = F.silu(v)
    if rowbias is not None:
        v = v + rowbias.double()[rows // rows_per_rowbias]
    if residual is not None:
        v = v + residual.double()
    return v * out_scale


def replay_gemm(inp: Dict, out: torch.Tensor) -> float:
    rows = _row_subset(inp["a"].shape[0])
    v = _gemm_ref(_take(inp["a"], rows), _cpu(inp["w"]), _take(inp["a2"], rows), _cpu(inp["bias"]), _cpu(inp["rowbias"]),
                  inp["rows_per_rowbias"], _take(inp["residual"], rows), inp["geglu"], inp["silu"], inp["out_scale"], rows)
    got = _take(out, rows)
    if inp.get("split_out"):
        n = v.shape[1]
        return rel_l2(got[:, :n].double() + got[:, n:2 * n].double(), v)
    return rel_l2(got[:, : v.shape[1]], _r(v, got))


def _conv_ref(x, wt, bias, stride, pad, pad_hi, upsample):
    cout, cin = wt.shape[0], x.shape[-1]
    w4 = wt.double().view(cout, 3, 3, cin).permute(0, 3, 1, 2)
    xi = x.double().permute(0, 3, 1, 2)
    if upsample:
        xi = F.interpolate(xi, scale_factor=2, mode="nearest")
    ph = pad if pad_hi is None else pad_hi
    return F.conv2d(F.pad(xi, (pad, ph, pad, ph)), w4, None if bias is None else bias.double(), stride=stride).permute(0, 2, 3, 1)


def _sample_subset(b: int, flops_per_sample: float = 0.0) -> torch.Tensor:
    """First and last sample of the batch; the last one alone when a sample costs more than 5 GFLOP to recompute in fp64."""
    if b <= MAX_SAMPLES and flops_per_sample <= 5e9:
        return torch.arange(b)
    return torch.tensor([b - 1]) if flops_per_sample > 5e9 else torch.tensor([0, b - 1])


def replay_conv3x3(inp: Dict, out: torch.Tensor) -> float:
    x = inp["x"]
    sel = _sample_subset(x.shape[0], 2.0 * out[0].numel() * inp["wt"].shape[1])
    v = _conv_ref(_take(x, sel), _cpu(inp["wt"]), _cpu(inp["bias"]), inp["stride"], inp["pad"], inp["pad_hi"], inp["upsample"])
    if inp["rowbias"] is not None:
        v = v + _take(inp["rowbias"], sel).double()[:, None, None, :]
    got = _take(out, sel)
    if inp["residual"] is not None:
        v = v + _take(inp["residual"].reshape(out.shape), sel).double()
    return rel_l2(got, _r(v * inp["out_scale"], got))


def replay_conv_up2x(inp: Dict, out: torch.Tensor) -> float:
    """The four 2 x 2 phase convolutions of the low-resolution input, weights wp [4][Cout][(dy, dx, ci)] as prepared at load."""
    wp = _cpu(inp["wp"])
    sel = _sample_subset(inp["x"].shape[0], 2.0 * out[0].numel() * wp.shape[2])
    xi = _take(inp["x"], sel).double().permute(0, 3, 1, 2)
    B, cin, H, W = xi.shape
    cout = wp.shape[1]
    y = torch.zeros(B, cout, 2 * H, 2 * W, dtype=torch.float64)
    for py in (0, 1):
        for px in (0, 1):
            w4 = wp[2 * py + px].double().view(cout, 2, 2, cin).permute(0, 3, 1, 2)
            y[:, :, py::2, px::2] = F.conv2d(F.pad(xi, (1 - px, px, 1 - py, py)), w4)
    if inp["bias"] is not None:
        y = y + _cpu(inp["bias"]).double()[None, :, None, None]
    got = _take(out, sel)
    return rel_l2(got, _r(y.permute(0, 2, 3, 1), got))


def replay_groupnorm(inp: Dict, out: torch.Tensor) -> float:
    sel = _sample_subset(inp["x1"].shape[0])
    x1, x2 = _take(inp["x1"], sel), _take(inp["x2"], sel)
    x = x1 if x2 is None else torch.cat([x1, x2], dim=-1)
    b, c = x.shape[0], x.shape[-1]
    v = F.group_norm(x.double().reshape(b, -1, c).permute(0, 2, 1), inp["groups"], _cpu(inp["gamma"]).double(), _cpu(inp["beta"]).double(),
                     inp["eps"]).permute(0, 2, 1)
    if inp["silu"]:
        v = F.silu(v)
    got = _take(out, sel).reshape(b, -1, c)
    return rel_l2(got, _r(v, got))


def replay_layernorm(inp: Dict, out: torch.Tensor) -> float:
    rows = _row_subset(inp["x"].shape[0])
    x = _take(inp["x"], rows)
    v = F.layer_norm(x.double(), (x.shape[1],), _cpu(inp["gamma"]).double(), _cpu(inp["beta"]).double(), inp["eps"])
    got = _take(out, rows)
    return rel_l2(got, _r(v, got))


def replay_attention(inp: Dict, out: torch.Tensor, first_tile: int = 64) -> float:
    """csrc/attention.hip: S in fp32, P = exp2(S' - m) with m the row maximum over the FIRST 64 keys (S' = S scale log2 e, already in
    the Q rows when q_scaled), P rounded to bf16 for P V, O / l rounded.  Row sums: of the unrounded P in attn_kernel; of the ROUNDED P in
    the hand-placed attn64_kernel (the sums run on the matrix pipe against the packed probabilities), which takes the launches with a
    pre-scaled Q and a whole number (>= 3) of 64-key tiles (attention_launch)."""
    batch, heads, seq, kv = inp["batch"], inp["heads"], inp["seq"], inp["kv_seq"]
    c = 1.0 if inp["q_scaled"] else (0.125 if inp["scale"] is None else inp["scale"]) * 1.4426950408889634
    hand_placed = bool(inp["q_scaled"]) and kv % 64 == 0 and kv >= 192
    qs = torch.linspace(0, seq - 1, min(seq, MAX_QUERIES)).long().unique()
    bs = range(batch) if batch <= 4 else (0, batch - 1)  # 2-D attention folds nothing: 32 - 48 sequences, two of them are recomputed
    want, have = [], []
    for b in bs:
        q, got = _take(inp["q"], b * seq + qs), _take(out, b * seq + qs)
        k, v = _cpu(inp["k"][b * kv:(b + 1) * kv]), _cpu(inp["v"][b * kv:(b + 1) * kv])
        for h in range(heads):
            cols = slice(h * 64, (h + 1) * 64)
            qq, kk, vv = q[:, cols].double(), k[:, cols].double(), v[:, cols].double()
            s = (qq @ kk.t()) * c
            p = torch.exp2(s - s[:, :first_tile].amax(dim=-1, keepdim=True))
            pr = p.to(BF).double()
            want.append((pr @ vv) / (pr if hand_placed else p).sum(dim=-1, keepdim=True))
            have.append(got[:, cols])
    want, have = torch.cat(want), torch.cat(have)
    return rel_l2(have, _r(want, have))


def _ff_ref(n, ln, w1, b1, w2, b2, residual):
    if ln is not None:
        n = F.layer_norm(n, (n.shape[1],), ln[0].double(), ln[1].double(), ln[2]).to(BF).double()  # norm3's output is a bf16 tensor
    u, g = (n @ w1.double().t() + (b1.double() if b1 is not None else 0.0)).chunk(2, dim=-1)
    hid = (u * F.gelu(g)).to(BF).double()  # the hidden tensor is rounded between the two products
    return hid @ w2.double().t() + (b2.double() if b2 is not None else 0.0) + residual


def replay_ff_fused(inp: Dict, out: torch.Tensor) -> float:
    rows = _row_subset(inp["n"].shape[0])
    ln = None if inp["ln"] is None else (_cpu(inp["ln"][0]), _cpu(inp["ln"][1]), inp["ln"][2])
    v = _ff_ref(_take(inp["n"], rows).double(), ln, _cpu(inp["w1"]), _cpu(inp["b1"]), _cpu(inp["w2"]), _cpu(inp["b2"]),
                _take(inp["residual"], rows).double())
    got = _take(out, rows)
    return rel_l2(got, _r(v, got))


def replay_attn_out_ff_fused(inp: Dict, out: torch.Tensor) -> float:
    """h = a wo^T + bo + x rounded once (it is stored), then ff(LayerNorm(h)) + h."""
    rows = _row_subset(inp["a"].shape[0])
    h = (_take(inp["a"], rows).double() @ _cpu(inp["wo"]).double().t() + (_cpu(inp["bo"]).double() if inp["bo"] is not None else 0.0)
         + _take(inp["x"], rows).double())
    h = h.to(BF).double()
    ln = (_cpu(inp["ln"][0]), _cpu(inp["ln"][1]), inp["ln"][2])
    v = _ff_ref(h, ln, _cpu(inp["w1"]), _cpu(inp["b1"]), _cpu(inp["w2"]), _cpu(inp["b2"]), h)
    got = _take(out, rows)
    return rel_l2(got, _r(v, got))


def replay_softmax_rows(inp: Dict, out: torch.Tensor) -> float:
    rows = _row_subset(inp["s"].shape[0])
    s = _take(inp["s"], rows)
    n = s.shape[1] if inp["n"] is None else inp["n"]
    v = torch.softmax(s[:, :n].double() * inp["scale"], dim=-1)
    got = _take(out, rows)[:, :n]
    return rel_l2(got, _r(v, got))


def replay_silu(inp: Dict, out: torch.Tensor) -> float:
    got = _cpu(out)
    return rel_l2(got, _r(F.silu(_cpu(inp["x"]).double()), got))


REPLAY = {"gemm": replay_gemm, "conv3x3": replay_conv3x3, "conv_up2x": replay_conv_up2x, "groupnorm": replay_groupnorm,
          "layernorm": replay_layernorm, "attention": replay_attention, "ff_fused": replay_ff_fused,
          "attn_out_ff_fused": replay_attn_out_ff_fused, "softmax_rows": replay_softmax_rows, "silu": replay_silu}


def replay_trace(trace: List[Tuple[str, Dict, torch.Tensor]], every: int = 1) -> Dict[str, Dict]:
    """Recompute every `every`-th launch of each operator in `trace`; -> {operator: {launches, checked, worst, mean, where}}."""
    stats: Dict[str, Dict] = {}
    seen: Dict[str, int] = {}
    for i, (name, inp, out) in enumerate(trace):
        st = stats.setdefault(name, {"launches": 0, "checked": 0, "worst": 0.0, "sum": 0.0, "where": None})
        st["launches"] += 1
        k = seen[name] = seen.get(name, 0) + 1
        if name not in REPLAY or (k - 1) % every != 0:
            continue
        e = REPLAY[name](inp, out)
        st["checked"] += 1
        st["sum"] += e
        if e > st["worst"]:
            st["worst"], st["where"] = e, f"launch {i} of the call, output {tuple(out.shape)}"
    for st in stats.values():
        st["mean"] = st.pop("sum") / max(1, st["checked"])
    return stats
