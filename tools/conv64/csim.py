#!/usr/bin/env python
"""The strip-convolution stream of tools/conv64/cgen.py on the numpy workgroup model of tools/attn64/sim.py, against a direct 3x3
convolution in float64.  `wave_inputs` restates what tools/conv64/conv64.h::conv64_kernel computes on the C++ side (the lane-constant operands of
conv_strip2_kernel).  Test infrastructure (tests/test_conv64_sim.py) -- never imported by the package."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent / "attn64"))
sys.path.insert(0, str(HERE))
import sim as asim  # noqa: E402
from cgen import BM, NW, ROWB, Cfg, ConvProgram, SG  # noqa: E402

U32 = np.uint32


class ConvWorkgroup(asim.Workgroup):
    def _val(self, w, m):
        kind, v = m
        if kind == "s":
            return w.s[v]
        if kind == "in":
            return int(w.inp[v]) & 0xFFFFFFFF
        return v & 0xFFFFFFFF

    def op_sop2(self, w, op, d, a, b):
        x, y = self._val(w, a), self._val(w, b)
        if op == "s_mov_b32":
            w.s[d] = x
        elif op == "s_add_u32":
            t = x + y
            w.s[d], w.scc = t & 0xFFFFFFFF, t >> 32
        elif op == "s_addc_u32":
            t = x + y + w.scc
            w.s[d], w.scc = t & 0xFFFFFFFF, t >> 32
        elif op == "s_sub_u32":
            w.scc = int(x < y)
            w.s[d] = (x - y) & 0xFFFFFFFF
        elif op == "s_lshl_b32":
            w.s[d] = (x << (y & 31)) & 0xFFFFFFFF
            w.scc = int(w.s[d] != 0)
        elif op == "s_mul_i32":
            w.s[d] = (x * y) & 0xFFFFFFFF
        elif op == "s_cselect_b32":
            w.s[d] = x if w.scc else y
        else:
            raise NotImplementedError(op)

    def op_smov64(self, w, d, a):
        if a[0] == "s":
            w.s[d], w.s[d + 1] = w.s[a[1]], w.s[a[1] + 1]
        else:
            v = int(w.inp[a[1]])
            w.s[d], w.s[d + 1] = v & 0xFFFFFFFF, v >> 32

    def op_scmp(self, w, op, a, b):
        x, y = self._val(w, a), self._val(w, b)
        w.scc = int({"s_cmp_eq_u32": x == y, "s_cmp_lg_u32": x != y}[op])

    def op_m0add(self, w, base, imm):
        w.m0 = (w.s[base] + imm) & 0xFFFFFFFF

    def op_save_m0(self, w, d):
        w.s[d] = w.m0

    def op_restore_m0(self, w, d):
        w.m0 = w.s[d]

    def op_dma2(self, w, voff, ptr):
        base = w.s[ptr] | (w.s[ptr + 1] << 32)
        off = w.v[voff[1]] if voff[0] == "v" else w.inp[voff[1]]
        addr = base + off.astype(np.int64)
        data = np.zeros((64, 16), np.uint8)
        for l in range(64):
            assert 0 <= addr[l] and addr[l] + 16 <= self.gmem.size, "DMA source outside the buffers"
            data[l] = self.gmem[addr[l]:addr[l] + 16]
        dst = w.m0
        self.dma_written_phase.append((dst, dst + 1024, w.wid))
        w.dma_pending.append((dst, data))
        w.vm_queue.append(("dma", None))
        if self.land == "issue":
            self.lds[dst:dst + 1024] = data.reshape(-1)

    def op_vadd_in(self, w, d, a, b):
        w.v[d] = (w.inp[a].astype(np.uint64) + w.inp[b].astype(np.uint64)).astype(U32)

    def op_vmov_in(self, w, d, a):
        w.v[d] = np.broadcast_to(np.asarray(w.inp[a], dtype=U32), (64,)).copy()

    def op_vmov_imm(self, w, d, c):
        w.v[d] = np.full(64, c, U32)

    def op_ds_b128v(self, w, d, addr, off):
        a = (w.v[addr[1]] if addr[0] == "v" else w.inp[addr[1]]).astype(np.int64) + off
        assert not (a & 15).any()
        out = np.zeros((4, 64), U32)
        for l in range(64):
            out[:, l] = self.lds_read(w, a[l], 16).view(U32)
        w.v[d:d + 4] = out


def wave_inputs(c: Cfg, wave, P):
    """P: dict(M, N, H, W, Cin, ldw, m0, n0, abase, wbase, bias (uint16[N] or None), lds0)."""
    lane = np.arange(64)
    l31, lh = lane & 31, lane >> 5
    wm, wn = wave // c.WN, wave % c.WN
    TM, TN = c.MI * 32, c.NI * 32
    xs = ys = -1
    d_row, d_pos = lane >> 3, lane & 7
    M, N, H, W, Cin, ldw, m0, n0, lds0 = (P[k] for k in ("M", "N", "H", "W", "Cin", "ldw", "m0", "n0", "lds0"))
    ZROW, BS0 = BM + 8, 2 * c.A_BYTES
    inp = {}
    for ky in range(3):
        for r in range(c.AW):
            row = (wave + NW * r) * 8 + d_row
            px = np.clip(m0 + xs + row + (ky + ys) * W, 0, M - 1)
            inp[f"avoff{ky}_{r}"] = (px * Cin * 2 + ((d_pos ^ ((row >> 1) & 7)) * 16)).astype(U32)
    for r in range(c.BW):
        row = (wave + NW * r) * 8 + d_row
        chunk = d_pos ^ ((row >> 1) & 7)
        n = np.minimum(n0 + row, N - 1)
        inp[f"wvoff{r}"] = ((n * ldw + chunk * 8) * 2).astype(U32)
    edge = []
    for i in range(c.MI):
        m = np.minimum(m0 + wm * TM + i * 32 + l31, M - 1)
        x, y = m % W, (m // W) % H
        edge.append((x == 0) * 1 | (x == W - 1) * 2 | (y == 0) * 4 | (y == H - 1) * 8)
    for kx in range(3):
        swa = ((l31 + kx) >> 1) & 7
        for ks in range(4):
            inp[f"aswz{kx}_{ks}"] = (((ks * 2 + lh) ^ swa) * 16).astype(U32)
    for ky in range(3):
        for kx in range(3):
            for i in range(c.MI):
                e = edge[i]
                ox, oy = kx + xs, ky + ys
                zero = ((ox < 0) & ((e & 1) != 0)) | ((ox > 0) & ((e & 2) != 0)) | ((oy < 0) & ((e & 4) != 0)) | ((oy > 0) & ((e & 8) != 0))
                inp[f"arow{ky}_{kx}_{i}"] = (lds0 + np.where(zero, ZROW * ROWB, (wm * TM + i * 32 + l31 + kx) * ROWB)).astype(U32)
    swb = (l31 >> 1) & 7
    for ks in range(4):
        inp[f"brd{ks}"] = (lds0 + BS0 + (wn * TN + l31) * ROWB + (((ks * 2 + lh) ^ swb) * 16)).astype(U32)
    one = 0x3C00 if c.h16 else 0x3F80
    inp["onew"] = np.where(lh == 0, one, 0).astype(U32)
    for j in range(c.NI):
        n = np.minimum(n0 + wn * TN + j * 32 + l31, N - 1)
        bits = P["bias"][n].astype(U32) if P["bias"] is not None else np.zeros(64, U32)
        inp[f"biasw{j}"] = np.where(lh == 0, bits, 0).astype(U32)
    inp.update(abase=P["abase"], wbase=P["wbase"], cin2=Cin * 2, nci=Cin // 64, adst=lds0 + wave * 1024, bdst=lds0 + BS0 + wave * 1024, wave0=int(wave == 0))
    return inp


def run_case(cfgname="256X128", B=1, H=6, W=20, Cin=128, Cout=128, tile=(0, 0), h16=False, seed=0, land="issue", order=(0, 1, 2, 3), bias=True):
    from cgen import CONFIGS
    c = Cfg(*CONFIGS[cfgname], h16=h16)
    rng = np.random.default_rng(seed)
    M, N = B * H * W, Cout
    x = (rng.standard_normal((M, Cin)) * 0.5).astype(np.float32)
    wt = (rng.standard_normal((N, 9 * Cin)) / np.sqrt(9 * Cin) * 2).astype(np.float32)
    bs = rng.standard_normal(N).astype(np.float32) if bias else None
    xh, wh = asim.f32_to_half(x, h16), asim.f32_to_half(wt, h16)
    bh = asim.f32_to_half(bs, h16) if bias else None
    gmem = rng.integers(0, 255, 1 << 23, dtype=np.uint8)
    abase, wbase = 1 << 16, 1 << 22
    gmem[abase:abase + xh.size * 2] = xh.reshape(-1).view(np.uint8)
    gmem[wbase:wbase + wh.size * 2] = wh.reshape(-1).view(np.uint8)
    tm, tn = tile
    P = dict(M=M, N=N, H=H, W=W, Cin=Cin, ldw=9 * Cin, m0=tm * BM, n0=tn * c.BN, abase=abase, wbase=wbase, bias=bh, lds0=0)
    prog = ConvProgram(c).build()
    wg = ConvWorkgroup(prog, gmem, [wave_inputs(c, w, P) for w in range(NW)], lds_bytes=160 * 1024, land=land)
    wg.lds[:] = rng.integers(1, 255, wg.lds.size, dtype=np.uint8)  # stale bytes everywhere ...
    for par in range(2):                                           # ... but the zero rows (the kernel writes them before the stream)
        z = par * c.A_BYTES + (BM + 8) * ROWB
        wg.lds[z:z + ROWB] = 0
    wg.run(order=order)
    # reference
    x64 = asim.half_to_f32(xh, h16).astype(np.float64).reshape(B, H, W, Cin)
    w64 = asim.half_to_f32(wh, h16).astype(np.float64).reshape(N, 3, 3, Cin)
    xp = np.zeros((B, H + 2, W + 2, Cin))
    xp[:, 1:-1, 1:-1] = x64
    ref = np.zeros((B, H, W, N))
    for ky in range(3):
        for kx in range(3):
            ref += xp[:, ky:ky + H, kx:kx + W] @ w64[:, ky, kx].T
    if bias:
        ref += asim.half_to_f32(bh, h16).astype(np.float64)
    ref = ref.reshape(M, N)
    worst, TM, TN = 0.0, c.MI * 32, c.NI * 32
    scale = np.abs(ref).max()
    for wave in range(NW):
        wm, wn = wave // c.WN, wave % c.WN
        for i in range(c.MI):
            for j in range(c.NI):
                blk = wg.acc32(wg.waves[wave], "a", c.ACC(i, j))  # [n within block][m within block]
                for col in range(32):
                    m = P["m0"] + wm * TM + i * 32 + col
                    if m >= M:
                        continue
                    n = P["n0"] + wn * TN + j * 32 + np.arange(32)
                    ok = n < N
                    worst = max(worst, float(np.abs(blk[ok, col] - ref[m, n[ok]]).max()) / scale)
    return worst, wg


if __name__ == "__main__":
    for cfgname, cout in (("256X128", 128), ("256X160", 320)):
        for kw in (dict(B=1, H=6, W=20, Cin=128), dict(B=2, H=9, W=20, Cin=64, tile=(1, 0)), dict(B=1, H=13, W=20, Cin=192, tile=(0, 1) if cout > 160 else (0, 0), land="wait",
                                                                                                     order=(3, 2, 1, 0))):
            err, wg = run_case(cfgname, Cout=cout, **kw)
            print(cfgname, kw, f"max err / max |ref| = {err:.2e}")
            assert err < 2e-6, err
    print("ok")
