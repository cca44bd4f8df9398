#!/usr/bin/env python
"""Generator of the hand-placed main loop of the stride-1 3x3 strip convolution (tools/conv64/conv64_asm.inc) for gfx950.

The kernel form (tools/conv64/conv64.h::conv64_kernel -- an EXPERIMENT of round 6, measured and not shipped: tools/conv64/README.md): the tiling, LDS image, DMA geometry, K order (ky, 64-channel slab, kx, 16-wide k step) and MFMA
operand roles of conv_strip2_kernel -- so every output element is the same sum of the same products in the same order -- on FOUR waves,
one per SIMD, with wave tiles of MI x NI 32 x 32 blocks twice the size of the 8-wave kernel's (256 x 128: 4 x 2; 256 x 160: 2 x 5) and an
instruction stream placed here instead of by the compiler:

  * accumulators in the accumulator file a[0 : 16 MI NI), two fragment register sets in v[0 : 8 (MI + NI)): the fragments of k step
    ks + 1 are read behind the MFMAs of k step ks, one ds_read_b128 per MFMA gap (MI + NI reads feed MI * NI MFMAs: 0.75 / 0.7 KB of LDS
    traffic per MFMA instead of 1);
  * ONE barrier per 64-channel step, placed in front of the step's LAST k step: its MFMAs (operands already in registers) cover the
    first fragment reads of the next step's buffers and the issue of the next DMA pieces;
  * every fragment address is a register (a_row + a_swz added once per kernel row, 3 times per tile) plus an immediate (buffer parity,
    column block): no vector ALU instruction in the loop; the strip loop is unrolled over the two buffer parities.

The C++ side (conv64.h) computes the lane-constant operands exactly as conv_strip2_kernel does, runs the epilogue, and keeps the
split / upsampling-phase / odd-shape cases on the 8-wave kernel.  tools/conv64/csim.py executes the stream on the numpy model of
tools/attn64/sim.py against a direct convolution (tests/test_conv64_sim.py).

`python tools/conv64/cgen.py --write` regenerates the .inc, `--check` compares it with the committed file.
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path
from typing import List

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(HERE.parent / "attn64"))
from gen import Ins, branch, check_hazards, insert_lgkm_waits, label, regs, rng, s_nop  # noqa: E402

OUT = HERE / "conv64_asm.inc"  # the experiment is not part of libdm4d.so (see README.md here)
ROWB = 128   # bytes of an LDS row (64 channels)
BM = 256
NW = 4

# fixed scalar registers of the stream
SG = dict(wrow=40, wp=42, abn=44, wrn=46, cin2=48, nci=49, cs=50, ky=51, gleft=52, adst=53, bdst=54, m0save=55, t0=56, t1=57, abase=58, wt=60,
          wave0=62, csn=63, kyn=64, wrapd=65)
SGPR_CLOBBER = list(range(40, 74))  # s[66:71]: s_memtime stamps of the timing build


class Cfg:
    def __init__(self, MI, NI, WM, WN, h16=False):
        self.MI, self.NI, self.WM, self.WN, self.h16 = MI, NI, WM, WN, h16
        self.BN = NI * 32 * WN
        assert MI * 32 * WM == BM and WM * WN == NW
        self.NA = (BM + 8) // 8
        self.AW = (self.NA + NW - 1) // NW
        self.NB = self.BN // 8
        assert self.NB % NW == 0
        self.BW = self.NB // NW
        self.SR = BM + 16
        self.A_BYTES, self.B_BYTES = self.SR * ROWB, self.BN * ROWB
        self.FS = 4 * (MI + NI)
        self.AA0 = 64
        self.AV0 = self.AA0 + 3 * MI * 4
        self.TMP = self.AV0 + 12
        self.NV = self.TMP + 8
        self.NACC = MI * NI * 16
        self.mfma = "v_mfma_f32_32x32x16_f16" if h16 else "v_mfma_f32_32x32x16_bf16"
        self.name = f"{BM}X{self.BN}_{'F16' if h16 else 'BF16'}"

    def AF(self, s, i):
        return s * self.FS + 4 * i

    def BF(self, s, j):
        return s * self.FS + 4 * self.MI + 4 * j

    def AADDR(self, kx, i, ks):
        return self.AA0 + (kx * self.MI + i) * 4 + ks

    def AVW(self, r):
        return self.AV0 + r

    def ACC(self, i, j):
        return (i * self.NI + j) * 16


# ---- scalar / vector helpers (text + model op) ---------------------------------------------------------------------------------------
def _src(x):
    """operand: int n -> SGPR n;  str -> named input;  ("imm", v)."""
    if isinstance(x, int):
        return f"s{x}", ("s", x)
    if isinstance(x, str):
        return f"%[{x}]", ("in", x)
    return str(x[1]), ("imm", x[1])


def sop2(op, d, a, b):
    ta, ma = _src(a)
    tb, mb = _src(b)
    return Ins(f"{op} s{d}, {ta}, {tb}", "salu", sim=("sop2", op, d, ma, mb))


def smov(d, a):
    ta, ma = _src(a)
    return Ins(f"s_mov_b32 s{d}, {ta}", "salu", sim=("sop2", "s_mov_b32", d, ma, ("imm", 0)))


def smov64(d, a):
    if isinstance(a, int):
        return Ins(f"s_mov_b64 s[{d}:{d + 1}], s[{a}:{a + 1}]", "salu", sim=("smov64", d, ("s", a)))
    return Ins(f"s_mov_b64 s[{d}:{d + 1}], %[{a}]", "salu", sim=("smov64", d, ("in", a)))


def scmp(op, a, b):
    ta, ma = _src(a)
    tb, mb = _src(b)
    return Ins(f"{op} {ta}, {tb}", "salu", sim=("scmp", op, ma, mb))


def set_m0_add(base, imm):
    return Ins(f"s_add_u32 m0, s{base}, {imm}", "salu", sim=("m0add", base, imm))


def dma_v(voff, ptr):
    """voff: VGPR number (working register) or input name."""
    t = f"v{voff}" if isinstance(voff, int) else f"%[{voff}]"
    return Ins(f"global_load_lds_dwordx4 {t}, s[{ptr}:{ptr + 1}]", "dma", reads=[("v", voff)] if isinstance(voff, int) else [],
               sim=("dma2", ("v", voff) if isinstance(voff, int) else ("in", voff), ptr))


def vadd_in(d, a, b):
    return Ins(f"v_add_u32 v{d}, %[{a}], %[{b}]", "valu", writes=[("v", d)], sim=("vadd_in", d, a, b))


def vmov_in(d, a):
    return Ins(f"v_mov_b32 v{d}, %[{a}]", "valu", writes=[("v", d)], sim=("vmov_in", d, a))


def vmov0(d):
    return Ins(f"v_mov_b32 v{d}, 0", "valu", writes=[("v", d)], sim=("vmov_imm", d, 0))


def ds_read(d, addr, off):
    """addr: VGPR number or input name."""
    t = f"v{addr}" if isinstance(addr, int) else f"%[{addr}]"
    return Ins(f"ds_read_b128 {rng('v', d, 4)}, {t} offset:{off}", "ds", reads=[("v", addr)] if isinstance(addr, int) else [],
               writes=regs("v", d, 4), sim=("ds_b128v", d, ("v", addr) if isinstance(addr, int) else ("in", addr), off))


class ConvProgram:
    def __init__(self, cfg: Cfg, timing=False, dma_blocks=1, ablate=()):
        self.c, self.timing, self.dma_blocks, self.ablate = cfg, timing, dma_blocks, set(ablate)
        self.ins: List[Ins] = []
        self.var = cfg  # (the model reads .h16 from here)

    def emit(self, *x):
        self.ins.extend(x)

    # -- pieces ---------------------------------------------------------------------------------------------------------------------
    def mfmas(self, st):
        c = self.c
        out = []
        for i in range(c.MI):
            for j in range(c.NI):
                a = c.ACC(i, j)
                out.append(Ins(f"{c.mfma} {rng('a', a, 16)}, {rng('v', c.BF(st, j), 4)}, {rng('v', c.AF(st, i), 4)}, {rng('a', a, 16)}", "mfma",
                               reads=regs("v", c.BF(st, j), 4) + regs("v", c.AF(st, i), 4) + regs("a", a, 16), writes=regs("a", a, 16),
                               sim=("mfma32", ("a", a), ("v", c.BF(st, j)), ("v", c.AF(st, i)), ("a", a))))
        return out

    def reads(self, st, kx, ks, pa, pb):
        c = self.c
        a = [ds_read(c.AF(st, i), c.AADDR(kx, i, ks), pa * c.A_BYTES) for i in range(c.MI)]
        b = [ds_read(c.BF(st, j), f"brd{ks}", pb * c.B_BYTES + j * 32 * ROWB) for j in range(c.NI)]
        return [a[0]] + b + a[1:]  # in the order the MFMAs (i major, j minor) first need them

    def place(self, anchors, fillers):
        """Fillers spread evenly behind the anchors (MFMAs).  A filler is an instruction or an ATOMIC group of instructions (a DMA piece:
        M0 write, wait state, DMA; a conditional piece with its branch and label): no MFMA is placed inside a group."""
        fillers = list(fillers)
        k = 0
        acc = 0.0
        per = len(fillers) / max(1, len(anchors))
        for a in anchors:
            self.emit(a)
            acc += per
            while k < len(fillers) and acc >= 1.0 - 1e-9:
                f = fillers[k]
                self.emit(*(f if isinstance(f, list) else [f]))
                k += 1
                acc -= 1.0
        for f in fillers[k:]:
            self.emit(*(f if isinstance(f, list) else [f]))

    def b_pieces(self, pbn):
        c, s = self.c, SG
        out = []
        for r in range(c.BW):
            out.append([set_m0_add(s["bdst"], pbn * c.B_BYTES + NW * r * 1024), s_nop(0), dma_v(f"wvoff{r}", s["wp"])])
        return out

    def a_pieces(self, pan, tag):
        c, s = self.c, SG
        out = []
        for r in range(c.AW):
            piece = [set_m0_add(s["adst"], pan * c.A_BYTES + NW * r * 1024), s_nop(0), dma_v(c.AVW(r), s["abn"])]
            if NW * (r + 1) <= c.NA:
                out.append(piece)
            else:  # the last round is partial: waves below NA - NW r (here: wave 0 alone)
                assert c.NA - NW * r == 1
                out.append([scmp("s_cmp_eq_u32", s["wave0"], ("imm", 0)), branch("s_cbranch_scc1", f"ASKIP{tag}"), *piece, label(f"ASKIP{tag}")])
        return out

    def strip_head(self, tag):
        """Once per strip, behind the barrier of its first step: next strip's coordinates and pointers; at a new kernel row the fragment
        addresses of that row, one strip before it the DMA offsets of its strips."""
        c, s = self.c, SG
        e = []
        # -- vector state --
        e += [scmp("s_cmp_lg_u32", s["cs"], ("imm", 0)), branch("s_cbranch_scc1", f"NOA{tag}")]
        for ky in (1, 2):
            e += [scmp("s_cmp_lg_u32", s["ky"], ("imm", ky)), branch("s_cbranch_scc1", f"NOA{ky}{tag}")]
            e += [vadd_in(c.AADDR(kx, i, ks), f"arow{ky}_{kx}_{i}", f"aswz{kx}_{ks}") for kx in range(3) for i in range(c.MI) for ks in range(4)]
            e += [label(f"NOA{ky}{tag}")]
        e += [label(f"NOA{tag}")]
        e += [sop2("s_add_u32", s["t0"], s["cs"], ("imm", 1)), scmp("s_cmp_lg_u32", s["t0"], s["nci"]), branch("s_cbranch_scc1", f"NOV{tag}")]
        for ky in (0, 1):
            e += [scmp("s_cmp_lg_u32", s["ky"], ("imm", ky)), branch("s_cbranch_scc1", f"NOV{ky}{tag}")]
            e += [vmov_in(c.AVW(r), f"avoff{ky + 1}_{r}") for r in range(c.AW)]
            e += [label(f"NOV{ky}{tag}")]
        e += [label(f"NOV{tag}")]
        # -- scalar state: (cs, ky) of the next strip, its weight row and its A base --
        e += [sop2("s_add_u32", s["csn"], s["cs"], ("imm", 1)),
              scmp("s_cmp_eq_u32", s["csn"], s["nci"]),
              sop2("s_cselect_b32", s["csn"], ("imm", 0), s["csn"]),
              sop2("s_cselect_b32", s["t0"], ("imm", 1), ("imm", 0)),
              sop2("s_add_u32", s["kyn"], s["ky"], s["t0"]),
              scmp("s_cmp_eq_u32", s["csn"], ("imm", 0)),
              sop2("s_cselect_b32", s["t0"], s["wrapd"], ("imm", 128)),
              scmp("s_cmp_eq_u32", s["gleft"], ("imm", 1)),
              sop2("s_cselect_b32", s["t0"], ("imm", 0), s["t0"]),      # the last strip re-reads itself (nobody consumes it)
              sop2("s_add_u32", s["wrn"], s["wrow"], s["t0"]),
              sop2("s_addc_u32", s["wrn"] + 1, s["wrow"] + 1, ("imm", 0)),
              sop2("s_lshl_b32", s["t0"], s["csn"], ("imm", 7)),
              sop2("s_add_u32", s["abn"], s["abase"], s["t0"]),
              sop2("s_addc_u32", s["abn"] + 1, s["abase"] + 1, ("imm", 0))]
        return e

    def step(self, pa, kx, tag):
        """Barrier, first fragment reads of the step, the previous step's last k step (deferred), k steps 0..2 of this one."""
        c, s = self.c, SG
        pb = pa ^ (kx & 1)
        self.emit(Ins("s_waitcnt vmcnt(0) lgkmcnt(0)", "wait", sim=("wait", 0, 0)), Ins("s_barrier", "barrier", sim=("barrier",)))
        if kx == 0:
            self.emit(*self.strip_head(tag))
        # DMA of the next step's weight slab (and, in a strip's last step, of the next strip's rows), with their pointer arithmetic
        if kx == 0:
            fill = [sop2("s_add_u32", s["wp"], s["wrow"], s["cin2"]), sop2("s_addc_u32", s["wp"] + 1, s["wrow"] + 1, ("imm", 0))]
        elif kx == 1:
            fill = [sop2("s_add_u32", s["wp"], s["wp"], s["cin2"]), sop2("s_addc_u32", s["wp"] + 1, s["wp"] + 1, ("imm", 0))]
        else:
            fill = [smov64(s["wp"], s["wrn"])]
        fill += self.b_pieces(pb ^ 1)
        if kx == 2:
            fill += self.a_pieces(pa ^ 1, tag)
            fill += [smov64(s["wrow"], s["wrn"]), smov(s["cs"], s["csn"]), smov(s["ky"], s["kyn"])]
        self.emit(*self.reads(0, kx, 0, pa, pb))
        # the DMA issue is spread over the whole step: a quarter behind the deferred MFMAs (k step 3 of the previous step: fragment set 1,
        # read before the barrier), the rest behind the fragment reads of k steps 0..2
        n, nb = len(fill), self.dma_blocks  # the DMA issue goes behind the first `dma_blocks` of the step's four MFMA blocks
        cut = [min(n, (q * n + nb - 1) // nb) for q in range(5)]
        self.place(self.mfmas(1), fill[cut[0]:cut[1]])
        for ks in range(3):
            self.place(self.mfmas(ks & 1), self.reads((ks + 1) & 1, kx, ks + 1, pa, pb) + fill[cut[ks + 1]:cut[ks + 2]])

    def stamp(self, sg):
        return [Ins(f"s_memtime s[{sg}:{sg + 1}]", "nop", sim=("nop", 0)), Ins("s_waitcnt lgkmcnt(0)", "nop", sim=("nop", 0))]

    def build(self):
        c, s = self.c, SG
        e = self.emit
        if self.timing:
            e(*self.stamp(66))
        e(Ins(f"s_mov_b32 s{s['m0save']}, m0", "salu", sim=("save_m0", s["m0save"])))
        for d, a in (("abase", "abase"), ("wt", "wbase")):
            e(smov64(s[d], a))
        e(smov(s["cin2"], "cin2"), smov(s["nci"], "nci"), smov(s["adst"], "adst"), smov(s["bdst"], "bdst"), smov(s["wave0"], "wave0"),
          smov(s["cs"], ("imm", 0)), smov(s["ky"], ("imm", 0)),
          sop2("s_mul_i32", s["gleft"], s["nci"], ("imm", 3)),
          # weight-row step at a kernel-row change: 3 cin rows on, (nci - 1) slabs back = 4 cin + 128 bytes
          sop2("s_lshl_b32", s["wrapd"], s["cin2"], ("imm", 1)), sop2("s_add_u32", s["wrapd"], s["wrapd"], ("imm", 128)),
          smov64(s["wrow"], s["wt"]))
        # accumulators: the bias as the first k step (acc_init of gemm_common.h: MFMA of the bias fragment against e_0)
        t = c.TMP
        e(vmov_in(t + 4, "onew"), vmov0(t + 5), vmov0(t + 6), vmov0(t + 7), vmov0(t + 1), vmov0(t + 2), vmov0(t + 3))
        for j in range(c.NI):
            e(vmov_in(t, f"biasw{j}"), s_nop(1))
            for i in range(c.MI):
                a = c.ACC(i, j)
                e(Ins(f"{c.mfma} {rng('a', a, 16)}, {rng('v', t, 4)}, {rng('v', t + 4, 4)}, 0", "mfma",
                      reads=regs("v", t, 8), writes=regs("a", a, 16), sim=("mfma32", ("a", a), ("v", t), ("v", t + 4), None)))
            e(s_nop(3))  # the fragment register is rewritten for the next column block (MFMA source reads are over by then)
        # kernel row 0: fragment addresses, DMA offsets; fragment set 1 = zeros (the first step's "deferred" MFMAs add nothing)
        e(*[vadd_in(c.AADDR(kx, i, ks), f"arow0_{kx}_{i}", f"aswz{kx}_{ks}") for kx in range(3) for i in range(c.MI) for ks in range(4)])
        e(*[vmov_in(c.AVW(r), f"avoff0_{r}") for r in range(c.AW)])
        e(*[vmov0(c.FS + r) for r in range(c.FS)])
        # strip 0 and its first weight slab -> buffers 0
        e(smov64(s["abn"], s["abase"]), smov64(s["wp"], s["wrow"]))
        e(*[x for g in self.a_pieces(0, "P") + self.b_pieces(0) for x in g])
        if self.timing:
            e(*self.stamp(68))
        e(label("LOOP"))
        for kx in range(3):
            self.step(0, kx, "E")
        e(sop2("s_sub_u32", s["gleft"], s["gleft"], ("imm", 1)), scmp("s_cmp_eq_u32", s["gleft"], ("imm", 0)), branch("s_cbranch_scc1", "EXIT"))
        for kx in range(3):
            self.step(1, kx, "O")
        e(sop2("s_sub_u32", s["gleft"], s["gleft"], ("imm", 1)), scmp("s_cmp_lg_u32", s["gleft"], ("imm", 0)), branch("s_cbranch_scc1", "LOOP"))
        e(label("EXIT"), Ins("s_waitcnt vmcnt(0) lgkmcnt(0)", "wait", sim=("wait", 0, 0)), *self.mfmas(1),
          Ins("s_barrier", "barrier", sim=("barrier",)), Ins(f"s_mov_b32 m0, s{s['m0save']}", "salu", sim=("restore_m0", s["m0save"])))
        if self.timing:
            e(*self.stamp(70), *[Ins(f"s_mov_b32 %[ts{i}], s{66 + i}", "salu", sim=("nop", 0)) for i in range(6)])
        e(s_nop(15))
        if self.ablate:  # timing experiments only (wrong results): kinds of loop instructions left out
            lo = next(i for i, x in enumerate(self.ins) if x.kind == "label" and x.sim[1] == "LOOP")
            hi = next(i for i, x in enumerate(self.ins) if x.kind == "label" and x.sim[1] == "EXIT")
            self.ins = [x for i, x in enumerate(self.ins) if not (lo < i < hi and x.kind in self.ablate)]
        insert_lgkm_waits(self.ins, strict=not self.ablate)
        return self


CONFIGS = {"256X128": (4, 2, 2, 2), "256X160": (2, 5, 4, 1)}


def render(prog) -> str:
    return " \\\n".join(f'  "{x.text}\\n\\t"' for x in prog.ins if x.kind != "touch")


def emit_file(opts=None) -> str:
    opts = dict(opts or {})
    parts = ["// GENERATED by tools/conv64/cgen.py -- do not edit; `python tools/conv64/cgen.py --write` after changing the generator.\n",
             "// The hand-placed main loop of tools/conv64/conv64.h::conv64_kernel (register map and schedule: tools/conv64/cgen.py header).\n",
             "#pragma once\n"]
    for name, (MI, NI, WM, WN) in CONFIGS.items():
        for h16 in (False, True):
            cfg = Cfg(MI, NI, WM, WN, h16)
            prog = ConvProgram(cfg, **opts).build()
            errs = check_hazards(prog.ins)
            if errs:
                raise SystemExit("hazards:\n" + "\n".join(errs[:40]))
            parts.append(f"#define CONV64_ASM_{cfg.name} \\\n" + render(prog) + "\n\n")
        ops = [f'[avoff{ky}_{r}] "v"(avoff[{ky}][{r}])' for ky in range(3) for r in range(cfg.AW)]
        ops += [f'[wvoff{r}] "v"(wvoff[{r}])' for r in range(cfg.BW)]
        ops += [f'[arow{ky}_{kx}_{i}] "v"(arow[{ky}][{kx}][{i}])' for ky in range(3) for kx in range(3) for i in range(cfg.MI)]
        ops += [f'[aswz{kx}_{ks}] "v"(aswz[{kx}][{ks}])' for kx in range(3) for ks in range(4)]
        ops += [f'[brd{ks}] "v"(brd[{ks}])' for ks in range(4)] + [f'[biasw{j}] "v"(biasw[{j}])' for j in range(cfg.NI)] + ['[onew] "v"(onew)']
        ops += [f'[{n}] "s"({n})' for n in ("abase", "wbase", "cin2", "nci", "adst", "bdst", "wave0")]
        parts.append(f"#define CONV64_OPERANDS_{name} " + ", ".join(ops) + "\n")
        clob = ", ".join([f'"v{i}"' for i in range(cfg.NV)] + [f'"a{i}"' for i in range(cfg.NACC)] + [f'"s{i}"' for i in SGPR_CLOBBER] + ['"scc"', '"memory"'])
        parts.append(f"#define CONV64_CLOBBERS_{name} {clob}\n")
    return "".join(parts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    text = emit_file()
    if a.write:
        OUT.write_text(text)
        print("wrote", OUT, len(text), "bytes")
    if a.check:
        if not OUT.exists() or OUT.read_text() != text:
            raise SystemExit(f"{OUT} is stale: run python tools/conv64/cgen.py --write")
        print("up to date")


if __name__ == "__main__":
    main()
