#!/usr/bin/env python
"""Generator of the hand-placed attention main loop (diffuman4d_amd/csrc/attn64_asm.inc) for gfx950.

The kernel form (attention.hip::attn64_kernel): a workgroup = 4 waves = 256 query rows, ONE wave per SIMD, every wave owns 64 query
rows (two 32-row blocks qb = 0, 1) and the whole 512-register file; head dim 64; K/V tiles of 64 keys arrive by LDS-DMA into two-stage
rings.  What this file decides is the ORDER of the instruction stream and the register file, neither of which hipcc can be made to
produce (guide: a compiler-placed stream of the same multiset costs +10 %):

  per 64-key tile and wave: 32 v_mfma_f32_32x32x16 (QK^T one tile ahead: 16, PV: 16) + 8 v_mfma_f32_16x16x32 (row sums on the matrix
  pipe: A = a constant lane pattern of ones, B = the packed probabilities; 16 cycles each instead of 64 v_add_f32), 64 v_exp_f32 (in
  place), 32 v_cvt_pk, 8 ds_read_b128 (K fragments, each feeding both row blocks), 16 ds_read_b64_tr_b16 (V), 4 DMA pieces, ~25 SALU.

  The stream is eight ROUNDS of five MFMAs (PV db0, QK qb0, PV db1, QK qb1, row sum), each round carrying the soft-max of the
  probability group two rounds ahead and the fragment reads two rounds ahead as fillers in the MFMA gaps; the step barrier sits in
  front of the last round's MFMAs, so that the first fragment reads of the new stages are covered by register-only MFMAs.

Registers (asm-owned, listed as clobbers of the statement):
  v[0:63], v[64:127]   S sets 0 / 1: block (kb, qb) = 16 registers, S^T[key][q] (swapped QK^T: a lane holds 16 scores of ITS row)
  v[128:159]           -m per row block (C operand of the first QK^T MFMA of a block)
  v[160:191]           P: 8 groups (kb, jj, qb) x 4 registers of packed bf16 / fp16
  v[192:199]           scratch
  a[0:63]              O^T accumulators (db, qb);  a[64:71] row sums (qb);  a[72:103] Q fragments;  a[104:135] K fragments;
  a[136:167]           V fragments;  a[168:171] the ones pattern
  s[40:61]             loop state

`python tools/attn64/gen.py --write` regenerates the .inc; `--check` verifies that the committed file is what the generator emits;
tools/attn64/sim.py executes the stream on a numpy model of the wave (tests/test_attn64_sim.py).
"""
from __future__ import annotations

import argparse
import sys
from dataclasses import dataclass, field
from pathlib import Path
from typing import Callable, List, Optional, Sequence, Tuple

ROOT = Path(__file__).resolve().parent.parent.parent
OUT = ROOT / "diffuman4d_amd" / "csrc" / "attn64_asm.inc"

Reg = Tuple[str, int]  # ("v", n) | ("a", n) | ("s", n)

# ---- register map -------------------------------------------------------------------------------------------------
S_BASE, NEGM_BASE, P_BASE, TMP_BASE = 0, 128, 160, 192
O_BASE, L_BASE, Q_BASE, KF_BASE, VF_BASE, ONES_BASE = 0, 64, 72, 104, 136, 168
LV_BASE = 200            # row-sum accumulators of the VALU row-sum forms: 4 per row block
N_VGPR_CLOBBER, N_AGPR_CLOBBER = 208, 172
SG = dict(kptr=40, vptr=42, kstride=44, vstride=45, rem=46, remk=47, remv=48, inc=49, mk=50, mv=54, t0=58, t1=59)
SGPR_CLOBBER = list(range(40, 68))  # s[62:67]: s_memtime stamps of the timing build (tools/attn64/ab.py)
KV_STAGE = 8192          # bytes of one [64 keys][64 d] stage
HALF_TILE = 4096         # 32 key rows


def S(set_, kb, qb):
    return S_BASE + set_ * 64 + (kb * 2 + qb) * 16


def NEGM(qb):
    return NEGM_BASE + 16 * qb


def P(p):
    return P_BASE + 4 * p


def O(db, qb):
    return O_BASE + (db * 2 + qb) * 16


def L(qb):
    return L_BASE + 4 * qb


def Q(qb, j):
    return Q_BASE + (qb * 4 + j) * 4


def KF(f):
    return KF_BASE + 4 * f


def VF(g, db):
    return VF_BASE + (g * 2 + db) * 4


def rng(file, base, n):
    return f"{file}[{base}:{base + n - 1}]" if n > 1 else f"{file}{base}"


def regs(file, base, n):
    return [(file, base + i) for i in range(n)]


@dataclass
class Ins:
    text: str
    kind: str                      # mfma | mfma16 | valu | trans | ds | dma | gload | salu | wait | barrier | nop | label | branch
    reads: List[Reg] = field(default_factory=list)
    writes: List[Reg] = field(default_factory=list)
    sim: Optional[tuple] = None    # (op, args...) for tools/attn64/sim.py
    cost: float = 1.0              # issue-slot weight used by the placer

    def size(self):  # wait states this instruction provides to hazards that follow it
        if self.kind == "nop":
            return self.sim[1] + 1
        return 0 if self.kind in ("label", "touch") else 1


class Variant:
    def __init__(self, h16: bool):
        self.h16 = h16
        self.mfma = "v_mfma_f32_32x32x16_f16" if h16 else "v_mfma_f32_32x32x16_bf16"
        self.mfma16 = "v_mfma_f32_16x16x32_f16" if h16 else "v_mfma_f32_16x16x32_bf16"
        self.cvt = "v_cvt_pk_f16_f32" if h16 else "v_cvt_pk_bf16_f32"
        self.name = "F16" if h16 else "BF16"


# ---- instruction constructors ---------------------------------------------------------------------------------------
def mfma(var, d_file, d, a_file, a, b_file, b, c_file, c):
    """D[16 regs] = A[4] x B[4] + C[16];  c = None -> the inline constant 0."""
    ctext = "0" if c is None else rng(c_file, c, 16)
    return Ins(f"{var.mfma} {rng(d_file, d, 16)}, {rng(a_file, a, 4)}, {rng(b_file, b, 4)}, {ctext}", "mfma",
               reads=regs(a_file, a, 4) + regs(b_file, b, 4) + ([] if c is None else regs(c_file, c, 16)), writes=regs(d_file, d, 16),
               sim=("mfma32", (d_file, d), (a_file, a), (b_file, b), None if c is None else (c_file, c)))


def mfma_rowsum(var, qb, p):
    return Ins(f"{var.mfma16} {rng('a', L(qb), 4)}, {rng('a', ONES_BASE, 4)}, {rng('v', P(p), 4)}, {rng('a', L(qb), 4)}", "mfma16",
               reads=regs("a", ONES_BASE, 4) + regs("v", P(p), 4) + regs("a", L(qb), 4), writes=regs("a", L(qb), 4),
               sim=("mfma16", ("a", L(qb)), ("a", ONES_BASE), ("v", P(p)), ("a", L(qb))))


def v_exp(r):
    return Ins(f"v_exp_f32 v{r}, v{r}", "trans", reads=[("v", r)], writes=[("v", r)], sim=("exp", r), cost=5.0 / 3.0)


def v_cvt(var, d, lo, hi):
    return Ins(f"{var.cvt} v{d}, v{lo}, v{hi}", "valu", reads=[("v", lo), ("v", hi)], writes=[("v", d)], sim=("cvt", d, lo, hi))


def ds_read_k(f, addr_name, offset):
    return Ins(f"ds_read_b128 {rng('a', KF(f), 4)}, %[{addr_name}] offset:{offset}", "ds", writes=regs("a", KF(f), 4),
               sim=("ds_b128", ("a", KF(f)), addr_name, offset))


def ds_read_v(g, db, half, addr_name, offset):
    d = VF(g, db) + 2 * half
    return Ins(f"ds_read_b64_tr_b16 {rng('a', d, 2)}, %[{addr_name}] offset:{offset}", "ds", writes=regs("a", d, 2),
               sim=("ds_tr", ("a", d), addr_name, offset))


def salu(text, sim, reads=(), writes=()):
    return Ins(text, "salu", reads=[("s", r) for r in reads], writes=[("s", w) for w in writes], sim=sim)


def s_nop(n):
    return Ins(f"s_nop {n}", "nop", sim=("nop", n))


def label(name):
    return Ins(f"{name}_%=:", "label", sim=("label", name))


def branch(op, name):
    return Ins(f"{op} {name}_%=", "branch", sim=("branch", op, name))


def dma(voff_name, ptr_sgpr):
    return Ins(f"global_load_lds_dwordx4 %[{voff_name}], s[{ptr_sgpr}:{ptr_sgpr + 1}]", "dma", reads=[("s", ptr_sgpr), ("s", ptr_sgpr + 1)],
               sim=("dma", voff_name, ptr_sgpr))


def set_m0(sgpr):
    return salu(f"s_mov_b32 m0, s{sgpr}", ("s_mov_m0", sgpr), reads=[sgpr])


# ---- the stream of one step ------------------------------------------------------------------------------------------
class Program:
    def __init__(self, var: Variant, gap_big=5.0, gap_small=2.2, defer_valu=False, bare_head=0, timing=False, dma_round=None, order="pqpqL", ablate=(), rowsum="mfma", merge_waits=True):
        self.var = var
        self.ins: List[Ins] = []
        self.gap_big, self.gap_small = gap_big, gap_small
        self.defer_valu, self.bare_head, self.timing, self.dma_round, self.order = defer_valu, bare_head, timing, dma_round, order
        self.rowsum, self.merge_waits = rowsum, merge_waits  # rowsum: mfma | pk (v_pk_add_f32) | add (v_add_f32)
        if rowsum != "mfma":
            self.order = self.order.replace("L", "")
        self.ablate = set(ablate)  # timing experiments only (wrong results): kinds of loop instructions left out of the stream

    def emit(self, *ins):
        self.ins.extend(ins)

    # -- pieces of a step ------------------------------------------------------------------------------------------
    def softmax_fillers(self, set_, p):
        """exp2 in place + packing of probability group p = (g = kb * 2 + jj, qb) read from S set `set_`."""
        g, qb = p >> 1, p & 1
        kb, jj = g >> 1, g & 1
        s0 = S(set_, kb, qb) + 8 * jj
        exps = [v_exp(s0 + i) for i in range(8)]
        cvts = [v_cvt(self.var, P(p) + i, s0 + 2 * i, s0 + 2 * i + 1) for i in range(4)]
        # order: exps first; every cvt at least two instructions behind the exps it reads (trans -> non-trans hazard)
        return exps[:4] + exps[4:6] + [cvts[0]] + exps[6:8] + [cvts[1], cvts[2], cvts[3]]

    def k_frag_read(self, f, stage):
        kb, j = f >> 2, f & 3
        return ds_read_k(f, f"kfa{j}", stage * KV_STAGE + kb * HALF_TILE)

    def v_frag_reads(self, g, stage):
        kb, jj = g >> 1, g & 1
        out = []
        for db in range(2):
            off = stage * KV_STAGE + (kb * 32 + jj * 16) * 128
            out.append(ds_read_v(g, db, 0, f"vfa{db}", off))
            out.append(ds_read_v(g, db, 1, f"vfa{db}", off + 8 * 128))
        return out

    def round_mfmas(self, par, r):
        """The five MFMAs of round r of a step of parity `par` (cur = S set par, nxt = 1 - par)."""
        var = self.var
        cur, nxt = par, 1 - par
        p, g, qb = r, r >> 1, r & 1
        kb2, j = r >> 2, r & 3
        out = [mfma(var, "a", O(0, qb), "a", VF(g, 0), "v", P(p), "a", O(0, qb))]
        for q in range(2):
            c = ("v", NEGM(q)) if j == 0 else ("v", S(nxt, kb2, q))
            m = mfma(var, "v", S(nxt, kb2, q), "a", KF(r), "a", Q(q, j), c[0], c[1])
            out.append(m)
            if q == 0:
                out.append(mfma(var, "a", O(1, qb), "a", VF(g, 1), "v", P(p), "a", O(1, qb)))
        out.append(mfma_rowsum(var, qb, p))  # PV db0, QK qb0, PV db1, QK qb1, rowsum
        pick = {"p": [out[0], out[2]], "q": [out[1], out[3]], "L": [out[4]]}
        res = [pick[c].pop(0) for c in self.order]
        if self.merge_waits:  # one counted wait in front of the round instead of one per fragment
            res.insert(0, Ins("", "touch", reads=sorted({r for m in res for r in m.reads if r[0] == "a" and r[1] >= KF_BASE and r[1] < ONES_BASE})))
        return res

    def rowsum_fillers(self, set_, p):
        """VALU row sums (rowsum = pk | add): the exponentials of probability group p, still in place in S set `set_`, into the four
        accumulators of its row block.  Issued in the round that CONSUMES the group, so a tile past the end is never summed."""
        if self.rowsum == "mfma":
            return []
        g, qb = p >> 1, p & 1
        kb, jj = g >> 1, g & 1
        s0 = S(set_, kb, qb) + 8 * jj
        lv = LV_BASE + 4 * qb
        if self.rowsum == "pk":
            return [Ins(f"v_pk_add_f32 v[{lv + 2 * (i & 1)}:{lv + 2 * (i & 1) + 1}], v[{lv + 2 * (i & 1)}:{lv + 2 * (i & 1) + 1}], v[{s0 + 2 * i}:{s0 + 2 * i + 1}]", "valu",
                        reads=regs("v", lv + 2 * (i & 1), 2) + regs("v", s0 + 2 * i, 2), writes=regs("v", lv + 2 * (i & 1), 2),
                        sim=("pkadd", lv + 2 * (i & 1), s0 + 2 * i)) for i in range(4)]
        return [Ins(f"v_add_f32 v{lv + (i & 3)}, v{lv + (i & 3)}, v{s0 + i}", "valu", reads=[("v", lv + (i & 3)), ("v", s0 + i)], writes=[("v", lv + (i & 3))],
                    sim=("add", lv + (i & 3), lv + (i & 3), s0 + i)) for i in range(8)]

    def place(self, anchors: Sequence[Ins], fillers: Sequence[Ins], lead: Sequence[Ins] = (), bare=0):
        """MFMA anchors in order, fillers spread behind them by issue-slot budget (big gap / small gap); `lead` goes first.
        `bare` = number of leading anchors that get no filler (MFMA-only head after a barrier)."""
        self.emit(*lead)
        anchors = list(anchors)
        while anchors and anchors[0].kind == "touch":
            self.emit(anchors.pop(0))
        fillers = list(fillers)
        total = sum(f.cost for f in fillers)
        caps = [(self.gap_small if a.kind == "mfma16" else self.gap_big) if i >= bare else 0.0 for i, a in enumerate(anchors)]
        scale = max(1.0, total / max(sum(caps), 1e-9))  # over budget: stretch every gap by the same factor
        acc = 0.0
        k = 0
        for a, cap in zip(anchors, caps):
            self.emit(a)
            acc += cap * scale
            while k < len(fillers) and acc - fillers[k].cost >= -1e-6:
                acc -= fillers[k].cost
                self.emit(fillers[k])
                k += 1
        self.emit(*fillers[k:])

    def dma_issue(self, par):
        """Head of a step of parity par (step t, t % 2 == par): K(t+2) -> K stage par, V(t+1) -> V stage 1 - par; pointers advance
        by one tile unless they already sit on the last tile (remk / remv == 0)."""
        s = SG
        mk = lambda st, pc: s["mk"] + st * 2 + pc  # noqa: E731
        mv = lambda st, pc: s["mv"] + st * 2 + pc  # noqa: E731
        kp, vp = s["kptr"], s["vptr"]
        return [
            set_m0(mk(par, 0)),
            salu(f"s_cmp_lg_u32 s{s['remk']}, 0", ("s_cmp_lg", s["remk"], 0), reads=[s["remk"]]),
            dma("dk0", kp),
            set_m0(mk(par, 1)),
            salu(f"s_cselect_b32 s{s['inc']}, s{s['kstride']}, 0", ("s_cselect", s["inc"], s["kstride"], 0), reads=[s["kstride"]], writes=[s["inc"]]),
            dma("dk1", kp),
            salu(f"s_subb_u32 s{s['remk']}, s{s['remk']}, 0", ("s_subb", s["remk"]), reads=[s["remk"]], writes=[s["remk"]]),
            salu(f"s_add_u32 s{kp}, s{kp}, s{s['inc']}", ("s_add", kp, kp, s["inc"]), reads=[kp, s["inc"]], writes=[kp]),
            salu(f"s_addc_u32 s{kp + 1}, s{kp + 1}, 0", ("s_addc", kp + 1), reads=[kp + 1], writes=[kp + 1]),
            set_m0(mv(1 - par, 0)),
            salu(f"s_cmp_lg_u32 s{s['remv']}, 0", ("s_cmp_lg", s["remv"], 0), reads=[s["remv"]]),
            dma("dv0", vp),
            set_m0(mv(1 - par, 1)),
            salu(f"s_cselect_b32 s{s['inc']}, s{s['vstride']}, 0", ("s_cselect", s["inc"], s["vstride"], 0), reads=[s["vstride"]], writes=[s["inc"]]),
            dma("dv1", vp),
            salu(f"s_subb_u32 s{s['remv']}, s{s['remv']}, 0", ("s_subb", s["remv"]), reads=[s["remv"]], writes=[s["remv"]]),
            salu(f"s_add_u32 s{vp}, s{vp}, s{s['inc']}", ("s_add", vp, vp, s["inc"]), reads=[vp, s["inc"]], writes=[vp]),
            salu(f"s_addc_u32 s{vp + 1}, s{vp + 1}, 0", ("s_addc", vp + 1), reads=[vp + 1], writes=[vp + 1]),
        ]

    def head_reads(self, par):
        """First fragment reads of a step of parity par: K(t+1) lives in K stage 1 - par, V(t) in V stage par."""
        v = self.v_frag_reads(0, par)
        return [self.k_frag_read(0, 1 - par), v[0], v[1], v[2], v[3], self.k_frag_read(1, 1 - par)]

    def step(self, par, entry_label=None):
        """HEAD + DEFERRED round 7 of the previous step + rounds 0..6 of a step of parity par."""
        old = 1 - par
        self.emit(Ins("s_waitcnt vmcnt(0) lgkmcnt(0)", "wait", sim=("wait", 0, 0)), Ins("s_barrier", "barrier", sim=("barrier",)))
        lead = self.head_reads(par)
        defer_fill = ([] if self.dma_round is not None else self.dma_issue(par)) + (self.softmax_fillers(par, 1) if self.defer_valu else []) + \
            self.rowsum_fillers(old, 7)
        self.place(self.round_mfmas(old, 7), defer_fill, lead=lead, bare=self.bare_head)
        if entry_label:
            self.emit(label(entry_label))
        self.rounds_0_6(par, extra0=[] if self.defer_valu else self.softmax_fillers(par, 1))

    def rounds_0_6(self, par, extra0=()):
        cur, nxt = par, 1 - par
        for r in range(7):
            fill = list(extra0) if r == 0 else []
            if r < 6:
                lds = [self.k_frag_read(r + 2, 1 - par)]
                if r % 2 == 0:
                    lds += self.v_frag_reads((r + 2) // 2, par)
                fill += lds + self.softmax_fillers(cur, r + 2)
            else:
                fill += self.softmax_fillers(nxt, 0)
            fill += self.rowsum_fillers(cur, r)
            if self.dma_round == r:
                fill = self.dma_issue(par) + fill
            self.place(self.round_mfmas(par, r), fill)

    # -- whole program -----------------------------------------------------------------------------------------------
    def rowsum_handover(self):
        """VALU row sums: the four accumulators of a row block -> one value per lane (its half of the row's keys) in a[64] / a[68]."""
        if self.rowsum == "mfma":
            return []
        out = []
        for qb in range(2):
            lv = LV_BASE + 4 * qb
            add = lambda d, a, b: Ins(f"v_add_f32 v{d}, v{a}, v{b}", "valu", reads=[("v", a), ("v", b)], writes=[("v", d)], sim=("add", d, a, b))  # noqa: E731
            out += [add(lv, lv, lv + 1), add(lv + 2, lv + 2, lv + 3), s_nop(0), add(lv, lv, lv + 2), s_nop(1),
                    Ins(f"v_accvgpr_write_b32 a{L(qb)}, v{lv}", "valu", reads=[("v", lv)], writes=[("a", L(qb))], sim=("acc_from_v", L(qb), lv))]
        return out

    @staticmethod
    def stamp(sg):
        """s_memtime into s[sg:sg+1] (an SMEM load: lgkmcnt(0) behind it; the stream has no LDS read in flight where stamps sit)."""
        return [Ins(f"s_memtime s[{sg}:{sg + 1}]", "nop", sim=("nop", 0)), Ins("s_waitcnt lgkmcnt(0)", "nop", sim=("nop", 0))]

    def prologue(self):
        var, s = self.var, SG
        e = self.emit
        if self.timing:
            e(*self.stamp(62))
        # loop state into the fixed SGPRs (M0 belongs to the compiler: saved here, restored behind the loop)
        e(salu(f"s_mov_b32 s{s['t0']}, m0", ("s_save_m0", s["t0"]), writes=[s["t0"]]),
          salu(f"s_mov_b64 s[{s['kptr']}:{s['kptr'] + 1}], %[kbase]", ("s_mov64_in", s["kptr"], "kbase"), writes=[s["kptr"], s["kptr"] + 1]),
          salu(f"s_mov_b64 s[{s['vptr']}:{s['vptr'] + 1}], %[vbase]", ("s_mov64_in", s["vptr"], "vbase"), writes=[s["vptr"], s["vptr"] + 1]),
          salu(f"s_mov_b32 s{s['kstride']}, %[kstride]", ("s_mov_in", s["kstride"], "kstride"), writes=[s["kstride"]]),
          salu(f"s_mov_b32 s{s['vstride']}, %[vstride]", ("s_mov_in", s["vstride"], "vstride"), writes=[s["vstride"]]),
          salu(f"s_mov_b32 s{s['rem']}, %[nt]", ("s_mov_in", s["rem"], "nt"), writes=[s["rem"]]),
          salu(f"s_sub_u32 s{s['remk']}, %[nt], 3", ("s_sub_in", s["remk"], "nt", 3), writes=[s["remk"]]),
          salu(f"s_sub_u32 s{s['remv']}, %[nt], 2", ("s_sub_in", s["remv"], "nt", 2), writes=[s["remv"]]))
        for st in range(2):
            for pc in range(2):
                off = st * KV_STAGE + pc * HALF_TILE
                e(salu(f"s_add_u32 s{s['mk'] + st * 2 + pc}, %[m0k], {off}", ("s_add_in", s["mk"] + st * 2 + pc, "m0k", off), writes=[s["mk"] + st * 2 + pc]),
                  salu(f"s_add_u32 s{s['mv'] + st * 2 + pc}, %[m0v], {off}", ("s_add_in", s["mv"] + st * 2 + pc, "m0v", off), writes=[s["mv"] + st * 2 + pc]))
        # Q fragments straight into the accumulator file
        for qb in range(2):
            for j in range(4):
                e(Ins(f"global_load_dwordx4 {rng('a', Q(qb, j), 4)}, %[qa{qb}], off offset:{32 * j}", "gload", writes=regs("a", Q(qb, j), 4),
                      sim=("gload", ("a", Q(qb, j)), f"qa{qb}", 32 * j)))
        # K(0) -> K stage 0, V(0) -> V stage 0, K(1) -> K stage 1
        kp, vp = s["kptr"], s["vptr"]
        adv = lambda p, st: [salu(f"s_add_u32 s{p}, s{p}, s{st}", ("s_add", p, p, st), reads=[p, st], writes=[p]),  # noqa: E731
                             salu(f"s_addc_u32 s{p + 1}, s{p + 1}, 0", ("s_addc", p + 1), reads=[p + 1], writes=[p + 1])]
        e(set_m0(s["mk"] + 0), s_nop(0), dma("dk0", kp), set_m0(s["mk"] + 1), s_nop(0), dma("dk1", kp), *adv(kp, s["kstride"]),
          set_m0(s["mv"] + 0), s_nop(0), dma("dv0", vp), set_m0(s["mv"] + 1), s_nop(0), dma("dv1", vp), *adv(vp, s["vstride"]),
          set_m0(s["mk"] + 2), s_nop(0), dma("dk0", kp), set_m0(s["mk"] + 3), s_nop(0), dma("dk1", kp), *adv(kp, s["kstride"]))
        # accumulators and the ones pattern while the loads fly
        for r in range(72):
            e(Ins(f"v_accvgpr_write_b32 a{O_BASE + r}, 0", "valu", writes=[("a", O_BASE + r)], sim=("acc_zero", O_BASE + r)))
        if self.rowsum != "mfma":
            for r in range(8):
                e(Ins(f"v_mov_b32 v{LV_BASE + r}, 0", "valu", writes=[("v", LV_BASE + r)], sim=("movc", LV_BASE + r, 0.0)))
        for r in range(4):
            e(Ins(f"v_accvgpr_write_b32 a{ONES_BASE + r}, %[ones]", "valu", writes=[("a", ONES_BASE + r)], sim=("acc_in", ONES_BASE + r, "ones")))
        e(Ins("s_waitcnt vmcnt(0)", "wait", sim=("wait", 0, None)), Ins("s_barrier", "barrier", sim=("barrier",)))
        # S(0) = K(0) Q^T into set 0 (C = 0)
        for f in range(8):
            e(self.k_frag_read(f, 0))
        for f in range(8):
            kb, j = f >> 2, f & 3
            e(Ins(f"s_waitcnt lgkmcnt({7 - f})", "wait", sim=("wait", None, 7 - f)))
            for qb in range(2):
                e(mfma(var, "v", S(0, kb, qb), "a", KF(f), "a", Q(qb, j), *((None, None) if j == 0 else ("v", S(0, kb, qb)))))
        e(s_nop(15))
        # row maxima of the first tile -> -m (fp16: -(m + H16_OFF)), S(0) -= m
        t = TMP_BASE
        for qb in range(2):
            vals = [S(0, kb, qb) + r for kb in range(2) for r in range(16)]
            m, m2 = t + 2 * qb, t + 2 * qb + 1
            e(Ins(f"v_max3_f32 v{m}, v{vals[0]}, v{vals[1]}, v{vals[2]}", "valu", reads=[("v", x) for x in vals[:3]], writes=[("v", m)],
                  sim=("max3", m, vals[0], vals[1], vals[2])))
            i = 3
            while i + 1 < len(vals):
                e(Ins(f"v_max3_f32 v{m}, v{m}, v{vals[i]}, v{vals[i + 1]}", "valu", reads=[("v", m), ("v", vals[i]), ("v", vals[i + 1])],
                      writes=[("v", m)], sim=("max3", m, m, vals[i], vals[i + 1])))
                i += 2
            if i < len(vals):
                e(Ins(f"v_max_f32 v{m}, v{m}, v{vals[i]}", "valu", reads=[("v", m), ("v", vals[i])], writes=[("v", m)], sim=("max", m, m, vals[i])))
            e(Ins(f"v_mov_b32 v{m2}, v{m}", "valu", reads=[("v", m)], writes=[("v", m2)], sim=("mov", m2, m)), s_nop(1),
              Ins(f"v_permlane32_swap_b32 v{m}, v{m2}", "valu", reads=[("v", m), ("v", m2)], writes=[("v", m), ("v", m2)], sim=("swap32", m, m2)),
              s_nop(1),
              Ins(f"v_max_f32 v{m}, v{m}, v{m2}", "valu", reads=[("v", m), ("v", m2)], writes=[("v", m)], sim=("max", m, m, m2)))
            if var.h16:
                e(Ins(f"v_add_f32 v{m}, 0x41000000, v{m}", "valu", reads=[("v", m)], writes=[("v", m)], sim=("addc", m, m, 8.0)))
            for r in range(16):
                e(Ins(f"v_sub_f32 v{NEGM(qb) + r}, 0, v{m}", "valu", reads=[("v", m)], writes=[("v", NEGM(qb) + r)], sim=("neg", NEGM(qb) + r, m)))
            for x in vals:
                e(Ins(f"v_sub_f32 v{x}, v{x}, v{m}", "valu", reads=[("v", x), ("v", m)], writes=[("v", x)], sim=("sub", x, x, m)))
        # everybody has read K stage 0: the first head may overwrite it
        e(Ins("s_waitcnt lgkmcnt(0)", "wait", sim=("wait", None, 0)), Ins("s_barrier", "barrier", sim=("barrier",)))
        e(*self.head_reads(0))
        if self.dma_round is None:  # else the first step issues its own
            e(*self.dma_issue(0))
        e(*self.softmax_fillers(0, 0), *(self.softmax_fillers(0, 1) if self.defer_valu else []), s_nop(1))

    def build(self):
        s = SG
        self.prologue()
        if self.timing:
            self.emit(*self.stamp(64))
        self.emit(branch("s_branch", "ENTRY0"))
        self.emit(label("LOOP"))
        self.step(0, entry_label="ENTRY0")
        self.emit(salu(f"s_sub_u32 s{s['rem']}, s{s['rem']}, 1", ("s_subi", s["rem"], 1), reads=[s["rem"]], writes=[s["rem"]]),
                  salu(f"s_cmp_eq_u32 s{s['rem']}, 0", ("s_cmp_eq", s["rem"], 0), reads=[s["rem"]]),
                  branch("s_cbranch_scc1", "EXIT0"))
        self.step(1)
        self.emit(salu(f"s_sub_u32 s{s['rem']}, s{s['rem']}, 1", ("s_subi", s["rem"], 1), reads=[s["rem"]], writes=[s["rem"]]),
                  salu(f"s_cmp_lg_u32 s{s['rem']}, 0", ("s_cmp_lg", s["rem"], 0), reads=[s["rem"]]),
                  branch("s_cbranch_scc1", "LOOP"))
        # the last step's deferred round (old parity 1 after falling out of step(1), old parity 0 at EXIT0)
        self.emit(Ins("s_waitcnt lgkmcnt(0)", "wait", sim=("wait", None, 0)), *self.round_mfmas(1, 7), *self.rowsum_fillers(1, 7), branch("s_branch", "DONE"))
        self.emit(label("EXIT0"), Ins("s_waitcnt lgkmcnt(0)", "wait", sim=("wait", None, 0)), *self.round_mfmas(0, 7), *self.rowsum_fillers(0, 7))
        self.emit(label("DONE"), Ins("s_waitcnt vmcnt(0) lgkmcnt(0)", "wait", sim=("wait", 0, 0)), Ins("s_barrier", "barrier", sim=("barrier",)),
                  salu(f"s_mov_b32 m0, s{s['t0']}", ("s_mov_m0", s["t0"]), reads=[s["t0"]]), *self.rowsum_handover(), *(self.stamp(66) + [Ins(f"s_mov_b32 %[ts{i}], s{62 + i}", "salu", sim=("nop", 0)) for i in range(6)] if self.timing else []), s_nop(15))
        if self.ablate:
            lo = next(i for i, x in enumerate(self.ins) if x.kind == "label" and x.sim[1] == "LOOP")
            hi = next(i for i, x in enumerate(self.ins) if x.kind == "label" and x.sim[1] == "EXIT0")
            self.ins = [x for i, x in enumerate(self.ins) if not (lo < i < hi and x.kind in self.ablate)]
        insert_lgkm_waits(self.ins, strict=not self.ablate)
        return self


# ---- counted LDS waits -----------------------------------------------------------------------------------------------------
def insert_lgkm_waits(ins: List[Ins], strict=True):
    """ds_read results return in order: in front of the first consumer of a fragment put s_waitcnt lgkmcnt(N), N = number of LDS
    reads issued behind the youngest one the consumer needs.  Linear scan; at a label the two ways in must agree on the reads in
    flight (or the label is followed by a full wait): the queue recorded at the branch is compared with the fall-through's."""
    out: List[Ins] = []
    queue: List[List[Reg]] = []  # outstanding reads, oldest first
    at_branch = {}
    dead = False  # behind an unconditional branch
    for k, x in enumerate(ins):
        if x.kind == "branch":
            at_branch.setdefault(x.sim[2], [list(w) for w in queue])
            dead = x.sim[1] == "s_branch"
        elif x.kind == "label":
            nxt = ins[k + 1]
            full = nxt.kind == "wait" and nxt.sim[2] == 0
            snap = at_branch.get(x.sim[1])
            if dead:
                queue = [list(w) for w in (snap or [])]
            elif snap is not None and not full and strict:
                assert snap == queue, f"reads in flight differ at {x.text}: {snap} vs {queue}"
            dead = False
        elif x.kind == "wait" and x.sim[2] is not None:
            del queue[: max(0, len(queue) - x.sim[2])]
        if x.kind == "ds":
            assert len(queue) < 15, "lgkmcnt is a 4-bit counter"
            queue.append(list(x.writes))
        elif x.kind not in ("label", "branch", "wait"):
            need = -1
            touched = set(x.reads) | set(x.writes)
            for i, w in enumerate(queue):
                if touched & set(w):
                    need = i
            if need >= 0:
                n = len(queue) - 1 - need
                out.append(Ins(f"s_waitcnt lgkmcnt({n})", "wait", sim=("wait", None, n)))
                del queue[: need + 1]
        out.append(x)
    ins[:] = out


# ---- static hazard check ---------------------------------------------------------------------------------------------------
def check_hazards(ins: List[Ins]):
    """Software wait states of gfx950 that matter here (calibrated against what hipcc pads, tools/attn64/README.md):
    VALU write -> MFMA read 2; trans write -> non-trans VALU read 1; MFMA (8 pass) write -> VALU read / write 12; MFMA (4 pass) -> 8;
    VALU write -> v_permlane 2; s_mov m0 -> LDS-DMA 1.  Linear scan (branches fall through; targets are preceded by waits)."""
    errs = []
    last_valu_w, last_trans_w, last_mfma_w, last_m0 = {}, {}, {}, None
    pos = 0
    for x in ins:
        if x.kind in ("label", "touch"):
            continue
        if x.kind == "branch" and x.sim[1] == "s_branch":  # what follows is reached from elsewhere (behind its own waits)
            last_valu_w, last_trans_w, last_mfma_w, last_m0 = {}, {}, {}, None
        if x.kind in ("mfma", "mfma16"):
            for r in x.reads:
                if r in last_valu_w and pos - last_valu_w[r] - 1 < 2:
                    errs.append(f"VALU->MFMA {r} before {x.text}")
        if x.kind in ("valu", "trans"):
            for r in x.reads + x.writes:
                if r in last_mfma_w:
                    p0, need = last_mfma_w[r]
                    if pos - p0 - 1 < need:
                        errs.append(f"MFMA->VALU {r} before {x.text} ({pos - p0 - 1} < {need})")
            if x.kind == "valu":
                for r in x.reads:
                    if r in last_trans_w and pos - last_trans_w[r] - 1 < 1:
                        errs.append(f"trans->VALU {r} before {x.text}")
            if "permlane" in x.text:
                for r in x.reads:
                    if r in last_valu_w and pos - last_valu_w[r] - 1 < 2:
                        errs.append(f"VALU->permlane {r}")
        if x.kind == "dma" and last_m0 is not None and pos - last_m0 - 1 < 1:
            errs.append(f"m0->DMA before {x.text}")
        # record
        n = x.size()
        if x.kind in ("valu", "trans"):
            for r in x.writes:
                (last_trans_w if x.kind == "trans" else last_valu_w)[r] = pos
                (last_valu_w if x.kind == "trans" else last_trans_w).pop(r, None)
                last_mfma_w.pop(r, None)
        if x.kind in ("mfma", "mfma16"):
            for r in x.writes:
                last_mfma_w[r] = (pos, 12 if x.kind == "mfma" else 8)
                last_valu_w.pop(r, None)
        if x.sim and x.sim[0] == "s_mov_m0":
            last_m0 = pos
        pos += n
    return errs


# ---- output ----------------------------------------------------------------------------------------------------------------------
def render(prog: Program) -> str:
    return " \\\n".join(f'  "{x.text}\\n\\t"' for x in prog.ins if x.kind != "touch")


def emit_file(opts=None) -> str:
    opts = dict(opts or {})
    parts = ["// GENERATED by tools/attn64/gen.py -- do not edit; `python tools/attn64/gen.py --write` after changing the generator.\n",
             "// The hand-placed main loop of attention.hip::attn64_kernel (register map and schedule: tools/attn64/gen.py header).\n",
             "#pragma once\n"]
    for h16 in (False, True):
        prog = Program(Variant(h16), **opts).build()
        errs = check_hazards(prog.ins)
        if errs:
            raise SystemExit("hazards:\n" + "\n".join(errs[:40]))
        parts.append(f"#define ATTN64_ASM_{prog.var.name} \\\n" + render(prog) + "\n\n")
    parts.append(f"#define ATTN64_ROWSUM_VALU {int(prog.rowsum != 'mfma')}  // 1: a[64] / a[68] hold each lane's half of its row's sum\n")
    clob = ", ".join([f'"v{i}"' for i in range(N_VGPR_CLOBBER)] + [f'"a{i}"' for i in range(N_AGPR_CLOBBER)] + [f'"s{i}"' for i in SGPR_CLOBBER] + ['"scc"', '"memory"'])
    parts.append(f"#define ATTN64_CLOBBERS {clob}\n")
    return "".join(parts)


def stats(prog: Program):
    from collections import Counter
    c = Counter()
    inside = False
    for x in prog.ins:
        if x.kind == "label" and x.sim[1] == "LOOP":
            inside = True
        if inside:
            c[x.kind] += 1
        if x.kind == "branch" and x.sim[2] == "LOOP":
            break
    return dict(c)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--stats", action="store_true")
    a = ap.parse_args()
    text = emit_file()
    if a.stats:
        print(stats(Program(Variant(False)).build()))
    if a.write:
        OUT.write_text(text)
        print("wrote", OUT, len(text), "bytes")
    if a.check:
        if not OUT.exists() or OUT.read_text() != text:
            raise SystemExit(f"{OUT} is stale: run python tools/attn64/gen.py --write")
        print("up to date")


if __name__ == "__main__":
    main()
