#!/usr/bin/env python
"""Functional model of the instruction stream tools/attn64/gen.py emits: four waves of one workgroup executed on numpy register
files, a shared LDS image and a flat global memory, with the instruction semantics the stream relies on (MFMA fragment layouts of
cdna_hip_programming.md section 3, ds_read_b64_tr_b16's gather, the lane-linear LDS-DMA, v_permlane32_swap's half exchange).

What it checks that a GPU run cannot tell apart from luck:
  * every LDS read of DMA-written bytes happens behind the issuing wave's vmcnt wait AND a barrier (two landing models: each piece
    lands when it is issued / only at its wave's wait; both must give the reference result, under two wave orders);
  * every register a ds_read / global load returns is covered by a counted wait before its first consumer (`pending` registers);
  * the result against a float64 soft-max attention of the same bf16 / fp16 inputs.
Test infrastructure (tests/test_attn64_sim.py) -- never imported by the package.
"""
from __future__ import annotations

import numpy as np

from gen import Ins, Program, Variant, SG, O, L, N_VGPR_CLOBBER  # noqa: F401

U32 = np.uint32


def f32(u):
    return u.view(np.float32)


def u32(f):
    return np.asarray(f, dtype=np.float32).view(U32)


def bf16_to_f32(h):  # h: uint16 array
    return (h.astype(U32) << 16).view(np.float32)


def f32_to_bf16(x):  # round to nearest even
    u = np.asarray(x, dtype=np.float32).view(U32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    return r


def half_to_f32(h, h16):
    return h.view(np.float16).astype(np.float32) if h16 else bf16_to_f32(h)


def f32_to_half(x, h16):
    return np.asarray(x, dtype=np.float32).astype(np.float16).view(np.uint16) if h16 else f32_to_bf16(x)


class Wave:
    def __init__(self, wg, wid, inputs):
        self.wg, self.wid = wg, wid
        self.v = np.zeros((256, 64), U32)
        self.a = np.zeros((256, 64), U32)
        self.s = {}
        self.scc = 0
        self.m0 = 0
        self.inp = inputs          # name -> int (scalar) | uint32[64] | uint64[64]
        self.pc = 0
        self.dma_pending = []      # (lds_dst, bytes[64, 16]) issued and not yet waited for (oldest first)
        self.reg_pending = {}      # (file, n) -> kind of the load that has not been waited for ("ds" | "vm")
        self.ds_queue, self.vm_queue = [], []
        self.done = False
        self.reads_since_barrier = []  # (lo, hi) LDS byte ranges

    def file(self, f):
        return self.v if f == "v" else self.a


class Workgroup:
    def __init__(self, prog: Program, gmem: np.ndarray, inputs_per_wave, lds_bytes=65536, land="issue"):
        self.prog, self.h16 = prog, prog.var.h16
        self.ins = prog.ins
        self.labels = {x.sim[1]: i for i, x in enumerate(self.ins) if x.kind == "label"}
        self.gmem = gmem
        self.lds = np.zeros(lds_bytes, np.uint8)
        self.waves = [Wave(self, w, inputs_per_wave[w]) for w in range(4)]
        self.land = land
        self.dma_written_phase = []   # LDS ranges a DMA issued in the current barrier phase may write
        self.count = {}

    # -- register helpers ---------------------------------------------------------------------------------------
    def _check_ready(self, w: Wave, x: Ins):
        for r in x.reads + (x.writes if x.kind not in ("ds", "gload") else []):
            if r in w.reg_pending:
                raise AssertionError(f"wave {w.wid}: {x.text} touches {r} before the wait that covers its {w.reg_pending[r]} load")

    def frag32(self, w, f, base):  # A or B operand of a 32x32x16 MFMA: [32][16] float32
        regs = w.file(f)[base:base + 4]                      # [4][64]
        halves = np.stack([regs & 0xFFFF, regs >> 16], axis=1).reshape(8, 64).astype(np.uint16)  # element e = 2 * reg + half
        vals = half_to_f32(halves, self.h16)                 # [8][64]
        m = np.zeros((32, 16), np.float32)
        for h in range(2):
            m[:, 8 * h:8 * h + 8] = vals[:, 32 * h:32 * h + 32].T
        return m

    def frag16(self, w, f, base):  # operand of a 16x16x32 MFMA: [16][32]
        regs = w.file(f)[base:base + 4]
        halves = np.stack([regs & 0xFFFF, regs >> 16], axis=1).reshape(8, 64).astype(np.uint16)
        vals = half_to_f32(halves, self.h16)
        m = np.zeros((16, 32), np.float32)
        for g in range(4):
            m[:, 8 * g:8 * g + 8] = vals[:, 16 * g:16 * g + 16].T
        return m

    def acc32(self, w, f, base):   # C / D of a 32x32 MFMA as [32 rows][32 cols]; lane l reg r: col l & 31, row (r&3) + 8 (r>>2) + 4 (l>>5)
        regs = f32(w.file(f)[base:base + 16])
        m = np.zeros((32, 32), np.float32)
        for r in range(16):
            for h in range(2):
                m[(r & 3) + 8 * (r >> 2) + 4 * h, :] = regs[r, 32 * h:32 * h + 32]
        return m

    def put32(self, w, f, base, m):
        out = np.zeros((16, 64), np.float32)
        for r in range(16):
            for h in range(2):
                out[r, 32 * h:32 * h + 32] = m[(r & 3) + 8 * (r >> 2) + 4 * h, :]
        w.file(f)[base:base + 16] = out.view(U32)

    # -- execution ---------------------------------------------------------------------------------------------------------
    def lds_read(self, w, addr, n):
        lo = int(addr)
        w.reads_since_barrier.append((lo, lo + n))
        return self.lds[lo:lo + n]

    def step_wave(self, w: Wave):
        """Run wave w up to (and including) its next barrier or the end of the program."""
        while True:
            if w.pc >= len(self.ins):
                w.done = True
                return
            x = self.ins[w.pc]
            w.pc += 1
            k = x.kind
            self.count[k] = self.count.get(k, 0) + 1
            if k in ("label", "nop", "touch"):
                continue
            op = x.sim
            if k == "barrier":
                return
            if k == "branch":
                if op[1] == "s_branch" or (op[1] == "s_cbranch_scc1" and w.scc) or (op[1] == "s_cbranch_scc0" and not w.scc):
                    w.pc = self.labels[op[2]]
                continue
            if k == "wait":
                vm, lg = op[1], op[2]
                if vm is not None:
                    n_done = max(0, len(w.vm_queue) - vm)
                    for item in w.vm_queue[:n_done]:
                        if item[0] == "dma":
                            dst, data = w.dma_pending.pop(0)
                            if self.land == "wait":
                                self.lds[dst:dst + 1024] = data.reshape(-1)
                        else:
                            for r in item[1]:
                                w.reg_pending.pop(r, None)
                    del w.vm_queue[:n_done]
                if lg is not None:
                    n_done = max(0, len(w.ds_queue) - lg)
                    for regs_ in w.ds_queue[:n_done]:
                        for r in regs_:
                            w.reg_pending.pop(r, None)
                    del w.ds_queue[:n_done]
                continue
            self._check_ready(w, x)
            getattr(self, "op_" + op[0])(w, *op[1:])
            if k == "ds":
                w.ds_queue.append(list(x.writes))
                for r in x.writes:
                    w.reg_pending[r] = "ds"
            elif k == "gload":
                w.vm_queue.append(("gload", list(x.writes)))
                for r in x.writes:
                    w.reg_pending[r] = "vm"

    def run(self, order=(0, 1, 2, 3), max_phases=100000):
        for _ in range(max_phases):
            for wid in order:
                w = self.waves[wid]
                if not w.done:
                    self.step_wave(w)
            # barrier reached by everybody (or program end): races of this phase
            for w in self.waves:
                for lo, hi in w.reads_since_barrier:
                    for dlo, dhi, who in self.dma_written_phase:
                        if lo < dhi and dlo < hi:
                            raise AssertionError(f"LDS race: wave {w.wid} read [{lo}, {hi}) in the phase in which wave {who} issued a DMA into [{dlo}, {dhi})")
                w.reads_since_barrier = []
            # a DMA still pending at the barrier may land in the next phase too
            self.dma_written_phase = [(d, d + 1024, w.wid) for w in self.waves for d, _ in w.dma_pending]
            if all(w.done for w in self.waves):
                assert len({w.pc for w in self.waves}) == 1
                return
        raise AssertionError("no end")

    # -- ops ---------------------------------------------------------------------------------------------------------------------------
    def op_mfma32(self, w, d, a, b, c):
        A, B = self.frag32(w, *a), self.frag32(w, *b)          # A[i][k], B as [col j][k]
        C = self.acc32(w, *c) if c is not None else np.zeros((32, 32), np.float32)
        D = (A.astype(np.float64) @ B.astype(np.float64).T + C).astype(np.float32)
        self.put32(w, d[0], d[1], D)

    def op_mfma16(self, w, d, a, b, c):
        A, B = self.frag16(w, *a), self.frag16(w, *b)          # [16][32] each; D[i][j] = sum_k A[i][k] B[j][k]
        regs = f32(w.file(c[0])[c[1]:c[1] + 4])                  # lane l reg r: col l & 15, row 4 (l >> 4) + r
        C = np.zeros((16, 16), np.float32)
        for r in range(4):
            for g in range(4):
                C[4 * g + r, :] = regs[r, 16 * g:16 * g + 16]
        D = (A.astype(np.float64) @ B.astype(np.float64).T + C).astype(np.float32)
        out = np.zeros((4, 64), np.float32)
        for r in range(4):
            for g in range(4):
                out[r, 16 * g:16 * g + 16] = D[4 * g + r, :]
        w.file(d[0])[d[1]:d[1] + 4] = out.view(U32)

    def op_exp(self, w, r):
        with np.errstate(over="ignore", under="ignore"):
            w.v[r] = u32(np.exp2(f32(w.v[r]).astype(np.float64)).astype(np.float32))

    def op_cvt(self, w, d, lo, hi):
        a, b = f32_to_half(f32(w.v[lo]), self.h16), f32_to_half(f32(w.v[hi]), self.h16)
        w.v[d] = a.astype(U32) | (b.astype(U32) << 16)

    def op_max3(self, w, d, a, b, c):
        w.v[d] = u32(np.maximum(np.maximum(f32(w.v[a]), f32(w.v[b])), f32(w.v[c])))

    def op_max(self, w, d, a, b):
        w.v[d] = u32(np.maximum(f32(w.v[a]), f32(w.v[b])))

    def op_mov(self, w, d, a):
        w.v[d] = w.v[a].copy()

    def op_swap32(self, w, d, s):  # lanes 32-63 of vdst <-> lanes 0-31 of src
        t = w.v[d][32:].copy()
        w.v[d][32:] = w.v[s][:32]
        w.v[s][:32] = t

    def op_addc(self, w, d, a, c):
        w.v[d] = u32(f32(w.v[a]) + np.float32(c))

    def op_neg(self, w, d, a):
        w.v[d] = u32(np.float32(0) - f32(w.v[a]))

    def op_sub(self, w, d, a, b):
        w.v[d] = u32(f32(w.v[a]) - f32(w.v[b]))

    def op_pkadd(self, w, d, a):
        for i in range(2):
            w.v[d + i] = u32(f32(w.v[d + i]) + f32(w.v[a + i]))

    def op_add(self, w, d, a, b):
        w.v[d] = u32(f32(w.v[a]) + f32(w.v[b]))

    def op_movc(self, w, d, c):
        w.v[d] = u32(np.full(64, c, np.float32))

    def op_acc_from_v(self, w, d, a):
        w.a[d] = w.v[a].copy()

    def op_acc_zero(self, w, r):
        w.a[r] = 0

    def op_acc_in(self, w, r, name):
        w.a[r] = w.inp[name]

    def op_ds_b128(self, w, d, addr_name, off):
        addr = w.inp[addr_name].astype(np.int64) + off
        assert not (addr & 15).any()
        out = np.zeros((4, 64), U32)
        for l in range(64):
            out[:, l] = self.lds_read(w, addr[l], 16).view(U32)
        w.file(d[0])[d[1]:d[1] + 4] = out

    def op_ds_tr(self, w, d, addr_name, off):
        """ds_read_b64_tr_b16: within each 16-lane group, lane i supplies the address of 4 consecutive 16-bit elements = row i >> 2,
        columns 4 (i & 3) .. of a [4][16] block; lane i receives column i of that block (rows 0..3)."""
        addr = w.inp[addr_name].astype(np.int64) + off
        assert not (addr & 7).any()
        out = np.zeros((2, 64), U32)
        for g in range(4):
            blk = np.zeros((4, 16), np.uint16)
            for i in range(16):
                blk[i >> 2, 4 * (i & 3):4 * (i & 3) + 4] = self.lds_read(w, addr[16 * g + i], 8).view(np.uint16)
            for i in range(16):
                col = blk[:, i].astype(U32)
                out[0, 16 * g + i] = col[0] | (col[1] << 16)
                out[1, 16 * g + i] = col[2] | (col[3] << 16)
        w.file(d[0])[d[1]:d[1] + 2] = out

    def op_gload(self, w, d, addr_name, off):
        addr = w.inp[addr_name].astype(np.int64) + off
        out = np.zeros((4, 64), U32)
        for l in range(64):
            out[:, l] = self.gmem[addr[l]:addr[l] + 16].view(U32)
        w.file(d[0])[d[1]:d[1] + 4] = out

    def op_dma(self, w, voff_name, ptr):
        base = w.s[ptr] | (w.s[ptr + 1] << 32)
        addr = base + w.inp[voff_name].astype(np.int64)
        data = np.zeros((64, 16), np.uint8)
        for l in range(64):
            data[l] = self.gmem[addr[l]:addr[l] + 16]
        dst = w.m0
        self.dma_written_phase.append((dst, dst + 1024, w.wid))
        w.dma_pending.append((dst, data))
        w.vm_queue.append(("dma", None))
        if self.land == "issue":  # the earliest a piece can land; "wait" = the latest (at the wait that covers it)
            self.lds[dst:dst + 1024] = data.reshape(-1)

    # scalar ops
    def op_s_mov_m0(self, w, r):
        w.m0 = w.s[r]

    def op_s_save_m0(self, w, d):
        w.s[d] = w.m0

    def op_s_mov64_in(self, w, d, name):
        w.s[d], w.s[d + 1] = int(w.inp[name]) & 0xFFFFFFFF, int(w.inp[name]) >> 32

    def op_s_mov_in(self, w, d, name):
        w.s[d] = int(w.inp[name]) & 0xFFFFFFFF

    def op_s_sub_in(self, w, d, name, c):
        w.s[d] = (int(w.inp[name]) - c) & 0xFFFFFFFF
        w.scc = int(int(w.inp[name]) < c)

    def op_s_add_in(self, w, d, name, c):
        w.s[d] = (int(w.inp[name]) + c) & 0xFFFFFFFF

    def op_s_add(self, w, d, a, b):
        t = w.s[a] + w.s[b]
        w.s[d], w.scc = t & 0xFFFFFFFF, t >> 32

    def op_s_addc(self, w, d):
        t = w.s[d] + w.scc
        w.s[d], w.scc = t & 0xFFFFFFFF, t >> 32

    def op_s_subi(self, w, d, c):
        w.scc = int(w.s[d] < c)
        w.s[d] = (w.s[d] - c) & 0xFFFFFFFF

    def op_s_subb(self, w, d):
        t = w.s[d] - w.scc
        w.scc = int(t < 0)
        w.s[d] = t & 0xFFFFFFFF

    def op_s_cmp_lg(self, w, a, c):
        w.scc = int(w.s[a] != c)

    def op_s_cmp_eq(self, w, a, c):
        w.scc = int(w.s[a] == c)

    def op_s_cselect(self, w, d, a, c):
        w.s[d] = w.s[a] if w.scc else c


# ---- the host side of attn64_kernel, restated: what the C++ wrapper hands to the asm statement ---------------------------------------
def wave_inputs(wave, q_base, k_base, v_base, ldq, ldk, ldv, q0, Lq, nt, h16, lds_k=0, lds_v=2 * 8192):
    lane = np.arange(64)
    l31, lh = lane & 31, lane >> 5
    inp = {}
    for j in range(4):
        inp[f"kfa{j}"] = (lds_k + (l31 * 64 + (((2 * j + lh) ^ ((l31 >> 1) & 7)) * 8)) * 2).astype(U32)
    for db in range(2):
        inp[f"vfa{db}"] = (lds_v + ((4 * lh + ((lane & 15) >> 2)) * 64 + ((4 * (db ^ ((lane >> 3) & 1)) + 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1)) * 8)
                                    + 4 * (lane & 1)) * 2).astype(U32)
    d_row, d_slot = wave * 8 + (lane >> 3), lane & 7
    k_chunk, v_chunk = d_slot ^ ((d_row >> 1) & 7), d_slot ^ (((d_row >> 1) & 1) << 2)
    for pc in range(2):
        inp[f"dk{pc}"] = (((d_row + 32 * pc) * ldk + k_chunk * 8) * 2).astype(U32)
        inp[f"dv{pc}"] = (((d_row + 32 * pc) * ldv + v_chunk * 8) * 2).astype(U32)
    for qb in range(2):
        q = np.minimum(q0 + wave * 64 + qb * 32 + l31, Lq - 1)
        inp[f"qa{qb}"] = (q_base + (q * ldq + lh * 8) * 2).astype(np.uint64)
    inp["ones"] = np.where((lane == 0) | (lane == 32) | (lane == 17) | (lane == 49), 0x3C003C00 if h16 else 0x3F803F80, 0).astype(U32)
    inp.update(kbase=k_base, vbase=v_base, kstride=64 * ldk * 2, vstride=64 * ldv * 2, nt=nt, m0k=lds_k + wave * 1024, m0v=lds_v + wave * 1024)
    return inp


def read_result(wg: Workgroup, wave: int):
    """-> (O [64 rows][64 d] un-normalised fp32, l [64]) of one wave, from the accumulator file as the C++ epilogue reads it."""
    w = wg.waves[wave]
    out = np.zeros((64, 64), np.float32)
    lsum = np.zeros(64, np.float32)
    for qb in range(2):
        for db in range(2):
            m = wg.acc32(w, "a", O(db, qb))  # [d within block][q within block]
            out[32 * qb:32 * qb + 32, 32 * db:32 * db + 32] = m.T
        lr = f32(w.a[L(qb):L(qb) + 2])
        if wg.prog.rowsum != "mfma":             # each lane's half of its row's keys
            lsum[32 * qb:32 * qb + 32] = lr[0, :32] + lr[0, 32:]
        else:                                    # lanes 0..15: reg 0 = rows 0..15, reg 1 = rows 16..31
            lsum[32 * qb:32 * qb + 16] = lr[0, :16]
            lsum[32 * qb + 16:32 * qb + 32] = lr[1, :16]
    return out, lsum


def run_case(Lq=256, Lk=256, h16=False, seed=0, land="issue", order=(0, 1, 2, 3), ldq=64, ldk=64, ldv=64, spike=None, opts=None, qscale=None):
    """One workgroup (query rows 0..255 of one head) against float64 attention.  Q carries scale * log2(e) already (FOLD)."""
    rng = np.random.default_rng(seed)
    nt = Lk // 64
    # fp16 probabilities overflow once a score outgrows the first tile's maximum by 24: the kernel then redoes the rows with its exact loop
    # (not modelled here), so the fp16 cases stay inside the optimistic range
    qf = (rng.standard_normal((Lq, 64)) * (qscale if qscale is not None else (0.5 if h16 else 1.5))).astype(np.float32)
    kf = rng.standard_normal((Lk, 64)).astype(np.float32)
    vf = rng.standard_normal((Lk, 64)).astype(np.float32)
    if spike is not None:
        kf[spike] *= 6.0
    qh, kh, vh = f32_to_half(qf, h16), f32_to_half(kf, h16), f32_to_half(vf, h16)
    gmem = np.zeros(1 << 22, np.uint8)
    q_base, k_base, v_base = 4096, 1 << 20, 2 << 20

    def put(base, arr, ld):
        buf = np.zeros((arr.shape[0], ld), np.uint16)
        buf[:, :64] = arr
        gmem[base:base + buf.size * 2] = buf.reshape(-1).view(np.uint8)
    put(q_base, qh, ldq)
    put(k_base, kh, ldk)
    put(v_base, vh, ldv)
    prog = Program(Variant(h16), **(opts or {})).build()
    inputs = [wave_inputs(w, q_base, k_base, v_base, ldq, ldk, ldv, 0, Lq, nt, h16) for w in range(4)]
    wg = Workgroup(prog, gmem, inputs, land=land)
    wg.run(order=order)
    q64, k64, v64 = (half_to_f32(x, h16).astype(np.float64) for x in (qh, kh, vh))
    s = q64 @ k64.T
    p = np.exp2(s - s.max(axis=1, keepdims=True))
    ref = (p @ v64) / p.sum(axis=1, keepdims=True)
    got = np.zeros((256, 64))
    for w in range(4):
        o, l = read_result(wg, w)
        got[64 * w:64 * w + 64] = o.astype(np.float64) / l.astype(np.float64)[:, None]
    n = min(Lq, 256)
    err = np.abs(got[:n] - ref[:n]).max() / np.abs(ref[:n]).max()
    return err, wg


if __name__ == "__main__":
    import sys
    for h16 in (False, True):
        for land in ("issue", "wait"):
            for order in ((0, 1, 2, 3), (3, 2, 1, 0)):
                for Lk in (192, 256, 448):
                    err, wg = run_case(Lk=Lk, h16=h16, land=land, order=order, seed=Lk)
                    print(f"h16={h16} land={land} order={order} Lk={Lk}: max rel err {err:.3e}")
                    assert err < (2e-3 if h16 else 1.5e-2), err
    print("ok")
