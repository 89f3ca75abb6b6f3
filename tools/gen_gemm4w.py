#!/usr/bin/env python3
"""Generator of the hand-scheduled four-wave K-loop of pclip_gemm4w.hip (proto-clip_amd/csrc/pclip_gemm4w_loop.inc).

The loop is ONE `asm volatile` statement (hipcc cannot schedule this shape: with 256 accumulators per lane it spills inside the
K-loop, profiles/r03_ab_rejected.txt) whose operands are ordinary C++ variables, so the compiler still allocates every register and
knows what is live; only the instruction ORDER inside the statement is fixed here.

Geometry: 256 x 256 x 64 tile, four waves (one per SIMD), wave (wm, wn) owns 128 x 128 = 8 x 8 accumulator tiles of
v_mfma_f32_16x16x32_f16 (256 AGPRs).  Per 32-wide k-step a wave reads 8 A + 8 B fragments (16 ds_read_b128) for 64 MFMAs — 0.25
LDS reads per MFMA against 0.375 in the eight-wave kernel.  Same operand roles, same fragment <-> lane mapping and the same (K-tile,
k-step) order per accumulator as pgemm::mainloop_sr: results are bit-identical to linear_fast_kernel.

LDS: a ring of FIVE 32 KB half-slots (all 160 KB).  The operands travel as ONE stream of half-tiles that runs on across output
tiles: h(2t) = A(t), h(2t+1) = B(t) of the current output tile, then A'(0), B'(0), A'(1) ... of the workgroup's NEXT output tile;
half-tile n sits in slot n mod 5 (image as pgemm::stage_tile: [256 rows][128 B], 16-byte chunks XOR-swizzled by (row >> 1) & 7).
ONE barrier per K-tile, B(t), between the two k-steps of K-tile t:

    B(t)   : every wave has read all of K-tile t (its k-step-1 fragments are in registers) and waited for its own pieces of
             K-tile t+1  ->  tile t's two slots are free, tile t+1 is visible
    block 1: MFMAs (t, ks 1)   || fragment reads (t+1, ks 0) || LDS-DMA of h(2t+5) = B(t+2) into A(t)'s slot
    block 2: MFMAs (t+1, ks 0) || fragment reads (t+1, ks 1) || LDS-DMA of h(2t+6) = A(t+3) into B(t)'s slot
    s_waitcnt vmcnt(8): everything up to B(t+2) has landed, the 8 pieces of A(t+3) stay in flight;  B(t+1)

so a weight half-tile has one K-tile of lead and an activation half-tile one and a half to two, with up to 96 KB in flight.

Statement per output tile (PCLIP_GEMM4W_LOOP).  On entry K-tile 0 is visible and A(1) requested (by the previous tile's statement,
or PCLIP_GEMM4W_COLD for a workgroup's first tile); the two slots behind them were the epilogue's staging buffers:
    next tile's bias fragment -> registers (global loads, older than every wait of the statement)
    block (0, ks 0) || reads (0, ks 1) || DMA B(1), A(2);  vmcnt(8);  B(0)
    steady iterations t = 0 .. nt-4
    t = nt-3: DMA B(nt-1), A'(0);  vmcnt(8);  B(nt-2)           A', B' = the NEXT output tile (descriptors rsa2 / rsb2)
    t = nt-2: DMA B'(0), A'(1);  no barrier
    last block (nt-1, ks 1);  vmcnt(8): A'(0), B'(0) have landed — the epilogue's first barrier publishes them
The slots of A(nt-1), B(nt-1) are free behind the statement: the epilogue stages its output slabs there (ring positions returned).
"""
import argparse


class Ops:
    """operand table: name -> (constraint, C++ expression); numbering in declaration order, outputs first"""
    def __init__(self):
        self.outs, self.ins = [], []

    def out(self, name, cons, expr): self.outs.append((name, cons, expr))
    def inp(self, name, cons, expr): self.ins.append((name, cons, expr))

    def index(self):
        return {n: i for i, (n, _, _) in enumerate(self.outs + self.ins)}


class Gen:
    def __init__(self, dma_spread=6, dma_first=9, read_stride=2, sleep=0, cold=False, b1=False, b1_early=False, first_vm=8):
        self.lines = []
        self.dma_spread = dma_spread      # MFMAs between two LDS-DMA pieces of a block
        self.dma_first = dma_first        # MFMA index behind which the first piece's M0 write sits
        self.read_stride = read_stride    # MFMAs between two fragment reads
        self.sleep = sleep                # race-stress build: s_sleep jitter around waits (0 = off)
        self.cold = cold or b1                # statements without MFMAs: the first tile's prefetch (cold) / the next tile's B'(1) out of the epilogue (b1)
        self.b1 = b1
        self.b1_early = b1_early              # B(1) is requested by the PREVIOUS tile's epilogue (PCLIP_GEMM4W_B1), not by this statement's first block
        # counted wait behind the statement's first block: 8 = only A(2) stays in flight; with b1_early on tiles whose last epilogue slab issues exactly eight stores per
        # wave (bias / QuickGELU tiles: one bounds-checked descriptor, no branch) 24 = those eight stores + the eight bias loads + A(2): B(1) is OLDER than all of them
        self.first_vm = first_vm
        cold = self.cold
        o = Ops()
        if not cold:
            for i in range(8):
                for j in range(8): o.out(f"acc{i}_{j}", "+a", f"acc.q[{i}][{j}]")
            for k in range(32): o.out(f"fr{k}", "=&v", f"fr[{k}]")
            for k in range(4): o.out(f"vr{k}", "=&v", f"vr[{k}]")
        if not b1:
            for k in range(8): o.out(f"nb{k}", "=&v", f"nb[{k}]")
        for k in range(8): o.out(f"so{k}", "=&s", f"so[{k}]")
        for n in ("cnt", "tmp", "scr"): o.out(n, "=&s", f"st_{n}")
        for n in ("wr", "rda", "rdb"): o.out(n, "+s", f"ring_{n}")
        if not cold:
            o.inp("rsa", "s", "rsa"); o.inp("rsb", "s", "rsb")
        o.inp("rsa2", "s", "rsa2"); o.inp("rsb2", "s", "rsb2")
        o.inp("row16a", "s", "row16a"); o.inp("row16b", "s", "row16b")
        if not cold: o.inp("nt", "s", "nt")
        o.inp("wbase", "s", "wbase")
        for n in ("voffa0", "voffa1", "voffb0", "voffb1"): o.inp(n, "v", n)
        if not b1: o.inp("biasp", "v", "biasp")
        if not cold:
            o.inp("lanea", "v", "lanea"); o.inp("laneb", "v", "laneb")
        self.ops = o
        self.ix = o.index()

    def r(self, name): return f"%{self.ix[name]}"
    def acc(self, i, j): return self.r(f"acc{i}_{j}")
    def frag(self, buf, op, idx): return self.r(f"fr{buf * 16 + (0 if op == 'A' else 8) + idx}")
    def so(self, op, q): return self.r(f"so{(0 if op == 'A' else 4) + q}")
    def voff(self, op, par): return self.r(f"voff{op.lower()}{par}")
    def e(self, s): self.lines.append(s)

    # ---- pieces ------------------------------------------------------------------------------------------------------------
    def dma_m0(self, p): return f"s_add_u32 m0, {self.r('tmp')}, {p * 1024}"

    def dma_load(self, op, p, nxt=False):
        rs = self.r(("rsa" if op == 'A' else "rsb") + ("2" if nxt else ""))
        return f"buffer_load_dwordx4 {self.voff(op, p & 1)}, {rs}, {self.so(op, p >> 1)} offen lds"

    def advance_k(self, op): return [f"s_add_u32 {self.so(op, q)}, {self.so(op, q)}, 128" for q in range(4)]

    def reset_k(self, op, k_tiles=0):
        row16 = self.r("row16a" if op == 'A' else "row16b")
        out = [f"s_mov_b32 {self.so(op, 0)}, {k_tiles * 128}"]
        for q in range(1, 4): out.append(f"s_add_u32 {self.so(op, q)}, {self.so(op, q - 1)}, {row16}")
        return out

    def ring_next_wr(self):
        wr = self.r("wr")
        return [f"s_add_u32 {wr}, {wr}, 0x8000", f"s_cmp_eq_u32 {wr}, 0x28000", f"s_cselect_b32 {wr}, 0, {wr}"]

    def ring_next_rd(self, name):
        reg, scr = self.r(name), self.r("scr")
        return [f"s_add_u32 {reg}, {reg}, 0x10000", f"s_sub_u32 {scr}, {reg}, 0x28000", f"s_cmp_ge_u32 {reg}, 0x28000", f"s_cselect_b32 {reg}, {scr}, {reg}"]

    def m0_base(self): return f"s_add_u32 {self.r('tmp')}, {self.r('wr')}, {self.r('wbase')}"

    def reads(self, buf, ks):
        ra, rb = self.r("vr0" if ks == 0 else "vr1"), self.r("vr2" if ks == 0 else "vr3")
        order = [('B', 0), ('A', 0)] + [('B', j) for j in range(1, 8)] + [('A', i) for i in range(1, 8)]
        return [f"ds_read_b128 {self.frag(buf, op, idx)}, {ra if op == 'A' else rb} offset:{idx * 2048}" for op, idx in order]

    def mfma_block(self, buf, fillers):
        for i in range(8):
            for j in range(8):
                self.e(f"v_mfma_f32_16x16x32_f16 {self.acc(i, j)}, {self.frag(buf, 'B', j)}, {self.frag(buf, 'A', i)}, {self.acc(i, j)}")
                for ins in fillers.get(i * 8 + j, []): self.e(ins)

    @staticmethod
    def place(fillers, m, ins): fillers.setdefault(min(m, 63), []).append(ins)

    def block(self, buf, read_buf, read_ks, dmas, salu):
        """one 64-MFMA block on fragment buffer `buf`: the 16 fragment reads of (read_buf, read_ks) behind the even MFMAs; the half-tiles `dmas` = [(op, next_tile,
        pre)] one after the other behind the odd ones — [pre, M0 base] + 8 x (M0 write, two gaps, piece) + [ring advance, k advance]; `salu` behind the last piece"""
        f = {}
        if read_buf is not None:
            for n, ins in enumerate(self.reads(read_buf, read_ks)): self.place(f, n * self.read_stride, ins)
        # two half-tiles in one block (the first block of a statement) fit with four gaps per piece, starting at the first gap
        g = self.dma_first if len(dmas) < 2 else 1
        spread = self.dma_spread if len(dmas) < 2 else min(self.dma_spread, 4)
        for op, nxt, pre in dmas:
            for ins in list(pre) + [self.m0_base()]: self.place(f, g, ins)
            for p in range(8):
                assert g + 2 <= 63, "a piece must not share its gap with its M0 write (one wait state between them)"
                self.place(f, g, self.dma_m0(p))
                self.place(f, g + 2, self.dma_load(op, p, nxt))
                g += spread
            g = max(g, g - spread + 4)                          # first gap behind the last piece
            for ins in self.ring_next_wr() + self.advance_k(op): self.place(f, g - 1, ins)
        for ins in salu: self.place(f, max(g, 33), ins)
        self.mfma_block(buf, f)

    def set_read_addresses(self):
        self.e(f"v_add_u32 {self.r('vr0')}, {self.r('rda')}, {self.r('lanea')}")
        self.e(f"v_add_u32 {self.r('vr2')}, {self.r('rdb')}, {self.r('laneb')}")
        self.e(f"v_xor_b32 {self.r('vr1')}, 64, {self.r('vr0')}")
        self.e(f"v_xor_b32 {self.r('vr3')}, 64, {self.r('vr2')}")

    def jitter(self, n):
        if self.sleep:
            # race-stress build: at jitter point n the wave on SIMD (-n mod 4) pauses — every wait / barrier is approached with one wave late, a different one
            # each time (tmp is dead here: it is re-formed before every use)
            t = self.r("tmp")
            self.e(f"s_getreg_b32 {t}, hwreg(HW_REG_HW_ID, 4, 2)")
            self.e(f"s_add_u32 {t}, {t}, {n}")
            self.e(f"s_and_b32 {t}, {t}, 3")
            self.e(f"s_cmp_lg_u32 {t}, 0")
            self.e("s_cbranch_scc1 9f")
            self.e(f"s_sleep {self.sleep}")
            self.e("9:")

    def iteration(self, dma1, dma2, barrier, extra1=(), extra2=()):
        """loop body for one K-tile, entered right behind barrier B(t); dma1 / dma2: (op, next_tile, pre) or None"""
        self.set_read_addresses()
        self.block(buf=1, read_buf=0, read_ks=0, dmas=[dma1] if dma1 else [], salu=list(extra1))
        self.e("s_waitcnt lgkmcnt(0)")
        self.block(buf=0, read_buf=1, read_ks=1, dmas=[dma2] if dma2 else [], salu=list(extra2) + self.ring_next_rd("rda") + self.ring_next_rd("rdb"))
        if barrier:
            self.jitter(1)
            self.e("s_waitcnt vmcnt(8) lgkmcnt(0)")                 # the A pieces stay in flight
            self.jitter(2)
            self.e("s_barrier")
        else:
            self.e("s_waitcnt lgkmcnt(0)")

    def bias_loads(self):
        for j in range(8): self.e(f"global_load_dwordx2 {self.r(f'nb{j}')}, {self.r('biasp')}, off offset:{j * 32}")

    def prefetch_plain(self):
        """A'(0), B'(0), A'(1) of a workgroup's FIRST tile, no MFMAs around (PCLIP_GEMM4W_COLD)"""
        e = self.e
        e("s_nop 4")
        self.bias_loads()                                       # the first tile's bias fragment (inside the statement: hipcc then has no load of its own to wait for
                                                                # at the loop header, where it would drain the previous tile's stores on every iteration)
        for ins in self.reset_k('A') + self.reset_k('B'): e(ins)
        for op in ('A', 'B', 'A'):
            e(self.m0_base())
            for p in range(8):
                e(self.dma_m0(p)); e("s_nop 0"); e(self.dma_load(op, p, nxt=True))
            for ins in self.ring_next_wr(): e(ins)
            if op == 'A':
                for ins in self.advance_k('A'): e(ins)
        self.jitter(7)
        e("s_waitcnt vmcnt(8)")

    def b1_plain(self):
        """B'(1) of the next output tile into the slot the epilogue has just finished with (issued from the epilogue, one interval before it ends)"""
        e = self.e
        e("s_nop 4")
        for ins in self.reset_k('B', 1): e(ins)
        e(self.m0_base())
        for p in range(8):
            e(self.dma_m0(p)); e("s_nop 0"); e(self.dma_load('B', p, nxt=True))
        for ins in self.ring_next_wr(): e(ins)

    def generate(self):
        self.lines = []
        if self.b1:
            self.b1_plain()
            return self.lines
        if self.cold:
            self.prefetch_plain()
            return self.lines
        e = self.e
        e("s_nop 4")                                            # SGPR operands fresh from v_readfirstlane -> buffer / global instructions
        # next tile's bias fragment (the caller turns it into the next accumulators' initial value): older than every wait below
        self.bias_loads()
        # A(0), B(0), A(1) were requested by the previous statement (and B(1) by the epilogue behind it): B continues at K-tile 1 (2), A at K-tile 2
        for ins in self.reset_k('A', 2) + self.reset_k('B', 2 if self.b1_early else 1): e(ins)
        self.set_read_addresses()
        for ins in self.reads(0, 0): e(ins)
        for ins in self.ring_next_rd("rda") + self.ring_next_rd("rdb"): e(ins)            # -> K-tile 1
        e("s_waitcnt lgkmcnt(0)")
        self.block(buf=0, read_buf=1, read_ks=1, dmas=[('A', False, [])] if self.b1_early else [('B', False, []), ('A', False, [])], salu=[])
        self.jitter(5)
        e(f"s_waitcnt vmcnt({self.first_vm}) lgkmcnt(0)")       # K-tile 1 landed (and everything older: the previous tile's output stores), A(2) may fly
        self.jitter(6)
        e("s_barrier")                                          # B(0)
        # ---- steady iterations t = 0 .. nt - 4
        cnt = self.r("cnt")
        e(f"s_sub_u32 {cnt}, {self.r('nt')}, 3")
        e(f"s_cmp_eq_u32 {cnt}, 0")
        e("s_cbranch_scc1 2f")
        e("1:")
        self.iteration(('B', False, []), ('A', False, []), barrier=True)
        e(f"s_sub_u32 {cnt}, {cnt}, 1")
        e(f"s_cmp_lg_u32 {cnt}, 0")
        e("s_cbranch_scc1 1b")
        e("2:")
        # ---- t = nt - 3: B(nt - 1), then the NEXT output tile's A'(0)
        self.iteration(('B', False, []), ('A', True, self.reset_k('A')), barrier=True)
        # ---- t = nt - 2: B'(0), A'(1); no barrier behind it (nothing is published or freed)
        self.iteration(('B', True, self.reset_k('B')), ('A', True, []), barrier=False)
        # ---- t = nt - 1: k-step 1 of the last K-tile
        self.mfma_block(1, {})
        self.jitter(8)
        e("s_waitcnt vmcnt(8)")                                 # A'(0), B'(0) landed; A'(1) may fly
        e("s_nop 15")                                           # MFMA results -> compiler code behind the statement
        return self.lines

    def emit_statement(self, name):
        lines = self.generate()
        body = "\n".join(f'        "{l}\\n\\t" \\' for l in lines)
        outs = ", ".join(f'"{c}"({x})' for _, c, x in self.ops.outs)
        ins = ", ".join(f'"{c}"({x})' for _, c, x in self.ops.ins)
        return (f"// GENERATED by tools/gen_gemm4w.py — do not edit.  {len(lines)} instructions.\n"
                f"#define {name}() \\\n    asm volatile( \\\n{body}\n        : {outs} \\\n        : {ins} \\\n        : \"memory\", \"scc\")\n")


# Variants compiled side by side (linear4w_kernel<ACT, HAS_BIAS, VAR>, selected per launch by PCLIP_GEMM4W_VAR): 0 = the product loop, 1 = its race-stress build
# (s_sleep jitter in front of every counted wait and barrier), 2 .. = schedule experiments measured in the same process (tools/gemm4w_check.py --variants)
# Measured (profiles/r05_gemm4w_variants.txt, same process, interleaved rounds, bit-identical): pieces behind most of the block's fragment reads (first at MFMA 9, every
# sixth) +0.5 ... +1.7 % over pieces from the first gap; a burst in the first sixteen gaps -2 ... -3 %; spread over the whole block +-0; B'(1) requested out of the previous
# tile's epilogue (one interval of lead instead of half a K-tile) -0.9 ... -1.5 %.
VARIANTS = {
    0: dict(),
    1: dict(sleep=3),
    # 2 .. 5: schedule / epilogue experiments of round 5, measured and removed from the product sources (git history: piece placement from the first gap or later, a burst,
    #         every gap; B'(1) out of the epilogue with and without a relaxed first wait; batched LDS reads on residual tiles; un-staged 8-byte stores) — all within +-1.5 %
    #         or slower: profiles/r05_gemm4w_variants.txt, r05_gemm4w_epilogue.txt
    6: dict(),                                # the product loop WITHOUT the epilogue (nothing is stored): ablation — what a fully hidden epilogue would buy at most
    8: dict(),                                # the product loop in a kernel that keeps time stamps of its tile phases (tools/gemm4w_stamps.py)
    # 7 (round 5, removed in round 6): K = 768 unrolled with the previous tile's second half stored from registers under the loop — bit-identical and 5 - 13 % SLOWER in
    #    every form (profiles/r05_gemm4w_defer.txt: the scattered stores cost the LDS-DMA stream more than the half epilogue they hide)
}
B1_EARLY = {v: kw.get("b1_early", False) for v, kw in VARIANTS.items()}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-o", default="proto-clip_amd/csrc/pclip_gemm4w_loop.inc")
    a = ap.parse_args()
    txt = ""
    for v, kw in VARIANTS.items():
        txt += Gen(**kw).emit_statement(f"PCLIP_GEMM4W_LOOP_V{v}") + "\n"
    txt += Gen(cold=True).emit_statement("PCLIP_GEMM4W_COLD") + "\n" + Gen(cold=True, sleep=3).emit_statement("PCLIP_GEMM4W_COLD_STRESS")
    txt += "\n" + Gen(b1=True).emit_statement("PCLIP_GEMM4W_B1")
    txt += "\n#define PCLIP_GEMM4W_NVAR %d\n" % (max(VARIANTS) + 1)
    txt += "#define PCLIP_GEMM4W_HAS_VAR(v) (%s)\n" % " || ".join(f"(v) == {v}" for v in VARIANTS)
    txt += "#define PCLIP_GEMM4W_B1_EARLY(v) (%s)\n" % (" || ".join(f"(v) == {v}" for v, on in B1_EARLY.items() if on) or "false")
    open(a.o, "w").write(txt)
    print("wrote", a.o)
