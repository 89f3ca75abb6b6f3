#!/usr/bin/env python3
"""Static check of the generated four-wave K-loop (tools/gen_gemm4w.py): interpret ONE wave's instruction stream — scalar registers, M0, branches — for a
given number of K-tiles and check the ring protocol event by event.  Every wave runs the same stream and the only inter-wave ordering is s_barrier, so the
protocol holds for the workgroup iff it holds for this stream with "another wave may be anywhere between the same two barriers":

  DMA order    the LDS-DMA pieces are the half-tile stream B(1) A(2) B(2) A(3) ... B(nt-1) A'(0) B'(0) A'(1) (cold: A'(0) B'(0) A'(1)), eight pieces each, with the
               right descriptor, k offset (scalar offset) and 16-row step, into consecutive ring slots (mod 5), piece p at wave base + p * 1024
  RAW          a fragment read of slot s happens behind a barrier that is itself behind a counted vmcnt covering every piece of the half-tile in s
               (every wave waited for its own pieces, the barrier publishes them)
  WAR          an LDS-DMA into slot s is issued behind a barrier that is behind an lgkmcnt(0) that is behind the last fragment read of the previous occupant
  M0           an M0 write and the piece that uses it are separated by at least one instruction (one wait state), no other M0 write in between
  reads        K-tile t is read from the slots of A(t) / B(t), 16 reads per k-step with the right k-step XOR and fragment offsets
  MFMA         accumulator (i, j) takes B fragment j and A fragment i of the fragment buffer the last 16 reads of that k-step filled, (t, ks) ascending
"""
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from gen_gemm4w import Gen, VARIANTS  # noqa: E402


class ProtocolError(AssertionError):
    pass


def simulate(gen: Gen, nt: int, wr0: int = 0, rda0: int = 0, rdb0: int = 0x8000, row16a: int = 16 * 768 * 2, row16b: int = 16 * 1024 * 2, wbase: int = 8192):
    lines = gen.generate()
    inv = {f"%{v}": k for k, v in gen.ix.items()}
    prog = [re.sub(r"%\d+", lambda m: inv[m.group(0)], l) for l in lines]
    labels = {}
    for pc, l in enumerate(prog):
        if l.endswith(":"): labels.setdefault(l[:-1], []).append(pc)
    S = {"row16a": row16a, "row16b": row16b, "crow16": 16 * 2304 * 2, "nt": nt, "wbase": wbase, "wr": wr0, "rda": rda0, "rdb": rdb0, "scc": 0, "m0": None}
    V = {"lanea": ("lane", 0), "laneb": ("lane", 0)}          # vector registers that matter: (tag, scalar part)
    events = []                                              # (kind, payload)
    pc, steps = 0, 0
    last_m0_write = None

    def val(x):
        x = x.strip()
        if x in S: return S[x]
        if x.startswith("0x"): return int(x, 16)
        return int(x)

    while pc < len(prog):
        steps += 1
        if steps > 2_000_000: raise ProtocolError("runaway loop")
        l = prog[pc]
        op, _, rest = l.partition(" ")
        a = [t.strip() for t in rest.split(",")] if rest else []
        nxt = pc + 1
        if l.endswith(":") or op in ("s_nop", "s_sleep", "s_getreg_b32"):
            pass
        elif op == "s_mov_b32": S[a[0]] = val(a[1])
        elif op == "s_add_u32":
            r = val(a[1]) + val(a[2]); S["scc"] = int(r >= 1 << 32); S[a[0]] = r & 0xffffffff
            if a[0] == "m0": last_m0_write = len(events); events.append(("m0", pc))
        elif op == "s_sub_u32":
            r = val(a[1]) - val(a[2]); S["scc"] = int(r < 0); S[a[0]] = r & 0xffffffff
        elif op == "s_and_b32": S[a[0]] = val(a[1]) & val(a[2]); S["scc"] = int(S[a[0]] != 0)
        elif op == "s_cmp_eq_u32": S["scc"] = int(val(a[0]) == val(a[1]))
        elif op == "s_cmp_lg_u32": S["scc"] = int(val(a[0]) != val(a[1]))
        elif op == "s_cmp_ge_u32": S["scc"] = int(val(a[0]) >= val(a[1]))
        elif op == "s_cselect_b32": S[a[0]] = val(a[1]) if S["scc"] else val(a[2])
        elif op == "s_cbranch_scc1":
            if S["scc"]:
                t = a[0]
                cands = labels[t[:-1]]
                nxt = max(c for c in cands if c < pc) if t.endswith("b") else min(c for c in cands if c > pc)
        elif op == "v_add_u32": V[a[0]] = ("addr", val(a[1]))
        elif op == "v_xor_b32": V[a[0]] = ("addr^64", V[a[2]][1])
        elif op == "s_waitcnt":
            vm = re.search(r"vmcnt\((\d+)\)", l); lg = re.search(r"lgkmcnt\((\d+)\)", l)
            events.append(("wait", (int(vm.group(1)) if vm else None, int(lg.group(1)) if lg else None)))
        elif op == "s_barrier": events.append(("barrier", None))
        elif op == "buffer_load_dwordx4":
            rs, so = a[1], a[2].split()[0]
            if last_m0_write is None or last_m0_write != max(i for i, e in enumerate(events) if e[0] in ("m0",)): raise ProtocolError("M0 bookkeeping")
            if prog[pc - 1].startswith("s_add_u32 m0"): raise ProtocolError(f"pc {pc}: piece directly behind its M0 write (needs one wait state)")
            events.append(("dma", dict(voff=a[0], rs=rs, soff=S[so], m0=S["m0"], pc=pc)))
        elif op == "global_load_dwordx2": events.append(("vmem", None))
        elif op == "buffer_store_dwordx4":
            events.append(("store", dict(reg=a[0], voff=a[1], rs=a[2], soff=S[a[3].split()[0]], off=int(re.search(r"offset:(\d+)", l).group(1)), nt=" nt" in l)))
        elif op == "s_lshl_b32": S[a[0]] = (val(a[1]) << val(a[2])) & 0xffffffff
        elif op == "ds_read_b128":
            off = int(re.search(r"offset:(\d+)", l).group(1)); base = a[1].split()[0]
            events.append(("read", dict(dst=a[0], kind=V[base][0], slot=V[base][1], off=off, reg=base)))
        elif op == "v_mfma_f32_16x16x32_f16": events.append(("mfma", dict(acc=a[0], b=a[1], a=a[2])))
        else: raise ProtocolError(f"unknown instruction: {l}")
        pc = nxt
    return events, S


def check(gen: Gen, nt: int, wr0=0, rda0=0, rdb0=0x8000, **kw):
    """raises ProtocolError; returns the ring state the statement leaves"""
    row16a, row16b, wbase = kw.get("row16a", 16 * 768 * 2), kw.get("row16b", 16 * 1024 * 2), kw.get("wbase", 8192)
    ev, S = simulate(gen, nt, wr0, rda0, rdb0, row16a, row16b, wbase)
    cold = gen.cold
    early = gen.b1_early and not cold
    # ---- expected half-tile stream of this statement
    if gen.b1: stream = [("B", 1, True)]
    elif cold: stream = [("A", 0, True), ("B", 0, True), ("A", 1, True)]
    else:
        # B(1) A(2) B(2) A(3) ... B(nt-1) A'(0) B'(0) A'(1); with b1_early B(1) was requested by the previous tile's epilogue
        stream = ([] if early else [("B", 1, False)]) + [("A", 2, False)] + \
                 [x for t in range(2, nt) for x in ([("B", t, False)] + ([("A", t + 1, False)] if t + 1 < nt else []))] + [("A", 0, True), ("B", 0, True), ("A", 1, True)]
    dmas = [e[1] for e in ev if e[0] == "dma"]
    if len(dmas) != 8 * len(stream): raise ProtocolError(f"{len(dmas)} pieces, expected {8 * len(stream)}")
    slot = wr0
    slot_of = {}                                             # (op, t, next) -> ring position
    for n, (op, t, nx) in enumerate(stream):
        for p in range(8):
            d = dmas[8 * n + p]
            exp = dict(voff=f"voff{op.lower()}{p & 1}", rs=("rsa" if op == "A" else "rsb") + ("2" if nx else ""), soff=t * 128 + (p >> 1) * (row16a if op == "A" else row16b),
                       m0=slot + wbase + p * 1024)
            got = {k: d[k] for k in exp}
            if got != exp: raise ProtocolError(f"piece {p} of {op}{'′' if nx else ''}({t}): {got} != {exp}")
        slot_of[(op, t, nx)] = slot
        slot = (slot + 0x8000) % 0x28000
    if S["wr"] != slot: raise ProtocolError("write position returned")
    if cold: return S["wr"], S["rda"], S["rdb"]
    # what the statement found in the ring: A(0), B(0), A(1) (and B(1)) in the slots before wr0
    back = 0x8000 if early else 0
    if early: slot_of[("B", 1, False)] = (wr0 - 0x8000) % 0x28000
    slot_of[("A", 1, False)] = (wr0 - back - 0x8000) % 0x28000
    slot_of[("B", 0, False)] = (wr0 - back - 0x10000) % 0x28000
    slot_of[("A", 0, False)] = (wr0 - back - 0x18000) % 0x28000
    if (rda0, rdb0) != (slot_of[("A", 0, False)], slot_of[("B", 0, False)]): raise ProtocolError("entry read positions do not match the write position")
    # ---- walk the events: ordering rules
    # vector-memory ops in issue order: the half-tile of a piece, None for anything else.  At entry the previous statement's A(1) pieces may still be in flight, with the
    # epilogue's output stores behind them
    issued = [] if cold else [("A", 1, False)] * 8 + [None] * 24 + ([("B", 1, False)] * 8 if early else []) + [None] * 8      # (... + the epilogue's last eight stores)
    landed_visible = {("A", 0, False), ("B", 0, False)}      # published by the previous statement + the caller's barrier
    waited = {("A", 0, False), ("B", 0, False)}              # this wave's pieces have landed (counted wait), not yet published by a barrier
    reads_of = {}                                            # slot -> state of its last read: "pending" (issued), "done" (behind lgkmcnt(0)), "free" (behind a barrier after that)
    occupant = {v: k for k, v in slot_of.items() if not k[2]}
    occupant = {slot_of[k]: k for k in [("A", 0, False), ("B", 0, False), ("A", 1, False)] + ([("B", 1, False)] if early else [])}
    ndma = 0
    frag_src = {}                                            # fragment register -> (op, t, ks, idx)
    mfma_seen = {}
    reads_in_step = []
    stores = [e[1] for e in ev if e[0] == "store"]
    if stores: raise ProtocolError("stores inside the K-loop statement")
    for kind, p in ev:
        if kind in ("vmem", "store"): issued.append(None)
        elif kind == "dma":
            h = stream[ndma // 8]; ndma += 1
            s = slot_of[h]
            st = reads_of.get(s, "free")
            if st != "free": raise ProtocolError(f"WAR: piece of {h} into slot {s:#x} whose last reads are '{st}' (no lgkmcnt(0) + barrier behind them)")
            occupant[s] = h
            issued.append(h)
        elif kind == "wait":
            vm, lg = p
            if vm is not None:
                done = issued[:len(issued) - vm] if vm else issued
                for h in set(x for x in done if x is not None):
                    if sum(1 for x in done if x == h) == 8: waited.add(h)          # all eight of this wave's pieces of the half-tile
            if lg == 0:
                for s in reads_of:
                    if reads_of[s] == "pending": reads_of[s] = "done"
        elif kind == "barrier":
            # a half-tile is published when EVERY piece of it has been waited for (all eight of this wave's)
            for h in list(waited):
                n_issued = sum(1 for x in issued if x == h)
                if h in (("A", 0, False), ("B", 0, False)) or n_issued == 8: landed_visible.add(h)
            for s in reads_of:
                if reads_of[s] == "done": reads_of[s] = "free"
        elif kind == "read":
            s = p["slot"]
            h = occupant.get(s)
            if h is None or h[2]: raise ProtocolError(f"read of slot {s:#x} which holds {h}")
            if h not in landed_visible: raise ProtocolError(f"RAW: read of {h} (slot {s:#x}) before a covering vmcnt + barrier")
            # all its pieces waited? (for the ones this statement issued)
            reads_of[s] = "pending"
            ks = 0 if p["kind"] == "addr" else 1
            idx, rem = divmod(p["off"], 2048)
            if rem or idx > 7: raise ProtocolError("fragment offset")
            frag_src[p["dst"]] = (h[0], h[1], ks, idx)
        elif kind == "mfma":
            m = re.match(r"acc(\d)_(\d)", p["acc"]); i, j = int(m.group(1)), int(m.group(2))
            fa, fb = frag_src.get(p["a"]), frag_src.get(p["b"])
            if fa is None or fb is None: raise ProtocolError("MFMA on a fragment register never read")
            if fa[0] != "A" or fb[0] != "B" or fa[3] != i or fb[3] != j or fa[1:3] != fb[1:3]: raise ProtocolError(f"MFMA {p}: fragments {fa} {fb}")
            step = fa[1] * 2 + fa[2]
            prev = mfma_seen.get((i, j), -1)
            if step != prev + 1: raise ProtocolError(f"accumulator ({i},{j}): k-step {step} after {prev}")
            mfma_seen[(i, j)] = step
    if any(v != 2 * nt - 1 for v in mfma_seen.values()) or len(mfma_seen) != 64: raise ProtocolError("not every accumulator saw every k-step")
    # a fragment read must have completed (lgkmcnt) before the MFMA that uses it: checked coarsely — every MFMA block is preceded by an lgkmcnt(0) after its reads
    # ---- exit state: A'(0), B'(0) waited for; read positions = their slots
    if ("A", 0, True) not in waited or ("B", 0, True) not in waited: raise ProtocolError("the next tile's K-tile 0 has not been waited for at the end")
    if (S["rda"], S["rdb"]) != (slot_of[("A", 0, True)], slot_of[("B", 0, True)]): raise ProtocolError("read positions returned")
    return S["wr"], S["rda"], S["rdb"]


def check_mfma_wait(gen: Gen, nt: int):
    """every fragment register is read (ds_read) and then waited for (lgkmcnt(0)) before the first MFMA that uses it, and not overwritten by a later read before its
    last use in that k-step (the double buffer)"""
    ev, _ = simulate(gen, nt)
    pending, ready = {}, {}
    for kind, p in ev:
        if kind == "read": pending[p["dst"]] = True; ready.pop(p["dst"], None)
        elif kind == "wait" and p[1] == 0:
            for r in pending: ready[r] = True
            pending = {}
        elif kind == "mfma":
            for r in (p["a"], p["b"]):
                if r in pending: raise ProtocolError(f"MFMA reads {r} while its ds_read is outstanding / it is being refilled")
                if r not in ready: raise ProtocolError(f"MFMA reads {r} never loaded")


def check_all(variants=None, nts=(3, 4, 5, 6, 7, 12, 13, 48)):
    variants = VARIANTS if variants is None else variants
    n = 0
    cold = Gen(cold=True)
    for wr0 in range(0, 0x28000, 0x8000):
        wr, rda, rdb = check(cold, 0, wr0=wr0)
    for v, kw in variants.items():
        for nt in nts:
            g = Gen(**kw)
            check_mfma_wait(g, nt)
            wr, rda, rdb = check(Gen(cold=True), 0, wr0=0)
            # chain five statements: the ring phase advances by 2 nt mod 5 from tile to tile
            for _ in range(5):
                if kw.get("b1_early", False): wr, _, _ = check(Gen(b1=True), 0, wr0=wr)          # out of the previous tile's epilogue (the first tile's: behind the cold prefetch)
                wr, rda, rdb = check(Gen(**kw), nt, wr0=wr, rda0=rda, rdb0=rdb)
                n += 1
    return n


if __name__ == "__main__":
    print("statements checked:", check_all())
