"""CPU check of the generated four-wave K-loop (tools/gen_gemm4w.py -> csrc/pclip_gemm4w_loop.inc): tools/sim_gemm4w.py interprets one wave's instruction
stream and checks the ring protocol event by event — piece order / addresses, RAW (counted vmcnt + barrier before every fragment read), WAR (lgkmcnt(0) +
barrier before a slot is refilled), the M0 wait state, fragment / accumulator pairing and k order — for every tail variant, every ring phase across chained
output tiles, the race-stress build and the schedule experiments; mutations of the stream must be caught; the committed .inc is what the generator emits."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_gemm4w  # noqa: E402
import sim_gemm4w  # noqa: E402


def test_protocol_holds_for_every_variant_and_tile_count():
    assert sim_gemm4w.check_all(nts=(3, 4, 5, 6, 12, 48)) == len(gen_gemm4w.VARIANTS) * 6 * 5


class _Mutant(gen_gemm4w.Gen):
    def __init__(self, fn, **kw):
        super().__init__(**kw)
        self.fn = fn

    def generate(self):
        return self.fn(super().generate())


@pytest.mark.parametrize("name,fn", [
    ("weaker counted wait", lambda L: [l.replace("vmcnt(8) lgkmcnt(0)", "vmcnt(9) lgkmcnt(0)") for l in L]),
    ("barrier dropped", lambda L: [l for i, l in enumerate(L) if not (l == "s_barrier" and L[:i].count("s_barrier") == 1)]),
    ("read wait dropped", lambda L: [l.replace("vmcnt(8) lgkmcnt(0)", "vmcnt(8)") for l in L]),
    ("piece on the wrong descriptor", lambda L: [l.replace("%s offen lds" % "", "") if False else l for l in L][:-1] + ["s_nop 0"]),
])
def test_mutations_are_caught(name, fn):
    if name == "piece on the wrong descriptor":
        g0 = gen_gemm4w.Gen()
        rsa, rsb = g0.r("rsa"), g0.r("rsb")
        def fn(L):                                           # noqa: F811 — first A piece of the statement reads through B's descriptor
            out, done = [], False
            for l in L:
                if not done and l.startswith("buffer_load") and f", {rsa}," in l:
                    l, done = l.replace(f", {rsa},", f", {rsb},"), True
                out.append(l)
            return out
    with pytest.raises(sim_gemm4w.ProtocolError):
        wr, rda, rdb = sim_gemm4w.check(gen_gemm4w.Gen(cold=True), 0)
        sim_gemm4w.check(_Mutant(fn), 6, wr0=wr, rda0=rda, rdb0=rdb)
        sim_gemm4w.check_mfma_wait(_Mutant(fn), 6)


def test_committed_inc_is_what_the_generator_emits(tmp_path):
    out = tmp_path / "loop.inc"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_gemm4w.py"), "-o", str(out)], check=True, capture_output=True)
    assert out.read_text() == open(os.path.join(ROOT, "proto-clip_amd", "csrc", "pclip_gemm4w_loop.inc")).read()
