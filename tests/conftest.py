import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---- observed-vs-bound ledger: every tolerance-graded assertion records what it actually measured; the summary is printed
# at the end of the run (also under -q) and — with PCLIP_OBSERVED_JSON=1 — written to gpurun_out/observed_tolerances.json, from
# where it is committed as profiles/rNN_observed_tolerances.json.  Bounds are kept at <= 2x the largest value observed on MI355X.
_OBSERVED = {}


def observe(key, value, bound):
    value, bound = float(value), float(bound)
    cur = _OBSERVED.get(key)
    if cur is None or value > cur[0]:
        _OBSERVED[key] = (value, bound, (cur[2] + 1) if cur else 1)
    else:
        _OBSERVED[key] = (cur[0], cur[1], cur[2] + 1)
    return value


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if not _OBSERVED:
        return
    tr = terminalreporter
    tr.write_line("")
    tr.write_line("observed maxima of tolerance-graded checks (value / bound, #checks):")
    for k in sorted(_OBSERVED):
        v, b, n = _OBSERVED[k]
        tr.write_line(f"  {k:<58s} {v:10.3e} / {b:9.3e}  ({v / b if b else 0:5.2f} of the bound, {n} checks)")
    if not os.environ.get("PCLIP_OBSERVED_JSON"):          # opt-in (tools/gpu_*.sh set it): a plain pytest run writes nothing into the repo
        return
    try:
        import json
        out = os.path.join(REPO, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "observed_tolerances.json"), "w") as f:
            json.dump({k: dict(observed=v, bound=b, checks=n) for k, (v, b, n) in sorted(_OBSERVED.items())}, f, indent=1)
    except OSError:
        pass


def golden(name):
    path = os.path.join(GOLDEN, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"fixture {name}.npz not generated")
    return np.load(path, allow_pickle=False)


sys.path.insert(0, GOLDEN)
from spec import ENCODERS, FEWSHOT, ODD, RESNETS, SMALL, TINY, fewshot_inputs, randomize_adapter_   # noqa: E402,F401


def adapter_sd(g):
    return {str(k): torch.from_numpy(g["adapter__" + str(k)]) for k in g["adapter_keys"]}


F16_ULP = 2.0 ** -10     # relative spacing of fp16


def ulp_diff(a16, b16):
    """max |a-b| in fp16 ulps, the ulp taken at max(|a|, |b|, rms of the row): an upstream 1-ulp flip of a
    row norm moves every element by one ulp of ITS size, so elements that cancel to ~0 in a mean are
    judged at the scale of the vector, not of themselves."""
    a, b = a16.float().cpu(), b16.float().cpu()
    rms = b.pow(2).mean(-1, keepdim=True).sqrt()
    scale = torch.maximum(torch.maximum(a.abs(), b.abs()), rms).clamp_min(2.0 ** -14)
    return ((a - b).abs() / (scale * F16_ULP)).max().item()


def assert_adapter_close(y16, ref16, tag="adapter"):
    """Adapter outputs pass two or three whole-tensor LayerNorms in fp16.  A single 1-ulp flip in an
    intermediate (fp32 mean/variance summation order — it happens for ~1e-4 of elements even between two
    CPU formulas of the same LayerNorm) is amplified by the next normalisation.  Bounds = 2x the largest value observed on
    MI355X over every adapter test (profiles/r02_observed_tolerances.json), see ADAPTER_BOUNDS."""
    a, b = y16.float().cpu(), ref16.float().cpu()
    rel = (a - b).norm(dim=-1) / b.norm(dim=-1)
    rms = b.pow(2).mean(-1, keepdim=True).sqrt()
    b_max, b_mean, b_elem = ADAPTER_BOUNDS
    v_max = observe(f"{tag}: row rel-L2 max", rel.max().item(), b_max)
    v_mean = observe(f"{tag}: row rel-L2 mean", rel.mean().item(), b_mean)
    v_elem = observe(f"{tag}: worst element / row rms", ((a - b).abs() / rms).max().item(), b_elem)
    assert v_max <= b_max, v_max
    assert v_mean <= b_mean, v_mean
    assert v_elem < b_elem, v_elem


# Bounds = at most 2x the largest value observed on MI355X (profiles/r02_observed_tolerances.json; round 1 had 1e-2 / 1e-3 / 0.05,
# 3 / 0.5 queries).  The worst adapter case is conv-2x against the reference's own rows (C6: conv3's output has a tiny variance,
# so one 1-ulp flip upstream moves a whole row by 2.4e-3); against the oracle the GPU rows agree to 3e-4.
# (row relative L2 max, row relative L2 mean, worst element as a fraction of the row rms) — observed 2.44e-3 / 2.5e-4 / 1.9e-2
ADAPTER_BOUNDS = (5e-3, 5e-4, 0.04)
# post-adapter (alpha, beta) grids: (max queries of difference at any grid point, mean queries) — observed 2 / 0.094
GRID_BOUNDS = (3.0, 0.2)
# adapter-free grids: (max queries of difference at any grid point, fraction of grid points that differ at all) — observed 1 / 0.0125
GRID_EXACT_BOUNDS = (1.0, 0.02)


def assert_grid_close(acc, ref_acc, n_queries, exact=False, tag="grid"):
    """(alpha, beta) accuracy grids against the reference's.  Without an adapter in the path (exact=True) the counts agree
    except where a query's top-2 probabilities tie to ~1e-7 (fp32 summation order of the 512-long dot products differs between
    the MFMA tile and the CPU BLAS).  Behind an adapter, the fp16 LayerNorm noise described above perturbs a few adapted
    queries, and a near-tied query then flips at some grid points (the reference's own CPU/GPU builds would differ the same
    way).  Every call records the observed differences (in queries)."""
    acc, ref_acc = np.asarray(acc, dtype=np.float64), np.asarray(ref_acc, dtype=np.float64)
    d = np.abs(acc - ref_acc) * n_queries
    if exact:
        b_max, b_frac = GRID_EXACT_BOUNDS
        v_max = observe(f"{tag} (no adapter): max queries differing at a grid point", d.max(), b_max)
        v_frac = observe(f"{tag} (no adapter): fraction of grid points differing", (d > 1e-3).mean(), b_frac)
        assert v_max <= b_max + 1e-3, v_max
        assert v_frac <= b_frac, v_frac
        return
    b_max, b_mean = GRID_BOUNDS
    v_max = observe(f"{tag} (behind adapter): max queries differing at a grid point", d.max(), b_max)
    v_mean = observe(f"{tag} (behind adapter): mean queries differing", d.mean(), b_mean)
    assert v_max <= b_max + 1e-3, v_max
    assert v_mean <= b_mean, v_mean
