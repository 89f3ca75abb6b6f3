"""The training oracle (oracle/train_oracle.py) against the reference's own training run (tests/golden/train_*.npz,
produced by tests/golden/make_golden.py from /root/reference): episode sampler, per-episode losses, first-step
gradients, AdamW updates and the trained parameters."""
import numpy as np
import pytest
import torch

from conftest import golden
from golden.spec import TRAIN, train_inputs
from oracle import train_oracle as to


def _load(name):
    g = golden("train_" + name)
    names = [str(n) for n in g["names"]]
    init = {n: torch.from_numpy(g["init__" + n]) for n in names}
    return g, names, init


def _adapter_sd(init):
    return {k: v for k, v in init.items() if k not in ("visual", "textual")}


@pytest.mark.parametrize("name", list(TRAIN))
def test_episode_sampler_matches_reference(name):
    g, _, _ = _load(name)
    N, K = int(g["meta"][0]), int(g["meta"][1])
    epochs = TRAIN[name][11]
    rng = np.random.RandomState(1)
    labels, sizes = [], []
    for _ in range(epochs):
        for _, q_idx, q_lab in to.sample_epoch(N, K, rng):
            labels.extend(q_lab)
            sizes.append(len(q_lab))
            assert all(i // K == l for i, l in zip(q_idx, q_lab))
    assert sizes == list(g["ep_sizes"])
    assert labels == list(g["ep_labels"].astype(int))


@pytest.mark.parametrize("name", list(TRAIN))
def test_training_run_matches_reference(name):
    g, names, init = _load(name)
    split, cfg = train_inputs(name)
    N, K = int(g["meta"][0]), int(g["meta"][1])
    torch.manual_seed(0)
    tr = to.Trainer(cfg, split.visual_memory_keys, split.textual_memory_bank, _adapter_sd(init), cfg["alpha"], cfg["beta"])
    assert torch.equal(tr.visual.data, init["visual"])
    rng = np.random.RandomState(1)
    ep = 0
    cur = {"visual": tr.visual, "textual": tr.textual, **tr.adapter}
    for _ in range(cfg["train_epoch"]):
        for _, q_idx, q_lab in to.sample_epoch(N, K, rng):
            m, loss, terms, grads = tr.step(q_idx, q_lab)
            assert m == g["ep_matches"][ep]
            assert abs(loss - g["ep_loss"][ep]) <= 2e-6 * max(1.0, abs(g["ep_loss"][ep]))
            if "L1" in terms:
                assert abs(terms["L1"] - g["ep_l1"][ep]) <= 2e-6 * max(1.0, abs(g["ep_l1"][ep]))
            if "L2" in terms:
                assert abs(terms["L2"] - g["ep_l2"][ep]) <= 2e-6 and abs(terms["L3"] - g["ep_l3"][ep]) <= 2e-6
            if ep < 3:
                for n in names:
                    key = f"grad{ep}__{n}"
                    if key in g:
                        assert torch.equal(grads[n], torch.from_numpy(g[key])), (ep, n)
                    else:
                        assert grads[n] is None
                    assert torch.equal(cur[n].data, torch.from_numpy(g[f"after{ep}__{n}"])), (ep, n)
            ep += 1
        tr.sched.step()
    assert ep == int(g["n_episodes"])
    for n in names:
        assert torch.equal(cur[n].data, torch.from_numpy(g["final__" + n])), n
