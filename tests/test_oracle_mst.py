"""The MST oracles (literal Python model and C restatement) against the golden
vectors produced by running the reference's maxspantree.py."""
import os

import numpy as np
import pytest

from oracle import c_mst, mst_oracle


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "mst_golden.npz"))


def test_python_oracle_matches_reference_outputs(gold):
    for name in gold["names"]:
        G = gold[f"G__{name}"]
        want = [tuple(int(v) for v in e) for e in gold[f"E__{name}"]]
        assert mst_oracle.mst_edges(G) == want, name


def test_c_oracle_matches_reference_outputs(gold):
    for name in gold["names"]:
        G = gold[f"G__{name}"]
        want = gold[f"E__{name}"]
        edges, count = c_mst.mst_full(G[None])
        assert int(count[0]) == len(want), name
        assert np.array_equal(edges[0, : len(want)], want), name
        assert np.all(edges[0, len(want):] == -1), name


def test_triu_path_matches_reference_outputs(gold):
    for name in gold["triu_names"]:
        row = gold[f"R__{name}"]
        n = gold[f"F__{name}"].shape[0]
        full = mst_oracle.full_from_triu(row, n)
        assert np.array_equal(full, gold[f"F__{name}"], equal_nan=True), name   # +inf diagonals expand to NaN
        want = gold[f"E__{name}"]
        assert mst_oracle.mst_edges_from_triu(row, n) == [tuple(int(v) for v in e) for e in want], name
        edges, count = c_mst.mst_triu(row[None], n)
        assert int(count[0]) == len(want) and np.array_equal(edges[0, : len(want)], want), name


def test_c_oracle_equals_python_model_on_random_and_extreme_inputs():
    rng = np.random.default_rng(5)
    specials = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, -1e19, -9.3e18, 1e-45, 0.5, 0.25], dtype=np.float32)
    for trial in range(200):
        n = int(rng.integers(1, 13))
        G = rng.uniform(-1, 1, size=(n, n)).astype(np.float32)
        mask = rng.uniform(size=(n, n)) < 0.35
        G[mask] = rng.choice(specials, size=int(mask.sum()))
        d = rng.uniform(0.2, 1.0, size=n).astype(np.float32)
        G[np.arange(n), np.arange(n)] = d
        want = mst_oracle.mst_edges(G)
        edges, count = c_mst.mst_full(G[None])
        got = [tuple(int(v) for v in e) for e in edges[0, : count[0]]]
        assert got == want, (trial, G)


def test_precision_recall_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "graphvae_golden.npz"))
    for G, gold_m, (r, p) in zip(g["stats__G"], g["stats__gold"], g["stats__recall_precision"]):
        gold_edges = [(int(a), int(b)) for a, b in zip(*np.nonzero(gold_m))]
        pred = mst_oracle.mst_edges(G)
        rr, pp = mst_oracle.edge_precision_recall(pred, gold_edges)
        assert rr == pytest.approx(r, abs=0) and pp == pytest.approx(p, abs=0)
