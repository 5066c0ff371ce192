"""Sparse problems of other shapes through the several-updates-per-step CSR form (sweep_csr_rows.hip) against
the one-update-at-a-time form (SGA_CSR_PAIR_AHEAD=0 | 4 | 8 in the environment): a 3-D +-J Edwards-Anderson
lattice (degree 6), a random 3-regular-like graph, a degree-16 graph; 4096 replicas, ladder 3 -> 0.3."""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spin_glass_anneal_rl_amd as sg  # noqa: E402
from spin_glass_anneal_rl_amd.engine import last_kernel  # noqa: E402


def lattice3d(L, seed):
    rng = np.random.RandomState(seed)
    idx = np.arange(L ** 3).reshape(L, L, L)
    rows, cols, vals = [], [], []
    for ax in range(3):
        nb = np.roll(idx, -1, axis=ax)
        v = rng.randint(0, 2, idx.size) * 2.0 - 1.0
        rows += [idx.ravel(), nb.ravel()]
        cols += [nb.ravel(), idx.ravel()]
        vals += [v, v]
    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(L ** 3,) * 2).tocsr()
    A.sort_indices()
    return A


def random_graph(n, half_degree, seed):
    rng = np.random.RandomState(seed)
    rows = np.repeat(np.arange(n), half_degree)
    cols = rng.randint(0, n, rows.size)
    keep = rows != cols
    lo, hi = np.minimum(rows[keep], cols[keep]), np.maximum(rows[keep], cols[keep])
    up = sp.coo_matrix((np.ones(lo.size), (lo, hi)), shape=(n, n)).tocsr()
    up.data[:] = rng.randint(0, 2, up.nnz) * 2.0 - 1.0
    A = (up + up.T).tocsr()
    A.sort_indices()
    return A


def gaussian(A, seed):
    """the same graph with symmetric Gaussian couplings (no accept table: fp64 row sums in the canonical order)"""
    rng = np.random.RandomState(seed)
    U = sp.triu(A, 1).tocoo()
    v = rng.randn(U.nnz)
    B = sp.coo_matrix((np.concatenate([v, v]), (np.concatenate([U.row, U.col]), np.concatenate([U.col, U.row]))), shape=A.shape).tocsr()
    B.sort_indices()
    return B


R = int(os.environ.get("R", 4096))
cases = [("3-D EA lattice L=22 (degree 6)", lattice3d(22, 1)),
         ("3-D Gaussian EA lattice L=22", gaussian(lattice3d(22, 1), 5)),
         ("Gaussian random graph, degree ~32", gaussian(random_graph(10000, 16, 3), 6)), ("random graph, degree ~4", random_graph(10000, 2, 2)),
         ("random graph, degree ~16", random_graph(10000, 8, 3)), ("random graph, degree ~32 (C3)", random_graph(10000, 16, 3))]
for name, A in cases:
    n = A.shape[0]
    with sg.AnnealEngine(0) as e:
        e.set_csr(A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float32), np.zeros(n, np.float32))
        e.init_replicas(R, seed=11)
        e.set_ladder(np.geomspace(3.0, 0.3, R))
        e.sweep(5)
        e.enable_timing(True)
        e.kernel_time(reset=True)
        a0 = e.stats()[0].sum()
        e.sweep(20)
        launches, ms = e.kernel_time(reset=True)
        acc = (e.stats()[0].sum() - a0) / (20.0 * n * R)
        print(f"{name:36s} max row {int(np.diff(A.indptr).max()):3d}  {ms / 20:7.3f} ms/sweep  "
              f"{R * n * 20 / (ms * 1e-3):.3e} attempts/s  acceptance {acc:.3f}  {last_kernel()}", flush=True)
