"""Stress of the several-updates-per-step forms against the one-update forms of the same engine (GPU vs GPU, so the
cases can be larger and longer than the oracle allows): random sparse problems (n = 20 ... 6000, degree 2 ... 250,
integer / half-integer / real-valued, int8 and bit spins, 4 | 8 rows per step) and random TSP instances in the
implicit form (5 ... 260 cities, 2 | 4 | 8 updates per step), hot and cold ladders, 10 ... 40 sweeps in launches of
random length; spins, accept counts and energies must be identical.
usage: multi_update_stress.py <seconds> [seed]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spin_glass_anneal_rl_amd as sg  # noqa: E402
from spin_glass_anneal_rl_amd import encoders as enc  # noqa: E402
from spin_glass_anneal_rl_amd.engine import last_kernel  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
cases = fails = 0


def run(setup, env, R, temps, blocks, seed):
    for k in ("SGA_CSR_PAIR_AHEAD", "SGA_TSP_PARALLEL", "SGA_FORCE_CSR_BIG"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with sg.AnnealEngine(0) as e:
        setup(e)
        e.init_replicas(R, seed=seed)
        e.set_temperatures(temps)
        for b in blocks:
            e.sweep(b)
        return e.spins().copy(), e.stats()[0].copy(), e.energies().copy(), e.best(with_spins=False)[0], last_kernel()


while time.time() < t_end:
    seed = int(rng.randint(1, 1 << 30))
    R = int(rng.choice([3, 8, 33, 130]))
    hot = rng.rand() < 0.5
    blocks = [int(b) for b in rng.randint(1, 12, rng.randint(1, 5))]
    if rng.rand() < 0.65:
        n = int(rng.choice([20, 64, 300, 1000, 2500, 6000]))
        deg = int(rng.choice([2, 4, 6, 12, 30, 58, 110, 210]))
        kind = str(rng.choice(["int", "half", "real"]))
        rows = np.repeat(np.arange(n), max(1, deg // 2))
        cols = rng.randint(0, n, rows.size)
        keep = rows != cols
        lo, hi = np.minimum(rows[keep], cols[keep]), np.maximum(rows[keep], cols[keep])
        up = sp.coo_matrix((np.ones(lo.size), (lo, hi)), shape=(n, n)).tocsr()
        up.data[:] = rng.randn(up.nnz) if kind == "real" else rng.choice([-2.0, -1.0, 1.0, 2.0], up.nnz)
        A = (up + up.T).tocsr()
        A.sort_indices()
        longest = int(np.diff(A.indptr).max())
        if longest > (64 if kind == "real" else 256):  # (longer rows: integer problems only)
            continue
        h = (rng.randn(n) if kind == "real" else rng.randint(-2, 3, n) + (0.5 if kind == "half" else 0.0)).astype(np.float32)
        csr = (A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float32))
        temps = np.geomspace(12.0 if hot else 2.0, 0.3, R)
        big = {"SGA_FORCE_CSR_BIG": "1"} if rng.rand() < 0.5 else {}
        setup = lambda e: e.set_csr(*csr, h)  # noqa: E731
        ref = run(setup, dict(big, SGA_CSR_PAIR_AHEAD="0"), R, temps, blocks, seed)
        got = run(setup, dict(big, SGA_CSR_PAIR_AHEAD=str(rng.choice([4, 8]))), R, temps, blocks, seed)
        what = f"csr n={n} deg={deg} {kind} R={R} hot={hot} blocks={blocks} big={bool(big)} seed={seed}"
        assert "sweep_csr_rows_kernel" in got[4] and "sweep_csr_rows_kernel" not in ref[4], (what, got[4], ref[4])
    else:
        nc = int(rng.choice([5, 9, 24, 40, 70, 130, 260]))
        integer = rng.rand() < 0.5
        xy = rng.rand(nc, 2) * 100.0
        d = np.hypot(xy[:, None, 0] - xy[None, :, 0], xy[:, None, 1] - xy[None, :, 1])
        if integer:
            d = np.rint(d / 4.0) * 4.0
        d32, Aw, Bw, hh, _ = enc.tsp_structure(d, 200.0, 120.0, auto_scale=not integer)
        R = min(R, 33)
        temps = np.geomspace(600.0 if hot else 100.0, 3.0, R)
        blocks = blocks[:2] if nc >= 130 else blocks
        setup = lambda e: e.set_tsp(d32, Aw, Bw, hh)  # noqa: E731
        ref = run(setup, {"SGA_TSP_PARALLEL": "0"}, R, temps, blocks, seed)
        got = run(setup, {"SGA_TSP_PARALLEL": str(rng.choice([2, 4, 8]))}, R, temps, blocks, seed)
        what = f"tsp cities={nc} integer={integer} R={R} hot={hot} blocks={blocks} seed={seed}"
        assert "sweep_tsp_par_kernel" in got[4] and got[4] != ref[4], (what, got[4], ref[4])
    ok = np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1]) and np.array_equal(ref[2], got[2]) and ref[3] == got[3]
    cases += 1
    if not ok:
        fails += 1
        print("MISMATCH", what, "|", got[4], flush=True)
    if cases % 50 == 0:
        print(f"... {cases} cases, {fails} failures", flush=True)
print(f"{cases} cases, {fails} failures")
