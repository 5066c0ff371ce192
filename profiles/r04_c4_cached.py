"""Round 4: BASELINE configs[3] (C4: 50 000-spin scheduling instance, 1024 replicas = one rank's share, ladder 500 -> 5)
with the cached-local-field sweep over CSR couplings (csrc/sweep_clf_csr.hip) against the row-per-proposal kernel;
also C2b (assignment 100 x 100) as CSR and C3.  Kernel ms per sweep after 20 warm-up sweeps, 10 sweeps per launch."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spin_glass_anneal_rl_amd as sg  # noqa: E402
from spin_glass_anneal_rl_amd import encoders as enc  # noqa: E402
from spin_glass_anneal_rl_amd.engine import last_kernel  # noqa: E402

dev = torch.device("cuda", 0)
which = sys.argv[1:] or ["c4", "c2b", "c3"]


def run(name, setter, n, R, t_hot, t_cold, n_ladders=1):
    ref = None
    for mode in ("off", "on", "auto"):
        with sg.AnnealEngine(0) as e:
            e.set_field_cache(mode)
            setter(e)
            t0 = time.perf_counter()
            e.init_replicas(R, seed=42)
            e.set_ladder(np.tile(bench.geometric_ladder(R // n_ladders, t_hot, t_cold), n_ladders), n_ladders)
            for _ in range(2):
                e.sweep(10)
                e.exchange(count=False)
            e.energies()
            t_setup = time.perf_counter() - t0
            e.enable_timing(True)
            e.kernel_time(reset=True)
            a0 = e.stats()[0].copy()
            t1 = time.perf_counter()
            for _ in range(3):
                e.sweep(10)
                e.exchange(count=False)
            en = e.energies()
            dt = time.perf_counter() - t1
            launches, ms = e.kernel_time(reset=True)
            per = (e.stats()[0] - a0) / 30.0 / n
            if ref is None:
                ref = en
            tracked = e.energies()
            e.recompute_energies()
            print(f"[{name}] cache {mode:4s}: {dt / 30 * 1e3:8.3f} ms/sweep wall, {ms / 30:8.3f} kernel ({R * n * 30 / dt:.3e} attempts/s) "
                  f"acceptance mean {per.mean():.3%} max {per.max():.3%} same chain {np.array_equal(ref, en)} tracked==recomputed "
                  f"{np.array_equal(tracked, e.energies())} (first 20 sweeps incl. seeding {t_setup:.2f} s)\n      {last_kernel()[:150]}\n      "
                  f"{e.describe()[-160:]}", flush=True)


if "c4" in which:
    bld = enc.scheduling_ising(np.full(500, 1.0), n_agents=1, time_horizon=100.0, time_discretization=100,
                               objective="total_time", penalty_weights={"assignment": 100.0, "capacity": 50.0})
    csr, hh = bld.to_csr(), bld.fields()
    run("C4", lambda e: e.set_csr(*csr, hh), bld.n, 1024, 500.0, 5.0)
if "c2b" in which:
    b = enc.assignment_ising(100, 100, weight=100.0)
    csr2, h2 = b.to_csr(), b.fields()
    run("C2b as CSR", lambda e: e.set_csr(*csr2, h2), 10000, 1024, 400.0, 1.0)
    Jd = torch.from_numpy(b.to_dense()).to(dev)
    run("C2b dense", lambda e: e.set_dense(Jd, h2), 10000, 1024, 400.0, 1.0)
if "c3" in which:
    csr3 = bench.make_sparse_instance(10000, 16, 3)
    run("C3", lambda e: e.set_csr(*csr3, np.zeros(10000, np.float32)), 10000, 4096, 10.0, 0.1)
