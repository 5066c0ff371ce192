"""C3 (10 000 spins, degree ~32, 4096 replicas): the sweep kernel's time when EVERY replica sits at one temperature --
which regime paces the mixed launch of the bench ladder (10 -> 0.1)?  (profiles/r05_experiments.md)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import spin_glass_anneal_rl_amd as sg  # noqa: E402

n, R = 10000, 4096
csr = bench.make_sparse_instance(n, 16, 3)
with sg.AnnealEngine(0) as e:
    e.set_tuning(sweeps_per_launch=1)
    e.set_csr(*csr, np.zeros(n, np.float32))
    e.set_field_cache("off")
    e.init_replicas(R, seed=42)
    for label, temps in [("ladder 10 -> 0.1", bench.geometric_ladder(R))] + [(f"all at T = {t:g}", np.full(R, t)) for t in (10.0, 3.0, 1.5, 1.0, 0.5, 0.1)]:
        e.init_replicas(R, seed=42)
        e.set_temperatures(temps)
        e.sweep(15)
        a0 = e.stats()[0].sum()
        e.enable_timing(True)
        e.kernel_time(reset=True)
        e.sweep(10)
        launches, ms = e.kernel_time(reset=True)
        e.enable_timing(False)
        acc = (e.stats()[0].sum() - a0) / (R * n * 10.0)
        print(f"{label:22s} {ms / launches:7.3f} ms per sweep  {R * n / (ms / launches) * 1e3:.3e} attempts/s  acceptance {acc:.3f}  {e.last_kernel()[:60]}", flush=True)
