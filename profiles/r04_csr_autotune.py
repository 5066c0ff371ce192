"""Round 4: what sga_autotune picks for the CSR workloads of bench.py (C3, C4, C5 at 100 cities) against the forms
sga_init_replicas chooses by its thresholds: kernel ms per sweep before and after."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spin_glass_anneal_rl_amd as sg  # noqa: E402
from spin_glass_anneal_rl_amd.engine import last_kernel  # noqa: E402


class A:  # the bench arguments build_workload reads
    spins, replicas, cities, implicit, storage, workload = 10000, 0, 100, False, "f32", ""


dev = torch.device("cuda", 0)
for name in (sys.argv[1:] or ["c3", "c4", "c5"]):
    a = A()
    a.workload = name
    wl = bench.build_workload(name, a, dev, 1)
    n, R, nl = wl["n"], wl["R"], wl["n_ladders"]
    with sg.AnnealEngine(0) as e:
        wl["load"](e)
        e.init_replicas(R, seed=42)
        e.set_ladder(np.tile(bench.geometric_ladder(R // nl, wl["t_hot"], wl["t_cold"]), nl), nl)
        e.enable_timing(True)

        def per_sweep(k=10):
            e.sweep(2)
            e.kernel_time(reset=True)
            e.sweep(k)
            return e.kernel_time(reset=True)[1] / k

        t0, k0, d0 = per_sweep(), last_kernel(), e.describe()
        en0 = e.energies()
        best = e.autotune()
        assert np.array_equal(en0, e.energies())
        t1 = per_sweep()
        print(f"[{name}] heuristic {t0:.4f} ms/sweep  {k0}\n      autotuned {t1:.4f} ms/sweep (trial best {best:.4f})  {last_kernel()}\n      {e.describe()}", flush=True)
