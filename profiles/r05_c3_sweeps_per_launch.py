"""C3 on the bench ladder: one sweep per launch (what bench.py times) against 2 / 5 / 10 sweeps per launch -- a launch ends
with its hottest replicas' waves running alone (the cold ones are done early), and that tail is paid once per launch."""
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
    e.set_csr(*csr, np.zeros(n, np.float32))
    e.set_field_cache("off")
    for spl in (1, 2, 5, 10, 20):
        e.set_tuning(sweeps_per_launch=spl)
        e.init_replicas(R, seed=42)
        e.set_temperatures(bench.geometric_ladder(R))
        e.sweep(20)
        e.enable_timing(True)
        e.kernel_time(reset=True)
        e.sweep(20)
        launches, ms = e.kernel_time(reset=True)
        e.enable_timing(False)
        print(f"{spl:3d} sweep(s) per launch: {launches:2d} launches, {ms / 20:7.3f} ms per sweep  {R * n * 20 / ms * 1e3:.3e} attempts/s", flush=True)
