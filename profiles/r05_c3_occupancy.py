"""C3's several-updates-per-step kernel against the number of resident waves: 10 000 spins, degree ~32, every replica cold
(T = 0.5: the kernel's base cost), R = 512 ... 8192 replicas = 0.5 ... 8 waves per SIMD wanted (4 per SIMD fit: LDS).
Chain bound: the time per sweep barely moves while waves are added; issue bound: it doubles with them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import spin_glass_anneal_rl_amd as sg  # noqa: E402

n = 10000
csr = bench.make_sparse_instance(n, 16, 3)
with sg.AnnealEngine(0) as e:
    e.set_tuning(sweeps_per_launch=1)
    e.set_csr(*csr, np.zeros(n, np.float32))
    e.set_field_cache("off")
    for R in (256, 512, 1024, 2048, 3072, 4096, 8192):
        e.init_replicas(R, seed=42)
        e.set_temperatures(np.full(R, 0.5))
        e.sweep(15)
        e.enable_timing(True)
        e.kernel_time(reset=True)
        e.sweep(10)
        launches, ms = e.kernel_time(reset=True)
        e.enable_timing(False)
        per = ms / launches
        print(f"R = {R:5d} ({R / 1024.0:4.2f} waves per SIMD wanted)  {per:7.3f} ms per sweep  {R * n / per * 1e3:.3e} attempts/s  "
              f"{per * 1e6 / (n / 4):7.1f} ns per step of a wave  {e.last_kernel()[-32:]}", flush=True)
