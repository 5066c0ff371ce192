"""Round 4: what the first sweeps of SGA_FIELD_CACHE_AUTO cost on the C2a instance (ladder 10 -> 0.1) per coupling
storage, against ON: AUTO walks its first sweeps on the row-per-proposal kernels unless the break-even acceptance of the
storage says the cached-field kernel wins at any plausible acceptance.  Kernel ms, sweeps 0..4 and 4..24."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spin_glass_anneal_rl_amd as sg  # noqa: E402
from spin_glass_anneal_rl_amd.engine import last_kernel  # noqa: E402

n, R = 10000, 1024
dev = torch.device("cuda", 0)
J = bench.make_sk_instance(n, 2, dev)
h = torch.zeros(n, device=dev)
ref = None
for storage in ("t2", "i8", "f32"):
    for cache in ("on", "auto"):
        with sg.AnnealEngine(0) as e:
            e.set_field_cache(cache)
            e.set_dense(J, h, storage=storage)
            e.init_replicas(R, seed=42)
            e.set_ladder(bench.geometric_ladder(R, 10.0, 0.1))
            e.enable_timing(True)
            out = []
            for k in (4, 20):
                e.kernel_time(reset=True)
                for _ in range(k // 4):
                    e.sweep(4)
                out.append(e.kernel_time(reset=True)[1])
            en = e.energies()
            ref = en if ref is None else ref
            print(f"{storage:4s} {cache:5s} sweeps 0..4: {out[0]:8.2f} ms   4..24: {out[1]:8.2f} ms   same chain: {np.array_equal(ref, en)}  [{last_kernel()[:50]}]", flush=True)
