"""Where the cached-field sweep pays: +-1 SK instances of n spins, 1024 replicas, a cold (glassy: T = 0.1 ..
0.001 sqrt(n)) and a hot (T = 3 .. 0.3 sqrt(n)) ladder; kernel ms per sweep with the field cache on / off
after 20 warm-up sweeps (storage picked by the engine)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spin_glass_anneal_rl_amd as sg  # noqa: E402

dev = torch.device("cuda", 0)
R = 1024
for n in (64, 256, 1024, 2048, 4096, 10000):
    J = bench.make_sk_instance(n, 2, dev)
    h = torch.zeros(n, device=dev)
    for name, (hi, lo) in (("cold", (0.1, 0.001)), ("hot", (3.0, 0.3))):
        temps = np.geomspace(hi * np.sqrt(n), lo * np.sqrt(n), R)
        out = {}
        for cache in ("on", "off"):
            with sg.AnnealEngine(0) as e:
                e.set_field_cache(cache)
                e.set_dense(J, h)
                e.init_replicas(R, seed=1)
                e.set_ladder(temps)
                e.sweep(20)
                a0 = e.stats()[0].sum()
                e.enable_timing(True)
                e.kernel_time(reset=True)
                e.sweep(10)
                _, ms = e.kernel_time(reset=True)
                out[cache] = (ms / 10, (e.stats()[0].sum() - a0) / (10.0 * R * n))
        print(f"n={n:6d} {name:4s} acceptance {out['on'][1]:.4f}: cache on {out['on'][0]:8.4f} ms/sweep, off {out['off'][0]:8.4f} "
              f"-> x{out['off'][0] / out['on'][0]:.2f}", flush=True)
