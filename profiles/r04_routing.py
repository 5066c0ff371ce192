"""Round 4: what per-replica routing of SGA_FIELD_CACHE_AUTO buys on ladders with and without a hot end.

C2a instance (10 000-spin +-1 SK, 1024 replicas), couplings as the engine stores them (bit-planes) and as int8:
field cache off / on / auto (per-replica routing: two concurrent launches) / auto with one launch for all
(option replica_routing = 0, round 3's rule), on three ladders.  Wall ms per sweep over 40 sweeps after 40
warm-up sweeps (sweep(10) + exchange, as the tempering classes drive the engine), and where the replicas ran.

    python profiles/r04_routing.py [storages=auto,i8] [ladders=cold,warm,hot]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spin_glass_anneal_rl_amd as sg  # noqa: E402
from spin_glass_anneal_rl_amd.engine import last_kernel  # noqa: E402

n, R = int(os.environ.get("N", 10000)), int(os.environ.get("R", 1024))
dev = torch.device("cuda", 0)
J = bench.make_sk_instance(n, 2, dev)
h = torch.zeros(n, device=dev)
storages = (sys.argv[1] if len(sys.argv) > 1 else "auto,i8").split(",")
ladders = {"cold": (10.0, 0.1), "warm": (100.0, 0.1), "hot": (400.0, 0.1)}
pick = (sys.argv[2] if len(sys.argv) > 2 else "cold,warm,hot").split(",")

for storage in storages:
    for lname in pick:
        t_hot, t_cold = ladders[lname]
        ref = None
        for mode in ("off", "on", "auto", "auto-one-launch"):
            with sg.AnnealEngine(0) as e:
                e.set_field_cache(mode.split("-")[0])
                if mode == "auto-one-launch":
                    e.set_option("replica_routing", 0)
                e.set_dense(J, h, storage=storage)
                e.init_replicas(R, seed=42)
                e.set_ladder(bench.geometric_ladder(R, t_hot, t_cold))
                for _ in range(4):
                    e.sweep(10)
                    e.exchange(count=False)
                a0 = e.stats()[0].copy()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(4):
                    e.sweep(10)
                    e.exchange(count=False)
                e.energies()
                dt = time.perf_counter() - t0
                per = (e.stats()[0] - a0) / 40.0 / n
                en = e.energies()
                if ref is None:
                    ref = en
                same = bool(np.array_equal(ref, en))
                print(f"[{storage:4s}] ladder {t_hot:g}->{t_cold:g} {mode:16s} {dt / 40 * 1e3:9.3f} ms/sweep "
                      f"{R * n * 40 / dt:.3e} attempts/s  acceptance mean {per.mean():.3%} max {per.max():.3%}  "
                      f"chain==off: {same}\n        {last_kernel()[:230]}\n        {e.describe()[-150:]}", flush=True)
