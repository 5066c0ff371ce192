"""What profiles/collect_r04.sh traces for the per-replica routing of SGA_FIELD_CACHE_AUTO: the C2a instance with int8
couplings on a ladder with a hot end (400 -> 0.1), 40 warm-up sweeps (the routing settles), then 30 sweeps as mixed
launches -- sweep_clf_kernel and sweep_dense_kernel side by side on two streams."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spin_glass_anneal_rl_amd as sg  # noqa: E402
from spin_glass_anneal_rl_amd.engine import last_kernel  # noqa: E402

n, R = 10000, 1024
dev = torch.device("cuda", 0)
J = bench.make_sk_instance(n, 2, dev)
with sg.AnnealEngine(0) as e:
    e.set_field_cache("auto")
    e.set_dense(J, torch.zeros(n, device=dev), storage="i8")
    e.init_replicas(R, seed=42)
    e.set_ladder(bench.geometric_ladder(R, 400.0, 0.1))
    for _ in range(7):
        e.sweep(10)
        e.exchange(count=False)
    e.energies()
    print(last_kernel())
    print(e.describe())
