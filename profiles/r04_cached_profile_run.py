"""What profiles/collect_r04.sh traces for the dense cached-field variant (field cache ON, the engine's own storage =
int8 rows): the C2a instance, ladder 10 -> 0.1, 120 sweeps in launches of 10 with exchange rounds -- first
sweep_clfb_kernel (several accepts per round while the hottest replica accepts more than ~1 %), then sweep_clf_kernel,
at eight waves once the launch is its hottest replica's chain.  Prints the accepted rows' bytes for the byte model."""
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
    e.set_field_cache("on")
    e.set_dense(J, torch.zeros(n, device=dev), storage="auto")
    e.init_replicas(R, seed=42)
    e.set_ladder(bench.geometric_ladder(R, 10.0, 0.1))
    seen = []
    for _ in range(12):
        e.sweep(10)
        k = last_kernel().split("<")[0] + " x " + last_kernel().split(" x ")[-1]
        if not seen or seen[-1] != k:
            seen.append(k)
        e.exchange(count=False)
    acc = int(e.stats()[0].sum())
    print("kernels in order:", " -> ".join(seen))
    print(f"accepted proposals: {acc} = {acc * 10112 / 1e9:.3f} GB of int8 rows (pitch 10112 B) over 120 sweeps")
    print(e.describe())
