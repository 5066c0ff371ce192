"""What profiles/collect_r03.sh traces for the cached-local-field variant: the C2a instance (10 000-spin
dense +-1 SK, 1024 replicas, ladder 10 -> 0.1, storage picked by the engine), W warm-up sweeps, then K
sweeps with an exchange round every 10 -- the sweeps bench.py's `variants.cached_local_fields` times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spin_glass_anneal_rl_amd as sg  # noqa: E402

W, K = int(os.environ.get("WARMUP", 3)), int(os.environ.get("STEPS", 20))
n, R = 10000, 1024
dev = torch.device("cuda", 0)
J = bench.make_sk_instance(n, 2, dev)
with sg.AnnealEngine(0) as e:
    e.set_field_cache("on")
    e.set_tuning(sweeps_per_launch=1)
    e.set_dense(J, torch.zeros(n, device=dev), storage=os.environ.get("STORAGE", "auto"))
    e.init_replicas(R, seed=42)
    e.set_ladder(bench.geometric_ladder(R))
    for k in range(1, W + K + 1):
        e.sweep(1)
        if k % 10 == 0:
            e.exchange(count=False)
    acc = e.stats()[0]
    print(e.describe())
    print(f"acceptance over {W + K} sweeps: {acc.sum() / (float(R) * n * (W + K)):.5f}; "
          f"hottest replica {acc.max() / (n * (W + K)):.5f}")
