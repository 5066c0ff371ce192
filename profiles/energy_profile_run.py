"""What profiles/collect_r03.sh traces for the all-replica field pass (energies of 1024 replicas from one
pass over J on the matrix cores): N spins, fp32 or int8 couplings; init_replicas evaluates the energies
once, then REPS more evaluations.  SGA_NO_MFMA_ENERGY=1 runs the per-replica kernel instead (the A/B)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spin_glass_anneal_rl_amd as sg  # noqa: E402

n, R, reps = int(os.environ.get("N", 10000)), int(os.environ.get("R", 1024)), int(os.environ.get("REPS", 5))
dev = torch.device("cuda", 0)
J = bench.make_sk_instance(n, 2, dev)
with sg.AnnealEngine(0) as e:
    e.set_dense(J, torch.zeros(n, device=dev), storage=os.environ.get("STORAGE", "f32"))
    del J
    e.init_replicas(R, seed=42)
    first = e.energies()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        e.recompute_energies()
    again = e.energies()
    dt = (time.perf_counter() - t0) / reps
    assert (first == again).all()
    print(f"{e.describe()}\nenergies of {R} replicas: {dt * 1e3:.3f} ms per evaluation (host wall, {reps} reps)")
