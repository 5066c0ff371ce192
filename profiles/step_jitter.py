"""Per-step wall time of the bench loop on a short-kernel workload (C3): where does host time go?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
import bench
import spin_glass_anneal_rl_amd as sg
from spin_glass_anneal_rl_amd.sharded import ShardedTempering
n, R = 10000, 4096
csr = bench.make_sparse_instance(n, 16, 3)
eng = sg.AnnealEngine(0)
eng.use_stream(torch.cuda.current_stream().cuda_stream)
eng.set_tuning(waves_per_replica=0, sweeps_per_launch=1)
eng.set_csr(*csr, torch.zeros(n, device="cuda"))
pt = ShardedTempering(eng, R_local=R, rank=0, world=1, seed=42, slot_temps=bench.geometric_ladder(R, 10.0, 0.1),
                      n_ladders=1, dist=None, device=torch.device("cuda"))
for rep in range(3):
    eng.enable_timing(rep % 2 == 0)
    ts = []
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    for i in range(20):
        t = time.perf_counter()
        pt.sweep(1)
        if (i + 1) % 10 == 0:
            pt.exchange()
        ts.append((time.perf_counter() - t) * 1e3)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t_all) * 1e3
    print(f"timing={'on' if rep % 2 == 0 else 'off'}: total {tot:.1f} ms for 20 steps; enqueue times ms: " + " ".join(f"{x:.2f}" for x in ts))
    eng.kernel_time()
