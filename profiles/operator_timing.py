"""Per-call cost of the drop-in operators (CUDAKernelManager API, one replica per call)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spin_glass_anneal_rl_amd as sg
for n in (64, 1000, 10000):
    rng = np.random.RandomState(n)
    J = np.triu(rng.randint(0, 2, (n, n)) * 2 - 1, 1).astype(np.float32); J = J + J.T
    Jt, h = torch.from_numpy(J).cuda(), torch.zeros(n).cuda()
    spins = torch.from_numpy((rng.randint(0, 2, n) * 2 - 1).astype(np.float32)).cuda()
    km = sg.CUDAKernelManager(torch.device("cuda:0"))
    km.metropolis_update_optimized(spins, Jt, h, 1.0, 1); km.compute_energy_optimized(spins, Jt, h)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20):
        km.metropolis_update_optimized(spins, Jt, h, 1.0, 1)
    torch.cuda.synchronize(); t1 = (time.perf_counter() - t) / 20
    t = time.perf_counter()
    for _ in range(20):
        km.compute_energy_optimized(spins, Jt, h)
    torch.cuda.synchronize(); t2 = (time.perf_counter() - t) / 20
    print(f"n={n}: metropolis_update_optimized {t1 * 1e3:.3f} ms/call (one sweep), compute_energy_optimized {t2 * 1e3:.3f} ms/call")
