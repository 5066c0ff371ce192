"""Dense sweep with real-valued (Gaussian) couplings: the fp64-accumulating general arithmetic."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spin_glass_anneal_rl_amd as sg
n, R = int(sys.argv[1]), int(sys.argv[2])
S = int(sys.argv[3]) if len(sys.argv) > 3 else 3
g = torch.Generator(device="cuda").manual_seed(1)
J = torch.randn(n, n, generator=g, device="cuda").triu(1)
J = J + J.T
with sg.AnnealEngine(0) as e:
    e.set_dense(J, np.zeros(n, np.float32))
    e.init_replicas(R, seed=1)
    e.set_temperatures(np.geomspace(10.0 * np.sqrt(n), 0.1 * np.sqrt(n), R))
    e.sweep(1)
    for tune in (False, True):
        if tune:
            e.autotune()
        e.enable_timing(True); e.kernel_time()
        e.sweep(S)
        e.energies()
        launches, ms = e.kernel_time()
        per = ms / S
        print(e.describe())
        print(f"{'autotuned' if tune else 'heuristic'}: {per:.2f} ms/sweep, {R * n / (per * 1e-3):.4g} attempts/s, "
              f"{R * n * n * 4 / (per * 1e-3) / 1e12:.2f} TB/s")
