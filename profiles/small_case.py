"""One small dense case for instruction-mix counters: n spins x R replicas x S sweeps through the
engine directly (int8 storage, one launch of S sweeps)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spin_glass_anneal_rl_amd as sg

n, R, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.RandomState(0)
J = np.triu(rng.randint(0, 2, (n, n)) * 2 - 1, 1).astype(np.float32)
J = J + J.T
with sg.AnnealEngine(0) as e:
    e.set_tuning(waves_per_replica=0, sweeps_per_launch=S)
    e.set_dense(J, np.zeros(n, np.float32), storage="i8")
    e.init_replicas(R, seed=1)
    e.set_temperatures(np.geomspace(10.0, 0.1, R))
    e.sweep(2)
    if len(sys.argv) > 4 and sys.argv[4] == "tune":
        e.autotune()
    e.enable_timing(True)
    t = time.time()
    e.sweep(S)
    e.energies()
    dt = time.time() - t
    launches, ms = e.kernel_time()
    print(e.describe())
    print(f"{n} spins x {R} replicas x {S} sweeps: {dt:.4f} s wall, kernel {ms:.2f} ms in {launches} launch(es), "
          f"{R * n * S / (ms * 1e-3):.4g} attempts/s, {ms * 1e-3 / (n * S) * 1e9:.0f} ns per update per replica")
