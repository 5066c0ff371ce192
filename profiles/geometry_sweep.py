"""Dense geometry sweep: for each (n, storage, R) time every feasible waves-per-replica against the
engine's own choice.  One process, integer +-1 couplings (look-ahead eligible where the chunks per
wave allow)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spin_glass_anneal_rl_amd as sg

def run(e, R, sweeps):
    e.init_replicas(R, seed=1)
    e.set_temperatures(np.geomspace(10.0, 0.1, R))
    e.sweep(1)
    e.enable_timing(True); e.kernel_time()
    e.sweep(sweeps)
    e.energies()
    launches, ms = e.kernel_time()
    e.enable_timing(False)
    return ms, e.describe()

cases = [(n, st, R) for R in (1024, 8192) for st in ("f32", "i8") for n in (256, 1024, 2048, 4096, 8192)]
for n, st, R in cases:
    if R == 8192 and n > 2048:
        continue
    rng = np.random.RandomState(n)
    J = np.triu(rng.randint(0, 2, (n, n)) * 2 - 1, 1).astype(np.float32)
    J = J + J.T
    sweeps = max(2, int(2e9 / (R * n * n)) if st == "f32" else int(8e9 / (R * n * n)))
    sweeps = min(sweeps, 200)
    epc = 256 if st == "f32" else 1024
    chunks = -(-n // epc)
    res = {}
    with sg.AnnealEngine(0) as e:
        e.set_tuning(waves_per_replica=0, sweeps_per_launch=sweeps)
        e.set_dense(J, np.zeros(n, np.float32), storage=st)
        ms, d = run(e, R, sweeps)
        auto = [t for t in d.split() if t.startswith("waves_per_replica=") or t.startswith("chunks_per_wave=") or t.startswith("look_ahead=")]
        res["auto"] = R * n * sweeps / (ms * 1e-3)
        for w in (1, 2, 3, 4, 6, 8, 12, 16):
            if w > chunks and w > 1:
                continue
            if -(-chunks // w) > 10:
                continue
            e.set_tuning(waves_per_replica=w, sweeps_per_launch=sweeps)
            e.set_dense(J, np.zeros(n, np.float32), storage=st)
            ms, d = run(e, R, sweeps)
            res[w] = R * n * sweeps / (ms * 1e-3)
    best = max((v, k) for k, v in res.items() if k != "auto")
    print(f"n={n:5d} {st} R={R:5d} auto[{' '.join(auto)}]={res['auto']:.3g}  best W={best[1]} {best[0]:.3g}  "
          + " ".join(f"W{k}:{v:.3g}" for k, v in res.items() if k != "auto"), flush=True)
