"""Dense geometry sweep: for each (n, storage, R) time every feasible waves-per-replica against the
engine's own choice.  One process, integer +-1 couplings (look-ahead eligible where the chunks per
wave allow).  usage: geometry_sweep.py [small|large]"""
import os, sys, time
import numpy as np
import torch
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

mode = sys.argv[1] if len(sys.argv) > 1 else "small"
if mode == "small":
    cases = [(n, st, R) for R in (1024, 8192) for st in ("f32", "i8") for n in (256, 1024, 2048, 4096, 8192)
             if not (R == 8192 and n > 2048)]
else:
    cases = [(n, st, R) for st in ("f32", "i8") for n in (6000, 10000, 16384, 24000, 32768) for R in (256, 1024, 4096)
             if not (R == 4096 and n > 16384)]
for n, st, R in cases:
    g = torch.Generator(device="cuda").manual_seed(n)
    J = (torch.randint(0, 2, (n, n), generator=g, device="cuda") * 2 - 1).float().triu(1)
    J = J + J.T
    h = np.zeros(n, np.float32)
    per_sweep = R * n * n * (4 if st == "f32" else 1) / 6e12          # seconds at ~6 TB/s
    sweeps = int(min(200, max(2, 0.05 / per_sweep)))
    epc = 256 if st == "f32" else 1024
    chunks = -(-n // epc)
    res = {}
    with sg.AnnealEngine(0) as e:
        e.set_tuning(waves_per_replica=0, sweeps_per_launch=sweeps)
        e.set_dense(J, h, storage=st)
        ms, d = run(e, R, sweeps)
        auto = [t.split("=")[1] for t in d.split() if t.startswith(("waves_per_replica=", "chunks_per_wave=", "look_ahead="))]
        res["auto"] = R * n * sweeps / (ms * 1e-3)
        for w in range(1, 17):
            if w > chunks and w > 1:
                continue
            if -(-chunks // w) > 10:
                continue
            e.set_tuning(waves_per_replica=w, sweeps_per_launch=sweeps)
            e.set_dense(J, h, storage=st)
            ms, d = run(e, R, sweeps)
            res[w] = R * n * sweeps / (ms * 1e-3)
    del J
    best = max((v, k) for k, v in res.items() if k != "auto")
    print(f"n={n:5d} {st} R={R:5d} C={chunks:3d} auto[W={auto[0]} cpw={auto[1]} look={auto[2]}]={res['auto']:.3g} ({res['auto'] / best[0]:.2f} of best W={best[1]} {best[0]:.3g})  "
          + " ".join(f"{k}:{v:.3g}" for k, v in res.items() if k != "auto"), flush=True)
