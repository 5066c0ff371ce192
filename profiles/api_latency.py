"""Latency of every engine call at n = 10^4 (dense +-1), for 1 and 1024 replicas."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spin_glass_anneal_rl_amd as sg
n = 10000
g = torch.Generator(device="cuda").manual_seed(1)
J = (torch.randint(0, 2, (n, n), generator=g, device="cuda") * 2 - 1).float().triu(1); J = J + J.T
h = np.zeros(n, np.float32)
def timed(label, fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    print(f"   {label:34s} {(time.perf_counter() - t) / reps * 1e3:9.3f} ms")
    return out
for R in (1, 1024):
    print(f"R = {R}")
    with sg.AnnealEngine(0) as e:
        timed("set_dense (f32, 400 MB on device)", lambda: e.set_dense(J, h, storage="f32"), 2)
        timed("init_replicas", lambda: e.init_replicas(R, seed=1), 2)
        temps = np.geomspace(10, 0.1, R) if R > 1 else np.asarray([1.0])
        timed("set_ladder", lambda: e.set_ladder(temps))
        timed("sweep(1)", lambda: e.sweep(1))
        timed("sweep(1, energy_trace)", lambda: e.sweep(1, energy_trace=True))
        timed("exchange()", lambda: e.exchange())
        timed("exchange(count=False)", lambda: e.exchange(count=False))
        timed("energies()", lambda: e.energies())
        timed("recompute_energies()", lambda: e.recompute_energies())
        timed("spins()", lambda: e.spins())
        timed("spins(0)", lambda: e.spins(0))
        timed("best()", lambda: e.best())
        timed("stats()", lambda: e.stats())
        timed("local_fields(0, [5])", lambda: e.local_fields(0, [5]))
        timed("local_fields(0, all)", lambda: e.local_fields(0, np.arange(n, dtype=np.int32)))
        timed("flip(0, 7)", lambda: e.flip(0, 7))
        timed("update(0, 7, T=1)", lambda: e.update(0, 7, 1.0, 0.5))
        blob = timed("export_state()", lambda: e.export_state())
        timed("import_state()", lambda: e.import_state(blob))
        timed("autotune()", lambda: e.autotune(), 1)
