import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
import spin_glass_anneal_rl_amd as sg
n, R = 10000, 1024
dev = torch.device("cuda", 0)
J = bench.make_sk_instance(n, 2, dev)
h = torch.zeros(n, device=dev)
temps = bench.geometric_ladder(R)
def check(e, tag):
    t = e.energies(); s0 = e.spins().copy(); e.recompute_energies(); r = e.energies()
    bad = np.flatnonzero(t != r)
    print(tag, "tracked != recomputed for", bad.size, "replicas", bad[:8], (t - r)[bad[:8]], flush=True)
with sg.AnnealEngine(0) as e:
    e.set_field_cache("on")
    e.set_tuning(sweeps_per_launch=1)
    e.set_dense(J, h, storage="i8")
    e.init_replicas(R, seed=42)
    e.set_ladder(temps)
    for block in (1, 1, 1, 2, 5, 10, 10, 30, 40):
        e.sweep(block)
    t_on = e.energies().copy()
    e.set_field_cache("off")
    e.sweep(2)
    check(e, "after 100 cached + 2 plain sweeps:")
with sg.AnnealEngine(0) as e:
    e.set_field_cache("on")
    e.set_tuning(sweeps_per_launch=1)
    e.set_dense(J, h, storage="i8")
    e.init_replicas(R, seed=42)
    e.set_ladder(temps)
    for block in (1, 1, 1, 2, 5, 10, 10, 30, 40):
        e.sweep(block)
        check(e, f"after block {block}:")
        e.set_field_cache("on")
