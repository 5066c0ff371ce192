"""BASELINE configs[1]'s own instance (C2b: 100 agents x 100 tasks assignment, 10 000 spins, 1024 replicas) handed
over as a DENSE matrix, three ways: the dense int8 streaming kernel (SGA_NO_SPARSE_ROUTE=1), the cached-local-
field sweep (field cache on), and what the engine does by itself -- 198 of 10 000 couplings per row are non-zero,
so sga_set_dense keeps the matrix as CSR and the sweeps work on four updates per step."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spin_glass_anneal_rl_amd as sg  # noqa: E402
from spin_glass_anneal_rl_amd import encoders as enc  # noqa: E402
from spin_glass_anneal_rl_amd.engine import last_kernel  # noqa: E402

b = enc.assignment_ising(100, 100, weight=100.0)
J, h = torch.from_numpy(b.to_dense()).cuda(), b.fields()
n, R = 10000, int(os.environ.get("R", 1024))
temps = np.asarray(sg.temperature_ladder(R, 1.0, 400.0))
res = {}
for mode in ("dense", "cached", "routed"):
    os.environ.pop("SGA_NO_SPARSE_ROUTE", None)
    if mode == "dense":
        os.environ["SGA_NO_SPARSE_ROUTE"] = "1"
    with sg.AnnealEngine(0) as e:
        e.set_field_cache("on" if mode == "cached" else "off")
        e.set_dense(J, h)
        e.init_replicas(R, seed=77)
        e.set_ladder(temps)
        e.sweep(3)
        e.enable_timing(True)
        e.kernel_time(reset=True)
        a0 = e.stats()[0].sum()
        e.sweep(10)
        launches, ms = e.kernel_time(reset=True)
        acc = (e.stats()[0].sum() - a0) / (10.0 * n * R)
        res[mode] = (e.spins().copy(), e.energies().copy())
        print(f"{mode:7s} {ms / 10:8.3f} ms/sweep  {R * n * 10 / (ms * 1e-3):.3e} attempts/s  acceptance {acc:.3f}  {last_kernel()}", flush=True)
assert all(np.array_equal(res["dense"][0], res[m][0]) and np.array_equal(res["dense"][1], res[m][1]) for m in res), "chains differ"
print("identical spins and energies in all three")
