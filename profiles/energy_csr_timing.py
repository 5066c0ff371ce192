"""Energies of all replicas of the CSR workloads: the one-pass kernel (csrc/fields_csr.hip) against the
per-replica kernels (SGA_NO_MFMA_ENERGY=1), host wall time per evaluation (synchronised).
usage: energy_csr_timing.py [c3] [c4] [c5] [c5_500] [c5_1000]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spin_glass_anneal_rl_amd as sg  # noqa: E402
from spin_glass_anneal_rl_amd import encoders as enc  # noqa: E402

dev = torch.device("cuda", 0)


def instance(name):
    if name == "c3":
        csr = bench.make_sparse_instance(10000, 16, 3)
        return csr, np.zeros(10000, np.float32), 4096
    if name == "c4":
        b = enc.scheduling_ising(np.full(500, 1.0), n_agents=1, time_horizon=100.0, time_discretization=100,
                                 objective="total_time", penalty_weights={"assignment": 100.0, "capacity": 50.0})
        return b.to_csr(), b.fields(), 1024
    cities = {"c5": 100, "c5_500": 500, "c5_1000": 1000}[name]
    rs = np.random.RandomState(5)
    xy = rs.rand(cities, 2) * 100.0
    d = np.hypot(xy[:, None, 0] - xy[None, :, 0], xy[:, None, 1] - xy[None, :, 1])
    t = enc.tsp_csr(d, city_visit=200.0, position_fill=200.0, device=dev)
    return (t[0], t[1], t[2]), t[3], 2048 if cities == 100 else 256


for name in (sys.argv[1:] or ["c3", "c4", "c5"]):
    csr, h, R = instance(name)
    res = {}
    for one_pass in (True, False):
        if one_pass:
            os.environ.pop("SGA_NO_MFMA_ENERGY", None)
        else:
            os.environ["SGA_NO_MFMA_ENERGY"] = "1"
        with sg.AnnealEngine(0) as e:
            e.set_csr(*csr, h)
            e.init_replicas(R, seed=42)
            first = e.energies()
            reps = 3
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                e.recompute_energies()
            again = e.energies()
            dt = (time.perf_counter() - t0) / reps
            assert np.array_equal(first, again)
            res[one_pass] = (dt, first)
            desc = e.describe()
    same = np.array_equal(res[True][1], res[False][1]) or np.allclose(res[True][1], res[False][1], rtol=1e-6)
    print(f"{name}: R={R} {desc[:60]}... one pass {res[True][0] * 1e3:.2f} ms, per replica {res[False][0] * 1e3:.2f} ms, "
          f"equal={same}", flush=True)
    del csr
    torch.cuda.empty_cache()
