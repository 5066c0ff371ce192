"""Round 4, decision data: how many proposals each REPLICA accepts per sweep on bench.py's own ladders.

The cached-local-field sweep costs ~1 us of a replica's serial chain per ACCEPTED proposal and a launch ends
with its hottest replica, so what decides (a) how much the C2a variant can gain from a shorter chain and
(b) whether a row-on-accept form pays for the sparse configurations (C4, C5 at 100 cities) is the
per-replica distribution of accepts, not the mean.  Prints mean / median / p90 / max accepts per replica
and sweep for windows of sweeps, and the acceptance by ladder slot (hot end first).

    python profiles/r04_acceptance.py [c2a] [c4] [c5] [c3]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spin_glass_anneal_rl_amd as sg  # noqa: E402
from spin_glass_anneal_rl_amd import encoders as enc  # noqa: E402

dev = torch.device("cuda", 0)
which = sys.argv[1:] or ["c2a", "c4", "c5", "c3"]


def windows(e, n, R, n_ladders, plan, exchange_every=10):
    done = 0
    for lo, hi in plan:
        while done < lo:
            e.sweep(1)
            done += 1
            if done % exchange_every == 0:
                e.exchange(count=False)
        a0 = e.stats()[0].copy()
        k = hi - lo
        for _ in range(k):
            e.sweep(1)
            done += 1
            if done % exchange_every == 0:
                e.exchange(count=False)
        per = (e.stats()[0] - a0) / k           # accepts per replica and sweep
        slots = e.slot_map()                    # slot -> replica
        L = R // n_ladders
        by_slot = per[slots].reshape(n_ladders, L).mean(0)
        q = np.percentile(per, [50, 90, 99])
        print(f"  sweeps {lo:4d}..{hi:4d}: accepts/replica/sweep mean {per.mean():9.2f} median {q[0]:9.2f} p90 {q[1]:9.2f} "
              f"p99 {q[2]:9.2f} max {per.max():9.2f}  = acceptance mean {per.mean() / n:.4%} max {per.max() / n:.4%}")
        idx = np.unique(np.linspace(0, L - 1, 9).astype(int))
        print("     by ladder slot (0 = hottest): " + "  ".join(f"[{i}] {by_slot[i] / n:.3%}" for i in idx), flush=True)


if "c2a" in which:
    n, R = 10000, 1024
    J = bench.make_sk_instance(n, 2, dev)
    with sg.AnnealEngine(0) as e:
        e.set_field_cache("on")
        e.set_dense(J, torch.zeros(n, device=dev), storage="auto")
        e.init_replicas(R, seed=42)
        e.set_ladder(bench.geometric_ladder(R))
        print(f"C2a: {e.describe()}")
        windows(e, n, R, 1, [(0, 1), (1, 3), (3, 5), (5, 25), (25, 50), (50, 100), (100, 120), (200, 300)])
    del J

if "c3" in which:
    n, R = 10000, 4096
    csr = bench.make_sparse_instance(n, 16, 3)
    with sg.AnnealEngine(0) as e:
        e.set_csr(*csr, np.zeros(n, np.float32))
        e.init_replicas(R, seed=42)
        e.set_ladder(bench.geometric_ladder(R))
        print(f"C3: {e.describe()}")
        windows(e, n, R, 1, [(0, 5), (5, 25), (100, 120)])

if "c4" in which:
    R = 1024
    bld = enc.scheduling_ising(np.full(500, 1.0), n_agents=1, time_horizon=100.0, time_discretization=100,
                               objective="total_time", penalty_weights={"assignment": 100.0, "capacity": 50.0})
    n = bld.n
    csr = bld.to_csr()
    vals = np.unique(csr[2])
    hh = bld.fields()
    print(f"C4: n = {n}, nnz = {len(csr[1])}, distinct J values {vals[:8]}, h values {np.unique(hh)[:8]}, "
          f"max row |J| sum + |h| = {max(np.abs(csr[2][csr[0][i]:csr[0][i + 1]]).sum() + abs(hh[i]) for i in range(0, n, 997)):.1f}")
    for rank in (0, 3, 7):  # ranks of the 8192-replica ladder over 8 GPUs
        full = bench.geometric_ladder(8192, 500.0, 5.0)
        with sg.AnnealEngine(0) as e:
            e.set_csr(*csr, hh)
            e.init_replicas(R, seed=42, R_global=8192, replica0=rank * R)
            e.set_temperatures(full[rank * R:(rank + 1) * R])
            print(f" rank {rank} of 8 (T {full[rank * R]:.1f} .. {full[(rank + 1) * R - 1]:.1f}): {e.describe()}")
            done = 0
            for lo, hi in [(0, 5), (5, 25), (100, 120)]:
                e.sweep(lo - done)
                a0 = e.stats()[0].copy()
                e.sweep(hi - lo)
                done = hi
                per = (e.stats()[0] - a0) / (hi - lo)
                q = np.percentile(per, [50, 90])
                print(f"  sweeps {lo:4d}..{hi:4d}: accepts/replica/sweep mean {per.mean():9.2f} median {q[0]:9.2f} "
                      f"p90 {q[1]:9.2f} max {per.max():9.2f} = acceptance mean {per.mean() / n:.4%} max {per.max() / n:.4%}",
                      flush=True)
    # the one-GPU bench line: the whole ladder 500 -> 5 on 1024 replicas
    with sg.AnnealEngine(0) as e:
        e.set_csr(*csr, hh)
        e.init_replicas(R, seed=42)
        e.set_ladder(bench.geometric_ladder(R, 500.0, 5.0))
        print(f" bench.py --workload c4 (ladder 500 -> 5 on one GPU): {e.describe()}")
        windows(e, n, R, 1, [(0, 5), (5, 25), (100, 120)])

if "c5" in which:
    cities, R, n_ladders = 100, 2048, 32
    rs = np.random.RandomState(5)
    xy = rs.rand(cities, 2) * 100.0
    dmat = np.hypot(xy[:, None, 0] - xy[None, :, 0], xy[:, None, 1] - xy[None, :, 1])
    tsp = enc.tsp_csr(dmat, city_visit=200.0, position_fill=200.0, device=dev)
    n = cities * cities
    with sg.AnnealEngine(0) as e:
        e.set_csr(tsp[0], tsp[1], tsp[2], tsp[3])
        e.init_replicas(R, seed=42)
        e.set_ladder(np.tile(bench.geometric_ladder(R // n_ladders, 200.0, 2.0), n_ladders), n_ladders)
        print(f"C5 (100 cities): {e.describe()}")
        windows(e, n, R, n_ladders, [(0, 5), (5, 25), (100, 120)])
