#!/usr/bin/env python3
"""The reference's annealing entry points on the MI355X engine: an import switch.

    # from spin_glass_rl.core.ising_model import IsingModel, IsingModelConfig
    # from spin_glass_rl.annealing.gpu_annealer import GPUAnnealer, GPUAnnealerConfig
    # from spin_glass_rl.annealing.parallel_tempering import ParallelTempering, ParallelTemperingConfig
    from spin_glass_anneal_rl_amd import ...

Needs an MI355X (there is no CPU fallback).  Run from the repository root:  python examples/hot_path_usage.py
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spin_glass_anneal_rl_amd import (GPUAnnealer, GPUAnnealerConfig, IsingModel, IsingModelConfig,  # noqa: E402
                                      ParallelTempering, ParallelTemperingConfig, SpinGlassScheduler)
from spin_glass_anneal_rl_amd import encoders  # noqa: E402


def random_pm1_model(n, seed):
    g = torch.Generator().manual_seed(seed)
    J = (torch.randint(0, 2, (n, n), generator=g) * 2 - 1).float().triu(1)
    model = IsingModel(IsingModelConfig(n_spins=n, use_sparse=False, device="cuda"))
    model.set_couplings_from_matrix(J + J.T)
    return model


def simulated_annealing():
    model = random_pm1_model(256, seed=1)
    print(f"initial energy {model.compute_energy():.1f}")
    result = GPUAnnealer(GPUAnnealerConfig(n_sweeps=2000, random_seed=7)).anneal(model)
    print(f"SA: best energy {result.best_energy:.1f} after {result.n_sweeps} sweeps in {result.total_time:.3f} s")


def parallel_tempering():
    model = random_pm1_model(64, seed=1)
    cfg = ParallelTemperingConfig(n_replicas=8, n_sweeps=1000, exchange_interval=10, random_seed=42)
    result = ParallelTempering(cfg).run(model)
    print(f"PT: best energy {result.best_energy:.1f} in {result.total_time:.3f} s")


def many_replicas():
    model = random_pm1_model(2048, seed=3)
    t = time.time()
    result = SpinGlassScheduler(device="cuda", random_seed=5).anneal(model, n_replicas=1024, n_sweeps=300)
    dt = time.time() - t
    print(f"1024 replicas x 2048 spins x 300 sweeps: best energy {result.best_energy:.1f}, "
          f"{1024 * 2048 * 300 / dt:.3g} spin-flip attempts/s")


def travelling_salesman(n_cities=12):
    rs = np.random.RandomState(0)
    xy = rs.rand(n_cities, 2)
    d = np.hypot(xy[:, None, 0] - xy[None, :, 0], xy[:, None, 1] - xy[None, :, 1])
    builder = encoders.tsp_ising(d, city_visit=4.0, position_fill=4.0)
    model = builder.to_model(sparse=False)
    result = SpinGlassScheduler(device="cuda", random_seed=1).anneal(
        model, n_replicas=256, n_sweeps=2000, beta_min=0.5, beta_max=50.0)
    x = (result.best_configuration.numpy().reshape(n_cities, n_cities) > 0)
    feasible = (x.sum(0) == 1).all() and (x.sum(1) == 1).all()
    length = result.best_energy + builder.penalty_energy_offset()
    print(f"TSP, {n_cities} cities: objective + penalties = {length:.3f}, one-hot constraints satisfied: {bool(feasible)}")
    if feasible:
        tour = [int(np.argmax(x[:, p])) for p in range(n_cities)]
        print("   tour:", tour)


def travelling_salesman_without_storing_couplings(n_cities=200, n_replicas=512, n_sweeps=200):
    """The same QUBO at a size where the couplings themselves become the cost (200 cities: 40 000
    spins, 32 M couplings; 1000 cities: 10^6 spins, 32 GB): the engine keeps distances + penalty
    weights and rebuilds a row's sum from the TSP structure (AnnealEngine.set_tsp), bit-identical
    to the stored-coupling chain."""
    from spin_glass_anneal_rl_amd import AnnealEngine
    rs = np.random.RandomState(1)
    xy = rs.rand(n_cities, 2) * 100.0
    d = np.hypot(xy[:, None, 0] - xy[None, :, 0], xy[:, None, 1] - xy[None, :, 1])
    dist, w_city, w_pos, h, constant = encoders.tsp_structure(d, city_visit=200.0, position_fill=200.0)
    n_ladders = 8
    ladder = np.tile(np.geomspace(200.0, 2.0, n_replicas // n_ladders), n_ladders)
    with AnnealEngine(0) as eng:
        eng.set_tsp(dist, w_city, w_pos, h)
        eng.init_replicas(n_replicas, seed=3)
        eng.set_ladder(ladder, n_ladders)
        t = time.time()
        for _ in range(n_sweeps // 10):
            eng.sweep(10)
            eng.exchange(count=False)
        best, spins, _ = eng.best()
        dt = time.time() - t
    print(f"TSP, {n_cities} cities without stored couplings: best objective + penalties = {best + constant:.1f}, "
          f"{n_replicas * n_cities ** 2 * n_sweeps / dt:.3g} spin-flip attempts/s ({eng_desc(n_cities)})")


def eng_desc(n_cities):
    return f"{n_cities ** 2} spins, {4 * (n_cities - 1) * n_cities ** 2 / 1e6:.0f} M couplings never materialised"


if __name__ == "__main__":
    if not torch.cuda.is_available():
        raise SystemExit("this example needs an MI355X")
    simulated_annealing()
    parallel_tempering()
    many_replicas()
    travelling_salesman()
    travelling_salesman_without_storing_couplings()
