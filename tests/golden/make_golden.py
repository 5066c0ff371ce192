#!/usr/bin/env python3
"""Golden-vector generator: runs the *imported reference* and records what it did.

Run in the build container only (the reference never travels to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference \
        python3 /root/repo/tests/golden/make_golden.py --out /root/repo/tests/golden

What is recorded (data only -- inputs and expected outputs, no reference text):

* the random stream the reference consumed, per single-spin update:
  ``site`` (torch.randint(0, N, (1,)), spin_dynamics.py:69) and ``u``
  (torch.rand(1), drawn only when dE > 0, spin_dynamics.py:145-146; NaN if not drawn);
* the decision and dE it returned (spin_dynamics.py:131-152);
* per-sweep energies (spin_dynamics.py:87), histories and the AnnealingResult fields of
  GPUAnnealer.anneal (gpu_annealer.py:96-183) and ParallelTempering.run
  (parallel_tempering.py:82-144), and the PT exchange log (parallel_tempering.py:214-258).

The reference is run in its only working mode here: IsingModelConfig(use_sparse=False),
ParallelTempering n_threads=1.
"""
import argparse
import os
import sys

import numpy as np
import torch

assert not any(p.rstrip("/") == "/root/repo" for p in sys.path if p), "run from /tmp, not the repo"

from spin_glass_rl.core.ising_model import IsingModel, IsingModelConfig  # noqa: E402
from spin_glass_rl.core.spin_dynamics import SpinDynamics, UpdateRule  # noqa: E402
from spin_glass_rl.annealing.gpu_annealer import GPUAnnealer, GPUAnnealerConfig  # noqa: E402
from spin_glass_rl.annealing.parallel_tempering import (  # noqa: E402
    ParallelTempering, ParallelTemperingConfig)
from spin_glass_rl.annealing.temperature_scheduler import (  # noqa: E402
    TemperatureScheduler, ScheduleType)
from spin_glass_rl.annealing.cuda_kernels import CUDAKernelManager  # noqa: E402


# --------------------------------------------------------------------------- recording
class Recorder:
    """Wraps the RNG entry points the hot path uses and logs every draw."""

    def __init__(self):
        self.sites, self.us, self.acc, self.dE = [], [], [], []
        self._pending_u = None
        self.all_u = []   # every torch.rand(1) draw, in order (the Wolff rule draws many per update)
        self.cluster = []  # Wolff: cluster size of each update
        self.np_log = []  # ("randint", v) / ("rand", v)
        self._orig = {}

    def __enter__(self):
        self._orig["randint"] = torch.randint
        self._orig["rand"] = torch.rand
        self._orig["np_randint"] = np.random.randint
        self._orig["np_rand"] = np.random.rand
        self._orig["metro"] = SpinDynamics._metropolis_update
        self._orig["glauber"] = SpinDynamics._glauber_update
        self._orig["heat"] = SpinDynamics._heat_bath_update
        self._orig["wolff"] = SpinDynamics._wolff_update
        rec = self

        def randint(*a, **k):
            out = rec._orig["randint"](*a, **k)
            if tuple(out.shape) == (1,) and "generator" not in k:
                rec.sites.append(int(out.item()))
            return out

        def rand(*a, **k):
            out = rec._orig["rand"](*a, **k)
            if tuple(out.shape) == (1,):
                rec._pending_u = float(out.item())
                rec.all_u.append(rec._pending_u)
            return out

        def np_randint(*a, **k):
            v = rec._orig["np_randint"](*a, **k)
            rec.np_log.append(("randint", float(v)))
            return v

        def np_rand(*a, **k):
            v = rec._orig["np_rand"](*a, **k)
            rec.np_log.append(("rand", float(v)))
            return v

        def wrap(key):
            def rule(self_dyn, site):
                rec._pending_u = None
                accepted, d = rec._orig[key](self_dyn, site)
                rec.us.append(np.nan if rec._pending_u is None else rec._pending_u)
                rec.acc.append(bool(accepted))
                rec.dE.append(float(d))
                return accepted, d
            return rule

        metro = wrap("metro")

        def wolff(self_dyn, site):
            before = self_dyn.n_accepted
            accepted, d = rec._orig["wolff"](self_dyn, site)
            rec.cluster.append(int(self_dyn.n_accepted - before))
            rec.us.append(np.nan)
            rec.acc.append(bool(accepted))
            rec.dE.append(float(d))
            return accepted, d

        torch.randint = randint
        torch.rand = rand
        np.random.randint = np_randint
        np.random.rand = np_rand
        SpinDynamics._metropolis_update = metro
        SpinDynamics._glauber_update = wrap("glauber")
        SpinDynamics._heat_bath_update = wrap("heat")
        SpinDynamics._wolff_update = wolff
        return self

    def __exit__(self, *exc):
        torch.randint = self._orig["randint"]
        torch.rand = self._orig["rand"]
        np.random.randint = self._orig["np_randint"]
        np.random.rand = self._orig["np_rand"]
        SpinDynamics._metropolis_update = self._orig["metro"]
        SpinDynamics._glauber_update = self._orig["glauber"]
        SpinDynamics._heat_bath_update = self._orig["heat"]
        SpinDynamics._wolff_update = self._orig["wolff"]

    def stream(self):
        assert len(self.sites) == len(self.us) == len(self.acc)
        return dict(site=np.asarray(self.sites, np.int32), u=np.asarray(self.us, np.float32),
                    accepted=np.asarray(self.acc, np.bool_), dE=np.asarray(self.dE, np.float64))


# --------------------------------------------------------------------------- instances
def pm1_couplings(n, seed):
    """SURVEY.md 8(c) recipe: symmetric +-1, zero diagonal."""
    g = torch.Generator().manual_seed(seed)
    J = (torch.randint(0, 2, (n, n), generator=g) * 2 - 1).float()
    J = torch.triu(J, 1)
    return J + J.T


def gaussian_couplings(n, seed):
    g = torch.Generator().manual_seed(seed)
    J = torch.randn(n, n, generator=g)
    J = torch.triu(J, 1)
    return J + J.T


def dense_model(J, h=None):
    n = J.shape[0]
    m = IsingModel(IsingModelConfig(n_spins=n, use_sparse=False))
    m.set_couplings_from_matrix(J)
    if h is not None:
        m.set_external_fields(h)
    return m


def i8(t):
    return t.detach().cpu().numpy().astype(np.int8)


# --------------------------------------------------------------------------- cases
def case_sweeps(name, J, h, T, n_sweeps, seed, out, rule=UpdateRule.METROPOLIS):
    """SpinDynamics.sweep() at fixed temperature (spin_dynamics.py:73-94)."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    m = dense_model(J, h)
    s0 = i8(m.spins)
    e0 = m.compute_energy()
    energies = []
    with Recorder() as rec:
        dyn = SpinDynamics(m, T, rule)  # binds the wrapped update
        for _ in range(n_sweeps):
            energies.append(dyn.sweep())
    st = rec.stream()
    np.savez_compressed(
        os.path.join(out, name + ".npz"), kind="sweeps", rule=rule.value, J=J.numpy(),
        h=m.external_fields.numpy(), s0=s0, e0=np.float64(e0), T=np.float64(T),
        n_sweeps=np.int32(n_sweeps), sweep_energy=np.asarray(energies, np.float64),
        s_final=i8(m.spins), n_accepted=np.int64(dyn.n_accepted),
        n_rejected=np.int64(dyn.n_rejected),
        # Wolff (spin_dynamics.py:193-255): cluster size per update and every torch.rand(1) draw in order
        cluster=np.asarray(rec.cluster, np.int32), all_u=np.asarray(rec.all_u, np.float32), **st)
    print(f"{name}: updates={len(st['site'])} E0={e0} E_end={energies[-1]} acc={dyn.n_accepted}")


def case_sa(name, J, h, cfg_kwargs, model_seed, out):
    """GPUAnnealer.anneal on the CPU path (gpu_annealer.py:96-183)."""
    torch.manual_seed(model_seed)
    m = dense_model(J, h)
    s0 = i8(m.spins)
    cfg = GPUAnnealerConfig(**cfg_kwargs)
    ann = GPUAnnealer(cfg)  # seeds torch/numpy with cfg.random_seed (gpu_annealer.py:74-76)
    with Recorder() as rec:
        res = ann.anneal(m)
    st = rec.stream()
    n = J.shape[0]
    assert len(st["site"]) == res.n_sweeps * n
    # temperatures the reference used per executed sweep
    sched = TemperatureScheduler.create_schedule(cfg.schedule_type, cfg.initial_temp,
                                                 cfg.final_temp, cfg.n_sweeps,
                                                 **cfg.schedule_params)
    T = np.asarray([max(sched.get_temperature(s), 1e-10) for s in range(res.n_sweeps)])
    np.savez_compressed(
        os.path.join(out, name + ".npz"), kind="sa", J=J.numpy(), h=m.external_fields.numpy(),
        s0=s0, n_sweeps_cfg=np.int32(cfg.n_sweeps), initial_temp=np.float64(cfg.initial_temp),
        final_temp=np.float64(cfg.final_temp), schedule=cfg.schedule_type.value,
        alpha=np.float64(cfg.schedule_params.get("alpha", 0.95)),
        record_interval=np.int32(cfg.record_interval),
        energy_tolerance=np.float64(cfg.energy_tolerance),
        random_seed=np.int64(-1 if cfg.random_seed is None else cfg.random_seed),
        T_per_sweep=T, n_sweeps=np.int32(res.n_sweeps), best_energy=np.float64(res.best_energy),
        best_configuration=i8(res.best_configuration),
        energy_history=np.asarray(res.energy_history, np.float64),
        temperature_history=np.asarray(res.temperature_history, np.float64),
        acceptance_rate_history=np.asarray(res.acceptance_rate_history, np.float64),
        s_final=i8(m.spins), **st)
    print(f"{name}: n_sweeps={res.n_sweeps} best={res.best_energy} hist={len(res.energy_history)}")


def case_pt(name, J, h, cfg_kwargs, out):
    """ParallelTempering.run, n_threads=1 (parallel_tempering.py:82-144)."""
    cfg = ParallelTemperingConfig(n_threads=1, **cfg_kwargs)
    pt = ParallelTempering(cfg)  # seeds (parallel_tempering.py:52-54)
    m = dense_model(J, h)
    snap = {}
    orig_init = ParallelTempering._initialize_replicas
    orig_exch = ParallelTempering._attempt_single_exchange
    exch_log = []

    def init(self, model, rule):
        orig_init(self, model, rule)
        snap["s0"] = np.stack([i8(r.spins) for r in self.replicas])

    def exch(self, i, j):
        ei, ej = self.replicas[i].compute_energy(), self.replicas[j].compute_energy()
        before = self.exchange_accepts[min(i, j)]
        orig_exch(self, i, j)
        exch_log.append((i, j, ei, ej, int(self.exchange_accepts[min(i, j)] - before)))

    ParallelTempering._initialize_replicas = init
    ParallelTempering._attempt_single_exchange = exch
    try:
        with Recorder() as rec:
            res = pt.run(m)
    finally:
        ParallelTempering._initialize_replicas = orig_init
        ParallelTempering._attempt_single_exchange = orig_exch
    st = rec.stream()
    R, n = cfg.n_replicas, J.shape[0]
    assert len(st["site"]) == cfg.n_sweeps * R * n
    # np.random log, nearest neighbour: one randint per exchange round, one rand per attempted
    # pair.  all_pairs (parallel_tempering.py:222-232): per pair (i < j) one gate draw
    # (`rand() < 0.1`) and, behind an open gate, the exchange's own draw -- the whole stream is
    # kept (np_rand_all) and exch_u are the exchange draws in attempt order.
    starts = np.asarray([v for k, v in rec.np_log if k == "randint"], np.int32)
    all_rand = np.asarray([v for k, v in rec.np_log if k == "rand"], np.float64)
    if cfg.exchange_method == "all_pairs":
        us, k = [], 0
        while k < len(all_rand):
            if all_rand[k] < 0.1:
                us.append(all_rand[k + 1])
                k += 2
            else:
                k += 1
        us = np.asarray(us, np.float64)
    else:
        us = all_rand
    assert len(us) == len(exch_log)
    ex = np.asarray(exch_log, np.float64).reshape(-1, 5)
    np.savez_compressed(
        os.path.join(out, name + ".npz"), kind="pt", J=J.numpy(), h=m.external_fields.numpy(),
        s0=snap["s0"], n_replicas=np.int32(R), n_sweeps=np.int32(cfg.n_sweeps),
        temp_min=np.float64(cfg.temp_min), temp_max=np.float64(cfg.temp_max),
        temp_distribution=cfg.temp_distribution, exchange_interval=np.int32(cfg.exchange_interval),
        record_interval=np.int32(cfg.record_interval), random_seed=np.int64(cfg.random_seed),
        temperatures=np.asarray(pt.temperatures, np.float64),
        # stream order: sweep-major, then replica (ladder slot), then update
        site=st["site"].astype(np.uint8 if n <= 256 else np.int32), u=st["u"],
        accepted=st["accepted"],
        exch_start=starts, exch_i=ex[:, 0].astype(np.int32), exch_j=ex[:, 1].astype(np.int32),
        exch_Ei=ex[:, 2], exch_Ej=ex[:, 3], exch_accepted=ex[:, 4].astype(np.bool_), exch_u=us,
        exchange_attempts=pt.exchange_attempts, exchange_accepts=pt.exchange_accepts,
        exchange_method=cfg.exchange_method, np_rand_all=all_rand,
        energy_histories=np.asarray(pt.energy_histories, np.float64),
        best_energy=np.float64(res.best_energy), best_configuration=i8(res.best_configuration),
        acceptance_rates=np.asarray(res.acceptance_rate_history, np.float64),
        s_final=np.stack([i8(r.spins) for r in pt.replicas]))
    print(f"{name}: best={res.best_energy} accepts={pt.exchange_accepts.tolist()} "
          f"attempts={pt.exchange_attempts.tolist()} hist0={pt.energy_histories[0][:4]}")


def case_wire_formats(out):
    """Files WRITTEN BY THE REFERENCE: AnnealingResult.save (annealing/result.py:147-165) and
    IsingModel.to_dict (core/ising_model.py:213-229), so that the build's readers are pinned to
    the reference's on-disk format (and its writers to what the reference reads back)."""
    import json
    J = pm1_couplings(12, 21)
    g = torch.Generator().manual_seed(22)
    h = torch.randint(-2, 3, (12,), generator=g).float()
    torch.manual_seed(23)                # the model draws its spins from the global generator: seeded HERE, so
    np.random.seed(23)                   # that the case does not depend on what ran before it
    m = dense_model(J, h)
    ann = GPUAnnealer(GPUAnnealerConfig(n_sweeps=40, initial_temp=3.0, final_temp=0.2, record_interval=4,
                                        random_seed=5))
    res = ann.anneal(m)
    res.total_time = 0.25                # (wall time of this run: pinned, so that regenerating changes nothing)
    path = os.path.join(out, "wire_result_reference.npz")
    res.save(path)                       # the reference's own writer
    back = type(res).load(path)          # and its own reader, as a sanity check of the file
    assert back.best_energy == res.best_energy
    d = m.to_dict()                      # {"config": {...}, "spins", "couplings", "external_fields"}
    assert sorted(d) == ["config", "couplings", "external_fields", "spins"]
    np.savez_compressed(os.path.join(out, "wire_model_reference.npz"), spins=d["spins"],
                        couplings=d["couplings"], external_fields=d["external_fields"],
                        config_json=json.dumps(d["config"], sort_keys=True))
    # expected values, for the readers' checks
    np.savez_compressed(os.path.join(out, "wire_expected.npz"), J=J.numpy(), h=h.numpy(),
                        spins=i8(m.spins), best_energy=np.float64(res.best_energy),
                        best_configuration=i8(res.best_configuration),
                        energy_history=np.asarray(res.energy_history, np.float64),
                        temperature_history=np.asarray(res.temperature_history, np.float64),
                        acceptance_rate_history=np.asarray(res.acceptance_rate_history, np.float64),
                        n_sweeps=np.int32(res.n_sweeps), total_time=np.float64(res.total_time),
                        final_temperature=np.float64(res.final_temperature),
                        final_acceptance_rate=np.float64(res.final_acceptance_rate),
                        energy_std=np.float64(res.energy_std), random_seed=np.int64(res.random_seed),
                        convergence_sweep=np.int64(-1 if res.convergence_sweep is None else res.convergence_sweep))
    print("wire formats:", sorted(d.keys()), "->", path)


def case_operator(name, J, h, T, seed, out):
    """CUDAKernelManager fallbacks = what the operator API really computes
    (cuda_kernels.py:371-443): sequential-order sweep, energy, PT exchange."""
    n = J.shape[0]
    torch.manual_seed(seed)
    mgr = CUDAKernelManager(torch.device("cpu"))
    s0 = (torch.randint(0, 2, (n,)) * 2 - 1).float()
    us = []
    orig_rand = torch.rand

    def rand(*a, **k):
        out_ = orig_rand(*a, **k)
        if tuple(out_.shape) == (1,):
            us.append(float(out_.item()))
        return out_

    spins = s0.clone()
    torch.rand = rand
    try:
        new_spins, accepted, dE = mgr.metropolis_update_optimized(spins, J, h, T, n_updates=2)
    finally:
        torch.rand = orig_rand
    energy = mgr.compute_energy_optimized(new_spins, J, h)
    # PT exchange operator on 6 replicas
    R = 6
    sp = (torch.randint(0, 2, (R, n)) * 2 - 1).float()
    en = torch.tensor([mgr.compute_energy_optimized(sp[r], J, h) for r in range(R)])
    temps = torch.tensor([10.0 * (0.1 / 10.0) ** (i / (R - 1)) for i in range(R)])
    sp_in, en_in = sp.clone(), en.clone()
    pus = []

    def rand2(*a, **k):
        out_ = orig_rand(*a, **k)
        if tuple(out_.shape) == (1,):
            pus.append(float(out_.item()))
        return out_

    torch.rand = rand2
    try:
        n_ex = mgr.parallel_tempering_exchange_optimized(sp, en, temps)
    finally:
        torch.rand = orig_rand
    np.savez_compressed(
        os.path.join(out, name + ".npz"), kind="operator", J=J.numpy(), h=h.numpy(), s0=i8(s0),
        T=np.float64(T), n_updates=np.int32(2), u=np.asarray(us, np.float32),
        s_out=i8(new_spins), accepted=np.int64(accepted), energy_changes=dE.numpy(),
        energy=np.float64(energy), pt_spins_in=i8(sp_in), pt_energies_in=en_in.numpy(),
        pt_temps=temps.numpy(), pt_u=np.asarray(pus, np.float32), pt_spins_out=i8(sp),
        pt_energies_out=en.numpy(), pt_exchanges=np.int64(n_ex))
    print(f"{name}: accepted={accepted} E={energy} pt_exchanges={n_ex} n_u={len(us)}")


def case_encoders(out):
    """What the reference's encoders write into J and h (core/constraints.py:51-158,360-377;
    problems/routing.py:250-328; problems/scheduling.py:67-285), on dense models (the mode in
    which its per-element accumulation works) and on its default sparse model (where
    constraints.py:376 overwrites instead of accumulating)."""
    from spin_glass_rl.core.constraints import ConstraintEncoder
    from spin_glass_rl.problems.routing import TSPProblem, Location
    from spin_glass_rl.problems.scheduling import SchedulingProblem, Task, Agent
    from spin_glass_rl.problems import base as pbase
    d = {}

    def apply_all(model):
        enc = ConstraintEncoder(model)
        enc.add_cardinality_constraint(list(range(0, 6)), k=2, penalty_weight=8.0)
        enc.add_equality_constraint([3, 4, 7, 9], [1.0, -2.0, 0.5, 3.0], 1.5, penalty_weight=2.0)
        enc.add_cardinality_constraint(list(range(6, 12)), k=1, penalty_weight=100.0)
        enc.add_inequality_constraint([0, 11], [1.0, 1.0], 0.0, penalty_weight=3.0)
        return enc

    md = IsingModel(IsingModelConfig(n_spins=12, use_sparse=False))
    enc = apply_all(md)
    d["con_dense_J"], d["con_dense_h"] = md.couplings.numpy().copy(), md.external_fields.numpy().copy()
    g = torch.Generator().manual_seed(4)
    probe = (torch.randint(0, 2, (6, 12), generator=g) * 2 - 1).float()
    d["con_probe_spins"] = probe.numpy().astype(np.int8)
    d["con_probe_violation"] = np.asarray(
        [enc.evaluate_all_constraints(p)["total_violation"] for p in probe], np.float64)
    ms = IsingModel(IsingModelConfig(n_spins=12, use_sparse=True))
    apply_all(ms)
    d["con_sparse_J"] = ms.couplings.to_dense().numpy().copy()
    d["con_sparse_h"] = ms.external_fields.numpy().copy()

    # problem encoders, forced onto dense models (their sparse default cannot index couplings)
    orig_create = pbase.ProblemTemplate.create_ising_model

    def dense_create(self, n_spins, device="cpu"):
        model = IsingModel(IsingModelConfig(n_spins=n_spins, use_sparse=False, device=device))
        self.constraint_encoder = ConstraintEncoder(model)
        return model

    pbase.ProblemTemplate.create_ising_model = dense_create
    try:
        tsp = TSPProblem()
        xy = [(0.0, 0.0), (3.0, 4.0), (6.0, 0.0), (3.0, -2.0), (1.0, 5.0)]
        for i, (x, y) in enumerate(xy):
            tsp.add_location(Location(i, f"c{i}", x, y))
        m = tsp.encode_to_ising(penalty_weights={"city_visit": 40.0, "position_fill": 30.0})
        d["tsp_xy"] = np.asarray(xy)
        d["tsp_dist"] = np.asarray(tsp.distance_matrix, np.float64)
        d["tsp_J"], d["tsp_h"] = m.couplings.numpy().copy(), m.external_fields.numpy().copy()

        sch = SchedulingProblem()
        sch.time_horizon, sch.time_discretization = 8.0, 4
        for i, (dur, due, pr) in enumerate([(2.0, None, 1.0), (4.0, 6.0, 2.0), (1.0, 3.0, 0.5)]):
            sch.add_task(Task(id=i, duration=dur, due_date=due, priority=pr))
        for j in range(2):
            sch.add_agent(Agent(id=j, name=f"a{j}"))
        m = sch.encode_to_ising(objective="makespan",
                                penalty_weights={"assignment": 100.0, "capacity": 50.0,
                                                 "time_window": 60.0})
        d["sched_durations"] = np.asarray([2.0, 4.0, 1.0])
        d["sched_due"] = np.asarray([np.nan, 6.0, 3.0])
        d["sched_J"], d["sched_h"] = m.couplings.numpy().copy(), m.external_fields.numpy().copy()
    finally:
        pbase.ProblemTemplate.create_ising_model = orig_create
    np.savez_compressed(os.path.join(out, "encoders.npz"), kind="encoders", **d)
    print("encoders: ok", {k: v.shape for k, v in d.items() if hasattr(v, "shape")})


def case_schedules(out):
    """TemperatureSchedule.get_temperature tables (temperature_scheduler.py:68-213)."""
    d = {}
    for st in (ScheduleType.LINEAR, ScheduleType.EXPONENTIAL, ScheduleType.GEOMETRIC,
               ScheduleType.LOGARITHMIC, ScheduleType.POWER_LAW, ScheduleType.FAST,
               ScheduleType.BOLTZMANN):
        s = TemperatureScheduler.create_schedule(st, 10.0, 0.01, 1000)
        d[st.value] = np.asarray([s.get_temperature(k) for k in range(0, 1200)], np.float64)
    # adaptive: driven by a fixed synthetic acceptance sequence
    s = TemperatureScheduler.create_schedule(ScheduleType.ADAPTIVE, 10.0, 0.01, 1000)
    rng = np.random.RandomState(0)
    accs = rng.rand(400)
    d["adaptive_acc"] = accs
    d["adaptive"] = np.asarray([s.update(k, acceptance_rate=float(accs[k])) for k in range(400)])
    np.savez_compressed(os.path.join(out, "schedules.npz"), kind="schedules", **d)
    print("schedules: ok")


def case_diagnostics(out):
    """SpinDynamics.get_autocorrelation_time / thermal_equilibrium_check (core/spin_dynamics.py:361-421)
    on recorded histories: a real run's energy / magnetisation series and synthetic ones (AR(1) series of
    several correlation lengths, a drifting series, short and constant ones)."""
    torch.manual_seed(5)                 # before the model: its initial spins come from the global generator
    np.random.seed(5)
    model = dense_model(pm1_couplings(48, 12))
    dyn = SpinDynamics(model, temperature=2.5)
    for _ in range(260):
        dyn.sweep()
    series = {"run_energy": np.asarray(dyn.energy_history, np.float64),
              "run_magnetization": np.asarray(dyn.magnetization_history, np.float64)}
    rng = np.random.RandomState(17)
    for name, phi in (("ar_02", 0.2), ("ar_08", 0.8), ("ar_097", 0.97)):
        x = np.zeros(400)
        for i in range(1, 400):
            x[i] = phi * x[i - 1] + rng.randn()
        series[name] = x
    series["drift"] = np.linspace(-50.0, -250.0, 300) + rng.randn(300)
    series["short"] = rng.randn(9)
    series["ten"] = rng.randn(10)
    series["constant"] = np.full(250, -17.0)
    d = {}
    for name, x in series.items():
        d[f"{name}__data"] = x
        for obs in ("energy", "magnetization"):
            dyn.energy_history = list(x) if obs == "energy" else []
            dyn.magnetization_history = list(x) if obs == "magnetization" else []
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                d[f"{name}__tau_{obs}"] = np.float64(dyn.get_autocorrelation_time(obs))
        dyn.energy_history = list(x)
        for win in (100, 50, 5):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                d[f"{name}__equilibrium_w{win}"] = np.bool_(dyn.thermal_equilibrium_check(win))
    np.savez_compressed(os.path.join(out, "diagnostics.npz"), kind="diagnostics", **d)
    print("diagnostics: ok", {k: v for k, v in d.items() if "__data" not in k and "run_" in k})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--only", default="")
    ap.add_argument("--compare", default="",
                    help="after writing, compare every array of every .npz in --out with the same file under this "
                         "directory (the committed fixtures): prints the differences, exit code 1 if there are any")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    torch.set_num_threads(1)
    want = lambda n: (not a.only) or a.only in n  # noqa: E731

    z = lambda n: torch.zeros(n)  # noqa: E731
    if want("sweeps_pm1_n8"):
        case_sweeps("sweeps_pm1_n8", pm1_couplings(8, 1), z(8), 1.5, 40, 42, a.out)
    if want("sweeps_pm1_n16"):
        case_sweeps("sweeps_pm1_n16", pm1_couplings(16, 1), z(16), 2.0, 40, 42, a.out)
    if want("sweeps_pm1_n64"):
        case_sweeps("sweeps_pm1_n64", pm1_couplings(64, 1), z(64), 3.0, 60, 42, a.out)
    if want("sweeps_pm1_n64_cold"):
        case_sweeps("sweeps_pm1_n64_cold", pm1_couplings(64, 1), z(64), 0.4, 60, 7, a.out)
    if want("sweeps_gauss_n64"):
        case_sweeps("sweeps_gauss_n64", gaussian_couplings(64, 5), z(64), 1.0, 40, 42, a.out)
    if want("sweeps_field_n64"):
        g = torch.Generator().manual_seed(9)
        h = (torch.randint(-2, 3, (64,), generator=g)).float()
        case_sweeps("sweeps_field_n64", pm1_couplings(64, 1), h, 2.0, 40, 42, a.out)
    if want("sweeps_pm1_n300"):
        # N not a multiple of 4/64/256: ragged row tail for the kernels
        case_sweeps("sweeps_pm1_n300", pm1_couplings(300, 3), z(300), 4.0, 8, 11, a.out)
    if want("sweeps_glauber_n64"):
        g = torch.Generator().manual_seed(9)
        h = (torch.randint(-2, 3, (64,), generator=g)).float()
        case_sweeps("sweeps_glauber_n64", pm1_couplings(64, 1), h, 2.5, 40, 21, a.out,
                    rule=UpdateRule.GLAUBER)
    if want("sweeps_heatbath_n64"):
        case_sweeps("sweeps_heatbath_n64", pm1_couplings(64, 1), z(64), 1.7, 40, 22, a.out,
                    rule=UpdateRule.HEAT_BATH)
    if want("sweeps_glauber_gauss_n32"):
        case_sweeps("sweeps_glauber_gauss_n32", gaussian_couplings(32, 8), z(32), 0.9, 30, 23,
                    a.out, rule=UpdateRule.GLAUBER)
    if want("sweeps_wolff_n24"):
        # mixed-sign couplings: the reference grows clusters over J < 0 bonds between aligned spins
        g = torch.Generator().manual_seed(31)
        h = (torch.randint(-1, 2, (24,), generator=g)).float()
        case_sweeps("sweeps_wolff_n24", pm1_couplings(24, 8), h, 2.5, 6, 33, a.out, rule=UpdateRule.WOLFF)
    if want("sweeps_wolff_gauss_n20"):
        case_sweeps("sweeps_wolff_gauss_n20", gaussian_couplings(20, 9), z(20), 1.2, 5, 34, a.out,
                    rule=UpdateRule.WOLFF)
    if want("sa_default_n64"):
        # defaults: geometric alpha=.95 floors at sweep 135, early stop at sweep 480
        case_sa("sa_default_n64", pm1_couplings(64, 1), z(64), dict(random_seed=42), 123, a.out)
    if want("sa_linear_n20"):
        case_sa("sa_linear_n20", pm1_couplings(20, 4), z(20),
                dict(n_sweeps=200, initial_temp=5.0, final_temp=0.05,
                     schedule_type=ScheduleType.LINEAR, schedule_params={}, random_seed=3,
                     record_interval=5), 5, a.out)
    if want("pt_c1_n64_r8"):
        # BASELINE.json configs[0]: 64-spin dense +-1, 8 replicas, 1000 sweeps
        case_pt("pt_c1_n64_r8", pm1_couplings(64, 1), z(64),
                dict(n_replicas=8, n_sweeps=1000, temp_min=0.1, temp_max=10.0,
                     exchange_interval=10, record_interval=10, random_seed=42), a.out)
    if want("pt_small_n16_r4"):
        case_pt("pt_small_n16_r4", pm1_couplings(16, 2), z(16),
                dict(n_replicas=4, n_sweeps=120, temp_min=0.5, temp_max=5.0,
                     temp_distribution="linear", exchange_interval=3, record_interval=4,
                     random_seed=7), a.out)
    if want("pt_allpairs_n16_r5"):
        case_pt("pt_allpairs_n16_r5", pm1_couplings(16, 2), z(16),
                dict(n_replicas=5, n_sweeps=90, temp_min=0.5, temp_max=5.0,
                     exchange_interval=3, record_interval=4, exchange_method="all_pairs",
                     random_seed=11), a.out)
    if want("wire"):
        case_wire_formats(a.out)
    if want("operator_n48"):
        g = torch.Generator().manual_seed(13)
        h = torch.randint(-1, 2, (48,), generator=g).float()
        case_operator("operator_n48", pm1_couplings(48, 6), h, 1.7, 21, a.out)
    if want("schedules"):
        case_schedules(a.out)
    if want("encoders"):
        case_encoders(a.out)
    if want("diagnostics"):
        case_diagnostics(a.out)
    if a.compare:
        sys.exit(compare_dirs(a.out, a.compare))


def compare_dirs(new, old):
    """Content comparison of two fixture directories (npz archives carry time stamps: bytes differ, arrays must not)."""
    import glob
    bad = 0
    names = sorted({os.path.basename(p) for d in (new, old) for p in glob.glob(os.path.join(d, "*.npz"))})
    for name in names:
        pa, pb = os.path.join(new, name), os.path.join(old, name)
        if not (os.path.exists(pa) and os.path.exists(pb)):
            print(f"DIFF {name}: only in {new if os.path.exists(pa) else old}")
            bad += 1
            continue
        a, b = np.load(pa, allow_pickle=True), np.load(pb, allow_pickle=True)
        for k in sorted(set(a.files) | set(b.files)):
            if k not in a.files or k not in b.files:
                print(f"DIFF {name}[{k}]: only in {'new' if k in a.files else 'old'}")
                bad += 1
            elif a[k].dtype != b[k].dtype or a[k].shape != b[k].shape or not np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"):
                print(f"DIFF {name}[{k}]")
                bad += 1
    print(f"compared {len(names)} fixtures: {bad} difference(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    main()
