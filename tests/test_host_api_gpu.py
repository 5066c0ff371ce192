"""GPU tests of the host-side mirror of the reference API (IsingModel, GPUAnnealer.anneal,
ParallelTempering.run, the CUDAKernelManager operators, SpinGlassScheduler, sharding) against
the golden runs captured from the reference and against the oracle."""
import numpy as np
import pytest
import torch

import oracle
from oracle_follow import follow, ladder_ends
from conftest import load_golden
from oracle_engine import OracleEngine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sg():
    import spin_glass_anneal_rl_amd as m
    return m


def model_from(sg, J, h, s0=None, sparse=False):
    n = J.shape[0]
    m = sg.IsingModel(sg.IsingModelConfig(n_spins=n, use_sparse=sparse))
    m.set_couplings_from_matrix(torch.from_numpy(np.asarray(J, np.float32)))
    m.set_external_fields(torch.from_numpy(np.asarray(h, np.float32)))
    if s0 is not None:
        m.set_spins(torch.from_numpy(np.asarray(s0, np.float32)))
    return m


# ----------------------------------------------------------------------------- IsingModel
@pytest.mark.parametrize("sparse", [False, True])
def test_model_arithmetic_identities(sg, sparse):
    # reference tests/unit/test_core_ising_model.py:95-107 (dE == E_new - E_old) and
    # :202-231 (dense == sparse energy)
    g = load_golden("sweeps_field_n64")
    m = model_from(sg, g["J"], g["h"], g["s0"], sparse=sparse)
    prob = oracle.Problem(J=g["J"], h=g["h"])
    s = g["s0"].copy()
    assert m.compute_energy() == float(g["e0"]) == oracle.energy(prob, s)
    for site in (0, 17, 63, 17):
        assert m.get_local_field(site) == oracle.local_field(prob, s, site)
        e_old = m.compute_energy()
        dE = m.flip_spin(site)
        s[site] = -s[site]
        assert m.compute_energy() - e_old == dE
        assert m.compute_energy() == oracle.energy(prob, s)
        assert np.array_equal(m.spins.numpy().astype(np.int8), s)
    m.spins[5] *= -1  # direct tensor edits are picked up
    s[5] = -s[5]
    assert m.compute_energy() == oracle.energy(prob, s)
    m.set_coupling(1, 2, 4.0)
    J2 = g["J"].copy()
    J2[1, 2] = J2[2, 1] = 4.0
    assert m.compute_energy() == oracle.energy(oracle.Problem(J=J2, h=g["h"]), s)
    with pytest.raises(ValueError):
        m.flip_spin(64)


def test_spin_dynamics_mirror(sg):
    g = load_golden("sweeps_pm1_n64")
    prob = oracle.Problem(J=g["J"], h=g["h"])
    m = model_from(sg, g["J"], g["h"], g["s0"])
    dyn = sg.SpinDynamics(m, temperature=2.0, random_seed=5)
    e0 = m.compute_energy()
    acc, dE = dyn.single_spin_update(3)
    s = m.spins.numpy().astype(np.int8)
    assert oracle.energy(prob, s) == e0 + dE and (acc or dE == 0.0)
    energies = [dyn.sweep() for _ in range(5)]
    assert energies[-1] == oracle.energy(prob, m.spins.numpy().astype(np.int8))
    assert dyn.total_flips == 1 + 5 * 64 and 0 < dyn.get_acceptance_rate() <= 1
    assert len(dyn.energy_history) == 5 and m.compute_energy() == energies[-1]
    # same seed, same trajectory
    m2 = model_from(sg, g["J"], g["h"], g["s0"])
    d2 = sg.SpinDynamics(m2, temperature=2.0, random_seed=5)
    d2.single_spin_update(3)
    assert [d2.sweep() for _ in range(5)] == energies


def test_update_rules_through_the_public_classes(sg):
    g = load_golden("sweeps_glauber_n64")
    prob = oracle.Problem(J=g["J"], h=g["h"])
    for rule in (sg.UpdateRule.GLAUBER, sg.UpdateRule.HEAT_BATH):
        m = model_from(sg, g["J"], g["h"], g["s0"])
        dyn = sg.SpinDynamics(m, temperature=1.5, update_rule=rule, random_seed=2)
        e = [dyn.sweep() for _ in range(4)]
        assert e[-1] == oracle.energy(prob, m.spins.numpy().astype(np.int8))
        r = sg.GPUAnnealer(sg.GPUAnnealerConfig(n_sweeps=50, random_seed=4)).anneal(
            model_from(sg, g["J"], g["h"], g["s0"]), update_rule=rule)
        assert oracle.energy(prob, r.best_configuration.numpy().astype(np.int8)) == r.best_energy
        p = sg.ParallelTempering(sg.ParallelTemperingConfig(n_replicas=6, n_sweeps=40,
                                                            random_seed=4))
        r2 = p.run(model_from(sg, g["J"], g["h"]), update_rule=rule)
        assert oracle.energy(prob, r2.best_configuration.numpy().astype(np.int8)) == r2.best_energy
    # the Wolff cluster rule runs through the same host loops (its own kernel)
    rw = sg.GPUAnnealer(sg.GPUAnnealerConfig(n_sweeps=8, random_seed=4)).anneal(
        model_from(sg, g["J"], g["h"], g["s0"]), update_rule=sg.UpdateRule.WOLFF)
    assert oracle.energy(prob, rw.best_configuration.numpy().astype(np.int8)) == rw.best_energy


# ----------------------------------------------------------------------------- GPUAnnealer
@pytest.mark.parametrize("name", ["sa_default_n64", "sa_linear_n20"])
def test_gpu_annealer_reproduces_reference_run(sg, name):
    g = load_golden(name)
    m = model_from(sg, g["J"], g["h"], g["s0"])
    seed = int(g["random_seed"])
    cfg = sg.GPUAnnealerConfig(
        n_sweeps=int(g["n_sweeps_cfg"]), initial_temp=float(g["initial_temp"]),
        final_temp=float(g["final_temp"]), schedule_type=sg.ScheduleType(str(g["schedule"])),
        schedule_params={"alpha": float(g["alpha"])} if str(g["schedule"]) == "geometric" else {},
        record_interval=int(g["record_interval"]), energy_tolerance=float(g["energy_tolerance"]),
        random_seed=None if seed < 0 else seed)
    u = np.nan_to_num(g["u"], nan=2.0)
    res = sg.GPUAnnealer(cfg).anneal(m, _replay=(g["site"], u))
    assert res.n_sweeps == int(g["n_sweeps"])
    assert res.best_energy == float(g["best_energy"])
    assert np.array_equal(res.best_configuration.numpy().astype(np.int8), g["best_configuration"])
    assert np.array_equal(np.asarray(res.energy_history), g["energy_history"])
    assert np.array_equal(np.asarray(res.temperature_history), g["temperature_history"])
    assert np.array_equal(np.asarray(res.acceptance_rate_history), g["acceptance_rate_history"])
    assert np.array_equal(m.spins.numpy().astype(np.int8), g["s_final"])
    assert res.algorithm == "simulated_annealing"


def test_gpu_annealer_philox_run_is_reproducible_and_consistent(sg):
    # reference tests/unit/test_annealing_gpu_annealer.py:222-235: same seed => same result
    g = load_golden("sweeps_pm1_n300")
    prob = oracle.Problem(J=g["J"], h=g["h"])
    out = []
    for storage in ("f32", "i8", "f32"):
        m = model_from(sg, g["J"], g["h"], g["s0"])
        cfg = sg.GPUAnnealerConfig(n_sweeps=120, random_seed=11, coupling_storage=storage)
        out.append(sg.GPUAnnealer(cfg).anneal(m))
    a, b, c = out
    assert a.best_energy == b.best_energy == c.best_energy
    assert a.energy_history == b.energy_history == c.energy_history
    best = a.best_configuration.numpy().astype(np.int8)
    assert oracle.energy(prob, best) == a.best_energy <= a.energy_history[0]
    # and the whole run equals the oracle driven with the same schedule and Philox key
    s = g["s0"].copy()[None, :]
    sched = sg.TemperatureScheduler.create_schedule(sg.ScheduleType.GEOMETRIC, 10.0, 0.01, 120,
                                                   alpha=0.95)
    temps = np.maximum(sched.table(0, 120), 1e-10)[:, None]
    ref = oracle.sweeps(prob, s, temps, 120, seed=11)
    assert ref["best_energy"][0] == a.best_energy
    assert list(ref["energy_trace"][::10, 0]) == a.energy_history[1:]


def test_gpu_annealer_sequential_order_and_adaptive(sg):
    g = load_golden("sweeps_pm1_n64")
    m = model_from(sg, g["J"], g["h"], g["s0"])
    r = sg.GPUAnnealer(sg.GPUAnnealerConfig(n_sweeps=60, random_seed=3,
                                            site_order="sequential")).anneal(m)
    assert r.n_sweeps == 60 and len(r.energy_history) == 7
    m = model_from(sg, g["J"], g["h"], g["s0"])
    r = sg.GPUAnnealer(sg.GPUAnnealerConfig(n_sweeps=130, random_seed=3,
                                            schedule_type=sg.ScheduleType.ADAPTIVE)).anneal(m)
    assert r.n_sweeps == 130 and r.best_energy <= r.energy_history[0]


# ----------------------------------------------------------------------------- ParallelTempering
@pytest.mark.parametrize("name", ["pt_small_n16_r4", "pt_c1_n64_r8"])
def test_parallel_tempering_reproduces_reference_run(sg, name):
    g = load_golden(name)
    n, R, ns = g["J"].shape[0], int(g["n_replicas"]), int(g["n_sweeps"])
    cfg = sg.ParallelTemperingConfig(
        n_replicas=R, n_sweeps=ns, temp_min=float(g["temp_min"]), temp_max=float(g["temp_max"]),
        temp_distribution=str(g["temp_distribution"]),
        exchange_interval=int(g["exchange_interval"]), record_interval=int(g["record_interval"]),
        random_seed=int(g["random_seed"]))
    pt = sg.ParallelTempering(cfg)
    replay = dict(s0=g["s0"], site=g["site"].astype(np.int32).reshape(ns, R, n),
                  u=np.nan_to_num(g["u"], nan=2.0).astype(np.float32).reshape(ns, R, n),
                  exch_start=g["exch_start"], exch_u=g["exch_u"])
    res = pt.run(model_from(sg, g["J"], g["h"]), _replay=replay)
    assert res.best_energy == float(g["best_energy"])
    assert np.array_equal(res.best_configuration.numpy().astype(np.int8), g["best_configuration"])
    assert np.array_equal(np.asarray(pt.energy_histories), g["energy_histories"])
    assert np.array_equal(np.asarray(res.energy_history), g["energy_histories"][0])
    assert np.array_equal(pt.exchange_attempts, g["exchange_attempts"])
    assert np.array_equal(pt.exchange_accepts, g["exchange_accepts"])
    assert np.allclose(res.acceptance_rate_history, g["acceptance_rates"], rtol=0, atol=1e-15)
    assert np.array_equal(pt.final_spins, g["s_final"])
    assert res.n_sweeps == ns and res.algorithm == "parallel_tempering"


def test_parallel_tempering_all_pairs_reproduces_reference_run(sg):
    """exchange_method="all_pairs" (parallel_tempering.py:222-232): the reference's gate and exchange
    draws are replayed from the recorded np.random stream; attempts run as one ordered pair list
    on the engine (sga_exchange_pairs)."""
    g = load_golden("pt_allpairs_n16_r5")
    assert str(g["exchange_method"]) == "all_pairs"
    n, R, ns = g["J"].shape[0], int(g["n_replicas"]), int(g["n_sweeps"])
    cfg = sg.ParallelTemperingConfig(
        n_replicas=R, n_sweeps=ns, temp_min=float(g["temp_min"]), temp_max=float(g["temp_max"]),
        temp_distribution=str(g["temp_distribution"]), exchange_method="all_pairs",
        exchange_interval=int(g["exchange_interval"]), record_interval=int(g["record_interval"]),
        random_seed=int(g["random_seed"]))
    pt = sg.ParallelTempering(cfg)
    replay = dict(s0=g["s0"], site=g["site"].astype(np.int32).reshape(ns, R, n),
                  u=np.nan_to_num(g["u"], nan=2.0).astype(np.float32).reshape(ns, R, n),
                  np_rand_all=g["np_rand_all"])
    res = pt.run(model_from(sg, g["J"], g["h"]), _replay=replay)
    assert g["exchange_attempts"].sum() > 0 and g["exchange_accepts"].sum() > 0
    assert res.best_energy == float(g["best_energy"])
    assert np.array_equal(res.best_configuration.numpy().astype(np.int8), g["best_configuration"])
    assert np.array_equal(np.asarray(pt.energy_histories), g["energy_histories"])
    assert np.array_equal(pt.exchange_attempts, g["exchange_attempts"])
    assert np.array_equal(pt.exchange_accepts, g["exchange_accepts"])
    assert np.array_equal(pt.final_spins, g["s_final"])
    # production mode (Philox uniforms, seeded host gate): runs, is reproducible, conserves the ladder
    runs = [sg.ParallelTempering(cfg).run(model_from(sg, g["J"], g["h"])) for _ in range(2)]
    assert runs[0].best_energy == runs[1].best_energy and runs[0].energy_history == runs[1].energy_history
    with pytest.raises(ValueError):
        bad = sg.ParallelTemperingConfig(n_replicas=R, n_sweeps=5, exchange_method="ring")
        sg.ParallelTempering(bad).run(model_from(sg, g["J"], g["h"]))


def test_exchange_pairs_equals_oracle(sg):
    rng = np.random.RandomState(3)
    R, n = 9, 40
    J = np.triu(rng.randint(-1, 2, (n, n)), 1).astype(np.float32)
    J = J + J.T
    temps = np.geomspace(8.0, 0.3, R)
    pairs = [(i, j) for i in range(R - 1) for j in range(i + 1, R) if rng.rand() < 0.4]
    with sg.AnnealEngine(0) as e:
        e.set_dense(J, np.zeros(n, np.float32))
        e.init_replicas(R, seed=5)
        e.set_ladder(temps)
        slot = np.arange(R, dtype=np.int32)
        att, acc = np.zeros(R, np.int64), np.zeros(R, np.int64)
        for rnd in range(3):
            e.sweep(2)
            en = e.energies()
            u = rng.rand(len(pairs)) if rnd == 1 else None     # recorded uniforms | Philox
            want = oracle.pt_exchange_pairs(temps, en, slot, pairs, u=u, seed=5, round_=rnd,
                                            attempts=att, accepts=acc)
            assert e.exchange_pairs(pairs, u=u) == want
            assert np.array_equal(e.slot_map(), slot)
            rep_T = np.empty(R)
            rep_T[slot] = temps
            assert np.array_equal(e.temperatures(), rep_T)
        a2, c2 = e.exchange_stats()
        assert np.array_equal(a2, att) and np.array_equal(c2, acc) and acc.sum() > 0
        with pytest.raises(sg.AnnealingError):
            e.exchange_pairs([(0, R)])


def test_parallel_tempering_philox_run_equals_oracle_driver(sg):
    g = load_golden("pt_c1_n64_r8")
    R, ns = 8, 200
    cfg = sg.ParallelTemperingConfig(n_replicas=R, n_sweeps=ns, random_seed=42)
    pt = sg.ParallelTempering(cfg)
    res = pt.run(model_from(sg, g["J"], g["h"]))
    # same protocol on the oracle-backed engine
    eng = OracleEngine(J=g["J"], h=g["h"])
    eng.init_replicas(R, seed=42)
    eng.set_ladder(np.asarray(pt.temperatures), 1)
    best, hist0, k = np.inf, [], 0
    while k < ns:
        stop = k
        while stop < ns - 1 and not pt._event(stop):
            stop += 1
        eng.sweep(stop - k + 1)
        if stop % 10 == 0 and stop > 0:
            eng.exchange()
        if stop % 10 == 0:
            en = eng.energies()[eng.slot_map()]
            hist0.append(en[0])
            best = min(best, en.min())
        k = stop + 1
    assert res.best_energy == best and res.energy_history == hist0
    assert pt.exchange_accepts.sum() == eng.ex_acc.sum() > 0


# ----------------------------------------------------------------------------- operators
def test_kernel_manager_operators_equal_reference(sg):
    g = load_golden("operator_n48")
    n, nu = 48, int(g["n_updates"])
    mgr = sg.CUDAKernelManager(torch.device("cuda"))
    J, h = torch.from_numpy(g["J"]).cuda(), torch.from_numpy(g["h"]).cuda()
    # per-update uniforms: the reference draws one only when dE > 0
    prob = oracle.Problem(J=g["J"], h=g["h"])
    s = g["s0"].copy()[None, :]
    ulist = np.concatenate([g["u"], np.full(4, 2.0, np.float32)])[None, :]
    tr = oracle.sweeps(prob, s, float(g["T"]), nu, site_mode=oracle.SITE_SEQUENTIAL,
                       arith=oracle.ARITH_F32, replay_u=ulist, u_compact=True, trace=True)
    consumed = (tr["accept_trace"][0] == 0) | (tr["dE_trace"][0] > 0)
    per_update = np.full(nu * n, 2.0, np.float32)
    per_update[consumed] = g["u"]
    spins = torch.from_numpy(g["s0"].astype(np.float32)).cuda()
    out, accepted, changes = mgr.metropolis_update_optimized(
        spins, J, h, float(g["T"]), n_updates=nu, _uniforms=per_update[None, :])
    assert out is spins and accepted == int(g["accepted"])
    assert np.array_equal(spins.cpu().numpy().astype(np.int8), g["s_out"])
    assert np.array_equal(changes.cpu().numpy(), g["energy_changes"])
    assert mgr.compute_energy_optimized(spins, J, h) == float(g["energy"])
    sp = torch.from_numpy(g["pt_spins_in"].astype(np.float32)).cuda()
    en = torch.from_numpy(g["pt_energies_in"].astype(np.float32)).cuda()
    k = mgr.parallel_tempering_exchange_optimized(sp, en, torch.from_numpy(g["pt_temps"]).cuda(),
                                                  _uniforms=g["pt_u"])
    assert k == int(g["pt_exchanges"])
    assert np.array_equal(sp.cpu().numpy().astype(np.int8), g["pt_spins_out"])
    assert np.array_equal(en.cpu().numpy(), g["pt_energies_out"])
    # CPU tensors are accepted too and updated in place; Philox uniforms when none are given
    sp2 = torch.from_numpy(g["pt_spins_in"].astype(np.float32))
    en2 = torch.from_numpy(g["pt_energies_in"].astype(np.float32))
    k2 = mgr.parallel_tempering_exchange_optimized(sp2, en2, torch.from_numpy(g["pt_temps"]))
    assert 0 <= k2 <= 5 and sorted(en2.tolist()) == sorted(g["pt_energies_in"].tolist())


# ----------------------------------------------------------------------------- scheduler / sharding
def test_spin_glass_scheduler_entry_point(sg):
    g = load_golden("pt_c1_n64_r8")
    prob = oracle.Problem(J=g["J"], h=g["h"])
    sch = sg.SpinGlassScheduler(device="cuda", random_seed=9)
    res = sch.anneal(model_from(sg, g["J"], g["h"]), n_replicas=64, n_sweeps=300,
                     beta_schedule="geometric")
    best = res.best_configuration.numpy().astype(np.int8)
    assert oracle.energy(prob, best) == res.best_energy
    assert res.best_energy <= -360.0  # the reference's best on this instance is -372
    res2 = sch.anneal(model_from(sg, g["J"], g["h"], sparse=True), n_replicas=64, n_sweeps=300,
                      beta_schedule="geometric")
    assert res2.best_energy == res.best_energy  # CSR path, same Philox key, same chain
    multi = sch.anneal(model_from(sg, g["J"], g["h"]), n_replicas=64, n_sweeps=100,
                       beta_schedule="linear", n_ladders=4)
    assert oracle.energy(prob, multi.best_configuration.numpy().astype(np.int8)) == multi.best_energy


def test_sharding_over_two_engines_equals_one(sg):
    """Two engines on the same GPU stand in for two GPUs: the sharded run must equal the
    single-engine run and the oracle-backed run bit for bit."""
    g = load_golden("sweeps_pm1_n300")
    R, temps = 16, np.asarray(sg.temperature_ladder(16, 0.3, 6.0))

    def run(group):
        swaps = []
        for _ in range(8):
            group.sweep(3)
            swaps.append(group.exchange())
        return swaps, group.gather_energies(), group.global_best()

    def engine():
        e = sg.AnnealEngine(0)
        e.set_dense(g["J"], g["h"])
        return e

    one = run(sg.LocalShardedTempering([engine()], R, seed=5, slot_temps=temps))
    two = run(sg.LocalShardedTempering([engine(), engine()], R // 2, seed=5, slot_temps=temps))
    four = run(sg.LocalShardedTempering([engine() for _ in range(4)], R // 4, seed=5,
                                        slot_temps=temps))
    ref = run(sg.LocalShardedTempering([OracleEngine(J=g["J"], h=g["h"])], R, seed=5,
                                       slot_temps=temps))
    for other in (two, four, ref):
        assert other[0] == one[0] and np.array_equal(other[1], one[1])
        assert other[2][0] == one[2][0] and other[2][2] == one[2][2]
        assert np.array_equal(other[2][1], one[2][1])
    assert sum(one[0]) > 0


def test_multi_gpu_annealer_interface(sg):
    g = load_golden("sweeps_pm1_n64")
    acfg = sg.GPUAnnealerConfig(n_sweeps=40, random_seed=1)
    with pytest.raises(sg.DeviceError):
        sg.MultiGPUAnnealer(sg.MultiGPUConfig(gpu_ids=[0, 99]), acfg)
    dp = sg.MultiGPUAnnealer(sg.MultiGPUConfig(gpu_ids=[0]), acfg)
    res = dp.anneal([model_from(sg, g["J"], g["h"], g["s0"]) for _ in range(3)])
    assert len(res) == 3 and res[0].best_energy == res[1].best_energy == res[2].best_energy
    with pytest.raises(sg.AnnealingError):
        dp.anneal(model_from(sg, g["J"], g["h"]))
    rx = sg.MultiGPUAnnealer(sg.MultiGPUConfig(gpu_ids=[0], strategy="replica_exchange",
                                               replicas_per_gpu=16), acfg)
    r = rx.anneal(model_from(sg, g["J"], g["h"]))
    prob = oracle.Problem(J=g["J"], h=g["h"])
    assert oracle.energy(prob, r.best_configuration.numpy().astype(np.int8)) == r.best_energy
    with pytest.raises(sg.AnnealingError):
        sg.MultiGPUAnnealer(sg.MultiGPUConfig(gpu_ids=[0], strategy="model_parallel"),
                            acfg).anneal(model_from(sg, g["J"], g["h"]))


def test_encoded_instances_anneal_to_feasible_solutions(sg):
    """Encoders -> engine: assignment (C2b-shaped) and scheduling (C4-shaped, CSR) instances
    built in the physical convention reach zero-penalty configurations."""
    from spin_glass_anneal_rl_amd import encoders as enc
    rng = np.random.RandomState(1)
    costs = rng.randint(1, 20, (12, 12)).astype(float)
    b = enc.assignment_ising(12, 12, weight=40.0, costs=costs)
    sch = sg.SpinGlassScheduler(device="cuda", random_seed=3)
    res = sch.anneal(b.to_model(sparse=False), n_replicas=128, n_sweeps=400, beta_min=0.02,
                     beta_max=2.0)
    x = (res.best_configuration.numpy().reshape(12, 12) > 0)
    assert np.all(x.sum(0) == 1) and np.all(x.sum(1) == 1), "not a permutation"
    assert res.best_energy + b.constant == pytest.approx(costs[x].sum())
    from scipy.optimize import linear_sum_assignment
    r, c = linear_sum_assignment(costs)
    assert costs[x].sum() <= 2.0 * costs[r, c].sum()  # heuristic: sanity bound only
    # scheduling: 40 unit tasks, 1 agent, 40 slots -> every task alone in a slot
    s = enc.scheduling_ising(np.full(40, 1.0), n_agents=1, time_horizon=40.0,
                             time_discretization=40,
                             penalty_weights={"assignment": 100.0, "capacity": 50.0})
    m = s.to_model(sparse=True)
    res = sch.anneal(m, n_replicas=256, n_sweeps=600, beta_min=0.01, beta_max=1.0)
    y = (res.best_configuration.numpy().reshape(40, 40) > 0)
    assert np.all(y.sum(1) == 1) and np.all(y.sum(0) <= 1)
    prob = oracle.Problem(csr=s.to_csr(), h=s.fields())
    # real-valued fields: the tracked energy is a double sum, the from-scratch evaluation
    # rounds to fp32 as torch does (core/ising_model.py:161-168): stated tolerance 1e-6 relative
    assert oracle.energy(prob, res.best_configuration.numpy().astype(np.int8)) == \
        pytest.approx(res.best_energy, rel=1e-6)


def test_batch_processor_many_models(sg):
    rng = np.random.RandomState(3)
    models, probs = [], []
    for n in (20, 48, 20, 48, 20):
        J = np.triu(rng.randint(0, 2, (n, n)) * 2 - 1, 1).astype(np.float32)
        J = J + J.T
        h = rng.randint(-1, 2, n).astype(np.float32)
        models.append(model_from(sg, J, h))
        probs.append(oracle.Problem(J=J, h=h))
    bp = sg.BatchProcessor(sg.GPUAnnealerConfig(n_sweeps=120, random_seed=8),
                           sg.BatchConfig(batch_size=2, replicas_per_model=4))
    res = bp.process_models_batch(models)
    assert len(res) == 5 and bp.get_processing_stats()["processed_models"] == 5
    for m, p, r in zip(models, probs, res):
        best = r.best_configuration.numpy().astype(np.int8)
        assert best.size == m.n_spins and oracle.energy(p, best) == r.best_energy
        assert r.best_energy <= r.energy_history[0] and len(r.energy_history) == 13
    again = sg.BatchProcessor(sg.GPUAnnealerConfig(n_sweeps=120, random_seed=8),
                              sg.BatchConfig(batch_size=2, replicas_per_model=4))
    streamed = [r for chunk in again.process_models_stream(iter(models)) for r in chunk]
    assert len(streamed) == 5
    with pytest.raises(ValueError):
        sg.BatchConfig(batch_size=0)


# ----------------------------------------------------------------------------- full size
def test_full_size_c2_properties(sg):
    """BASELINE configs[1] at full size (10 000 spins, 1024 replicas): size-independent
    properties -- the incrementally tracked energies equal a from-scratch evaluation, the
    int8 and fp32 coupling layouts run the identical chain, bests never exceed the start."""
    n, R = 10000, 1024
    gen = torch.Generator(device="cuda").manual_seed(2)
    J = torch.triu((torch.randint(0, 2, (n, n), generator=gen, device="cuda", dtype=torch.int8)
                    * 2 - 1), 1)
    J = (J + J.T).float()
    h = torch.zeros(n, device="cuda")
    temps = np.asarray(sg.temperature_ladder(R, 0.1, 10.0))
    runs = {}
    for storage in ("f32", "i8"):
        with sg.AnnealEngine(0) as e:
            e.set_dense(J, h, storage=storage)
            e.init_replicas(R, seed=123)
            e0 = e.energies()
            e.set_ladder(temps)
            e.sweep(2)
            k = e.exchange()
            e.sweep(1)
            tracked = e.energies()
            e.recompute_energies()
            assert np.array_equal(e.energies(), tracked)
            acc, att = e.stats()
            assert np.all(att == 3 * n) and np.all(acc <= att) and acc[0] > acc[-1]
            best = np.asarray([e.best(r, with_spins=False)[0] for r in (0, 500, 1023)])
            assert np.all(best <= e0[[0, 500, 1023]])
            runs[storage] = (tracked, k, e.slot_map(), e.spins(1023))
    assert np.array_equal(runs["f32"][0], runs["i8"][0]) and runs["f32"][1] == runs["i8"][1]
    assert np.array_equal(runs["f32"][2], runs["i8"][2])
    assert np.array_equal(runs["f32"][3], runs["i8"][3])


def test_c4_shaped_instance_full_size(sg):
    """BASELINE configs[3] shape: 500 tasks x 100 slots = 50 000 spins of constraint-compiled
    penalties, CSR (degree 598), replicas on one ladder.  Size-independent properties: tracked
    energy == from-scratch energy (integer penalties), dense-int8 and CSR layouts run the same
    chain, and the first sweeps match the oracle on a few replicas."""
    from spin_glass_anneal_rl_amd import encoders as enc
    b = enc.scheduling_ising(np.full(500, 1.0), n_agents=1, time_horizon=100.0,
                             time_discretization=100, objective="total_time",
                             penalty_weights={"assignment": 100.0, "capacity": 50.0})
    csr, h = b.to_csr(), b.fields()
    n, R, seed = 50000, 64, 31
    temps = np.asarray(sg.temperature_ladder(R, 5.0, 500.0))
    with sg.AnnealEngine(0) as e:
        e.set_csr(*csr, h)
        e.init_replicas(R, seed=seed)
        assert "n=50000" in e.describe() and "waves_per_replica=2" in e.describe()
        e.set_ladder(temps)
        out = e.sweep(2, energy_trace=True)
        e.exchange()
        e.sweep(1)
        tracked, spins_csr = e.energies(), e.spins()
        e.recompute_energies()
        # |E| ~ 1e9 here: the from-scratch value is rounded to fp32 as torch.dot rounds it
        # (core/ising_model.py:161-168), the tracked one is an exact double sum
        assert np.allclose(e.energies(), tracked, rtol=1e-6, atol=0)
    prob = oracle.Problem(csr=csr, h=h)
    s = oracle.init_spins(n, 4, seed)
    ref = oracle.sweeps(prob, s, temps[:4], 2, seed=seed, n_threads=4)
    assert np.array_equal(out["energy_trace"][:, :4], ref["energy_trace"])
    J = np.zeros((n, n), np.float32)                  # same couplings, dense int8 path
    rows = np.repeat(np.arange(n), np.diff(csr[0]))
    J[rows, csr[1]] = csr[2]
    with sg.AnnealEngine(0) as e:
        e.set_dense(torch.from_numpy(J).cuda(), h)
        del J
        e.init_replicas(R, seed=seed)
        assert "storage=i8" in e.describe()
        e.set_ladder(temps)
        e.sweep(2)
        e.exchange()
        e.sweep(1)
        assert np.array_equal(e.energies(), tracked) and np.array_equal(e.spins(), spins_csr)


@pytest.mark.parametrize("n_cities,force_bits", [(40, False), (420, False), (60, True)])
def test_c5_tsp_rows_written_on_the_device(sg, n_cities, force_bits):
    """BASELINE configs[4] shape (TSP QUBO, n = cities^2, degree 4(cities - 1), 32 ladders): the
    structured CSR is produced on the GPU with 64-bit extents and handed over as device pointers.
    420 cities = 176 400 spins is past the int8 LDS capacity: spins live as bits (the form the
    1000-city instance runs in).  First sweep == oracle on the same rows; exchanges stay inside
    their ladders."""
    opts = {}  # engine options (sga_set_option): which kernel form runs, never what it computes
    from spin_glass_anneal_rl_amd import encoders as enc
    if force_bits:
        opts["force_csr_bits"] = 1
    rs = np.random.RandomState(n_cities)
    xy = rs.rand(n_cities, 2) * 100.0
    d = np.hypot(xy[:, None, 0] - xy[None, :, 0], xy[:, None, 1] - xy[None, :, 1])
    rowptr, col, val, h, const = enc.tsp_csr(d, city_visit=200.0, position_fill=200.0, device="cuda:0")
    assert rowptr.is_cuda and rowptr.dtype == torch.int64
    n, R, seed, n_ladders = n_cities ** 2, 64, 9, 32
    bits = force_bits or n > 160_000
    temps = np.tile(np.asarray(sg.temperature_ladder(R // n_ladders, 2.0, 200.0)), n_ladders)
    with sg.AnnealEngine(0) as e:
        e.set_options(opts)
        e.set_csr(rowptr, col, val, h)
        e.init_replicas(R, seed=seed)
        assert ("spins=lds-bits" in e.describe()) == bits and "recomputed" not in e.describe()
        e.set_ladder(temps, n_ladders)
        e0 = e.energies()
        out = e.sweep(1, energy_trace=True)
        e.exchange()
        e.exchange()
        att, acc = e.exchange_stats()
        slots = e.slot_map()
        assert 0 < att.sum() <= 2 * n_ladders and acc.sum() <= att.sum()
        assert sorted(slots) == list(range(R))
        assert np.array_equal(slots // (R // n_ladders), np.arange(R) // (R // n_ladders))  # ladders stay apart
    csr = (rowptr.cpu().numpy().astype(np.int32), col.cpu().numpy(), val.cpu().numpy())
    prob = oracle.Problem(csr=csr, h=h.cpu().numpy())
    k = 3
    s = oracle.init_spins(n, k, seed)
    assert np.array_equal(e0[:k], [oracle.energy(prob, s[r]) for r in range(k)])
    # the rule's temperatures are the replicas' current ones: slot i of the ladder at start
    ref = oracle.sweeps(prob, s, temps[:k], 1, seed=seed, n_threads=k)
    assert np.array_equal(out["energy_trace"][:, :k], ref["energy_trace"])


def test_scheduler_result_does_not_depend_on_autotune(sg):
    rng = np.random.RandomState(5)
    n = 1500
    J = np.triu(rng.randint(0, 2, (n, n)) * 2 - 1, 1).astype(np.float32)
    J = J + J.T
    m = sg.IsingModel(sg.IsingModelConfig(n_spins=n, use_sparse=False))
    m.set_couplings_from_matrix(torch.from_numpy(J))
    runs = [sg.SpinGlassScheduler(device="cuda", random_seed=11).anneal(
        m, n_replicas=32, n_sweeps=30, autotune=flag) for flag in (False, True)]
    assert runs[0].best_energy == runs[1].best_energy
    assert torch.equal(runs[0].best_configuration, runs[1].best_configuration)
    assert runs[0].energy_history == runs[1].energy_history


@pytest.mark.parametrize("cache", ["off", "on", "sparse"])
def test_c2b_assignment_instance_full_size(sg, cache):
    """BASELINE configs[1] parity instance: 100 agents x 100 tasks one-hot
    penalties (lambda = 100), 10 000 spins dense, 1024 replicas -- with one coupling-row read per
    proposal, with the cached-local-field sweep (a hot ladder: most proposals are accepted), and as the
    engine takes the matrix by itself: 198 of 10 000 couplings per row are non-zero, so `sga_set_dense`
    keeps it as CSR and the sweeps work on four updates per step ("sparse")."""
    opts = {}  # engine options (sga_set_option): which kernel form runs, never what it computes
    from spin_glass_anneal_rl_amd import encoders as enc
    from spin_glass_anneal_rl_amd.engine import last_kernel
    b = enc.assignment_ising(100, 100, weight=100.0)
    J, h = torch.from_numpy(b.to_dense()).cuda(), b.fields()
    n, R, seed = 10000, 1024, 77
    temps = np.asarray(sg.temperature_ladder(R, 1.0, 400.0))
    if cache == "off":
        opts["sparse_route"] = 0
    with sg.AnnealEngine(0) as e:
        e.set_options(opts)
        e.set_field_cache("off" if cache == "sparse" else cache)
        e.set_dense(J, h)
        if cache == "sparse":
            assert "csr n=10000 nnz=1980000" in e.describe() and "source=dense-matrix" in e.describe(), e.describe()
        else:
            assert "storage=i8" in e.describe()          # penalties are +-25 / -50: integer
        e.init_replicas(R, seed=seed)
        assert ("sweep=cached-local-fields" in e.describe()) == (cache == "on"), e.describe()
        if cache == "sparse":
            assert "updates_per_step=4" in e.describe(), e.describe()
        e.set_ladder(temps)
        ns, followed = 10, ladder_ends(R)
        out = e.sweep(ns, energy_trace=True)
        after = {r: e.spins(r) for r in followed}
        acc = e.stats()[0]
        if cache == "sparse":
            assert "sweep_csr_rows_kernel" in last_kernel() and "16 entries per lane" in last_kernel(), last_kernel()
        e.exchange()
        e.sweep(2)
        tracked = e.energies()
        e.recompute_energies()
        assert np.array_equal(e.energies(), tracked)
        cold = e.spins(R - 1)
    # the oracle follows the hot end (T = 400), the middle and the COLD end (T = 1 against penalties of 25..50: nearly
    # every uphill proposal is refused) of the ladder for all ten sweeps
    prob = oracle.Problem(J=b.to_dense(), h=h)
    for r, (trace, s, n_acc) in follow(prob, n, seed, temps, followed, ns).items():
        assert np.array_equal(out["energy_trace"][:, r], trace), r
        assert np.array_equal(after[r], s), r
        assert acc[r] == n_acc, r
    x = cold.reshape(100, 100) > 0       # the coldest replica is already nearly one-hot
    assert abs(int(x.sum()) - 100) <= 30


def test_cli_ising_command(sg, tmp_path, capsys):
    from spin_glass_anneal_rl_amd.__main__ import main
    out = str(tmp_path / "res.npz")
    for pattern in ("random", "nearest_neighbor", "fully_connected"):
        assert main(["ising", "--n-spins", "60", "--pattern", pattern, "--sweeps", "120",
                     "-o", out, "-v"]) == 0
        text = capsys.readouterr().out
        assert "Final energy:" in text and "Energy improvement:" in text
        back = sg.AnnealingResult.load(out)
        assert back.n_sweeps == 120 and back.best_configuration.numel() == 60


def test_engine_calls_are_serialised_across_threads(sg):
    import threading
    g = load_golden("sweeps_pm1_n64")
    with sg.AnnealEngine(0) as e:
        e.set_dense(g["J"], g["h"])
        e.init_replicas(8, seed=1)
        e.set_temperatures(np.full(8, 2.0))
        errs = []

        def work():
            try:
                for _ in range(20):
                    e.sweep(1)
                    e.energies()
            except Exception as exc:  # noqa: BLE001
                errs.append(exc)

        ts = [threading.Thread(target=work) for _ in range(4)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errs and e.counters()[0] == 80
        tracked = e.energies()
        e.recompute_energies()
        assert np.array_equal(e.energies(), tracked)


def test_energy_computer_mirror(sg):
    g = load_golden("sweeps_field_n64")
    prob = oracle.Problem(J=g["J"], h=g["h"])
    m = model_from(sg, g["J"], g["h"], g["s0"])
    ec = sg.EnergyComputer(m, sg.ComputeMode.VECTORIZED)
    s = g["s0"]
    assert ec.compute_total_energy() == oracle.energy(prob, s)
    assert ec.compute_energy_change(7) == 2.0 * s[7] * oracle.local_field(prob, s, 7)
    grad = ec.compute_energy_gradient().numpy()
    assert np.array_equal(grad, np.asarray([-oracle.local_field(prob, s, i) for i in range(64)],
                                           np.float32))
    st = ec.compute_energy_stats()
    assert st.total_energy == oracle.energy(prob, s)
    assert st.field_energy == -float(np.dot(g["h"].astype(np.float64), s))
    assert float(st.per_spin_energy.sum()) == pytest.approx(st.total_energy)
    cfgs = torch.from_numpy(oracle.init_spins(64, 5, 3).astype(np.float32))
    batch = ec.compute_batch_energies(cfgs).numpy()
    assert np.array_equal(batch, oracle.energy(prob, cfgs.numpy().astype(np.int8)).astype(np.float32))
    assert ec.compute_total_energy(cfgs[2]) == batch[2]


def test_checkpoint_resume_is_bit_exact(sg):
    g = load_golden("sweeps_pm1_n300")
    R, temps = 12, np.asarray(sg.temperature_ladder(12, 0.4, 5.0))

    def fresh():
        e = sg.AnnealEngine(0)
        e.set_dense(g["J"], g["h"])
        e.init_replicas(R, seed=77)
        e.set_ladder(np.tile(temps[:6], 2), n_ladders=2)
        return e

    def advance(e, rounds):
        for _ in range(rounds):
            e.sweep(3)
            e.exchange(count=False)

    a = fresh()
    advance(a, 4)
    blob = a.export_state()
    advance(a, 5)
    b = fresh()
    with pytest.raises(sg.AnnealingError):
        b.import_state(blob[:-8])
    b.import_state(blob)
    assert b.counters() == (12, 4)
    advance(b, 5)
    assert np.array_equal(a.spins(), b.spins()) and np.array_equal(a.energies(), b.energies())
    assert np.array_equal(a.slot_map(), b.slot_map()) and a.counters() == b.counters()
    assert all(np.array_equal(x, y) for x, y in zip(a.exchange_stats(), b.exchange_stats()))
    assert all(np.array_equal(x, y) for x, y in zip(a.stats(), b.stats()))
    for r in (0, 5, 11):
        ea, sa, _ = a.best(r)
        eb, sb, _ = b.best(r)
        assert ea == eb and np.array_equal(sa, sb)
    other = sg.AnnealEngine(0)
    other.set_dense(g["J"], g["h"])
    other.init_replicas(R + 1, seed=77)
    with pytest.raises(sg.AnnealingError):
        other.import_state(blob)
    for e in (a, b, other):
        e.close()


def test_integration_md_binding_stub_runs_as_written(sg):
    """INTEGRATION.md section 2 shows the ctypes binding a reference maintainer would add.  That code block is
    executed here as it stands (only the library's file name is made absolute): the three operators against the
    reference's operator fixture -- the energy of the reference's output spins, a sequential fp32 sweep equal to
    the oracle's under the same Philox uniforms (seed 0, as the stub passes), and the exchange operator equal to
    this package's HIPKernelManager with the same seed."""
    import re
    from conftest import ROOT
    import os
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text[text.index("## 2. Reference-side binding"):]
    code = re.search(r"```python\n(.*?)```", section, re.S).group(1)
    assert 'C.CDLL("libsga.so")' in code and "class CUDAKernelManager" in code
    code = code.replace('C.CDLL("libsga.so")', f'C.CDLL({sg._native.library_path()!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md#2", "exec"), ns)
    g = load_golden("operator_n48")
    n, nu = 48, int(g["n_updates"])
    dev = torch.device("cuda", 0)
    mgr = ns["CUDAKernelManager"](dev)
    J, h = torch.from_numpy(g["J"]).to(dev), torch.from_numpy(g["h"]).to(dev)
    # compute_energy_optimized: the energy the reference computed for its own output spins
    s_out = torch.from_numpy(g["s_out"].astype(np.float32)).to(dev)
    assert mgr.compute_energy_optimized(s_out, J, h) == float(g["energy"])
    # metropolis_update_optimized: sequential sites, fp32 arithmetic, Philox uniforms of seed 0 / replica 0
    prob = oracle.Problem(J=g["J"], h=g["h"])
    s = g["s0"].copy()[None, :]
    u = np.asarray([[oracle.stream_u(0, 0, k, t) for k in range(nu) for t in range(n)]], np.float32)
    ref = oracle.sweeps(prob, s, float(g["T"]), nu, site_mode=oracle.SITE_SEQUENTIAL, arith=oracle.ARITH_F32,
                        replay_u=u, trace=True)   # (the engine's uniforms when none are handed over: Philox, seed 0)
    spins = torch.from_numpy(g["s0"].astype(np.float32)).to(dev)
    out, accepted, changes = mgr.metropolis_update_optimized(spins, J, h, float(g["T"]), n_updates=nu)
    assert out is spins and accepted == int(ref["accept_trace"].sum())
    assert np.array_equal(spins.cpu().numpy().astype(np.int8), s[0])
    assert np.array_equal(changes.cpu().numpy(), ref["dE_trace"][0].reshape(nu, n).sum(0).astype(np.float32))
    # parallel_tempering_exchange_optimized: in place on fp32 tensors; the package's own manager, same seed
    sp = torch.from_numpy(g["pt_spins_in"].astype(np.float32)).to(dev)
    en = torch.from_numpy(g["pt_energies_in"].astype(np.float32)).to(dev)
    tt = torch.from_numpy(g["pt_temps"]).to(dev)
    k = mgr.parallel_tempering_exchange_optimized(sp, en, tt)
    own = sg.CUDAKernelManager(dev, seed=0)
    sp2 = torch.from_numpy(g["pt_spins_in"].astype(np.float32)).to(dev)
    en2 = torch.from_numpy(g["pt_energies_in"].astype(np.float32)).to(dev)
    k2 = own.parallel_tempering_exchange_optimized(sp2, en2, tt)
    assert k == k2 and torch.equal(sp, sp2) and torch.equal(en, en2)
    assert sorted(en.tolist()) == sorted(g["pt_energies_in"].tolist())
