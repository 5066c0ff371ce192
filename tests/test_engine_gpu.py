"""GPU parity tests proper: the HIP kernels, called through the C ABI (libsga.so), against the
CPU oracle on the same seeded inputs and against the golden vectors captured from the
reference.  Integer-valued couplings: bit-exact.  Real-valued couplings: the kernels form the
row sum in fp64 and round once to fp32 exactly as the oracle does, so decisions and energies
are compared exactly too; against the reference's own fp32 (MKL) sums the stated tolerance
is 1e-5 relative (tests/test_oracle_golden.py).
"""
import numpy as np
import pytest

import oracle
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sg():
    import spin_glass_anneal_rl_amd as m
    return m


def pm1(n, seed):
    rng = np.random.RandomState(seed)
    J = np.triu(rng.randint(0, 2, (n, n)) * 2 - 1, 1).astype(np.float32)
    return J + J.T


def gauss(n, seed):
    rng = np.random.RandomState(seed)
    J = np.triu(rng.randn(n, n), 1).astype(np.float32)
    return J + J.T


def csr_of(J):
    n = J.shape[0]
    rowptr = np.concatenate([[0], np.cumsum((J != 0).sum(1))]).astype(np.int32)
    colidx = np.concatenate([np.nonzero(J[i])[0] for i in range(n)]).astype(np.int32)
    val = np.concatenate([J[i][J[i] != 0] for i in range(n)]).astype(np.float32)
    return rowptr, colidx, val


def ladder(R, tmax=10.0, tmin=0.1):
    return np.asarray([tmax * (tmin / tmax) ** (i / max(R - 1, 1)) for i in range(R)])


# ----------------------------------------------------------------------------- basics
def test_library_reports_geometry(sg):
    with sg.AnnealEngine(0) as e:
        e.set_dense(pm1(300, 1), np.zeros(300, np.float32), storage="f32")
        e.init_replicas(4, seed=1)
        d = e.describe()
        assert "dense n=300" in d and "storage=f32" in d


@pytest.mark.parametrize("n,R", [(8, 3), (64, 8), (300, 5), (1000, 4)])
def test_init_spins_and_energy_match_oracle(sg, n, R):
    J = pm1(n, 3)
    h = np.random.RandomState(4).randint(-2, 3, n).astype(np.float32)
    prob = oracle.Problem(J=J, h=h)
    for storage in ("f32", "i8"):
        with sg.AnnealEngine(0) as e:
            e.set_dense(J, h, storage=storage)
            e.init_replicas(R, seed=0xABCDEF0123, replica0=0)
            s = e.spins()
            assert np.array_equal(s, oracle.init_spins(n, R, 0xABCDEF0123))
            assert np.array_equal(e.energies(), oracle.energy(prob, s))


def test_energy_real_valued_and_csr(sg):
    n, R = 257, 6
    J = gauss(n, 8)
    J[np.abs(J) < 1.0] = 0.0
    h = np.random.RandomState(5).randn(n).astype(np.float32)
    prob = oracle.Problem(J=J, h=h)
    s0 = oracle.init_spins(n, R, 77)
    with sg.AnnealEngine(0) as e:
        e.set_dense(J, h)
        e.init_replicas(R, seed=77)
        assert np.allclose(e.energies(), oracle.energy(prob, s0), rtol=0, atol=1e-9)
    with sg.AnnealEngine(0) as e:
        e.set_csr(*csr_of(J), h)
        e.init_replicas(R, seed=77)
        assert np.allclose(e.energies(), oracle.energy(prob, s0), rtol=0, atol=1e-9)


# ----------------------------------------------------------------------------- golden replay
SWEEP_CASES = ["sweeps_pm1_n8", "sweeps_pm1_n16", "sweeps_pm1_n64", "sweeps_pm1_n64_cold",
               "sweeps_field_n64", "sweeps_pm1_n300"]


@pytest.mark.parametrize("storage", ["f32", "i8", "csr"])
@pytest.mark.parametrize("name", SWEEP_CASES)
def test_reference_sweeps_replayed_on_gpu(sg, name, storage):
    g = load_golden(name)
    n, ns = g["J"].shape[0], int(g["n_sweeps"])
    u = np.nan_to_num(g["u"], nan=2.0).astype(np.float32)
    with sg.AnnealEngine(0) as e:
        if storage == "csr":
            e.set_csr(*csr_of(g["J"]), g["h"])
        else:
            e.set_dense(g["J"], g["h"], storage=storage)
        e.init_replicas(1, seed=0, s0=g["s0"][None, :])
        assert e.energies()[0] == float(g["e0"])
        e.set_temperatures([float(g["T"])])
        out = e.sweep(ns, site_mode=sg._native.SITE_REPLAY, replay_site=g["site"][None, :],
                      replay_u=u[None, :], energy_trace=True, trace=True)
        assert np.array_equal(out["accept_trace"][0].astype(bool), g["accepted"])
        assert np.array_equal(out["dE_trace"][0], g["dE"])
        assert np.array_equal(out["energy_trace"][:, 0], g["sweep_energy"])
        assert np.array_equal(e.spins(0), g["s_final"])
        acc, att = e.stats()
        assert acc[0] == int(g["n_accepted"]) and att[0] == ns * n
        e.recompute_energies()
        assert e.energies()[0] == g["sweep_energy"][-1]


def test_reference_gaussian_sweeps_replayed_on_gpu(sg):
    g = load_golden("sweeps_gauss_n64")
    ns = int(g["n_sweeps"])
    u = np.nan_to_num(g["u"], nan=2.0).astype(np.float32)
    with sg.AnnealEngine(0) as e:
        e.set_dense(g["J"], g["h"])
        e.init_replicas(1, seed=0, s0=g["s0"][None, :])
        assert "acc=f64" in e.describe()
        e.set_temperatures([float(g["T"])])
        out = e.sweep(ns, site_mode=sg._native.SITE_REPLAY, replay_site=g["site"][None, :],
                      replay_u=u[None, :], energy_trace=True, trace=True)
        assert np.array_equal(out["accept_trace"][0].astype(bool), g["accepted"])
        assert np.allclose(out["dE_trace"][0], g["dE"], rtol=1e-5, atol=1e-5)  # vs MKL fp32 sums
        assert np.array_equal(e.spins(0), g["s_final"])
        e.recompute_energies()
        assert e.energies()[0] == pytest.approx(g["sweep_energy"][-1], rel=1e-5)


RULE_CASES = [("sweeps_glauber_n64", 1), ("sweeps_heatbath_n64", 2), ("sweeps_glauber_gauss_n32", 1)]


@pytest.mark.parametrize("storage", ["dense", "csr"])
@pytest.mark.parametrize("name,rule", RULE_CASES)
def test_reference_glauber_heatbath_replayed_on_gpu(sg, name, rule, storage):
    """Reference core/spin_dynamics.py:154-191 replayed update by update."""
    g = load_golden(name)
    exact = "gauss" not in name
    ns = int(g["n_sweeps"])
    with sg.AnnealEngine(0) as e:
        if storage == "csr":
            e.set_csr(*csr_of(g["J"]), g["h"])
        else:
            e.set_dense(g["J"], g["h"])
        e.set_update_rule(rule)
        e.init_replicas(1, seed=0, s0=g["s0"][None, :])
        e.set_temperatures([float(g["T"])])
        out = e.sweep(ns, site_mode=sg._native.SITE_REPLAY, replay_site=g["site"][None, :],
                      replay_u=g["u"][None, :], energy_trace=True, trace=True)
        assert np.array_equal(out["accept_trace"][0].astype(bool), g["accepted"])
        assert np.array_equal(e.spins(0), g["s_final"])
        assert e.stats()[0][0] == int(g["n_accepted"])
        if exact:
            assert np.array_equal(out["dE_trace"][0], g["dE"])  # heat bath: minus the change
            assert np.array_equal(out["energy_trace"][:, 0], g["sweep_energy"])
        else:
            assert np.allclose(out["dE_trace"][0], g["dE"], rtol=1e-5, atol=1e-5)
        with pytest.raises(sg.AnnealingError):
            e.sweep(1, arith=sg._native.ARITH_F32)


@pytest.mark.parametrize("rule", [1, 2])
@pytest.mark.parametrize("n,R,storage,waves", [(64, 6, "f32", 0), (1100, 4, "f32", 5),
                                               (2500, 3, "i8", 3), (700, 5, "csr", 0)])
def test_philox_glauber_heatbath_match_oracle(sg, n, R, storage, waves, rule):
    J = pm1(n, 40 + n)
    if storage == "csr":
        J = J * (np.random.RandomState(n).rand(n, n) < 0.03)
        J = np.triu(J, 1) + np.triu(J, 1).T
    h = np.random.RandomState(n).randint(-1, 2, n).astype(np.float32)
    prob = oracle.Problem(J=J, h=h)
    ns, seed = 6, 8080 + n
    temps = ladder(R, 5.0, 0.4)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, rule=rule, seed=seed, n_threads=8)
    with sg.AnnealEngine(0) as e:
        e.set_tuning(waves_per_replica=waves)
        if storage == "csr":
            e.set_csr(*csr_of(J), h)
        else:
            e.set_dense(J, h, storage=storage)
        e.set_update_rule(rule)
        e.init_replicas(R, seed=seed)
        e.set_temperatures(temps)
        out = e.sweep(ns, energy_trace=True)
        assert np.array_equal(out["energy_trace"], ref["energy_trace"])
        assert np.array_equal(e.spins(), s)
        assert np.array_equal(e.stats()[0], ref["n_accepted"])


@pytest.mark.parametrize("storage", ["f32", "i8", "csr"])
@pytest.mark.parametrize("name,exact", [("sweeps_wolff_n24", True), ("sweeps_wolff_gauss_n20", False)])
def test_reference_wolff_cluster_moves_replayed_on_gpu(sg, name, exact, storage):
    """UpdateRule.WOLFF (core/spin_dynamics.py:193-255): the reference's start sites and every
    candidate-bond uniform replayed; cluster sizes (through n_accepted), per-move dE =
    compute_energy() after - before, per-sweep energies and final spins."""
    g = load_golden(name)
    if storage == "i8" and not exact:
        pytest.skip("int8 storage needs integer couplings")
    n, ns = g["J"].shape[0], int(g["n_sweeps"])
    with sg.AnnealEngine(0) as e:
        if storage == "csr":
            e.set_csr(*csr_of(g["J"]), g["h"])
        else:
            e.set_dense(g["J"], g["h"], storage=storage)
        e.set_update_rule(sg._native.RULE_WOLFF)
        e.init_replicas(1, seed=0, s0=g["s0"][None, :])
        e.set_temperatures([float(g["T"])])
        e.set_wolff_replay(g["all_u"][None, :])
        out = e.sweep(ns, site_mode=sg._native.SITE_REPLAY, replay_site=g["site"][None, :],
                      replay_u=np.zeros((1, ns * n), np.float32), energy_trace=True, trace=True)
        assert out["accept_trace"].all()
        assert int(e.stats()[0][0]) == int(g["n_accepted"]) == int(g["cluster"].sum())
        assert np.array_equal(e.spins(0), g["s_final"])
        if exact:
            assert np.array_equal(out["dE_trace"][0], g["dE"])
            assert np.array_equal(out["energy_trace"][:, 0], g["sweep_energy"])
        else:  # Gaussian J: against the reference's fp32 MKL sums the stated tolerance
            assert np.allclose(out["dE_trace"][0], g["dE"], rtol=1e-5, atol=1e-4)
            assert np.allclose(out["energy_trace"][:, 0], g["sweep_energy"], rtol=1e-5, atol=1e-4)
        # single-site updates are not defined for a cluster rule
        with pytest.raises(sg.AnnealingError):
            e.update(0, 1, 1.0, 0.5)


@pytest.mark.parametrize("kind", ["dense", "csr"])
@pytest.mark.parametrize("integer", [True, False])
def test_philox_wolff_sweeps_match_oracle(sg, kind, integer):
    n, R, ns, seed = 300, 5, 2, 2024
    rng = np.random.RandomState(6)
    mask = np.triu(rng.rand(n, n) < 0.05, 1)
    J = (mask * (rng.randint(-2, 3, (n, n)) if integer else rng.randn(n, n))).astype(np.float32)
    J = J + J.T
    h = (rng.randint(-1, 2, n) if integer else rng.randn(n)).astype(np.float32)
    temps = ladder(R, 6.0, 1.0)
    prob = oracle.Problem(J=J, h=h) if kind == "dense" else oracle.Problem(csr=csr_of(J), h=h)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, rule=oracle.RULE_WOLFF, seed=seed, recompute_energy=True,
                        n_threads=R)
    with sg.AnnealEngine(0) as e:
        if kind == "dense":
            e.set_dense(J, h)
        else:
            e.set_csr(*csr_of(J), h)
        e.set_update_rule(sg._native.RULE_WOLFF)
        e.init_replicas(R, seed=seed)
        e.set_temperatures(temps)
        out = e.sweep(ns, energy_trace=True)
        assert np.array_equal(e.spins(), s)
        assert np.array_equal(e.stats()[0], ref["n_accepted"]) and ref["n_accepted"].min() > ns * n
        if integer:
            assert np.array_equal(out["energy_trace"], ref["energy_trace"])
        else:
            assert np.allclose(out["energy_trace"], ref["energy_trace"], rtol=1e-6, atol=1e-5)
        for r in range(R):
            be, bs, _ = e.best(r)
            assert np.isclose(be, ref["best_energy"][r], rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("name", ["sa_default_n64", "sa_linear_n20"])
def test_reference_sa_run_replayed_on_gpu(sg, name):
    g = load_golden(name)
    n, ns = g["J"].shape[0], int(g["n_sweeps"])
    u = np.nan_to_num(g["u"], nan=2.0).astype(np.float32)
    with sg.AnnealEngine(0) as e:
        e.set_dense(g["J"], g["h"])
        e.init_replicas(1, seed=0, s0=g["s0"][None, :])
        out = e.sweep(ns, site_mode=sg._native.SITE_REPLAY, sched=g["T_per_sweep"],
                      replay_site=g["site"][None, :], replay_u=u[None, :], energy_trace=True,
                      trace=True)
        assert np.array_equal(out["accept_trace"][0].astype(bool), g["accepted"])
        be, bs, _ = e.best(0)
        assert be == float(g["best_energy"]) and np.array_equal(bs, g["best_configuration"])
        ri = int(g["record_interval"])
        assert np.array_equal(out["energy_trace"][::ri, 0], g["energy_history"][1:])
        assert np.array_equal(e.spins(0), g["s_final"])


def test_operator_fallback_semantics_on_gpu(sg):
    g = load_golden("operator_n48")
    n, nu = g["J"].shape[0], int(g["n_updates"])
    prob = oracle.Problem(J=g["J"], h=g["h"])
    # expand the reference's compact uniform list to one value per update with the oracle
    s = g["s0"].copy()[None, :]
    ulist = np.concatenate([g["u"], np.full(4, 2.0, np.float32)])[None, :]
    tr = oracle.sweeps(prob, s, float(g["T"]), nu, site_mode=oracle.SITE_SEQUENTIAL,
                       arith=oracle.ARITH_F32, replay_u=ulist, u_compact=True, trace=True)
    # the reference draws a uniform only when dE > 0 (cuda_kernels.py:390): rejected updates
    # always consumed one, accepted ones only if their dE was positive
    consumed = (tr["accept_trace"][0] == 0) | (tr["dE_trace"][0] > 0)
    assert consumed.sum() == len(g["u"])
    per_update = np.full(nu * n, 2.0, np.float32)
    per_update[consumed] = g["u"]
    with sg.AnnealEngine(0) as e:
        e.set_dense(g["J"], g["h"])
        e.init_replicas(1, seed=0, s0=g["s0"][None, :])
        e.set_temperatures([float(g["T"])])
        out = e.sweep(nu, site_mode=sg._native.SITE_SEQUENTIAL, arith=sg._native.ARITH_F32,
                      replay_u=per_update[None, :], trace=True)
        assert np.array_equal(e.spins(0), g["s_out"])
        assert e.stats()[0][0] == int(g["accepted"])
        ech = out["dE_trace"][0].reshape(nu, n).sum(0).astype(np.float32)
        assert np.array_equal(ech, g["energy_changes"])
        e.recompute_energies()
        assert e.energies()[0] == float(g["energy"])
    sp = g["pt_spins_in"].astype(np.float32)
    en = g["pt_energies_in"].astype(np.float32).copy()
    k = sg.op_pt_exchange(0, sp, en, g["pt_temps"], g["pt_u"])
    assert k == int(g["pt_exchanges"])
    assert np.array_equal(sp.astype(np.int8), g["pt_spins_out"])
    assert np.array_equal(en, g["pt_energies_out"])


# ----------------------------------------------------------------------------- Philox stream
PHILOX_CASES = [
    # (n, R, storage, waves)  -- waves=0: heuristic
    (64, 8, "f32", 0), (64, 8, "i8", 0), (63, 5, "f32", 0), (300, 6, "f32", 2),
    (1000, 7, "f32", 1), (1000, 7, "f32", 4), (1100, 4, "f32", 5), (2500, 6, "f32", 10),
    (2500, 6, "f32", 3), (2500, 3, "i8", 3), (4100, 3, "i8", 5), (4000, 4, "f32", 16),
    (10000, 2, "f32", 4), (10000, 2, "f32", 8), (10000, 2, "i8", 0),
    # the geometries the autotuner has picked for the graded run (BENCH_r02: 13 waves x 4 chunks; 9 x 5)
    (10000, 2, "f32", 13), (10000, 2, "f32", 9),
]


@pytest.mark.parametrize("n,R,storage,waves", PHILOX_CASES)
def test_philox_sweeps_match_oracle(sg, n, R, storage, waves):
    J = pm1(n, 10 + n)
    h = np.random.RandomState(n).randint(-1, 2, n).astype(np.float32)
    prob = oracle.Problem(J=J, h=h)
    ns = 4 if n >= 2500 else 12
    temps = ladder(R, 6.0, 0.5)
    seed = 0x5EED0000 + n
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=8)
    with sg.AnnealEngine(0) as e:
        e.set_tuning(waves_per_replica=waves)
        e.set_dense(J, h, storage=storage)
        e.init_replicas(R, seed=seed)
        e.set_temperatures(temps)
        out = e.sweep(ns, energy_trace=True)
        assert np.array_equal(out["energy_trace"], ref["energy_trace"]), e.describe()
        assert np.array_equal(e.spins(), s)
        assert np.array_equal(e.energies(), ref["energy"])
        assert np.array_equal(e.stats()[0], ref["n_accepted"])
        for r in range(R):
            be, bs, _ = e.best(r)
            assert be == ref["best_energy"][r] and np.array_equal(bs, ref["best_spins"][r])
        e.recompute_energies()
        assert np.array_equal(e.energies(), ref["energy"])


@pytest.mark.parametrize("n,R,waves", [(64, 4, 0), (700, 5, 3), (2500, 3, 5)])
def test_philox_sweeps_real_couplings_match_oracle(sg, n, R, waves):
    J = gauss(n, n)
    h = np.random.RandomState(n + 1).randn(n).astype(np.float32)
    prob = oracle.Problem(J=J, h=h)
    ns, seed = 6, 99 + n
    temps = ladder(R, 3.0, 0.3)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, trace=True, n_threads=8)
    with sg.AnnealEngine(0) as e:
        e.set_tuning(waves_per_replica=waves)
        e.set_dense(J, h)
        e.init_replicas(R, seed=seed)
        e.set_temperatures(temps)
        out = e.sweep(ns, energy_trace=True, trace=True)
        assert np.array_equal(out["accept_trace"], ref["accept_trace"])
        assert np.array_equal(out["dE_trace"], ref["dE_trace"])
        assert np.array_equal(e.spins(), s)
        assert np.allclose(out["energy_trace"], ref["energy_trace"], rtol=0, atol=1e-9)


def test_schedule_tables_and_split_launches(sg):
    n, R, ns = 300, 5, 9
    J, h = pm1(n, 2), np.zeros(n, np.float32)
    prob = oracle.Problem(J=J, h=h)
    seed = 4242
    sched = np.linspace(5.0, 0.2, ns)[:, None] * np.linspace(1.0, 2.0, R)[None, :]
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, sched, ns, seed=seed)
    for spl in (0, 1, 2, 4):
        with sg.AnnealEngine(0) as e:
            e.set_tuning(sweeps_per_launch=spl)
            e.set_dense(J, h)
            e.init_replicas(R, seed=seed)
            out = e.sweep(ns, sched=sched, energy_trace=True)
            assert np.array_equal(out["energy_trace"], ref["energy_trace"])
            assert np.array_equal(e.spins(), s)
    # shared 1-D schedule, two consecutive calls continue the same stream
    s1 = oracle.init_spins(n, R, seed)
    ref1 = oracle.sweeps(prob, s1, np.repeat(sched[:, :1], R, 1), ns, seed=seed)
    with sg.AnnealEngine(0) as e:
        e.set_dense(J, h)
        e.init_replicas(R, seed=seed)
        a = e.sweep(4, sched=sched[:4, 0], energy_trace=True)
        b = e.sweep(ns - 4, sched=sched[4:, 0], energy_trace=True)
        assert np.array_equal(np.vstack([a["energy_trace"], b["energy_trace"]]),
                              ref1["energy_trace"])
        assert e.counters()[0] == ns


def test_sequential_order_matches_oracle(sg):
    n, R, ns = 130, 4, 5
    J, h = pm1(n, 6), np.random.RandomState(1).randint(-1, 2, n).astype(np.float32)
    prob = oracle.Problem(J=J, h=h)
    seed = 31337
    rng = np.random.RandomState(0)
    u = rng.rand(R, ns * n).astype(np.float32)
    for arith in (oracle.ARITH_F64, oracle.ARITH_F32):
        s = oracle.init_spins(n, R, seed)
        ref = oracle.sweeps(prob, s, 1.3, ns, site_mode=oracle.SITE_SEQUENTIAL, arith=arith,
                            replay_u=u, seed=seed)
        with sg.AnnealEngine(0) as e:
            e.set_dense(J, h, storage="f32")
            e.init_replicas(R, seed=seed)
            e.set_temperatures(np.full(R, 1.3))
            out = e.sweep(ns, site_mode=sg._native.SITE_SEQUENTIAL, arith=arith, replay_u=u,
                          energy_trace=True)
            assert np.array_equal(out["energy_trace"], ref["energy_trace"])
            assert np.array_equal(e.spins(), s)


# ----------------------------------------------------------------------------- CSR
@pytest.mark.parametrize("bits", [False, True])
@pytest.mark.parametrize("n,deg,R", [(200, 6, 9), (3000, 32, 10), (1500, 100, 3)])
def test_csr_sweeps_match_oracle(sg, n, deg, R, bits):
    """Short rows: one replica per wave, several replicas per workgroup; `bits` = the same with the
    spins as bits in LDS (the form for 53k < n <= 1.3M spins)."""
    opts = {}  # engine options (sga_set_option): which kernel form runs, never what it computes
    if bits:
        opts["force_csr_bits"] = 1
    rng = np.random.RandomState(n)
    J = np.zeros((n, n), np.float32)
    for i in range(n):
        for j in rng.choice(n, deg // 2, replace=False):
            if i != j:
                J[i, j] = J[j, i] = rng.choice([-1.0, 1.0])
    h = rng.randint(-1, 2, n).astype(np.float32)
    csr = csr_of(J)
    prob = oracle.Problem(csr=csr, h=h)
    ns, seed = 8, 5150 + n
    temps = ladder(R, 4.0, 0.4)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=8)
    with sg.AnnealEngine(0) as e:
        e.set_options(opts)
        e.set_csr(*csr, h)
        e.init_replicas(R, seed=seed)
        d = e.describe()
        assert ("spins=lds-bits" in d) == bits and "replicas_per_block=4" in d
        e.set_temperatures(temps)
        out = e.sweep(ns, energy_trace=True)
        assert np.array_equal(out["energy_trace"], ref["energy_trace"])
        assert np.array_equal(e.spins(), s)
        assert np.array_equal(e.stats()[0], ref["n_accepted"])
        be, bs, idx = e.best()
        assert be == ref["best_energy"].min()
        out2 = e.sweep(2, energy_trace=True, trace=True)       # general (traced) variant
        ref2 = oracle.sweeps(prob, s, temps, 2, seed=seed, sweep0=ns, energy=ref["energy"], trace=True)
        assert np.array_equal(out2["accept_trace"], ref2["accept_trace"])
        assert np.array_equal(out2["energy_trace"], ref2["energy_trace"])


@pytest.mark.parametrize("bits", [False, True])
@pytest.mark.parametrize("n,deg,amp,half_h,dups", [
    (5, 4, 1, False, False),      # tiny: both updates of a pair hit the same site / neighbours all the time
    (17, 16, 3, False, True),     # every row nearly full, duplicate entries that add up, odd n (last pair has no B)
    (64, 63, 2, True, False),     # complete graph, rows of 63 entries, half-integer fields
    (65, 64, 1, False, False),    # rows of exactly 64 entries
    (1000, 32, 1, False, False),  # the C3 shape
    (301, 9, 5, True, True),
    (400, 6, 1, False, False),    # lattice-like degree: one entry per lane in the several-updates-per-step builds
    (120, 30, 2, False, False),   # rows of 17..32 entries
])
def test_csr_pair_look_ahead_equals_one_update_at_a_time(sg, n, deg, amp, half_h, dups, bits):
    """Narrow table form (integer problems, rows of <= 64 entries): the two updates of a Philox pair are
    reduced together and the chain replayed on scalars (fix-up by the entries of row B at site A).  Against
    the oracle and against the one-update-at-a-time form of the same kernel.  (Opt-in, SGA_CSR_PAIR_AHEAD:
    measured -1 ... +3 % on BASELINE configs[2], so the default stays one update at a time.)"""
    opts = {}  # engine options (sga_set_option): which kernel form runs, never what it computes
    if bits:
        opts["force_csr_bits"] = 1
    rng = np.random.RandomState(7 * n + deg)
    J = np.zeros((n, n), np.float32)
    for i in range(n):
        for j in rng.choice(n, min(deg // 2 + 1, n), replace=False):
            if i != j:
                J[i, j] = J[j, i] = float(rng.choice([v for v in range(-amp, amp + 1) if v != 0]))
    J[np.abs(J).sum(1) == 0, :] = 0
    h = rng.randint(-2, 3, n).astype(np.float32) + (0.5 if half_h else 0.0)
    rowptr, col, val = csr_of(J)
    if dups:  # split some entries into two that add up (unsorted rows, duplicate columns)
        rp, ci, vv = [0], [], []
        for i in range(n):
            for e_ in range(rowptr[i], rowptr[i + 1]):
                if (e_ % 3 == 0) and (rowptr[i + 1] - rowptr[i]) < 40:
                    ci += [col[e_], col[e_]]
                    vv += [val[e_] + 2.0, -2.0]
                else:
                    ci.append(col[e_])
                    vv.append(val[e_])
            rp.append(len(ci))
        rowptr, col, val = np.asarray(rp, np.int32), np.asarray(ci, np.int32), np.asarray(vv, np.float32)
    assert np.diff(rowptr).max() <= 64
    csr = (rowptr, col, val)
    prob = oracle.Problem(csr=csr, h=h)
    R, ns, seed = 7, 10, 1717 + n
    temps = ladder(R, 6.0 * amp, 0.3 * amp)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=7)
    got = {}
    # "4" | "8" = that many updates per step (sweep_csr_rows.hip: one update per row of 16 | 8 lanes, replayed one
    # at a time when an accepted update touches a later one of the step) -- the default where it applies (None)
    for ahead in ("1", "2", "4", "8", None, "0"):
        if ahead is not None:
            opts["csr_updates_per_step"] = int(ahead)
        else:
            opts.pop("csr_updates_per_step", None)
        with sg.AnnealEngine(0) as e:
            e.set_options(opts)
            e.set_csr(*csr, h)
            e.init_replicas(R, seed=seed)
            assert "fast" in e.describe(), e.describe()
            e.set_temperatures(temps)
            out = e.sweep(ns, energy_trace=True)
            from spin_glass_anneal_rl_amd.engine import last_kernel
            assert ("sweep_csr_rows_kernel" in last_kernel()) == (ahead in ("4", "8", None)), (ahead, last_kernel())
            assert ("bit spins" in last_kernel()) == bits, last_kernel()
            assert np.array_equal(out["energy_trace"], ref["energy_trace"]), (ahead, e.describe())
            assert np.array_equal(e.spins(), s)
            assert np.array_equal(e.stats()[0], ref["n_accepted"])
            assert np.array_equal(e.energies(), ref["energy"])
            e.recompute_energies()
            assert np.array_equal(e.energies(), ref["energy"])
            got[ahead] = (out["energy_trace"], e.best()[0])
    assert all(np.array_equal(got[k][0], got["0"][0]) and got[k][1] == got["0"][1] for k in got)


@pytest.mark.parametrize("bits", [False, True])
@pytest.mark.parametrize("n,deg,amp,half_h", [
    (300, 100, 1, False),    # rows of ~80-130 entries: 8 entries per lane
    (420, 200, 2, True),     # rows of ~170-250 entries: 16 entries per lane; half-integer fields
    (144, 22, 1, False),     # (control: short rows through the same test)
    (300, 100, 60, False),   # penalties: moves beyond the accept table's 2048 entries (computed, not looked up)
    (144, 22, 120, True),
])
def test_csr_several_updates_per_step_medium_rows(sg, n, deg, amp, half_h, bits):
    """Integer problems whose rows hold 65 ... 256 entries (assignment / small scheduling instances) run four
    updates per step with 8 | 16 entries per lane: the oracle's chain, equal to the one-update form."""
    opts = {}  # engine options (sga_set_option): which kernel form runs, never what it computes
    from spin_glass_anneal_rl_amd.engine import last_kernel
    if bits:
        opts["force_csr_bits"] = 1
    rng = np.random.RandomState(3 * n + deg)
    J = np.zeros((n, n), np.float32)
    for i in range(n):
        for j in rng.choice(n, deg // 2, replace=False):
            if i != j and np.count_nonzero(J[i]) < 250 and np.count_nonzero(J[j]) < 250:
                J[i, j] = J[j, i] = float(rng.choice([v for v in range(-amp, amp + 1) if v != 0]))
    h = rng.randint(-2, 3, n).astype(np.float32) + (0.5 if half_h else 0.0)
    csr = csr_of(J)
    longest = int(np.diff(csr[0]).max())
    assert longest <= 256
    prob = oracle.Problem(csr=csr, h=h)
    R, ns, seed = 6, 6, 909 + n
    temps = ladder(R, 8.0 * amp, 0.5 * amp)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=6)
    for ahead in ("4", None, "0"):
        if ahead is not None:
            opts["csr_updates_per_step"] = int(ahead)
        else:
            opts.pop("csr_updates_per_step", None)
        with sg.AnnealEngine(0) as e:
            e.set_options(opts)
            e.set_csr(*csr, h)
            e.init_replicas(R, seed=seed)
            e.set_temperatures(temps)
            out = e.sweep(ns, energy_trace=True)
            k = last_kernel()
            assert ("sweep_csr_rows_kernel" in k) == (ahead != "0"), (ahead, k, e.describe())
            if ahead != "0" and longest > 64:
                assert ("16 entries per lane" if longest > 128 else "8 entries per lane") in k, (k, e.describe())
            assert np.array_equal(out["energy_trace"], ref["energy_trace"]), (ahead, k, e.describe())
            assert np.array_equal(e.spins(), s), (ahead, k)
            assert np.array_equal(e.stats()[0], ref["n_accepted"]), (ahead, k)
            e.recompute_energies()
            assert np.array_equal(e.energies(), ref["energy"]), (ahead, k)


@pytest.mark.parametrize("bits", [False, True])
@pytest.mark.parametrize("n,deg,grid", [
    (6, 4, False),       # every step hits
    (400, 6, False),     # lattice-like degree, Gaussian couplings: one entry per lane
    (300, 24, True),     # couplings on a 2^-10 grid (the fp64-exact row-sum class), rows of up to ~40 entries
    (150, 60, False),    # rows of up to 64 entries
])
def test_csr_several_updates_per_step_with_real_valued_couplings(sg, n, deg, grid, bits):
    """The several-updates-per-step form without the accept table: fp64 row sums in the canonical order (a lane's
    consecutive entries are one subtree of the 64-lane adjacent-pairs tree, the DPP steps over the row's lanes
    continue it), energies added in chain order -- bit for bit the oracle's chain, for 4 and 8 rows per step, on
    int8 and on bit spins, and equal to the one-update-at-a-time form."""
    opts = {}  # engine options (sga_set_option): which kernel form runs, never what it computes
    from spin_glass_anneal_rl_amd.engine import last_kernel
    if bits:
        opts["force_csr_bits"] = 1
    rng = np.random.RandomState(11 * n + deg)
    J = np.zeros((n, n), np.float32)
    for i in range(n):
        for j in rng.choice(n, min(deg // 2, n), replace=False):
            if i != j and np.count_nonzero(J[i]) < 62 and np.count_nonzero(J[j]) < 62:
                v = rng.randn()
                J[i, j] = J[j, i] = np.float32(np.rint(v * 1024.0) / 1024.0 if grid else v)
    h = rng.randn(n).astype(np.float32)
    csr = csr_of(J)
    assert np.diff(csr[0]).max() <= 64
    prob = oracle.Problem(csr=csr, h=h)
    R, ns, seed = 6, 8, 4242 + n
    temps = ladder(R, 3.0, 0.2)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=6)
    for ahead in ("4", "8", None, "0"):
        if ahead is not None:
            opts["csr_updates_per_step"] = int(ahead)
        else:
            opts.pop("csr_updates_per_step", None)
        with sg.AnnealEngine(0) as e:
            e.set_options(opts)
            e.set_csr(*csr, h)
            e.init_replicas(R, seed=seed)
            e.set_temperatures(temps)
            out = e.sweep(ns, energy_trace=True)
            k = last_kernel()
            assert ("sweep_csr_rows_kernel" in k and "fp64 canonical sums" in k) == (ahead != "0"), (ahead, k)
            assert np.array_equal(e.spins(), s), (ahead, k)
            assert np.array_equal(e.stats()[0], ref["n_accepted"]), (ahead, k)
            assert np.array_equal(out["energy_trace"], ref["energy_trace"]), (ahead, k)


@pytest.mark.parametrize("waves", [1, 2, 4, 8])
@pytest.mark.parametrize("integer", [True, False])
@pytest.mark.parametrize("big", [False, True])
def test_csr_wide_rows_match_oracle(sg, waves, integer, big):
    """Rows of a few hundred entries dealt to several waves per replica (C4/C5 shape); `big`
    sends the same case through the form for n > 160k (spins as bits in LDS, 64-bit extents)."""
    opts = {}  # engine options (sga_set_option): which kernel form runs, never what it computes
    if big:
        opts["force_csr_bits"] = 1
    n, R = 900, 5
    rng = np.random.RandomState(17)
    mask = np.triu(rng.rand(n, n) < 0.35, 1)
    vals = (rng.randint(0, 2, (n, n)) * 2 - 1) if integer else rng.randn(n, n)
    J = (mask * vals).astype(np.float32)
    J = J + J.T
    h = (rng.randint(-2, 3, n) if integer else rng.randn(n)).astype(np.float32)
    csr = csr_of(J)
    prob = oracle.Problem(csr=csr, h=h)
    ns, seed = 5, 6006
    temps = ladder(R, 30.0, 3.0)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=8)
    with sg.AnnealEngine(0) as e:
        e.set_options(opts)
        e.set_tuning(waves_per_replica=waves)
        e.set_csr(*csr, h)
        e.init_replicas(R, seed=seed)
        assert f"waves_per_replica={waves}" in e.describe()
        assert ("spins=lds-bits" in e.describe()) == big
        e.set_temperatures(temps)
        out = e.sweep(ns, energy_trace=True)
        assert np.array_equal(out["energy_trace"], ref["energy_trace"])
        assert np.array_equal(e.spins(), s)
        for r in range(R):
            be, bs, _ = e.best(r)
            assert be == ref["best_energy"][r] and np.array_equal(bs, ref["best_spins"][r])
        out2 = e.sweep(2, energy_trace=True, trace=True)       # general (traced) variant
        ref2 = oracle.sweeps(prob, s, temps, 2, seed=seed, sweep0=ns, energy=ref["energy"],
                             trace=True)
        assert np.array_equal(out2["accept_trace"], ref2["accept_trace"])
        assert np.array_equal(out2["energy_trace"], ref2["energy_trace"])


@pytest.mark.parametrize("every", [1, 7, 64])
@pytest.mark.parametrize("waves,integer", [(2, True), (4, False), (8, False)])
def test_csr_wide_rows_with_zero_slots_inside_the_layout(sg, waves, integer, every):
    """Head slots past a row's end read an all-zero slot; layouts beyond 1 GB carry such slots INSIDE
    (one per 2^21 slots).  SGA_ZERO_SLOT_EVERY puts them into a small layout: rows of very different
    lengths (empty ones too), so that most head slots are redirected, against the oracle."""
    opts = {}  # engine options (sga_set_option): which kernel form runs, never what it computes
    opts["zero_slot_every"] = int(every)
    opts["force_csr_bits"] = 1
    n, R = 700, 4
    rng = np.random.RandomState(5 + every)
    dens = rng.choice([0.0, 0.02, 0.2, 0.9], n)[:, None]
    mask = np.triu(rng.rand(n, n) < np.minimum(dens, dens.T), 1)
    vals = (rng.randint(0, 2, (n, n)) * 2 - 1) if integer else np.rint(rng.randn(n, n) * 256.0) / 256.0
    J = (mask * vals).astype(np.float32)
    J = J + J.T
    h = (rng.randint(-2, 3, n) if integer else rng.randn(n)).astype(np.float32)
    csr = csr_of(J)
    prob = oracle.Problem(csr=csr, h=h)
    ns, seed = 4, 8128
    temps = ladder(R, 20.0, 1.0)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=8)
    with sg.AnnealEngine(0) as e:
        e.set_options(opts)
        e.set_tuning(waves_per_replica=waves)
        e.set_csr(*csr, h)
        e.init_replicas(R, seed=seed)
        assert f"waves_per_replica={waves}" in e.describe() and "spins=lds-bits" in e.describe()
        assert np.array_equal(e.energies(), oracle.energy(prob, oracle.init_spins(n, R, seed)))
        e.set_temperatures(temps)
        out = e.sweep(ns, energy_trace=True)
        assert np.array_equal(out["energy_trace"], ref["energy_trace"])
        assert np.array_equal(e.spins(), s)


@pytest.mark.parametrize("waves", [1, 2, 4, 8])
@pytest.mark.parametrize("every", [0, 3])
def test_csr_packed_entries_run_the_same_chain(sg, waves, every):
    """Integer couplings with |J| <= 127: the bit-spin wide forms keep one dword per entry (24-bit column,
    8-bit value) and accumulate integers -- the chain of the (column, fp32 value) entries and of the
    oracle; larger values and real values keep the unpacked entries; "packed" demanded for them fails."""
    opts = {}  # engine options (sga_set_option): which kernel form runs, never what it computes
    opts["force_csr_bits"] = 1
    if every:
        opts["zero_slot_every"] = int(every)
    n, R = 1200, 4
    rng = np.random.RandomState(40 + waves)
    dens = rng.choice([0.0, 0.3, 0.6, 0.95], n)[:, None]   # long rows (mean degree > 192), empty ones too
    mask = np.triu(rng.rand(n, n) < np.minimum(dens, dens.T), 1)
    J = (mask * rng.randint(-127, 128, (n, n))).astype(np.float32)
    J = J + J.T
    h = (rng.randint(-5, 6, n) + (0.5 if waves & 2 else 0.0)).astype(np.float32)   # half-integer fields: no table
    csr = csr_of(J)
    prob = oracle.Problem(csr=csr, h=h)
    ns, seed = 3, 4242
    temps = ladder(R, 4000.0, 50.0)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=8)
    for storage in ("auto", "f32", "packed"):
        with sg.AnnealEngine(0) as e:
            e.set_options(opts)
            e.set_tuning(waves_per_replica=waves)
            e.set_csr_storage(storage)
            e.set_csr(*csr, h)
            e.init_replicas(R, seed=seed)
            assert ("entries=packed-32bit" in e.describe()) == (storage != "f32"), e.describe()
            e.set_temperatures(temps)
            out = e.sweep(ns, energy_trace=True)
            assert np.array_equal(out["energy_trace"], ref["energy_trace"]), storage
            assert np.array_equal(e.spins(), s), storage
            assert np.array_equal(e.stats()[0], ref["n_accepted"]), storage
            out2 = e.sweep(1, energy_trace=True, trace=True)   # traced build: unpacked entries, same state
            assert out2["accept_trace"].shape == (R, n)
    for bad in (J * 2.0, J + np.where(J != 0, 0.25, 0.0).astype(np.float32)):   # |J| up to 254; quarter offsets
        c2 = csr_of(bad.astype(np.float32))
        p2 = oracle.Problem(csr=c2, h=h)
        s2 = oracle.init_spins(n, R, seed)
        r2 = oracle.sweeps(p2, s2, temps, 1, seed=seed, n_threads=8)
        with sg.AnnealEngine(0) as e:
            e.set_options(opts)
            e.set_tuning(waves_per_replica=waves)
            e.set_csr(*c2, h)
            e.init_replicas(R, seed=seed)
            assert "entries=packed-32bit" not in e.describe()
            e.set_temperatures(temps)
            e.sweep(1)
            assert np.array_equal(e.spins(), s2)
        with sg.AnnealEngine(0) as e:
            e.set_options(opts)
            e.set_tuning(waves_per_replica=waves)
            e.set_csr_storage("packed")
            e.set_csr(*c2, h)
            with pytest.raises(sg.AnnealingError):
                e.init_replicas(R, seed=seed)


def random_sparse_pm1(n, deg, seed):
    """Symmetric +-1 couplings, ~deg entries per row, canonical CSR (sorted, no duplicates)."""
    import scipy.sparse as sp
    rng = np.random.RandomState(seed)
    m = n * deg // 2
    i, j = rng.randint(0, n, m), rng.randint(0, n, m)
    keep = i != j
    lo, hi = np.minimum(i, j)[keep], np.maximum(i, j)[keep]
    _, first = np.unique(lo.astype(np.int64) * n + hi, return_index=True)
    lo, hi = lo[first], hi[first]
    v = (rng.randint(0, 2, lo.size) * 2 - 1).astype(np.float32)
    A = sp.coo_matrix((np.concatenate([v, v]), (np.concatenate([lo, hi]), np.concatenate([hi, lo]))),
                      shape=(n, n)).tocsr()
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float32)


@pytest.mark.parametrize("n,deg,waves,wide_extents", [(200_000, 12, 0, False), (170_001, 300, 4, True)])
def test_csr_beyond_int8_lds_capacity_matches_oracle(sg, n, deg, waves, wide_extents):
    """n > 160k spins (BASELINE config 5's regime): spins live in LDS as bits, picked up
    automatically; 64-bit row extents through sga_set_csr64."""
    rowptr, colidx, val = random_sparse_pm1(n, deg, n)
    rng = np.random.RandomState(3)
    h = rng.randint(-1, 2, n).astype(np.float32)
    prob = oracle.Problem(csr=(rowptr, colidx, val), h=h)
    R, ns, seed = 3, 2, 77 + n
    temps = ladder(R, 6.0, 0.8)
    s = oracle.init_spins(n, R, seed)
    e0 = np.asarray([oracle.energy(prob, s[r]) for r in range(R)])
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=8)
    with sg.AnnealEngine(0) as e:
        if waves:
            e.set_tuning(waves_per_replica=waves)
        e.set_csr(rowptr.astype(np.int64) if wide_extents else rowptr, colidx, val, h)
        e.init_replicas(R, seed=seed)
        d = e.describe()
        assert "spins=lds-bits" in d and "path=integer-fast" in d and "recomputed" not in d
        assert np.array_equal(e.energies(), e0)
        e.set_temperatures(temps)
        out = e.sweep(ns, energy_trace=True)
        assert np.array_equal(out["energy_trace"], ref["energy_trace"])
        assert np.array_equal(e.spins(), s)
        assert np.array_equal(e.stats()[0], ref["n_accepted"])
        for r in range(R):
            be, bs, _ = e.best(r)
            assert be == ref["best_energy"][r] and np.array_equal(bs, ref["best_spins"][r])
        # single-site operators and a recompute read the same replica from HBM
        assert e.local_fields(0, [0, n // 2, n - 1]).shape == (3,)
        e.recompute_energies()
        assert np.array_equal(e.energies(), ref["energy"])


def test_csr_structure_is_validated_on_the_device(sg):
    n = 500
    rowptr, colidx, val = random_sparse_pm1(n, 10, 1)
    h = np.zeros(n, np.float32)
    with sg.AnnealEngine(0) as e:
        bad = rowptr.copy()
        bad[10] = bad[11] + 1
        with pytest.raises(sg.AnnealingError, match="rowptr"):
            e.set_csr(bad, colidx, val, h)
        bad = colidx.copy()
        bad[5] = n
        with pytest.raises(sg.AnnealingError, match="column"):
            e.set_csr(rowptr, bad, val, h)
        with pytest.raises(sg.AnnealingError, match="rowptr"):
            e.set_csr(rowptr.astype(np.int64) + 1, colidx, val, h)
        # symmetric: incremental energies; one entry changed: energies recomputed per sweep
        e.set_csr(rowptr, colidx, val, h)
        e.init_replicas(2, seed=1)
        assert "recomputed" not in e.describe()
        asym = val.copy()
        asym[0] = -asym[0]
        e.set_csr(rowptr, colidx, asym, h)
        e.init_replicas(2, seed=1)
        assert "recomputed" in e.describe()
        # rows in arbitrary column order (and 64-bit extents): still recognised as symmetric
        perm_c, perm_v = colidx.copy(), val.copy()
        for i in range(n):
            b, t = rowptr[i], rowptr[i + 1]
            p = np.random.RandomState(i).permutation(t - b)
            perm_c[b:t], perm_v[b:t] = colidx[b:t][p], val[b:t][p]
        e.set_csr(rowptr.astype(np.int64), perm_c, perm_v, h)
        e.init_replicas(2, seed=1)
        assert "recomputed" not in e.describe() and "integer-fast" in e.describe()
        ea = e.energies()
        e.set_csr(rowptr, colidx, val, h)
        e.init_replicas(2, seed=1)
        assert np.array_equal(e.energies(), ea)


# ----------------------------------------------------------------------------- exchange
@pytest.mark.parametrize("R,n_ladders", [(8, 1), (9, 1), (24, 3), (64, 4)])
def test_exchange_rounds_match_oracle(sg, R, n_ladders):
    n = 64
    J, h = pm1(n, 1), np.zeros(n, np.float32)
    prob = oracle.Problem(J=J, h=h)
    seed = 2024
    L = R // n_ladders
    temps = np.concatenate([ladder(L, 8.0, 0.3) for _ in range(n_ladders)])
    s = oracle.init_spins(n, R, seed)
    slot_to_rep = np.arange(R, dtype=np.int32)
    att, acc = np.zeros(R, np.int64), np.zeros(R, np.int64)
    energy = oracle.energy(prob, s)
    with sg.AnnealEngine(0) as e:
        e.set_dense(J, h)
        e.init_replicas(R, seed=seed)
        e.set_ladder(temps, n_ladders)
        for rnd in range(12):
            rep_t = np.empty(R)
            rep_t[slot_to_rep] = temps
            assert np.array_equal(e.temperatures(), rep_t)
            ref = oracle.sweeps(prob, s, rep_t, 2, seed=seed, sweep0=2 * rnd, energy=energy)
            energy = ref["energy"]
            e.sweep(2)
            assert np.array_equal(e.energies(), energy)
            n_acc = 0
            for l in range(n_ladders):
                sl = slice(l * L, (l + 1) * L)
                local = slot_to_rep[sl] - 0
                view = slot_to_rep[sl].copy()
                a_l, c_l = att[sl].copy(), acc[sl].copy()
                n_acc += oracle.pt_exchange_round(temps[sl], energy, view, start=-1, seed=seed,
                                                  round_=rnd, ladder=l, attempts=a_l, accepts=c_l)
                slot_to_rep[sl], att[sl], acc[sl] = view, a_l, c_l
            assert e.exchange() == n_acc
            assert np.array_equal(e.slot_map(), slot_to_rep)
        ea, ec = e.exchange_stats()
        assert np.array_equal(ea, att) and np.array_equal(ec, acc)
        assert acc.sum() > 0


@pytest.mark.parametrize("name", ["pt_small_n16_r4", "pt_c1_n64_r8"])
def test_reference_pt_run_replayed_on_gpu(sg, name):
    """BASELINE configs[0]: the reference's ParallelTempering.run, replayed update by update
    and exchange by exchange on the GPU (recorded torch / numpy random streams)."""
    g = load_golden(name)
    n, R, ns = g["J"].shape[0], int(g["n_replicas"]), int(g["n_sweeps"])
    temps = g["temperatures"]
    site = g["site"].astype(np.int32).reshape(ns, R, n)
    u = np.nan_to_num(g["u"], nan=2.0).astype(np.float32).reshape(ns, R, n)
    acc_ref = g["accepted"].reshape(ns, R, n)
    ei, ri = int(g["exchange_interval"]), int(g["record_interval"])
    hist = [[] for _ in range(R)]
    best_e, best_cfg, rnd, ucur = np.inf, None, 0, 0
    with sg.AnnealEngine(0) as e:
        e.set_dense(g["J"], g["h"])
        e.init_replicas(R, seed=0, s0=g["s0"])
        e.set_ladder(temps, 1)
        slot_to_rep = np.arange(R, dtype=np.int32)
        for k in range(ns):
            inv = np.argsort(slot_to_rep)  # storage replica -> slot
            out = e.sweep(1, site_mode=sg._native.SITE_REPLAY, replay_site=site[k][inv],
                          replay_u=u[k][inv], trace=True)
            assert np.array_equal(out["accept_trace"].astype(bool), acc_ref[k][inv])
            if k % ei == 0 and k > 0:
                start = int(g["exch_start"][rnd])
                npairs = len(range(start, R - 1, 2))
                uu = np.zeros(R // 2)
                uu[:npairs] = g["exch_u"][ucur:ucur + npairs]
                k_acc = e.exchange(start=[start], u=uu)
                assert k_acc == int(g["exch_accepted"][ucur:ucur + npairs].sum())
                slot_to_rep = e.slot_map()
                ucur += npairs
                rnd += 1
            if k % ri == 0:
                en = e.energies()
                for i in range(R):
                    hist[i].append(en[slot_to_rep[i]])
                i_best = int(np.argmin(en[slot_to_rep]))
                if en[slot_to_rep[i_best]] < best_e:
                    best_e = float(en[slot_to_rep[i_best]])
                    best_cfg = e.spins(int(slot_to_rep[i_best]))
        ea, ec = e.exchange_stats()
        assert np.array_equal(ea[:R - 1], g["exchange_attempts"].astype(np.int64))
        assert np.array_equal(ec[:R - 1], g["exchange_accepts"].astype(np.int64))
        assert np.array_equal(np.asarray(hist), g["energy_histories"])
        assert best_e == float(g["best_energy"])
        assert np.array_equal(best_cfg, g["best_configuration"])
        assert np.array_equal(e.spins()[slot_to_rep], g["s_final"])


# ----------------------------------------------------------------------------- errors
def test_error_paths(sg):
    with pytest.raises(sg.DeviceError):
        sg.AnnealEngine(9999)
    with sg.AnnealEngine(0) as e:
        with pytest.raises(sg.AnnealingError):
            e.init_replicas(4)  # no couplings yet
        e.set_dense(pm1(32, 1), np.zeros(32, np.float32))
        with pytest.raises(sg.AnnealingError):
            e.sweep(1)  # no replicas
        e.init_replicas(4, seed=1)
        with pytest.raises(sg.AnnealingError):
            e.exchange()  # no ladder
        with pytest.raises(sg.AnnealingError):
            e.sweep(1, site_mode=sg._native.SITE_REPLAY)  # replay arrays missing
        with pytest.raises(sg.AnnealingError):
            e.set_dense(gauss(16, 1), np.zeros(16, np.float32), storage="i8")
        with pytest.raises(sg.AnnealingError):
            e.set_csr(np.asarray([0, 2, 1], np.int32), np.asarray([0, 1], np.int32),
                      np.ones(2, np.float32), np.zeros(2, np.float32))


# ----------------------------------------------------------------------------- streaming form
@pytest.mark.parametrize("n,storage,waves", [(3000, "f32", 1), (45000, "i8", 4), (6000, "f32", 2)])
def test_streaming_form_matches_oracle(sg, n, storage, waves):
    """Rows longer than waves x 10 chunks take the streaming form of the dense kernel
    (forced here with few waves so that small instances exercise it)."""
    rng = np.random.RandomState(n)
    if n > 20000:  # keep the host-side instance cheap: banded +-1 couplings
        J = np.zeros((n, n), np.float32)
        for d in (1, 7, 300):
            v = (rng.randint(0, 2, n - d) * 2 - 1).astype(np.float32)
            J[np.arange(n - d), np.arange(d, n)] = v
            J[np.arange(d, n), np.arange(n - d)] = v
    else:
        J = pm1(n, n)
    h = rng.randint(-1, 2, n).astype(np.float32)
    prob = oracle.Problem(J=J, h=h)
    R, ns, seed = 3, 2, 7000 + n
    temps = ladder(R, 4.0, 0.5)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=8)
    with sg.AnnealEngine(0) as e:
        e.set_tuning(waves_per_replica=waves)
        e.set_dense(J, h, storage=storage)
        e.init_replicas(R, seed=seed)
        assert "(streaming)" in e.describe()
        e.set_temperatures(temps)
        out = e.sweep(ns, energy_trace=True)
        assert np.array_equal(out["energy_trace"], ref["energy_trace"])
        assert np.array_equal(e.spins(), s)
        out2 = e.sweep(1, energy_trace=True, trace=True)   # general (traced) variant too
        ref2 = oracle.sweeps(prob, s, temps, 1, seed=seed, sweep0=ns, energy=ref["energy"])
        assert np.array_equal(out2["energy_trace"], ref2["energy_trace"])


# ----------------------------------------------------------------------------- model batches
@pytest.mark.parametrize("n,M,k,storage", [(64, 5, 3, "auto"), (300, 4, 2, "f32"), (1100, 3, 4, "i8")])
def test_many_model_batch_matches_per_model_oracle(sg, n, M, k, storage):
    """sga_set_dense_batch: M stacked models, k replicas each, one launch; every replica must
    follow the oracle run on ITS model (reference BatchProcessor workload)."""
    Js = np.stack([pm1(n, 100 + m) for m in range(M)])
    hs = np.stack([np.random.RandomState(m).randint(-1, 2, n).astype(np.float32) for m in range(M)])
    R, ns, seed = M * k, 5, 999 + n
    temps = np.tile(ladder(k, 3.0, 0.5), M)
    with sg.AnnealEngine(0) as e:
        e.set_dense_batch(Js, hs, storage=storage)
        e.init_replicas(R, seed=seed)
        e.set_ladder(temps, n_ladders=M)
        out = e.sweep(ns, energy_trace=True)
        swaps = e.exchange()
        spins, energies, slot_map = e.spins(), e.energies(), e.slot_map()
        f = e.local_fields(R - 1, [0, n - 1])
        e.recompute_energies()
        assert np.array_equal(e.energies(), energies)
    n_acc = 0
    for m in range(M):
        prob = oracle.Problem(J=Js[m], h=hs[m])
        sl = slice(m * k, (m + 1) * k)
        s = oracle.init_spins(n, k, seed, replica0=m * k)
        ref = oracle.sweeps(prob, s, temps[sl], ns, seed=seed, replica0=m * k)
        assert np.array_equal(out["energy_trace"][:, sl], ref["energy_trace"])
        assert np.array_equal(spins[sl], s)
        view = np.arange(m * k, (m + 1) * k, dtype=np.int32)
        full_e = np.zeros(R)
        full_e[sl] = ref["energy"]
        n_acc += oracle.pt_exchange_round(temps[sl], full_e, view, seed=seed, round_=0, ladder=m)
        assert np.array_equal(slot_map[sl], view)
    assert swaps == n_acc
    prob = oracle.Problem(J=Js[M - 1], h=hs[M - 1])
    assert f[0] == oracle.local_field(prob, spins[R - 1], 0)
    assert f[1] == oracle.local_field(prob, spins[R - 1], n - 1)


# ----------------------------------------------------------------------------- physics
@pytest.mark.parametrize("rule", [0, 1, 2])
@pytest.mark.parametrize("storage", ["dense", "csr"])
def test_samples_follow_the_boltzmann_distribution(sg, rule, storage):
    """Independent of the oracle: at fixed T every rule must sample exp(-E/T).  The exact mean
    energy of a 14-spin instance (16384 states, enumerated) is compared with the average over
    4096 replicas after equilibration."""
    n, R, T = 14, 4096, 2.5
    rng = np.random.RandomState(3)
    J = np.triu(rng.randint(-2, 3, (n, n)), 1).astype(np.float32)
    J = J + J.T
    h = rng.randint(-1, 2, n).astype(np.float32)
    states = ((np.arange(1 << n)[:, None] >> np.arange(n)[None, :]) & 1) * 2.0 - 1.0
    E = -0.5 * np.einsum("si,ij,sj->s", states, J.astype(np.float64), states) - states @ h
    w = np.exp(-(E - E.min()) / T)
    exact_mean = float((E * w).sum() / w.sum())
    exact_var = float((E * E * w).sum() / w.sum() - exact_mean ** 2)
    with sg.AnnealEngine(0) as e:
        if storage == "csr":
            e.set_csr(*csr_of(J), h)
        else:
            e.set_dense(J, h)
        e.set_update_rule(rule)
        e.init_replicas(R, seed=1234 + rule)
        e.set_temperatures(np.full(R, T))
        e.sweep(60)                       # equilibrate
        samples = []
        for _ in range(8):
            e.sweep(5)
            samples.append(e.energies())
    est = float(np.mean(samples))
    sigma = np.sqrt(exact_var / R)        # per snapshot; snapshots are correlated: be generous
    assert abs(est - exact_mean) < 5 * sigma, (est, exact_mean, sigma)


def test_replica_exchange_preserves_every_slots_distribution(sg):
    """Label swapping must leave slot i at temperature T_i in equilibrium: with 512 ladders of
    8 temperatures the per-slot mean energies are compared with the exact Boltzmann means."""
    n, L, ladders = 12, 8, 512
    R = L * ladders
    rng = np.random.RandomState(5)
    J = np.triu(rng.randint(-2, 3, (n, n)), 1).astype(np.float32)
    J = J + J.T
    h = rng.randint(-1, 2, n).astype(np.float32)
    temps_one = ladder(L, 6.0, 0.8)
    states = ((np.arange(1 << n)[:, None] >> np.arange(n)[None, :]) & 1) * 2.0 - 1.0
    E = -0.5 * np.einsum("si,ij,sj->s", states, J.astype(np.float64), states) - states @ h
    exact, var = [], []
    for T in temps_one:
        w = np.exp(-(E - E.min()) / T)
        m = (E * w).sum() / w.sum()
        exact.append(m)
        var.append((E * E * w).sum() / w.sum() - m * m)
    with sg.AnnealEngine(0) as e:
        e.set_dense(J, h)
        e.init_replicas(R, seed=99)
        e.set_ladder(np.tile(temps_one, ladders), n_ladders=ladders)
        for _ in range(30):               # equilibrate with exchanges
            e.sweep(2)
            e.exchange(count=False)
        acc = np.zeros(L)
        rounds = 12
        for _ in range(rounds):
            e.sweep(2)
            e.exchange(count=False)
            en = e.energies()[e.slot_map()].reshape(ladders, L)   # energy found in each slot
            acc += en.mean(0)
        att, ok = e.exchange_stats()
        assert ok.sum() > 0.2 * att.sum()  # exchanges really happen
        assert np.array_equal(np.sort(e.slot_map().reshape(ladders, L), axis=1) % L,
                              np.tile(np.arange(L), (ladders, 1)))  # swaps stay inside a ladder
    est = acc / rounds
    sigma = np.sqrt(np.asarray(var) / ladders)
    assert np.all(np.abs(est - np.asarray(exact)) < 5 * sigma), (est, exact, sigma)


# ----------------------------------------------------------------------------- bit-plane couplings
T2_CASES = [(64, 4, 0, 1.0), (300, 5, 0, 0.3), (5000, 3, 0, 1.0), (10000, 3, 2, 1.0),
            (20000, 2, 3, 0.05), (8193, 3, 0, 1.0), (70000, 2, 0, 0.001)]


@pytest.mark.parametrize("n,R,waves,density", T2_CASES)
def test_ternary_bit_plane_storage_matches_oracle(sg, n, R, waves, density):
    """J in {-1, 0, +1} held as two bit-planes (popcount row sums) must run the oracle's chain."""
    rng = np.random.RandomState(n)
    if n <= 10000:
        J = np.triu((rng.randint(0, 2, (n, n)) * 2 - 1) * (rng.rand(n, n) < density), 1).astype(np.float32)
        J = J + J.T
        prob_args = dict(J=J)
    else:  # large n: a banded / scattered ternary matrix, oracle through CSR
        import scipy.sparse as sp
        m = int(n * n * density / 2)
        i, j = rng.randint(0, n, m), rng.randint(0, n, m)
        keep = i < j
        up = sp.coo_matrix(((rng.randint(0, 2, keep.sum()) * 2 - 1).astype(np.float32),
                            (i[keep], j[keep])), shape=(n, n)).tocsr()
        up.data = np.sign(up.data).astype(np.float32)       # merged duplicates back to +-1 / 0
        A = (up + up.T).tocsr()
        A.eliminate_zeros()
        A.sort_indices()
        J = torch_dense = None
        prob_args = dict(csr=(A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data))
    h = rng.randint(-1, 2, n).astype(np.float32)
    prob = oracle.Problem(h=h, **prob_args)
    ns, seed = 3, 4321 + n
    temps = ladder(R, 4.0, 0.6)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=8)
    with sg.AnnealEngine(0) as e:
        e.set_tuning(waves_per_replica=waves)
        if J is None:
            import torch
            Jt = torch.sparse_csr_tensor(torch.from_numpy(A.indptr.astype(np.int64)),
                                         torch.from_numpy(A.indices.astype(np.int64)),
                                         torch.from_numpy(A.data), size=(n, n)).to_dense().cuda()
            e.set_dense(Jt, h, storage="t2")
            del Jt
        else:
            e.set_dense(J, h, storage="t2")
        e.init_replicas(R, seed=seed)
        assert "storage=t2" in e.describe()
        e.set_temperatures(temps)
        out = e.sweep(ns, energy_trace=True)
        assert np.array_equal(out["energy_trace"], ref["energy_trace"]), e.describe()
        assert np.array_equal(e.spins(), s)
        assert np.array_equal(e.stats()[0], ref["n_accepted"])
        for r in range(R):
            be, bs, _ = e.best(r)
            assert be == ref["best_energy"][r] and np.array_equal(bs, ref["best_spins"][r])
        e.recompute_energies()
        assert np.array_equal(e.energies(), ref["energy"])
        # traced sweeps fall back to the int8 layout of the same couplings
        out2 = e.sweep(1, energy_trace=True, trace=True)
        ref2 = oracle.sweeps(prob, s, temps, 1, seed=seed, sweep0=ns, energy=ref["energy"])
        assert np.array_equal(out2["energy_trace"], ref2["energy_trace"])


@pytest.mark.parametrize("n,R,storage,waves", [
    (3, 3, "f32", 0), (7, 5, "i8", 0), (64, 8, "f32", 0), (65, 4, "t2", 0), (257, 6, "f32", 2),
    (700, 5, "f32", 3), (1023, 4, "i8", 1), (2500, 3, "i8", 3), (3000, 3, "t2", 1), (9000, 2, "t2", 2),
    # four chunks per wave (two for bit-planes): the 256-thread builds
    (1000, 3, "f32", 1), (2040, 2, "f32", 2), (4000, 3, "i8", 1), (9000, 2, "t2", 1), (30000, 2, "t2", 2),
    # five / six chunks per wave (three / four for bit-planes): two updates per batch
    (1500, 3, "f32", 1), (2600, 2, "f32", 2), (5000, 2, "i8", 1), (20000, 2, "t2", 1), (25000, 2, "t2", 1),
])
def test_look_ahead_form_equals_one_update_at_a_time(sg, n, R, storage, waves):
    """Integer problems with short rows reduce four consecutive updates together and replay the
    chain on scalars (sweep_dense_impl.h).  Same spins, energies, counters and best states as the
    oracle's strictly sequential chain and as the kernel's own one-at-a-time form -- including
    repeated sites inside a batch (tiny n), odd n, fields, and several waves per replica."""
    opts = {}  # engine options (sga_set_option): which kernel form runs, never what it computes
    rng = np.random.RandomState(n)
    J = pm1(n, 100 + n)
    if storage == "t2":
        J = J * (rng.rand(n, n) < 0.6)
        J = np.triu(J, 1)
        J = (J + J.T).astype(np.float32)
    h = rng.randint(-2, 3, n).astype(np.float32)
    prob = oracle.Problem(J=J, h=h)
    ns, seed = 6, 31337 + n
    temps = ladder(R, 3.0 * np.sqrt(n), 0.3)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=8)
    got = {}
    for look in (True, False):
        if look:
            opts.pop("look_ahead", None)
        else:
            opts["look_ahead"] = 0
        with sg.AnnealEngine(0) as e:
            e.set_options(opts)
            if waves:
                e.set_tuning(waves_per_replica=waves)
            e.set_dense(J, h, storage=storage)
            e.init_replicas(R, seed=seed)
            batched = "look_ahead=4" in e.describe() or "look_ahead=2" in e.describe()
            assert batched == look, e.describe()
            e.set_temperatures(temps)
            out = e.sweep(ns, energy_trace=True)
            assert np.array_equal(out["energy_trace"], ref["energy_trace"])
            assert np.array_equal(e.spins(), s)
            assert np.array_equal(e.stats()[0], ref["n_accepted"])
            for r in range(R):
                be, bs, _ = e.best(r)
                assert be == ref["best_energy"][r] and np.array_equal(bs, ref["best_spins"][r])
            got[look] = e.energies()
    assert np.array_equal(got[True], got[False])


@pytest.mark.parametrize("n,R,storage", [(700, 6, "f32"), (3000, 5, "i8"), (9000, 3, "t2"), (5000, 4, "f32")])
def test_autotune_keeps_the_chain_and_the_state(sg, n, R, storage):
    """sga_autotune times every feasible waves-per-replica on the live replicas.  Called in the
    middle of a run it must leave spins, energies, best states, acceptance counters and the random
    stream exactly where they were: the continued run equals the oracle's uninterrupted one."""
    rng = np.random.RandomState(n)
    J = pm1(n, 7 + n)
    if storage == "t2":
        J = np.triu(J * (rng.rand(n, n) < 0.5), 1)
        J = (J + J.T).astype(np.float32)
    if storage == "f32" and n == 5000:
        J = gauss(n, 3)                                   # real couplings: general path
    h = (rng.randn(n) if n == 5000 else rng.randint(-1, 2, n)).astype(np.float32)
    prob = oracle.Problem(J=J, h=h)
    seed = 4242 + n
    temps = ladder(R, 2.0 * np.sqrt(n), 0.5)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, 5, seed=seed, n_threads=8)
    with sg.AnnealEngine(0) as e:
        e.set_dense(J, h, storage=storage)
        e.init_replicas(R, seed=seed)
        e.set_ladder(temps)
        a = e.sweep(2, energy_trace=True)
        before = e.describe()
        ms = e.autotune()
        assert ms > 0.0
        b = e.sweep(3, energy_trace=True)
        assert np.array_equal(np.concatenate([a["energy_trace"], b["energy_trace"]]), ref["energy_trace"])
        assert np.array_equal(e.spins(), s)
        assert np.array_equal(e.stats()[0], ref["n_accepted"])
        assert e.counters()[0] == 5
        for r in range(R):
            be, bs, _ = e.best(r)
            assert be == ref["best_energy"][r] and np.array_equal(bs, ref["best_spins"][r])
        assert e.describe().split("waves_per_replica")[0] == before.split("waves_per_replica")[0]


def test_bit_plane_storage_needs_ternary_couplings(sg):
    with sg.AnnealEngine(0) as e:
        with pytest.raises(sg.AnnealingError):
            e.set_dense(pm1(32, 1) * 2.0, np.zeros(32, np.float32), storage="t2")
        e.set_dense(pm1(5000, 1), np.zeros(5000, np.float32))       # auto: ternary and n >= 4096
        e.init_replicas(2, seed=1)
        assert "storage=t2" in e.describe()



def test_two_engines_of_one_process_run_different_forms(sg):
    """sga_set_option is per engine: the same CSR problem swept by two engines side by side -- several updates per
    step on int8 spins in one, one update at a time on bit spins in the other -- gives the same chain; unknown
    keys and values out of range are refused."""
    from spin_glass_anneal_rl_amd.engine import last_kernel, option_names
    n, R, ns, seed = 700, 8, 4, 2024
    rng = np.random.RandomState(n)
    J = np.zeros((n, n), np.float32)
    for i in range(n):
        for j in rng.choice(n, 5, replace=False):
            if i != j:
                J[i, j] = J[j, i] = float(rng.choice([-2.0, -1.0, 1.0, 2.0]))
    h = rng.randint(-1, 2, n).astype(np.float32)
    csr = csr_of(J)
    temps = ladder(R, 6.0, 0.3)
    prob = oracle.Problem(csr=csr, h=h)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=R)
    with sg.AnnealEngine(0) as a, sg.AnnealEngine(0) as b:
        assert a.get_option("csr_updates_per_step") == -1 and a.get_option("look_ahead") == 1
        b.set_options(csr_updates_per_step=0, force_csr_bits=1)
        assert b.get_option("csr_updates_per_step") == 0 and a.get_option("csr_updates_per_step") == -1
        kernels = []
        for e in (a, b):
            e.set_csr(*csr, h)
            e.init_replicas(R, seed=seed)
            e.set_temperatures(temps)
        for e in (a, b):      # interleaved calls: nothing is latched per process
            out = e.sweep(ns, energy_trace=True)
            kernels.append(last_kernel())
            assert np.array_equal(out["energy_trace"], ref["energy_trace"]), kernels[-1]
            assert np.array_equal(e.spins(), s)
        assert "sweep_csr_rows_kernel" in kernels[0] and "sweep_csr_rows_kernel" not in kernels[1], kernels
        assert "spins=lds-bits" in b.describe() and "spins=lds-int8" in a.describe()
        with pytest.raises(sg.AnnealingError, match="unknown option"):
            a.set_option("no_such_option", 1)
        with pytest.raises(sg.AnnealingError, match="outside"):
            a.set_option("clf_waves", 99)  # (beyond the documented range)
        assert set(option_names()) >= {"look_ahead", "clf_waves", "sparse_route", "batched_energy"}


@pytest.mark.parametrize("longest", [100, 200, 256])
def test_csr_medium_rows_with_a_short_last_row(sg, longest):
    """Rows of 65 ... 256 entries run four updates per step with 8 | 16 entries per lane, every lane of a row loading
    its entries without a bounds test: a row that starts within the last 255 entries of the layout reaches past the
    array -- into the 256 zeroed entries the engine keeps behind it.  The LAST row here holds one entry, the rows
    before it few, the first one `longest`: every update at the late sites reads beyond the last entry."""
    from spin_glass_anneal_rl_amd.engine import last_kernel
    n = longest + 40
    rng = np.random.RandomState(longest)
    J = np.zeros((n, n), np.float32)
    hub = 0                                    # one long row (and its column)
    for j in rng.choice(np.arange(1, n - 1), longest - 1, replace=False):
        J[hub, j] = J[j, hub] = float(rng.choice([-2.0, -1.0, 1.0, 3.0]))
    J[hub, n - 1] = J[n - 1, hub] = 1.0        # the last row: exactly this one entry
    for i in range(1, n - 1, 7):               # a few more short rows
        j = (i * 5) % (n - 2) + 1
        if i != j:
            J[i, j] = J[j, i] = -1.0
    h = rng.randint(-1, 2, n).astype(np.float32)
    csr = csr_of(J)
    lens = np.diff(csr[0])
    assert lens.max() == longest and lens[-1] == 1
    R, ns, seed = 5, 8, 31 + longest
    temps = ladder(R, 6.0, 0.5)
    prob = oracle.Problem(csr=csr, h=h)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=R)
    with sg.AnnealEngine(0) as e:
        e.set_csr(*csr, h)
        e.init_replicas(R, seed=seed)
        e.set_temperatures(temps)
        out = e.sweep(ns, energy_trace=True)
        k = last_kernel()
        assert "sweep_csr_rows_kernel" in k and ("16 entries per lane" if longest > 128 else "8 entries per lane") in k, k
        assert np.array_equal(out["energy_trace"], ref["energy_trace"]), k
        assert np.array_equal(e.spins(), s)
        assert np.array_equal(e.stats()[0], ref["n_accepted"])


@pytest.mark.parametrize("n,deg,R", [(1200, 30, 48), (900, 220, 40), (2500, 700, 24)])
def test_csr_autotune_keeps_the_chain_and_the_state(sg, n, deg, R):
    """sga_autotune on a CSR problem: every admissible sweep form (waves per replica, several updates per step or one)
    is timed on the live replicas through the geometry-independent state blob; the run continues as if nothing had
    happened -- spins, energies, bests, counters, ladder permutation -- and stays on the oracle's chain."""
    from spin_glass_anneal_rl_amd.engine import last_kernel
    rng = np.random.RandomState(n + deg)
    J = np.zeros((n, n), np.float32)
    for i in range(n):
        for j in rng.choice(n, deg // 2, replace=False):
            if i != j:
                J[i, j] = J[j, i] = float(rng.choice([-2.0, -1.0, 1.0, 2.0]))
    h = rng.randint(-1, 2, n).astype(np.float32)
    csr = csr_of(J)
    temps = ladder(R, 2.0 * np.sqrt(deg), 0.1 * np.sqrt(deg))
    seed = 99 + n
    prob = oracle.Problem(csr=csr, h=h)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, 3, seed=seed, n_threads=8)
    with sg.AnnealEngine(0) as e:
        e.set_csr(*csr, h)
        e.init_replicas(R, seed=seed)
        e.set_ladder(temps)
        e.sweep(3)
        e.exchange()
        before = (e.energies(), e.spins(), e.stats()[0], e.slot_map(), e.counters(), e.best()[0], e.temperatures())
        k0 = last_kernel()
        ms = e.autotune()
        assert ms > 0.0
        after = (e.energies(), e.spins(), e.stats()[0], e.slot_map(), e.counters(), e.best()[0], e.temperatures())
        for x, y in zip(before, after):
            assert np.array_equal(np.asarray(x), np.asarray(y))
        assert np.array_equal(before[1], s) and np.array_equal(before[0], ref["energy"])
        # the run goes on, on whatever form was picked, exactly like one that was never tuned
        e.sweep(3)
        with sg.AnnealEngine(0) as plain:
            plain.set_csr(*csr, h)
            plain.init_replicas(R, seed=seed)
            plain.set_ladder(temps)
            plain.sweep(3)
            plain.exchange()
            plain.sweep(3)
            assert np.array_equal(plain.energies(), e.energies()) and np.array_equal(plain.spins(), e.spins())
        print(n, deg, "before:", k0, "| picked:", last_kernel(), f"| {ms:.4f} ms per sweep")
