"""The cached-local-field sweep (sga_set_field_cache, csrc/sweep_clf_impl.h) and the all-replica
field pass on the matrix cores (csrc/fields_dense.hip), through the C ABI, against the CPU oracle
and the reference's golden vectors.

The variant reads a coupling row only when a proposal is accepted, but its chain must be the
one-row-per-proposal chain bit for bit: every test here compares it with the same oracle runs and
reference fixtures as tests/test_engine_gpu.py does for the row-per-proposal kernels.
"""
import numpy as np
import pytest

import oracle
from oracle_follow import follow, ladder_ends
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


def int_couplings(n, seed, amp, density=1.0):
    rng = np.random.RandomState(seed)
    J = np.triu(rng.randint(-amp, amp + 1, (n, n)) * (rng.rand(n, n) < density), 1).astype(np.float32)
    return J + J.T


def ladder(R, tmax=10.0, tmin=0.1):
    return np.asarray([tmax * (tmin / tmax) ** (i / max(R - 1, 1)) for i in range(R)])


def check_against(e, ref, s, out):
    assert np.array_equal(out["energy_trace"], ref["energy_trace"]), e.describe()
    assert np.array_equal(e.spins(), s)
    assert np.array_equal(e.energies(), ref["energy"])
    assert np.array_equal(e.stats()[0], ref["n_accepted"])
    for r in range(len(s)):
        be, bs, _ = e.best(r)
        assert be == ref["best_energy"][r] and np.array_equal(bs, ref["best_spins"][r])
    e.recompute_energies()
    assert np.array_equal(e.energies(), ref["energy"])


# ----------------------------------------------------------------------------- Philox streams
CLF_CASES = [
    # (n, R, storage, clf waves: 0 = heuristic)
    (64, 8, "f32", 0), (64, 8, "i8", 0), (63, 5, "f32", 0), (1, 3, "f32", 0), (2, 4, "i8", 0),
    (129, 33, "i8", 0), (300, 6, "f32", 2), (1000, 40, "f32", 0), (1000, 7, "i8", 3),
    (1100, 130, "f32", 5), (2500, 6, "f32", 0), (2500, 3, "i8", 2), (4100, 3, "i8", 0),
    (4000, 4, "f32", 16), (10000, 2, "f32", 0), (10000, 3, "i8", 0), (10000, 2, "i8", 1),
    (10000, 2, "t2", 0), (10000, 2, "f32", 4),  # (the last: rows longer than one batch of chunks per wave)
]


@pytest.mark.parametrize("n,R,storage,waves", CLF_CASES)
def test_cached_field_sweeps_match_oracle(sg, n, R, storage, waves):
    opts = {}  # engine options (sga_set_option): which kernel form runs, never what it computes
    if waves:
        opts["clf_waves"] = int(waves)
    J = pm1(n, 10 + n)
    h = np.random.RandomState(n).randint(-1, 2, n).astype(np.float32)
    prob = oracle.Problem(J=J, h=h)
    ns = 4 if n >= 2500 else 12
    temps = ladder(R, 6.0, 0.5)
    seed = 0xC1F0000 + n
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=8)
    from spin_glass_anneal_rl_amd.engine import last_kernel
    # one accept per round (sweep_clf_kernel) and several (sweep_clfb_kernel: what the default -- clf_batched = 2, by the
    # hottest replica's acceptance -- starts a run with) walk the same cases
    forms = [dict(opts, clf_batched=0), dict(opts, clf_batched=1), opts]
    for form in forms:
        with sg.AnnealEngine(0) as e:
            e.set_options(form)
            e.set_field_cache("on")
            e.set_dense(J, h, storage=storage)
            e.init_replicas(R, seed=seed)
            assert "sweep=cached-local-fields" in e.describe(), e.describe()
            e.set_temperatures(temps)
            out = e.sweep(ns, energy_trace=True)
            # (the default picks by the hottest replica's acceptance: either windowed form)
            want = ("sweep_clf" if "clf_batched" not in form else
                    "sweep_clfb_kernel" if form["clf_batched"] else "sweep_clf_kernel")
            assert last_kernel().startswith(want), (want, last_kernel())
            check_against(e, ref, s, out)


@pytest.mark.parametrize("amp,n,bits", [(3, 500, 16), (100, 1000, 32), (127, 2000, 32)])
@pytest.mark.parametrize("storage", ["f32", "i8"])
def test_cached_fields_int16_and_int32(sg, amp, n, bits, storage):
    """Row sums beyond int16 keep the fields as int32; hot and cold replicas; non-zero h."""
    J = int_couplings(n, amp + n, amp, density=0.7)
    h = np.random.RandomState(amp).randint(-amp, amp + 1, n).astype(np.float32)
    prob = oracle.Problem(J=J, h=h)
    R, ns, seed = 6, 5, 77 + amp
    temps = ladder(R, 40.0 * amp, 0.5 * amp)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=6)
    for form in ({"clf_batched": 0}, {"clf_batched": 1}):          # one accept per round | several
        with sg.AnnealEngine(0) as e:
            e.set_options(form)
            e.set_field_cache("on")
            e.set_dense(J, h, storage=storage)
            e.init_replicas(R, seed=seed)
            assert f"cached-local-fields(int{bits}" in e.describe(), e.describe()
            e.set_temperatures(temps)
            out = e.sweep(ns, energy_trace=True)
            check_against(e, ref, s, out)


@pytest.mark.parametrize("amp,n,bits", [(2, 300, 16), (120, 700, 32)])
def test_cached_fields_with_half_integer_fields(sg, amp, n, bits):
    """h in multiples of 1/2 (penalty encodings of 0/1 variables): the fields are kept doubled (scale 2),
    the accept table at twice the resolution; dE values are integers and half-integers' doubles."""
    J = int_couplings(n, 3 * n, amp, density=0.6)
    h = np.random.RandomState(n).randint(-2 * amp, 2 * amp + 1, n).astype(np.float32) / 2.0
    assert np.any(h != np.rint(h))
    prob = oracle.Problem(J=J, h=h)
    R, ns, seed = 5, 6, 909 + amp
    temps = ladder(R, 30.0 * amp, 0.4 * amp)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, trace=True, n_threads=5)
    for storage, form in (("f32", {"clf_batched": 0}), ("i8", {"clf_batched": 0}), ("f32", {"clf_batched": 1}), ("i8", {"clf_batched": 1})):
        with sg.AnnealEngine(0) as e:
            e.set_options(form)
            e.set_field_cache("on")
            e.set_dense(J, h, storage=storage)
            e.init_replicas(R, seed=seed)
            assert f"cached-local-fields(int{bits}" in e.describe(), e.describe()
            e.set_temperatures(temps)
            out = e.sweep(ns, energy_trace=True)                      # production build (table)
            assert np.array_equal(out["energy_trace"], ref["energy_trace"])
            assert np.array_equal(e.spins(), oracle_spins_after(prob, R, n, seed, temps, ns))
        with sg.AnnealEngine(0) as e:
            e.set_field_cache("on")
            e.set_dense(J, h, storage=storage)
            e.init_replicas(R, seed=seed)
            e.set_temperatures(temps)
            out = e.sweep(ns, energy_trace=True, trace=True)           # general build (per-update records)
            assert np.array_equal(out["accept_trace"], ref["accept_trace"])
            assert np.array_equal(out["dE_trace"], ref["dE_trace"])
            assert np.array_equal(e.spins(), s)


def oracle_spins_after(prob, R, n, seed, temps, ns):
    s = oracle.init_spins(n, R, seed)
    oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=R)
    return s


@pytest.mark.parametrize("n,R,amp,hot", [(97, 6, 1, 3.0), (700, 9, 1, 0.6), (2100, 5, 3, 2.0), (5000, 4, 1, 0.3)])
def test_windowed_forms_walk_the_same_chain_on_a_hot_ladder(sg, n, R, amp, hot):
    """The windowed form (one accept per round), its several-accepts-per-round variant and the default (picked by the
    hottest replica's acceptance) give the oracle's chain on ladders with a hot end; small n: every site is proposed
    several times per window."""
    from spin_glass_anneal_rl_amd.engine import last_kernel
    J = int_couplings(n, 7 * n, amp, density=0.8) if amp > 1 else pm1(n, 3 * n)
    h = np.random.RandomState(n).randint(-amp, amp + 1, n).astype(np.float32)
    prob = oracle.Problem(J=J, h=h)
    ns, seed = 6, 0xC4A1 + n
    temps = ladder(R, hot * amp * np.sqrt(n), 0.02 * amp * np.sqrt(n))
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=8)
    assert ref["n_accepted"].max() > n // 2         # a hot end: several accepts per window there
    for opts in ({}, {"clf_batched": 0}, {"clf_batched": 1}):
        with sg.AnnealEngine(0) as e:
            e.set_options(opts)
            e.set_field_cache("on")
            e.set_dense(J, h)
            e.init_replicas(R, seed=seed)
            e.set_temperatures(temps)
            out = e.sweep(ns, energy_trace=True)
            k = last_kernel()
            assert k.startswith("sweep_clf" if "clf_batched" not in opts else
                                "sweep_clfb_kernel" if opts["clf_batched"] else "sweep_clf_kernel"), k
            check_against(e, ref, s, out)


# ----------------------------------------------------------------------------- golden replay
SWEEP_CASES = ["sweeps_pm1_n8", "sweeps_pm1_n16", "sweeps_pm1_n64", "sweeps_pm1_n64_cold",
               "sweeps_field_n64", "sweeps_pm1_n300"]


@pytest.mark.parametrize("storage", ["f32", "i8"])
@pytest.mark.parametrize("name", SWEEP_CASES)
def test_reference_sweeps_replayed_with_cached_fields(sg, name, storage):
    """The reference's own recorded streams (sites, uniforms) -> its decisions, dE and energies."""
    g = load_golden(name)
    n, ns = g["J"].shape[0], int(g["n_sweeps"])
    u = np.nan_to_num(g["u"], nan=2.0).astype(np.float32)
    with sg.AnnealEngine(0) as e:
        e.set_field_cache("on")
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
        assert e.stats()[0][0] == int(g["n_accepted"])


RULE_CASES = [("sweeps_glauber_n64", 1), ("sweeps_heatbath_n64", 2)]


@pytest.mark.parametrize("name,rule", RULE_CASES)
def test_reference_glauber_heatbath_replayed_with_cached_fields(sg, name, rule):
    g = load_golden(name)
    n, ns = g["J"].shape[0], int(g["n_sweeps"])
    with sg.AnnealEngine(0) as e:
        e.set_field_cache("on")
        e.set_dense(g["J"], g["h"])
        e.init_replicas(1, seed=0, s0=g["s0"][None, :])
        e.set_temperatures([float(g["T"])])
        e.set_update_rule(rule)
        out = e.sweep(ns, site_mode=sg._native.SITE_REPLAY, replay_site=g["site"][None, :],
                      replay_u=g["u"].astype(np.float32)[None, :], energy_trace=True, trace=True)
        assert np.array_equal(out["accept_trace"][0].astype(bool), g["accepted"])
        assert np.array_equal(out["dE_trace"][0], g["dE"])  # heat bath: minus the change
        assert np.array_equal(out["energy_trace"][:, 0], g["sweep_energy"])
        assert np.array_equal(e.spins(0), g["s_final"])


@pytest.mark.parametrize("rule", [1, 2])
@pytest.mark.parametrize("n,R,storage", [(64, 6, "f32"), (1100, 4, "i8")])
def test_cached_fields_glauber_heatbath_match_oracle(sg, n, R, storage, rule):
    J = pm1(n, 3 + n)
    h = np.random.RandomState(n).randint(-2, 3, n).astype(np.float32)
    prob = oracle.Problem(J=J, h=h)
    ns, seed = 5, 555 + n
    temps = ladder(R, 30.0, 3.0)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, rule=rule, trace=True, n_threads=4)
    with sg.AnnealEngine(0) as e:
        e.set_field_cache("on")
        e.set_dense(J, h, storage=storage)
        e.init_replicas(R, seed=seed)
        e.set_temperatures(temps)
        e.set_update_rule(rule)
        out = e.sweep(ns, energy_trace=True, trace=True)
        assert np.array_equal(out["accept_trace"], ref["accept_trace"])
        assert np.array_equal(out["dE_trace"], ref["dE_trace"])
        assert np.array_equal(out["energy_trace"], ref["energy_trace"])
        assert np.array_equal(e.spins(), s)


def test_cached_fields_sequential_order_and_operator_arithmetic(sg):
    """The operator fallback's order and fp32 arithmetic (annealing/cuda_kernels.py:381-390)."""
    n, R, ns = 130, 4, 5
    J, h = pm1(n, 6), np.random.RandomState(1).randint(-1, 2, n).astype(np.float32)
    prob = oracle.Problem(J=J, h=h)
    seed = 31337
    u = np.random.RandomState(0).rand(R, ns * n).astype(np.float32)
    for arith in (oracle.ARITH_F64, oracle.ARITH_F32):
        s = oracle.init_spins(n, R, seed)
        ref = oracle.sweeps(prob, s, 1.3, ns, site_mode=oracle.SITE_SEQUENTIAL, arith=arith,
                            replay_u=u, seed=seed, trace=True)
        with sg.AnnealEngine(0) as e:
            e.set_field_cache("on")
            e.set_dense(J, h, storage="f32")
            e.init_replicas(R, seed=seed)
            e.set_temperatures(np.full(R, 1.3))
            out = e.sweep(ns, site_mode=sg._native.SITE_SEQUENTIAL, arith=arith, replay_u=u,
                          energy_trace=True, trace=True)
            assert np.array_equal(out["accept_trace"], ref["accept_trace"])
            assert np.array_equal(out["dE_trace"], ref["dE_trace"])
            assert np.array_equal(out["energy_trace"], ref["energy_trace"])
            assert np.array_equal(e.spins(), s)


# ----------------------------------------------------------------------------- state handling
def test_cached_fields_survive_everything_that_moves_spins(sg):
    """Single-site operators, set_spins, exchanges, checkpoints and switching the cache off and on
    between calls: the chain stays the row-per-proposal chain."""
    n, R, seed = 700, 5, 2024
    J = int_couplings(n, 5, 4)
    h = np.random.RandomState(2).randint(-3, 4, n).astype(np.float32)
    temps = ladder(R, 60.0, 4.0)

    def run(cache_plan):
        with sg.AnnealEngine(0) as e:
            e.set_dense(J, h)
            e.init_replicas(R, seed=seed)
            e.set_ladder(temps)
            log = []
            e.set_field_cache(cache_plan[0])
            log.append(e.sweep(3, energy_trace=True)["energy_trace"])
            e.flip(1, 17)
            acc, dE = e.update(2, 5, 3.0, 0.25)
            log.append(np.asarray([float(acc), dE]))
            snew = -e.spins(3)
            e.set_spins(3, snew)
            e.set_field_cache(cache_plan[1])
            log.append(e.sweep(2, energy_trace=True)["energy_trace"])
            log.append(np.asarray([e.exchange()], float))
            blob = e.export_state()
            e.set_field_cache(cache_plan[2])
            log.append(e.sweep(2, energy_trace=True)["energy_trace"])
            after = e.energies().copy()
            e.import_state(blob)
            again = e.sweep(2, energy_trace=True)["energy_trace"]
            assert np.array_equal(again, log[-1]) and np.array_equal(e.energies(), after)
            tracked = e.energies()
            e.recompute_energies()
            assert np.array_equal(e.energies(), tracked)
            return log, e.spins(), e.stats()[0]

    base = run(("off", "off", "off"))
    for plan in (("on", "on", "on"), ("on", "off", "on"), ("off", "on", "auto")):
        got = run(plan)
        for a, b in zip(base[0], got[0]):
            assert np.array_equal(a, b), plan
        assert np.array_equal(base[1], got[1]) and np.array_equal(base[2], got[2])


def test_cached_fields_need_an_integer_symmetric_problem(sg):
    n = 96
    rng = np.random.RandomState(0)
    Jg = np.triu(rng.randn(n, n), 1).astype(np.float32)
    Jg = Jg + Jg.T
    Ja = pm1(n, 1)
    Ja[3, 5] += 1.0  # asymmetric
    Jd = pm1(n, 2)
    Jd[4, 4] = 2.0   # diagonal
    hz = np.zeros(n, np.float32)
    for J, h in ((Jg, hz), (Ja, hz), (Jd, hz), (pm1(n, 3), hz + 0.25)):  # (h + 0.5 would do: next test)
        with sg.AnnealEngine(0) as e:
            e.set_dense(J, h)
            e.init_replicas(3, seed=1)
            e.set_field_cache("on")
            with pytest.raises(sg.AnnealingError, match="cached local fields"):
                e.sweep(1)
            e.set_field_cache("auto")  # falls back to the row-per-proposal kernels
            assert "cached-local-fields" not in e.describe()
            a = e.sweep(2, energy_trace=True)["energy_trace"]
        with sg.AnnealEngine(0) as e:
            e.set_dense(J, h)
            e.init_replicas(3, seed=1)
            assert np.array_equal(e.sweep(2, energy_trace=True)["energy_trace"], a)
    with sg.AnnealEngine(0) as e:  # the Wolff rule keeps its own kernel
        e.set_dense(pm1(n, 3), hz)
        e.init_replicas(2, seed=1)
        e.set_field_cache("on")
        e.set_update_rule(3)
        e.sweep(1)
    with sg.AnnealEngine(0) as e:
        with pytest.raises(sg.AnnealingError):
            sg._native.check(e._lib.sga_set_field_cache(e._h, 7), "sga_set_field_cache")


def test_auto_routes_every_replica_by_its_own_acceptance(sg):
    """SGA_FIELD_CACHE_AUTO starts on the row-per-proposal kernels, looks at the acceptance counters every few
    sweeps and then routes each replica by ITS OWN acceptance: a cold ladder ends on the cached-field sweep, a hot
    one stays on the row kernels, and a ladder with a hot end runs as a MIXED launch -- the cached-field kernel and
    the row-per-proposal kernel side by side over disjoint replica lists -- with the chain of "off" in every case.
    Option "replica_routing" = 0 keeps one launch, decided by the hottest replica."""
    from spin_glass_anneal_rl_amd.engine import last_kernel
    n, R, seed = 1500, 48, 99
    J = pm1(n, 4)
    h = np.zeros(n, np.float32)
    rt = np.sqrt(n)

    def kind(k):
        return "mixed" if k.startswith("mixed launch") else k.split("<")[0]

    runs = {}
    for name, temps in (("cold", ladder(R, 0.05 * rt, 0.002 * rt)), ("hot", ladder(R, 40.0 * rt, 10.0 * rt)),
                        ("mixed", ladder(R, 20.0 * rt, 0.002 * rt))):
        for cache in ("auto", "off", "auto-one-launch"):
            with sg.AnnealEngine(0) as e:
                e.set_field_cache(cache.split("-")[0])
                if cache == "auto-one-launch":
                    e.set_option("replica_routing", 0)
                e.set_dense(J, h)
                e.init_replicas(R, seed=seed)
                e.set_ladder(temps)
                kernels, trace = [], []
                for _ in range(12):
                    trace.append(e.sweep(2, energy_trace=True)["energy_trace"])
                    kernels.append(kind(last_kernel()))
                    e.exchange(count=False)
                tracked = e.energies()
                e.recompute_energies()
                assert np.array_equal(e.energies(), tracked), (name, cache)
                runs[name, cache] = (np.vstack(trace), e.spins(), kernels, e.describe(), last_kernel(), e.stats()[0])
        for cache in ("auto", "auto-one-launch"):
            assert np.array_equal(runs[name, cache][0], runs[name, "off"][0]), (name, cache)
            assert np.array_equal(runs[name, cache][1], runs[name, "off"][1]), (name, cache)
            assert np.array_equal(runs[name, cache][5], runs[name, "off"][5]), (name, cache)
        assert set(runs[name, "off"][2]) == {"sweep_dense_kernel"}
    assert runs["cold", "auto"][2][0] == "sweep_dense_kernel"          # until the acceptance is known
    # (either windowed form: several accepts per round while the hottest replica still accepts more than 1 %)
    assert runs["cold", "auto"][2][-1] in ("sweep_clf_kernel", "sweep_clfb_kernel") and f"now: {R} of {R} replica(s) cached" in runs["cold", "auto"][3]
    assert set(runs["hot", "auto"][2]) == {"sweep_dense_kernel"} and f"now: 0 of {R} replica(s)" in runs["hot", "auto"][3]
    # the ladder with a hot end: both kernels in one sweep call, each on its own replicas
    assert runs["mixed", "auto"][2][-1] == "mixed", runs["mixed", "auto"][2]
    last = runs["mixed", "auto"][4]
    assert "sweep_clf" in last and "sweep_dense_kernel" in last and "||" in last, last
    import re
    m = re.search(r"now: (\d+) of", runs["mixed", "auto"][3])
    assert m and 0 < int(m.group(1)) < R, runs["mixed", "auto"][3]
    # one launch for all: the hot end keeps everybody on the row kernels
    assert set(runs["mixed", "auto-one-launch"][2]) == {"sweep_dense_kernel"}, runs["mixed", "auto-one-launch"][2]


def test_field_cache_request_after_a_sparse_matrix_was_taken_as_csr(sg):
    """sga_set_dense keeps a sparse integer matrix as CSR while the field cache is OFF (the C ABI's default) and
    releases the dense source.  A later request for the cached-field sweep is served by the CSR form of that sweep
    where the problem qualifies (the same chain as the dense forms); where it does not, the sweep says why -- the
    order of the two calls -- and handing the matrix over again after the request takes the dense forms."""
    from spin_glass_anneal_rl_amd.engine import last_kernel
    n = 4200
    rng = np.random.RandomState(n)
    h = np.zeros(n, np.float32)
    temps = np.asarray([6.0, 3.0, 1.0, 0.3])
    for big in (False, True):
        J = np.zeros((n, n), np.float32)
        for i in range(0, n - 2, 2):
            J[i, i + 1] = J[i + 1, i] = float(rng.choice([-3.0, 2.0])) * (9000.0 if big else 1.0)
            J[i, i + 2] = J[i + 2, i] = float(rng.choice([-1.0, 1.0])) * (9000.0 if big else 1.0)
        with sg.AnnealEngine(0) as e:
            e.set_dense(J, h)                       # field cache OFF: taken as CSR
            assert "source=dense-matrix" in e.describe()
            e.init_replicas(4, seed=1)
            e.set_temperatures(temps * (9000.0 if big else 1.0))
            e.set_field_cache("auto")               # AUTO may fall back: sweeps run on the CSR forms
            e.sweep(2)
            e.set_field_cache("on")
            if big:                                 # sum_j |J_ij| = 45 000: beyond the int16 fields of the CSR form
                with pytest.raises(sg.AnnealingError, match="before sga_set_dense"):
                    e.sweep(1)
            else:
                e.sweep(3)
                assert last_kernel().startswith("sweep_clf_csr_kernel"), last_kernel()
                as_csr = (e.energies(), e.spins())
            e.set_dense(J, h)                       # the request first, then the matrix: the dense forms
            e.init_replicas(4, seed=1)
            e.set_temperatures(temps * (9000.0 if big else 1.0))
            e.sweep(5)
            assert "sweep=cached-local-fields(int" in e.describe() and last_kernel().startswith("sweep_clf")
            if not big:
                assert np.array_equal(e.energies(), as_csr[0]) and np.array_equal(e.spins(), as_csr[1])


def test_all_replica_field_pass_in_tiles(sg):
    """The scratch of the all-replica pass (Y = S J^T) is bounded (option "fields_scratch_mb"): beyond it the
    replicas go through in tiles of whole 128-replica blocks -- energies and seeded fields as in one pass."""
    n, R, seed = 1000, 300, 5
    J = int_couplings(n, n, 3)
    h = np.random.RandomState(1).randint(-2, 3, n).astype(np.float32)
    prob = oracle.Problem(J=J, h=h)
    s0 = oracle.init_spins(n, R, seed)
    want = oracle.energy(prob, s0)
    temps = ladder(R, 30.0, 1.0)
    ref = oracle.sweeps(prob, s0.copy(), temps, 2, seed=seed, n_threads=8)
    with sg.AnnealEngine(0) as e:
        e.set_option("fields_scratch_mb", 1)    # 256 replicas of 1024 fp32 per tile: two tiles
        e.set_field_cache("on")
        e.set_dense(J, h, storage="i8")
        e.init_replicas(R, seed=seed)
        assert np.array_equal(e.energies(), want)
        e.set_temperatures(temps)
        out = e.sweep(2, energy_trace=True)
        assert np.array_equal(out["energy_trace"], ref["energy_trace"])


# ----------------------------------------------------------------------------- BASELINE configs[1]
def test_c2a_at_full_size_with_cached_fields(sg):
    """10 000-spin dense +-1 SK instance, 1024 replicas (bench.py's variant): the oracle follows
    replicas 0..2 (Philox streams are keyed by the global replica id); all replicas: tracked
    energy == energy from scratch, and the row-per-proposal kernels give the same energies.  123 sweeps with exchange
    rounds: the default form ends at eight waves per replica (the launch is its hottest replica's chain by then), the
    several-accepts-per-round form at the standard waves walks the same chain.  The default (clf_batched = 2) starts on
    the several-accepts form -- from random spins every replica accepts several per cent -- and leaves it once the
    hottest replica accepts less than 1 %."""
    import torch
    import bench
    from spin_glass_anneal_rl_amd.engine import last_kernel
    n, R, seed, ns = 10000, 1024, 42, 10
    followed = ladder_ends(R)
    J = bench.make_sk_instance(n, 2, torch.device("cuda", 0))
    h = torch.zeros(n, device="cuda:0")
    temps = ladder(R, 10.0, 0.1)
    res, long_run = {}, {}
    for cache in ("on", "on-batched", "off"):
        with sg.AnnealEngine(0) as e:
            if cache == "on-batched":
                e.set_options({"clf_batched": 1, "clf_tail_waves": 0})
            e.set_field_cache(cache.split("-")[0])
            e.set_dense(J, h, storage="f32" if cache == "off" else "auto")
            e.init_replicas(R, seed=seed)
            e.set_ladder(temps)
            out = e.sweep(ns, energy_trace=True)
            first_kernel = last_kernel()
            tracked = e.energies()
            e.recompute_energies()
            assert np.array_equal(e.energies(), tracked)
            res[cache] = (out["energy_trace"], e.spins()[followed].copy(), e.stats()[0])
            if cache != "off":
                # a longer run of the multi-wave forms (barriers between the waves of a replica, ~10^7 accepts in
                # all): any ordering slip between the waves -- round 3 had one: the flipped spin read after the
                # barrier by a slower wave -- leaves fields, spins and the tracked energy inconsistent
                for _ in range(12):
                    e.sweep(10)
                    e.exchange(count=False)
                tracked = e.energies()
                e.recompute_energies()
                assert np.array_equal(e.energies(), tracked)
                long_run[cache] = (tracked, e.stats()[0], last_kernel(), e.describe(), first_kernel)
                e.set_field_cache("off")  # the row-per-proposal kernel continues from the same state
                e.sweep(1)
                tracked = e.energies()
                e.recompute_energies()
                assert np.array_equal(e.energies(), tracked)
    for cache in ("on", "on-batched"):
        assert np.array_equal(res[cache][0], res["off"][0])
        assert np.array_equal(res[cache][1], res["off"][1]) and np.array_equal(res[cache][2], res["off"][2])
    assert np.array_equal(long_run["on"][0], long_run["on-batched"][0]) and np.array_equal(long_run["on"][1], long_run["on-batched"][1])
    assert long_run["on"][4].startswith("sweep_clfb_kernel"), long_run["on"][4]
    assert long_run["on"][2].startswith("sweep_clf_kernel") and "x 8 wave" in long_run["on"][2], long_run["on"][2:]
    assert long_run["on-batched"][2].startswith("sweep_clfb_kernel") and "x 4 wave" in long_run["on-batched"][2]
    # the oracle follows the hot end, the middle and the COLD end (T = 0.1) of the ladder for all ten sweeps: at the
    # cold end the accept table's boundary, the look-ahead replay and the eight-wave tail form are on the checked path
    prob = oracle.Problem(J=J.cpu().numpy(), h=np.zeros(n, np.float32))
    ref = follow(prob, n, seed, temps, followed, ns, exact_f32=True)
    for i, r in enumerate(followed):
        for cache in ("on", "on-batched", "off"):
            assert np.array_equal(res[cache][0][:, r], ref[r][0]), (cache, r)
            assert np.array_equal(res[cache][1][i], ref[r][1]), (cache, r)
            assert res[cache][2][r] == ref[r][2], (cache, r)


# ----------------------------------------------------------------------------- all-replica field pass
@pytest.mark.parametrize("kind", ["pm1", "int", "gauss"])
@pytest.mark.parametrize("n,R", [(257, 32), (1000, 130), (3000, 200), (64, 64)])
def test_matrix_core_energies_match_oracle(sg, n, R, kind):
    """Energies of >= 32 replicas come from one pass over J on the matrix cores (i8 / f32 / f64 MFMA)."""
    if kind == "pm1":
        J = pm1(n, n)
    elif kind == "int":
        J = int_couplings(n, n, 90, density=0.5)
    else:
        rng = np.random.RandomState(n)
        J = np.triu(rng.randn(n, n), 1).astype(np.float32)
        J = J + J.T
    h = (np.random.RandomState(5).randn(n) if kind == "gauss"
         else np.random.RandomState(5).randint(-3, 4, n)).astype(np.float32)
    prob = oracle.Problem(J=J, h=h)
    s0 = oracle.init_spins(n, R, 99)
    want = oracle.energy(prob, s0)
    for storage in (("f32", "i8") if kind != "gauss" else ("f32",)):
        with sg.AnnealEngine(0) as e:
            e.set_dense(J, h, storage=storage)
            e.init_replicas(R, seed=99)
            got = e.energies()
            if kind == "gauss":  # fp64 accumulation in another order: equal to the rounding of the last bits
                assert np.allclose(got, want, rtol=1e-6, atol=1e-6)
            else:
                assert np.array_equal(got, want)


@pytest.mark.parametrize("kind", ["pm1", "int", "gauss"])
@pytest.mark.parametrize("n,deg,R", [(300, 8, 64), (1000, 40, 130), (2000, 200, 70), (65, 64, 97)])
def test_csr_energies_of_many_replicas_in_one_pass(sg, n, deg, R, kind):
    """64 and more replicas of a CSR problem: spins transposed to a bit matrix, every entry read once for 32
    replicas per lane (csrc/fields_csr.hip) -- against the oracle and against the per-replica kernel."""
    opts = {}  # engine options (sga_set_option): which kernel form runs, never what it computes
    rng = np.random.RandomState(n + deg)
    J = np.zeros((n, n), np.float32)
    for i in range(n):
        for j in rng.choice(n, min(deg // 2 + 1, n - 1), replace=False):
            if i != j:
                v = {"pm1": rng.choice([-1.0, 1.0]), "int": float(rng.randint(-90, 91)), "gauss": rng.randn()}[kind]
                J[i, j] = J[j, i] = v
    h = (rng.randn(n) if kind == "gauss" else rng.randint(-3, 4, n)).astype(np.float32)
    rowptr = np.concatenate([[0], np.cumsum((J != 0).sum(1))]).astype(np.int32)
    col = np.concatenate([np.nonzero(J[i])[0] for i in range(n)]).astype(np.int32)
    val = np.concatenate([J[i][J[i] != 0] for i in range(n)]).astype(np.float32)
    prob = oracle.Problem(csr=(rowptr, col, val), h=h)
    s0 = oracle.init_spins(n, R, 321)
    want = oracle.energy(prob, s0)
    got = {}
    for one_pass in (True, False):
        if not one_pass:
            opts["batched_energy"] = 0
        with sg.AnnealEngine(0) as e:
            e.set_options(opts)
            e.set_csr(rowptr, col, val, h)
            e.init_replicas(R, seed=321)
            got[one_pass] = e.energies()
    if kind == "gauss":  # fp64 accumulation in another order: equal to the rounding of the last bits
        assert np.allclose(got[True], want, rtol=1e-6, atol=1e-6) and np.allclose(got[True], got[False], rtol=1e-6, atol=1e-6)
    else:
        assert np.array_equal(got[True], want) and np.array_equal(got[False], want)


# ----------------------------------------------------------------------------- sparse couplings (CSR)
def _csr_of(J):
    rowptr = np.concatenate([[0], np.cumsum((J != 0).sum(1))]).astype(np.int32)
    col = np.concatenate([np.nonzero(J[i])[0] for i in range(J.shape[0])] + [np.zeros(0, int)]).astype(np.int32)
    val = np.concatenate([J[i][J[i] != 0] for i in range(J.shape[0])] + [np.zeros(0)]).astype(np.float32)
    return rowptr, col, val


CSR_CLF_CASES = [
    # n, mean degree, |J| <= amp, half-integer h, R
    (60, 6, 1, False, 5), (500, 12, 3, False, 6), (1000, 40, 2, True, 7), (2000, 250, 1, False, 4),
    (3000, 700, 2, True, 3), (1500, 1400, 1, False, 3), (130, 100, 50, True, 6),
]


@pytest.mark.parametrize("n,deg,amp,half_h,R", CSR_CLF_CASES)
def test_cached_fields_over_csr_couplings_match_oracle(sg, n, deg, amp, half_h, R):
    """sga_set_csr problems with integer couplings: the dynamic part of the local fields (J s) as int16 in LDS, h read
    beside it, a row's entries read only on accept (csrc/sweep_clf_csr.hip) -- the oracle's chain: short rows, rows of
    several hundred entries padded to 64-entry slots, rows beyond 512 entries (several per thread), half-integer fields
    (penalty encodings), hot and cold replicas.  AUTO ends on it once the run is cold; the row-per-proposal CSR kernels
    give the same energies."""
    from spin_glass_anneal_rl_amd.engine import last_kernel
    rng = np.random.RandomState(5 * n + deg)
    J = np.zeros((n, n), np.float32)
    for i in range(n):
        for j in rng.choice(n, max(1, deg // 2), replace=False):
            if i != j:
                J[i, j] = J[j, i] = float(rng.choice([v for v in range(-amp, amp + 1) if v != 0]))
    h = (rng.randint(-4 * amp, 4 * amp + 1, n) / (2.0 if half_h else 1.0)).astype(np.float32)
    csr = _csr_of(J)
    prob = oracle.Problem(csr=csr, h=h)
    ns, seed = 6, 0x5C5 + n
    scale = amp * np.sqrt(max(deg, 1))
    temps = ladder(R, 3.0 * scale, 0.05 * scale)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=8)
    with sg.AnnealEngine(0) as e:
        e.set_field_cache("on")
        e.set_csr(*csr, h)
        e.init_replicas(R, seed=seed)
        assert "sweep=cached-local-fields(int16 dynamic fields" in e.describe(), e.describe()
        e.set_temperatures(temps)
        out = e.sweep(ns, energy_trace=True)
        assert last_kernel().startswith("sweep_clf_csr_kernel"), last_kernel()
        check_against(e, ref, s, out)
        # traced sweeps take the row-per-proposal kernels (the same chain) and the cache is seeded anew after them
        more = oracle.sweeps(prob, s, temps, 2, seed=seed, sweep0=ns, energy=ref["energy"], trace=True, n_threads=8)
        out2 = e.sweep(2, trace=True)
        assert not last_kernel().startswith("sweep_clf_csr_kernel")
        assert np.array_equal(out2["accept_trace"], more["accept_trace"]) and np.array_equal(e.spins(), s)
        last = oracle.sweeps(prob, s, temps, 2, seed=seed, sweep0=ns + 2, energy=more["energy"], n_threads=8)
        e.sweep(2)
        assert last_kernel().startswith("sweep_clf_csr_kernel")
        assert np.array_equal(e.spins(), s) and np.array_equal(e.energies(), last["energy"])


def test_cached_fields_over_csr_what_does_not_qualify(sg):
    """Real-valued couplings, duplicate entries in a row, dynamic fields beyond int16: ON says why, AUTO runs on the
    row-per-proposal kernels."""
    n = 300
    rng = np.random.RandomState(9)
    J = np.zeros((n, n), np.float32)
    for i in range(n):
        for j in rng.choice(n, 4, replace=False):
            if i != j:
                J[i, j] = J[j, i] = float(rng.choice([-1.0, 1.0]))
    h = np.zeros(n, np.float32)
    rp, col, val = _csr_of(J)
    cases = {"real": (rp, col, (val * 0.37).astype(np.float32)),
             "big": (rp, col, (val * 20000.0).astype(np.float32))}
    rp2 = np.concatenate([[0], np.cumsum(np.diff(rp) * 2)]).astype(np.int32)   # every entry twice, halved: duplicates
    col2 = np.concatenate([np.repeat(col[rp[i]:rp[i + 1]], 2) for i in range(n)]).astype(np.int32)
    val2 = np.concatenate([np.repeat(val[rp[i]:rp[i + 1]] * 2.0, 2) / 2.0 for i in range(n)]).astype(np.float32)
    cases["duplicates"] = (rp2, col2, val2)
    for name, csr in cases.items():
        with sg.AnnealEngine(0) as e:
            e.set_field_cache("on")
            e.set_csr(*csr, h)
            e.init_replicas(4, seed=3)
            e.set_temperatures(np.full(4, 2.0))
            with pytest.raises(sg.AnnealingError, match="cached local fields over CSR"):
                e.sweep(1)
            e.set_field_cache("auto")
            e.sweep(2)
            assert "cached" not in e.describe() or "sweep=auto" not in e.describe(), (name, e.describe())



@pytest.mark.parametrize("cache", ["on", "auto"])
@pytest.mark.parametrize("form", [{"clf_batched": 0}, {"clf_batched": 1}, {"clf_batched": 0, "one_call": True}])
def test_every_replica_gets_eight_waves_once_the_launch_is_one_replicas_chain(sg, cache, form):
    """Option "clf_tail_waves" (default): a launch of the cached-field kernel ends with its hottest replica's chain; once
    the mean acceptance is below 0.28 of the hottest replica's, every replica runs at eight waves (the workgroups
    of the others are gone early anyway).  Same chain as with the option off and as the oracle's.  A single call of
    40 sweeps is walked in pieces of 16 (the counters are looked at when a call starts), so it gets there as well."""
    from spin_glass_anneal_rl_amd.engine import last_kernel
    form = dict(form)
    one_call = form.pop("one_call", False)
    n, R, seed, ns = 1600, 16, 0x807, 40
    J = pm1(n, 77)
    h = np.zeros(n, np.float32)
    temps = ladder(R, 0.8 * np.sqrt(n), 0.002 * np.sqrt(n))   # a hot end; most of the ladder accepts next to nothing
    prob = oracle.Problem(J=J, h=h)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=8)
    acc = ref["n_accepted"] / ns
    assert acc.max() > 24 and acc.mean() < 0.25 * acc.max(), acc
    seen = {}
    for tail in (1, 0):
        with sg.AnnealEngine(0) as e:
            e.set_options(dict(form, clf_tail_waves=tail))
            e.set_field_cache(cache)
            e.set_dense(J, h, storage="f32")
            e.init_replicas(R, seed=seed)
            e.set_temperatures(temps)
            kernels = []
            if one_call:
                out = e.sweep(ns, energy_trace=True)
                assert np.array_equal(out["energy_trace"], ref["energy_trace"])
                kernels = ["(first piece)", last_kernel()]
            else:
                for _ in range(ns // 4):
                    e.sweep(4)
                    kernels.append(last_kernel())
            assert np.array_equal(e.energies(), ref["energy"]) and np.array_equal(e.spins(), s)
            assert np.array_equal(e.stats()[0], ref["n_accepted"])
            tracked = e.energies()
            e.recompute_energies()
            assert np.array_equal(e.energies(), tracked)
            seen[tail] = (kernels, e.describe())
    kernels, desc = seen[1]
    assert "x 8 wave" in kernels[-1] and ("now 8" in desc or "at 8 waves each" in desc), (kernels, desc)
    assert "x 8 wave" not in kernels[0]
    assert not any("x 8 wave" in k for k in seen[0][0]), seen[0][0]
