"""BASELINE.json configs at the size AND in the kernel form bench.py really takes
(configs[2] at 4096 replicas, configs[3] at 1024 replicas per GPU -- the bit-spin
wide form picked by LDS residency -- and configs[4] at 1000 cities), plus the regression tests
for the round-1 parity loose ends: launch geometries with more waves than chunks, real-valued
couplings under every geometry (canonical summation order), checkpoints across geometries.

Everything goes through the C ABI; the oracle follows a few replicas (Philox streams are keyed
by the global replica id, so replicas 0..k-1 of a big run are reproduced by a k-replica oracle
run), the rest is covered by size-independent properties: tracked energy == energy recomputed
from scratch, acceptance counters within range, exchanges stay inside their ladders.
"""
import numpy as np
import pytest
import torch

import oracle
from oracle_follow import follow, ladder_ends

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sg():
    import spin_glass_anneal_rl_amd as m
    return m


def ladder(R, tmax, tmin):
    return np.asarray([tmax * (tmin / tmax) ** (i / max(R - 1, 1)) for i in range(R)])


def csr_of(J):
    n = J.shape[0]
    rowptr = np.concatenate([[0], np.cumsum((J != 0).sum(1))]).astype(np.int32)
    colidx = np.concatenate([np.nonzero(J[i])[0] for i in range(n)]).astype(np.int32)
    val = np.concatenate([J[i][J[i] != 0] for i in range(n)]).astype(np.float32)
    return rowptr, colidx, val


# ----------------------------------------------------------------------------- configs[2]
def test_c3_sparse_instance_at_4096_replicas(sg):
    """10 000 spins, CSR degree ~32, 4096 replicas (bench.py --workload c3)."""
    import bench
    csr = bench.make_sparse_instance(10000, 16, 3)
    n, R, seed, ns = 10000, 4096, 42, 10
    h = np.zeros(n, np.float32)
    temps = ladder(R, 10.0, 0.1)
    with sg.AnnealEngine(0) as e:
        e.set_csr(*csr, h)
        e.init_replicas(R, seed=seed)
        d = e.describe()
        assert "R=4096" in d and "path=integer-fast" in d and "spins=lds-int8" in d, d
        e.set_ladder(temps)
        out = e.sweep(ns, energy_trace=True)
        tracked = e.energies()
        acc, att = e.stats()
        assert np.all(att == ns * n) and np.all(acc <= att) and acc[0] > acc[-1]  # hot end moves more
        e.recompute_energies()
        assert np.array_equal(e.energies(), tracked)            # +-1 couplings: exact
        swaps = e.exchange()
        assert 0 < swaps <= R // 2 and sorted(e.slot_map()) == list(range(R))
        spins = e.spins()
    # the oracle follows the hot end, the middle and the COLD end of the ladder (T = 0.1: accept table at its
    # boundary, nearly every uphill proposal refused) for all ten sweeps
    prob = oracle.Problem(csr=csr, h=h)
    for r, (trace, s, n_acc) in follow(prob, n, seed, temps, ladder_ends(R, extra=(1, 2, 3, R - 2)), ns).items():
        assert np.array_equal(out["energy_trace"][:, r], trace), r
        assert np.array_equal(spins[r], s), r
        assert acc[r] == n_acc, r


# ----------------------------------------------------------------------------- configs[3]
def test_c4_scheduling_instance_at_1024_replicas_per_gpu(sg):
    """50 000-spin scheduling penalties (degree 598), 1024 replicas = one GPU's share of the 8192:
    LDS residency makes the engine hold the spins as bits, two waves per replica."""
    from spin_glass_anneal_rl_amd import encoders as enc
    b = enc.scheduling_ising(np.full(500, 1.0), n_agents=1, time_horizon=100.0,
                             time_discretization=100, objective="total_time",
                             penalty_weights={"assignment": 100.0, "capacity": 50.0})
    csr, h = b.to_csr(), b.fields()
    n, R, Rg, seed = 50000, 1024, 8192, 31
    temps_g = ladder(Rg, 500.0, 5.0)
    runs = {}
    ns = 10
    for replica0 in (0, 3 * R, 7 * R):  # rank 0's, rank 3's and rank 7's (the coldest) share of the global ladder
        with sg.AnnealEngine(0) as e:
            e.set_csr(*csr, h)
            e.init_replicas(R, seed=seed, R_global=Rg, replica0=replica0)
            d = e.describe()
            assert "spins=lds-bits" in d and "waves_per_replica=2" in d and "R=1024" in d, d
            # integer couplings, half-integer fields: packed entries, the accept table at twice the resolution
            assert "entries=packed-32bit" in d and "path=half-integer-fast" in d, d
            e.set_ladder(temps_g)
            assert np.array_equal(e.temperatures(), temps_g[replica0:replica0 + R])
            out = e.sweep(ns, energy_trace=True)
            tracked = e.energies()
            e.recompute_energies()
            # |E| ~ 1e9: the from-scratch value is rounded to fp32 as torch.dot rounds it
            # (core/ising_model.py:161-168), the tracked one is a double sum of exact fp32 dE's
            assert np.allclose(e.energies(), tracked, rtol=1e-6, atol=0)
            runs[replica0] = (out["energy_trace"], e.spins(), e.stats()[0])
    # the oracle follows the hot end, the middle and the cold end of EVERY share for all ten sweeps: replica
    # 7 * 1024 + 1023 = 8191 is the coldest of the whole 8192-temperature ladder (T = 5 against couplings of 50..100)
    prob = oracle.Problem(csr=csr, h=h)
    for replica0, (trace, spins, acc) in runs.items():
        ids = [replica0 + r for r in ladder_ends(R, extra=(1,))]
        for g, (ref_trace, s, n_acc) in follow(prob, n, seed, temps_g, ids, ns).items():
            assert np.array_equal(trace[:, g - replica0], ref_trace), g
            assert np.array_equal(spins[g - replica0], s), g
            assert acc[g - replica0] == n_acc, g


# ----------------------------------------------------------------------------- configs[4]
def _tsp_distances(n_cities, seed, integer):
    rs = np.random.RandomState(seed)
    xy = rs.rand(n_cities, 2) * 100.0
    d = np.hypot(xy[:, None, 0] - xy[None, :, 0], xy[:, None, 1] - xy[None, :, 1])
    # integer: distances in multiples of 4 (and integer penalties: every J and h is an integer)
    return np.rint(d / 4.0) * 4.0 if integer else d


def _tsp(n_cities, seed, integer, device="cuda:0"):
    from spin_glass_anneal_rl_amd import encoders as enc
    return enc.tsp_csr(_tsp_distances(n_cities, seed, integer), city_visit=200.0, position_fill=200.0,
                       auto_scale=not integer, device=device)


@pytest.mark.parametrize("integer", [True, False])
def test_c5_tsp_1000_cities_full_size(sg, integer):
    """examples/tsp_example.py at BASELINE size: 1000 cities = 10^6 spins, 3.996e9 entries
    (32 GB of CSR written on the device, 64-bit extents), spins as bits, 8 waves per replica.
    16 replicas in 2 ladders, one sweep, on the integer-valued instance (accept-table path, fp32
    row sums) and on real distances (fp64 row sums in the canonical order): tracked energy ==
    energy recomputed from scratch to its fp32 rounding."""
    rowptr, col, val, h, _ = _tsp(1000, 5, integer)
    n, R, n_ladders, seed = 10 ** 6, 16, 2, 11
    assert rowptr.dtype == torch.int64 and int(rowptr[-1]) == 3996 * n
    temps = np.tile(ladder(R // n_ladders, 200.0, 2.0), n_ladders)
    with sg.AnnealEngine(0) as e:
        e.set_csr(rowptr, col, val, h)
        del col, val
        torch.cuda.empty_cache()
        e.init_replicas(R, seed=seed)
        d = e.describe()
        assert "nnz=3996000000" in d and "spins=lds-bits" in d and "waves_per_replica=8" in d, d
        assert ("path=integer-fast" in d) == integer and "recomputed" not in d, d
        e.set_ladder(temps, n_ladders)
        e0 = e.energies()
        assert np.all(np.isfinite(e0)) and len(set(e0)) == R
        s_init = e.spins(0)
        e.sweep(1)
        tracked = e.energies()
        acc, att = e.stats()
        assert np.all(att == n) and np.all(acc > 0) and np.all(acc <= att)
        assert not np.array_equal(e.spins(0), s_init)
        e.recompute_energies()
        # |E| ~ 1e11: the from-scratch value is rounded to fp32 as torch.dot rounds it
        # (core/ising_model.py:161-168); the tracked one is a double sum of the dE's
        assert np.allclose(e.energies(), tracked, rtol=1e-6, atol=0)
        e.exchange()
        slots = e.slot_map()
        assert sorted(slots) == list(range(R))
        assert np.array_equal(slots // (R // n_ladders), np.arange(R) // (R // n_ladders))
        spins_csr = e.spins()
    # the same problem with the couplings never stored (sga_set_tsp): identical chain
    from spin_glass_anneal_rl_amd import encoders as enc
    d32, A, B, h2, _ = enc.tsp_structure(_tsp_distances(1000, 5, integer), 200.0, 200.0,
                                         auto_scale=not integer)
    assert np.array_equal(h2, h.cpu().numpy())
    with sg.AnnealEngine(0) as e:
        e.set_tsp(d32, A, B, h2)
        e.init_replicas(R, seed=seed)
        d = e.describe()
        assert "couplings=implicit" in d and ("acc=f32-exact" if integer else "acc=f64-exact") in d, d
        e.set_ladder(temps, n_ladders)
        assert np.allclose(e.energies(), e0, rtol=1e-6, atol=0)
        e.sweep(1)
        assert np.array_equal(e.spins(), spins_csr)
        assert np.array_equal(e.stats()[0], acc)
        assert np.allclose(e.energies(), tracked, rtol=1e-6, atol=0)


@pytest.mark.parametrize("integer", [True, False])
@pytest.mark.parametrize("n_cities", [3, 5, 40, 130, 300, 600])
def test_tsp_implicit_form_equals_stored_couplings_and_oracle(sg, n_cities, integer):
    """sga_set_tsp against sga_set_csr on the couplings encoders.tsp_csr writes, and both against
    the oracle running on the CSR its own restatement writes (oracle.tsp_to_csr): energies,
    local fields, every decision and dE of two sweeps, asymmetric distances included."""
    from spin_glass_anneal_rl_amd import encoders as enc
    if n_cities == 600 and integer:
        pytest.skip("the 600-city case (3 waves per replica) runs once, on real distances")
    dist = _tsp_distances(n_cities, 100 + n_cities, integer)
    if n_cities == 40:
        dist = dist + (np.rint(np.random.RandomState(1).rand(n_cities, n_cities) * 5) * 4 if integer
                       else np.random.RandomState(1).rand(n_cities, n_cities))   # asymmetric
    d32, A, B, h, _ = enc.tsp_structure(dist, 200.0, 120.0, auto_scale=not integer)
    n, R, ns, seed = n_cities ** 2, 3, (2 if n_cities <= 130 else 1), 31 + n_cities
    csr = oracle.tsp_to_csr(d32, A, B)
    rowptr, col, val, h_enc, _ = enc.tsp_csr(dist, 200.0, 120.0, auto_scale=not integer)
    assert np.array_equal(rowptr.numpy(), csr[0]) and np.array_equal(col.numpy(), csr[1])
    assert np.array_equal(val.numpy(), csr[2]) and np.array_equal(h_enc.numpy(), h)
    temps = ladder(R, 150.0, 3.0)
    prob = oracle.Problem(csr=(csr[0].astype(np.int32), csr[1], csr[2]), h=h)
    s = oracle.init_spins(n, R, seed)
    s0 = s.copy()
    e_ref = np.asarray([oracle.energy(prob, s[r]) for r in range(R)])
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, energy=e_ref.copy(), trace=True, n_threads=R)
    sites = [0, 1, n_cities, n // 2, n - 1]
    for form in ("implicit", "csr"):
        with sg.AnnealEngine(0) as e:
            if form == "implicit":
                e.set_tsp(d32, A, B, h)
            else:
                e.set_csr(csr[0], csr[1], csr[2], h)
            e.init_replicas(R, seed=seed)
            d = e.describe()
            if form == "implicit":
                assert f"tsp n_cities={n_cities} " in d and ("acc=f32-exact" if integer else "acc=f64-exact") in d, d
            if integer:
                assert np.array_equal(e.energies(), e_ref), d
            else:
                assert np.allclose(e.energies(), e_ref, rtol=1e-6, atol=1e-6), d
            assert np.array_equal(e.local_fields(1, sites), [oracle.local_field(prob, s0[1], i) for i in sites]), d
            e.set_temperatures(temps)
            e.sweep(ns)                                   # production variant
            assert np.array_equal(e.spins(), s), d
            assert np.array_equal(e.stats()[0], ref["n_accepted"]), d
            be, bs, br = e.best()
            assert np.array_equal(bs, ref["best_spins"][br]), d
            e.init_replicas(R, seed=seed)
            e.set_temperatures(temps)
            out = e.sweep(ns, trace=True)                 # general (traced) variant
            assert np.array_equal(out["accept_trace"], ref["accept_trace"]), d
            assert np.array_equal(out["dE_trace"], ref["dE_trace"]), d
            if form == "implicit":
                with pytest.raises(sg.AnnealingError):
                    e.flip(0, 1)


@pytest.mark.parametrize("integer", [True, False])
@pytest.mark.parametrize("n_cities", [3, 6, 17, 40, 260])
def test_tsp_implicit_form_several_updates_per_step(sg, n_cities, integer):
    """The production sweep of the implicit TSP form works on 2 | 4 | 8 consecutive updates at once, one per
    wave, and replays a step one update at a time when an accepted update shares its city or a neighbouring
    position with a later one (always, at a handful of cities; a few per cent of the steps at hundreds):
    every setting walks the oracle's chain."""
    opts = {}  # engine options (sga_set_option): which kernel form runs, never what it computes
    from spin_glass_anneal_rl_amd import encoders as enc
    from spin_glass_anneal_rl_amd.engine import last_kernel
    dist = _tsp_distances(n_cities, 300 + n_cities, integer)
    d32, A, B, h, _ = enc.tsp_structure(dist, 200.0, 120.0, auto_scale=not integer)
    n, R, ns, seed = n_cities ** 2, 5, (6 if n_cities <= 40 else 1), 77 + n_cities
    csr = oracle.tsp_to_csr(d32, A, B)
    temps = ladder(R, 300.0, 3.0)
    prob = oracle.Problem(csr=(csr[0].astype(np.int32), csr[1], csr[2]), h=h)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=R)
    for par in ("0", "2", "4", "8", None):
        if par is None:
            opts.pop("tsp_updates_per_step", None)
        else:
            opts["tsp_updates_per_step"] = int(par)
        with sg.AnnealEngine(0) as e:
            e.set_options(opts)
            e.set_tsp(d32, A, B, h)
            e.init_replicas(R, seed=seed)
            e.set_temperatures(temps)
            out = e.sweep(ns, energy_trace=True)
            k = last_kernel()
            want = int(par) if par is not None else (8 if n_cities >= 256 else 4 if n_cities >= 64 else 2 if n_cities >= 24 else 0)
            assert ("sweep_tsp_par_kernel" in k and f"x {want} updates" in k) if want >= 2 else k.startswith("sweep_tsp_kernel"), k
            assert np.array_equal(e.spins(), s), (par, k)
            assert np.array_equal(e.stats()[0], ref["n_accepted"]), (par, k)
            if integer:
                assert np.array_equal(out["energy_trace"], ref["energy_trace"]), (par, k)
                assert np.array_equal(e.energies(), ref["energy"]), (par, k)
            else:
                assert np.allclose(out["energy_trace"], ref["energy_trace"], rtol=1e-9, atol=1e-6), (par, k)
            be, bs, br = e.best()
            assert np.array_equal(bs, ref["best_spins"][br]), (par, k)


@pytest.mark.parametrize("n_cities,splits", [(300, [(1, 2), (2, 1)]), (900, [(2, 2), (1, 4), (4, 1)])])
def test_tsp_implicit_form_waves_and_passes_give_one_chain(sg, n_cities, splits):
    """The implicit TSP sweep deals a distance row to waves x passes of 256 cities (default: two passes
    from four waves up); every split runs one and the same chain -- at 300 cities that of the stored
    couplings too (the 1000-city default is checked against its 32 GB CSR form above)."""
    from spin_glass_anneal_rl_amd import encoders as enc
    dist = _tsp_distances(n_cities, 7 + n_cities, False)
    d32, A, B, h, _ = enc.tsp_structure(dist, 200.0, 120.0)
    R, seed = 3, 99
    temps = ladder(R, 150.0, 3.0)
    out = []
    forms = [("implicit", w, p) for w, p in splits] + ([("csr", 0, 0)] if n_cities <= 300 else [])
    for form, waves, passes in forms:
        with sg.AnnealEngine(0) as e:
            if form == "implicit":
                e.set_tuning(waves_per_replica=waves)
                e.set_tsp(d32, A, B, h)
            else:
                e.set_csr(*oracle.tsp_to_csr(d32, A, B), h)
            e.init_replicas(R, seed=seed)
            if form == "implicit":
                assert f"waves_per_replica={waves} passes={passes}" in e.describe(), e.describe()
            e.set_temperatures(temps)
            e.sweep(1)
            out.append((e.spins().copy(), e.energies().copy(), e.stats()[0].copy()))
    for o in out[1:]:
        assert np.array_equal(out[0][0], o[0]) and np.array_equal(out[0][2], o[2])
        assert np.allclose(out[0][1], o[1], rtol=1e-9, atol=1e-6)


def test_tsp_implicit_form_checkpoint_ladders_and_errors(sg):
    from spin_glass_anneal_rl_amd import encoders as enc
    d32, A, B, h, _ = enc.tsp_structure(_tsp_distances(30, 3, False), 200.0, 200.0)
    R, n_ladders = 12, 3
    temps = np.tile(ladder(R // n_ladders, 100.0, 2.0), n_ladders)

    def fresh():
        e = sg.AnnealEngine(0)
        e.set_tsp(d32, A, B, h)
        e.init_replicas(R, seed=4)
        e.set_ladder(temps, n_ladders)
        return e

    a = fresh()
    for _ in range(3):
        a.sweep(2)
        a.exchange(count=False)
    blob = a.export_state()
    a.sweep(3)
    b = fresh()
    b.import_state(blob)
    b.sweep(3)
    assert np.array_equal(a.spins(), b.spins()) and np.array_equal(a.energies(), b.energies())
    tracked = a.energies()
    a.recompute_energies()
    assert np.allclose(a.energies(), tracked, rtol=1e-6, atol=1e-6)
    a.close()
    b.close()
    with sg.AnnealEngine(0) as e:
        with pytest.raises(sg.AnnealingError):
            e.set_tsp(np.zeros((2, 2), np.float32), 1.0, 1.0, np.zeros(4, np.float32))
        with pytest.raises(sg.AnnealingError):
            e.set_tsp(np.zeros((5, 5), np.float32), 1.0, 1.0, np.zeros(7, np.float32))


def test_c5_tsp_500_cities_against_the_oracle(sg):
    """The largest TSP instance whose CSR the oracle can hold (nnz = 5e8 < 2^31): the same
    bit-spin wide form as the 1000-city run (64-bit extents from the device encoder, 4 waves per
    replica), initial energies and the first sweep against the oracle on two replicas."""
    rowptr, col, val, h, _ = _tsp(500, 7, integer=False)
    n, R, seed = 250000, 8, 23
    temps = ladder(R, 200.0, 2.0)
    with sg.AnnealEngine(0) as e:
        e.set_csr(rowptr, col, val, h)
        e.init_replicas(R, seed=seed)
        d = e.describe()
        assert "spins=lds-bits" in d and "waves_per_replica=4" in d, d
        e.set_ladder(temps)
        e0 = e.energies()
        out = e.sweep(1, energy_trace=True, trace=False)
        spins = e.spins()
    csr = (rowptr.cpu().numpy().astype(np.int32), col.cpu().numpy(), val.cpu().numpy())
    del rowptr, col, val
    torch.cuda.empty_cache()
    prob = oracle.Problem(csr=csr, h=h.cpu().numpy())
    k = 2
    s = oracle.init_spins(n, k, seed)
    # real-valued distances: the initial energies agree to the fp32 rounding of the row sums'
    # total; the sweep's decisions and spins are bit-identical (canonical summation order)
    e_ref = np.asarray([oracle.energy(prob, s[r]) for r in range(k)])
    assert np.allclose(e0[:k], e_ref, rtol=1e-6, atol=0)
    ref = oracle.sweeps(prob, s, temps[:k], 1, seed=seed, energy=e0[:k].copy(), n_threads=k)
    assert np.array_equal(spins[:k], s)
    assert np.array_equal(out["energy_trace"][:, :k], ref["energy_trace"])


# ----------------------------------------------------------------------------- round-1 loose ends
@pytest.mark.parametrize("integer", [True, False])
@pytest.mark.parametrize("n", [33, 65, 129])
@pytest.mark.parametrize("waves", [2, 3, 4])
def test_more_waves_than_chunks_is_clamped(sg, n, waves, integer):
    """A forced wave count beyond the row's chunk count used to reach, through the streaming
    fallback of the geometry choice, a launch in which whole waves held nothing but pad lanes
    (the configuration of the one unreproduced round-1 fuzz mismatch: n = 65, 2 waves, sequential
    sites, fp32 operator arithmetic, traced general kernel).  It is clamped to the chunk count."""
    rng = np.random.RandomState(n * 10 + waves)
    J = np.triu((rng.randint(-2, 3, (n, n)) if integer else rng.randn(n, n)) * (rng.rand(n, n) < 0.3), 1)
    J = (J + J.T).astype(np.float32)
    h = (rng.randint(-2, 3, n) if integer else rng.randn(n)).astype(np.float32)
    R, ns, seed = 7, 2, 1000 + n
    temps = ladder(R, 3.0 * np.sqrt(n), 0.2)
    u = rng.rand(R, ns * n).astype(np.float32)
    prob = oracle.Problem(J=J, h=h)
    for site_mode, arith in ((oracle.SITE_SEQUENTIAL, oracle.ARITH_F32),
                             (oracle.SITE_SEQUENTIAL, oracle.ARITH_F64),
                             (oracle.SITE_RANDOM, oracle.ARITH_F32)):
        uu = u if site_mode == oracle.SITE_SEQUENTIAL else None
        s = oracle.init_spins(n, R, seed)
        ref = oracle.sweeps(prob, s, temps, ns, site_mode=site_mode, arith=arith, seed=seed,
                            replay_u=uu, trace=True, n_threads=4)
        with sg.AnnealEngine(0) as e:
            e.set_tuning(waves_per_replica=waves)
            e.set_dense(J, h, storage="f32")
            e.init_replicas(R, seed=seed)
            assert "waves_per_replica=1 " in e.describe(), e.describe()   # one 256-element chunk
            e.set_temperatures(temps)
            out = e.sweep(ns, site_mode=site_mode, arith=arith, replay_u=uu, energy_trace=True,
                          trace=True)
            assert np.array_equal(out["accept_trace"], ref["accept_trace"])
            assert np.array_equal(out["dE_trace"], ref["dE_trace"])
            assert np.array_equal(e.spins(), s)


def test_forced_waves_up_to_the_chunk_count_are_kept(sg):
    J = np.triu(np.random.RandomState(3).randint(-200, 201, (700, 700)), 1).astype(np.float32)
    J = J + J.T
    for waves, expect in ((2, 2), (3, 3), (4, 3), (16, 3)):  # 700 fp32 = 3 chunks of 256
        with sg.AnnealEngine(0) as e:
            e.set_tuning(waves_per_replica=waves)
            e.set_dense(J, np.zeros(700, np.float32), storage="f32")
            e.init_replicas(4, seed=1)
            assert f"waves_per_replica={expect} " in e.describe() and "acc=f32" in e.describe(), e.describe()


@pytest.mark.parametrize("n,geometries", [(700, (1, 2, 3)), (2500, (1, 3, 5, 8, 10)),
                                          (5000, (1, 2, 7, 16)), (6000, (2,))])
def test_real_valued_chain_does_not_depend_on_the_geometry(sg, n, geometries):
    """Gaussian couplings: the fp64 row sum is formed in one canonical order (256-element chunks,
    adjacent-pairs tree, chunk order) whatever the waves-per-replica, so every decision and every
    dE equals the oracle's bit for bit under each geometry -- including the streaming form (more
    than two super-chunks per wave: n = 2500 and 5000 on one wave, n = 5000 and 6000 on two)."""
    J = np.triu(np.random.RandomState(n).randn(n, n), 1).astype(np.float32)
    J = J + J.T
    h = np.random.RandomState(n + 1).randn(n).astype(np.float32)
    R, ns, seed = 3, 2, 77 + n
    temps = ladder(R, 3.0, 0.3)
    prob = oracle.Problem(J=J, h=h)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, trace=True, n_threads=R)
    for g in geometries:
        with sg.AnnealEngine(0) as e:
            e.set_tuning(waves_per_replica=g)
            e.set_dense(J, h)
            e.init_replicas(R, seed=seed)
            d = e.describe()
            # canonical-order builds: a wave owns whole 1024-element super-chunks, one or two of them
            # in registers; a forced wave count is clamped to the super-chunk count
            supers = (n + 1023) // 1024
            w_eff = min(g, supers)
            assert "acc=f64-canonical" in d and f"waves_per_replica={w_eff} " in d, d
            assert ("(streaming)" in d) == (-(-supers // w_eff) > 2), d
            e.set_temperatures(temps)
            for traced in (True, False):   # general and production variants
                if not traced:
                    e.init_replicas(R, seed=seed)
                    e.set_temperatures(temps)
                out = e.sweep(ns, energy_trace=True, trace=traced)
                if traced:
                    assert np.array_equal(out["accept_trace"], ref["accept_trace"]), d
                    assert np.array_equal(out["dE_trace"], ref["dE_trace"]), d
                assert np.array_equal(e.spins(), s), d
                assert np.array_equal(e.stats()[0], ref["n_accepted"])


@pytest.mark.parametrize("forced", [False, True])
def test_real_valued_dense_with_exact_fp64_sums_takes_the_cheap_order(sg, forced):
    """Real-valued couplings whose set bits span few binary places (distances on a 2^-12 grid,
    weights with a few decimals): the fp64 sum of a row is exact, so the kernels keep one tree per
    update in whatever order the geometry gives -- and still equal the oracle's canonical sum bit for
    bit under every geometry.  `forced`: the same problem through the canonical-order build."""
    opts = {}  # engine options (sga_set_option): which kernel form runs, never what it computes
    if forced:
        opts["force_dense_canonical"] = 1
    n, R, ns, seed = 2600, 3, 2, 404
    rng = np.random.RandomState(n)
    J = np.triu(np.rint(rng.rand(n, n) * 141.0 * 1024.0) / 1024.0 / 4.0 * (rng.rand(n, n) < 0.7), 1).astype(np.float32)
    J = J + J.T
    h = rng.randn(n).astype(np.float32)
    temps = ladder(R, 300.0, 30.0)
    prob = oracle.Problem(J=J, h=h)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, trace=True, n_threads=R)
    for g in (1, 3, 4, 11):
        with sg.AnnealEngine(0) as e:
            e.set_options(opts)
            e.set_tuning(waves_per_replica=g)
            e.set_dense(J, h)
            e.init_replicas(R, seed=seed)
            d = e.describe()
            w_eff = min(g, (n + 1023) // 1024) if forced else g   # canonical builds: whole super-chunks
            assert ("acc=f64-canonical" if forced else "acc=f64-exact") in d and f"waves_per_replica={w_eff} " in d, d
            e.set_temperatures(temps)
            out = e.sweep(ns, trace=True)
            assert np.array_equal(out["accept_trace"], ref["accept_trace"]), d
            assert np.array_equal(out["dE_trace"], ref["dE_trace"]), d
            e.init_replicas(R, seed=seed)
            e.set_temperatures(temps)
            e.sweep(ns)
            assert np.array_equal(e.spins(), s), d


@pytest.mark.parametrize("big", [False, True])
def test_real_valued_csr_chain_does_not_depend_on_the_wave_count(sg, big):
    """Same for CSR rows of a few hundred real-valued entries dealt to 1, 2, 4 or 8 waves (a
    request for 3 runs as 4: the canonical order's wide builds exist per power of two)."""
    opts = {}  # engine options (sga_set_option): which kernel form runs, never what it computes
    if big:
        opts["force_csr_bits"] = 1
    n, R, ns, seed = 1200, 4, 3, 515
    rng = np.random.RandomState(8)
    J = (np.triu(rng.rand(n, n) < 0.5, 1) * rng.randn(n, n)).astype(np.float32)
    J = J + J.T
    h = rng.randn(n).astype(np.float32)
    csr = csr_of(J)
    assert np.diff(csr[0]).max() > 512          # rows longer than one pass of the 512 virtual lanes
    prob = oracle.Problem(csr=csr, h=h)
    temps = ladder(R, 30.0, 3.0)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, trace=True, n_threads=R)
    for waves, runs_as in ((1, 1), (2, 2), (3, 4), (4, 4), (8, 8)):
        with sg.AnnealEngine(0) as e:
            e.set_options(opts)
            e.set_tuning(waves_per_replica=waves)
            e.set_csr(*csr, h)
            e.init_replicas(R, seed=seed)
            assert f"waves_per_replica={runs_as} " in e.describe() and "path=general" in e.describe()
            e.set_temperatures(temps)
            out = e.sweep(ns, energy_trace=True, trace=True)
            assert np.array_equal(out["accept_trace"], ref["accept_trace"]), e.describe()
            assert np.array_equal(out["dE_trace"], ref["dE_trace"]), e.describe()
            assert np.array_equal(e.spins(), s)
            e.init_replicas(R, seed=seed)       # production variant, untraced
            e.set_temperatures(temps)
            e.sweep(ns)
            assert np.array_equal(e.spins(), s), e.describe()
            # single-site operators form the same sums
            f = e.local_fields(0, [0, 5, n - 1])
            assert np.array_equal(f, [oracle.local_field(prob, s[0], i) for i in (0, 5, n - 1)])


@pytest.mark.parametrize("big", [False, True])
@pytest.mark.parametrize("kind", ["integer", "half_integer_h", "fixed_point", "gaussian"])
def test_csr_row_sum_forms_all_reproduce_the_oracle(sg, kind, big):
    """How a CSR row sum is formed is chosen at set time from what is exact: fp32 (+ accept table)
    for integer problems, fp32 without the table when only h is fractional (BASELINE configs[3]),
    fp64 in any order when the values' binary exponents span few enough places (TSP distances),
    the canonical fp64 order otherwise.  Each form, and each slower form forced onto the same
    problem, gives the oracle's chain bit for bit at 1 to 8 waves per replica."""
    opts = {}  # engine options (sga_set_option): which kernel form runs, never what it computes
    if big:
        opts["force_csr_bits"] = 1
    n, R, ns, seed = 1100, 4, 2, 99
    rng = np.random.RandomState(4)
    mask = np.triu(rng.rand(n, n) < 0.45, 1)
    if kind in ("integer", "half_integer_h"):
        vals = rng.randint(-3, 4, (n, n)).astype(np.float64)
    elif kind == "fixed_point":
        vals = np.rint(rng.rand(n, n) * 141.0 * 1024.0) / 1024.0 / 4.0   # distances / 4 on a 2^-12 grid
    else:
        vals = rng.randn(n, n)
    J = (mask * vals).astype(np.float32)
    J = J + J.T
    h = rng.randint(-2, 3, n).astype(np.float32) + (0.5 if kind == "half_integer_h" else 0.0)
    if kind in ("fixed_point", "gaussian"):
        h = rng.randn(n).astype(np.float32)
    expect = {"integer": "path=integer-fast", "half_integer_h": "path=half-integer-fast",
              "fixed_point": "acc=f64-exact", "gaussian": "acc=f64-canonical"}[kind]
    csr = csr_of(J)
    prob = oracle.Problem(csr=csr, h=h)
    temps = ladder(R, 40.0, 2.0)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, trace=True, n_threads=R)
    first = {"integer": 0, "half_integer_h": 1, "fixed_point": 2, "gaussian": 3}[kind]
    for force in range(first, 4):
        if force > first:
            opts["force_csr_acc"] = int(force)
        for waves in (1, 2, 3, 8):
            with sg.AnnealEngine(0) as e:
                e.set_options(opts)
                e.set_tuning(waves_per_replica=waves)
                e.set_csr(*csr, h)
                e.init_replicas(R, seed=seed)
                d = e.describe()
                if force == first:
                    assert expect in d, d
                if force == 3:
                    assert "acc=f64-canonical" in d and f"waves_per_replica={4 if waves == 3 else waves} " in d, d
                e.set_temperatures(temps)
                e.sweep(ns)                                       # production variant
                assert np.array_equal(e.spins(), s), d
                assert np.array_equal(e.stats()[0], ref["n_accepted"]), d
                e.init_replicas(R, seed=seed)
                e.set_temperatures(temps)
                out = e.sweep(ns, trace=True)                     # general (traced) variant
                assert np.array_equal(out["accept_trace"], ref["accept_trace"]), d
                assert np.array_equal(out["dE_trace"], ref["dE_trace"]), d
    opts.pop("force_csr_acc", None)


@pytest.mark.parametrize("kind", ["integer", "fixed_point"])
@pytest.mark.parametrize("n,dens", [(901, 0.4), (257, 0.9), (4001, 0.05)])
def test_csr_wide_bit_forms_on_awkward_shapes(sg, kind, n, dens):
    """The bit-spin wide builds on shapes the other CSR tests do not hit: odd n (the last Philox pair
    of a sweep is half used), nearly dense rows (consecutive sites are neighbours most of the time),
    tiny n (the same site twice in a row ~ once per 257 updates), rows beyond the eight head slots
    (n = 4001 at one wave).  (Written for a two-update look-ahead form that was measured slower and
    dropped, profiles/r02_experiments.md; `SGA_NO_LOOK_AHEAD` is a no-op for CSR problems.)"""
    opts = {}  # engine options (sga_set_option): which kernel form runs, never what it computes
    opts["force_csr_bits"] = 1
    rng = np.random.RandomState(n)
    mask = np.triu(rng.rand(n, n) < dens, 1)
    vals = rng.randint(-3, 4, (n, n)).astype(np.float64) if kind == "integer" else \
        np.rint(rng.rand(n, n) * 141.0 * 1024.0) / 1024.0 / 4.0
    J = (mask * vals).astype(np.float32)
    J = J + J.T
    h = rng.randint(-2, 3, n).astype(np.float32) if kind == "integer" else rng.randn(n).astype(np.float32)
    csr = csr_of(J)
    R, ns, seed = 3, 3, 7 + n
    temps = ladder(R, 30.0, 1.0)
    prob = oracle.Problem(csr=csr, h=h)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=R)
    for waves in (1, 2, 8):
        got = {}
        for look in (True, False):
            if look:
                opts.pop("look_ahead", None)
            else:
                opts["look_ahead"] = 0
            with sg.AnnealEngine(0) as e:
                e.set_options(opts)
                e.set_tuning(waves_per_replica=waves)
                e.set_csr(*csr, h)
                e.init_replicas(R, seed=seed)
                assert "spins=lds-bits" in e.describe() and f"waves_per_replica={waves} " in e.describe()
                e.set_temperatures(temps)
                out = e.sweep(ns, energy_trace=True)
                got[look] = (out["energy_trace"], e.spins(), e.stats()[0])
                assert np.array_equal(e.spins(), s), (e.describe(), look)
                assert np.array_equal(out["energy_trace"], ref["energy_trace"]), (e.describe(), look)
                assert np.array_equal(e.stats()[0], ref["n_accepted"])
        assert all(np.array_equal(x, y) for x, y in zip(got[True], got[False]))
    opts.pop("look_ahead", None)


def test_single_site_operators_use_the_canonical_order_dense(sg):
    n = 3000
    rng = np.random.RandomState(12)
    J = np.triu(rng.randn(n, n), 1).astype(np.float32)
    J = J + J.T
    h = rng.randn(n).astype(np.float32)
    prob = oracle.Problem(J=J, h=h)
    s = oracle.init_spins(n, 2, 5)
    sites = [0, 1, 255, 256, 1500, n - 1]
    with sg.AnnealEngine(0) as e:
        e.set_dense(J, h)
        e.init_replicas(2, seed=5)
        assert np.array_equal(e.local_fields(1, sites), [oracle.local_field(prob, s[1], i) for i in sites])
        for site, u in ((7, 0.3), (2999, 0.9), (7, 0.01)):
            proposed = 2.0 * float(s[1][site]) * oracle.local_field(prob, s[1], site)
            acc, dE = e.update(1, site, 1.7, u)
            ra, rd = oracle.metropolis_update(prob, s[1], site, 1.7, u)
            # (the engine reports the proposed dE, the oracle 0 for a rejected move)
            assert acc == ra and dE == proposed and (not ra or rd == proposed)
        assert np.array_equal(e.spins(1), s[1])


@pytest.mark.parametrize("kind", ["pm1", "int8", "gauss"])
def test_device_matrix_with_a_row_stride_is_packed_where_it_lies(sg, kind):
    """sga_set_dense scans and packs a device matrix in place (no engine copy): a strided view
    (row stride > n) of a larger tensor gives the same chain as its contiguous copy and the oracle."""
    n, big, R, ns, seed = 1500, 2304, 3, 2, 88
    rng = np.random.RandomState(2)
    vals = {"pm1": rng.randint(0, 2, (n, n)) * 2 - 1, "int8": rng.randint(-90, 91, (n, n)),
            "gauss": rng.randn(n, n)}[kind]
    J = np.triu(vals, 1).astype(np.float32)
    J = J + J.T
    h = rng.randn(n).astype(np.float32) if kind == "gauss" else rng.randint(-2, 3, n).astype(np.float32)
    canvas = torch.full((big, big), 7.5, device="cuda")      # what lies beside the view must not matter
    canvas[:n, :n] = torch.from_numpy(J).cuda()
    view = canvas[:n, :n]
    assert view.stride(0) == big and not view.is_contiguous()
    temps = ladder(R, 20.0 if kind == "int8" else 4.0, 0.5)
    prob = oracle.Problem(J=J, h=h)
    s = oracle.init_spins(n, R, seed)
    ref = oracle.sweeps(prob, s, temps, ns, seed=seed, n_threads=R)
    for src in (view, view.contiguous(), J):
        with sg.AnnealEngine(0) as e:
            e.set_dense(src, h)
            e.init_replicas(R, seed=seed)
            d = e.describe()
            assert {"pm1": "storage=i8", "int8": "storage=i8", "gauss": "acc=f64-canonical"}[kind] in d, d
            e.set_temperatures(temps)
            out = e.sweep(ns, energy_trace=True)
            assert np.array_equal(e.spins(), s), d
            if kind != "gauss":
                assert np.array_equal(out["energy_trace"], ref["energy_trace"])
    assert float(canvas[n, n]) == 7.5 and float(canvas[0, n]) == 7.5


@pytest.mark.parametrize("real", [False, True])
def test_checkpoint_moves_between_launch_geometries(sg, real):
    """A state exported after sga_autotune / with one waves-per-replica continues bit-exactly
    in an engine laid out for another (the blob carries unpadded spins; the chain does not depend
    on the geometry, for real-valued couplings either)."""
    n, R = 2600, 6
    rng = np.random.RandomState(21)
    J = np.triu(rng.randn(n, n) if real else rng.randint(-1, 2, (n, n)), 1).astype(np.float32)
    J = J + J.T
    h = (rng.randn(n) if real else rng.randint(-1, 2, n)).astype(np.float32)
    temps = ladder(R, 4.0, 0.4)

    def fresh(waves):
        e = sg.AnnealEngine(0)
        e.set_tuning(waves_per_replica=waves)
        e.set_dense(J, h, storage="f32")
        e.init_replicas(R, seed=9)
        e.set_ladder(temps)
        return e

    def advance(e, rounds):
        for _ in range(rounds):
            e.sweep(2)
            e.exchange(count=False)

    a = fresh(0)
    a.autotune()
    advance(a, 2)
    blob = a.export_state()
    advance(a, 3)
    seen = set()
    for waves in ((1, 2, 3) if real else (1, 4, 11)):  # (Gaussian J: 3 super-chunks of 1024 elements)
        b = fresh(waves)
        seen.add(b.describe())
        b.import_state(blob)
        advance(b, 3)
        assert np.array_equal(a.spins(), b.spins()) and np.array_equal(a.energies(), b.energies())
        assert np.array_equal(a.slot_map(), b.slot_map()) and a.counters() == b.counters()
        assert all(np.array_equal(x, y) for x, y in zip(a.stats(), b.stats()))
        ea, sa, _ = a.best()
        eb, sb, _ = b.best()
        assert ea == eb and np.array_equal(sa, sb)
        b.close()
    assert len(seen) == 3
    a.close()
