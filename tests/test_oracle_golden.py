"""Pins the CPU oracle (oracle/sg_oracle.c) to vectors captured from the imported reference
(tests/golden/, generator tests/golden/make_golden.py) and to published known answers."""
import math

import numpy as np
import pytest

import oracle
from conftest import load_golden


# ----------------------------------------------------------------------------- Philox KAT
# Random123 kat_vectors, philox4x32-10
@pytest.mark.parametrize("ctr,key,out", [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
])
def test_philox_known_answers(ctr, key, out):
    assert [int(x) for x in oracle.philox(ctr, key)] == out


def test_stream_helpers_consistent():
    seed, n = 0x1234567890ABCDEF, 1000
    for t in range(10):
        o = oracle.philox([t >> 1, 7, 3, 0], [seed & 0xffffffff, seed >> 32])
        w_site, w_u = int(o[2 * (t & 1)]), int(o[2 * (t & 1) + 1])
        assert oracle.stream_site(seed, 3, 7, t, n) == (w_site * n) >> 32
        assert oracle.stream_u(seed, 3, 7, t) == (w_u >> 8) * 2.0 ** -24
    s = oracle.init_spins(300, 2, seed, replica0=5)
    assert set(np.unique(s)) == {-1, 1}
    o = oracle.philox([1, 0, 6, 2], [seed & 0xffffffff, seed >> 32])  # replica 6, spins 128..255
    bit = (int(o[(130 >> 5) & 3]) >> (130 & 31)) & 1
    assert s[1, 130] == (1 if bit else -1)


# ----------------------------------------------------------------------------- exp
def _ulp_diff32(a, b):
    ia = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    ib = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    return np.abs(ia - ib)


def test_expf_within_one_ulp_of_libm():
    xs = np.concatenate([np.linspace(-104, 88.7, 200001), -np.logspace(-8, 2, 20001),
                         np.asarray([0.0, -0.0, -87.3, -87.4, -103.9, 88.72])]).astype(np.float32)
    got = np.asarray([oracle.expf(x) for x in xs], np.float32)
    ref = np.exp(xs.astype(np.float64)).astype(np.float32)  # correctly rounded reference
    assert _ulp_diff32(got, ref).max() <= 1
    assert oracle.expf(-200.0) == 0.0 and oracle.expf(100.0) == math.inf
    assert oracle.expf(0.0) == 1.0


def test_exp_double_close_to_libm():
    xs = np.concatenate([np.linspace(-745, 709, 100001), -np.logspace(-12, 2, 20001)])
    got = np.asarray([oracle.exp(x) for x in xs])
    ref = np.exp(xs)
    nz = ref > 1e-300
    assert np.max(np.abs(got[nz] - ref[nz]) / ref[nz]) < 4.5e-16
    assert oracle.exp(-800.0) == 0.0 and oracle.exp(0.0) == 1.0


# ----------------------------------------------------------------------------- sweeps
SWEEP_CASES = ["sweeps_pm1_n8", "sweeps_pm1_n16", "sweeps_pm1_n64", "sweeps_pm1_n64_cold",
               "sweeps_field_n64", "sweeps_pm1_n300", "sweeps_gauss_n64"]


@pytest.mark.parametrize("name", SWEEP_CASES)
def test_sweeps_replay_matches_reference(name):
    g = load_golden(name)
    exact = "gauss" not in name
    prob = oracle.Problem(J=g["J"], h=g["h"])
    n, ns = prob.n, int(g["n_sweeps"])
    s = g["s0"].copy()[None, :]
    assert oracle.energy(prob, s[0]) == pytest.approx(float(g["e0"]), abs=0 if exact else 1e-4)
    u = np.nan_to_num(g["u"], nan=2.0)
    out = oracle.sweeps(prob, s, float(g["T"]), ns, site_mode=oracle.SITE_REPLAY,
                        replay_site=g["site"], replay_u=u, recompute_energy=True, trace=True)
    assert np.array_equal(out["accept_trace"][0].astype(bool), g["accepted"])
    assert np.array_equal(s[0], g["s_final"])
    assert int(out["n_accepted"][0]) == int(g["n_accepted"])
    if exact:
        assert np.array_equal(out["dE_trace"][0], g["dE"])
        assert np.array_equal(out["energy_trace"][:, 0], g["sweep_energy"])
    else:  # fp32 summation order differs from MKL sdot: tolerance 1e-5 relative
        assert np.allclose(out["dE_trace"][0], g["dE"], rtol=1e-5, atol=1e-5)
        assert np.allclose(out["energy_trace"][:, 0], g["sweep_energy"], rtol=1e-5, atol=1e-4)
    # the uniform is consumed exactly where the reference drew one (spin_dynamics.py:145)
    drew = ~np.isnan(g["u"])
    assert np.array_equal(drew, (g["dE"] > 0) | (~g["accepted"]))


@pytest.mark.parametrize("name,rule", [("sweeps_glauber_n64", oracle.RULE_GLAUBER),
                                       ("sweeps_heatbath_n64", oracle.RULE_HEAT_BATH),
                                       ("sweeps_glauber_gauss_n32", oracle.RULE_GLAUBER)])
def test_other_update_rules_replay_matches_reference(name, rule):
    g = load_golden(name)
    exact = "gauss" not in name
    prob = oracle.Problem(J=g["J"], h=g["h"])
    s = g["s0"].copy()[None, :]
    assert not np.isnan(g["u"]).any()  # these rules always draw (spin_dynamics.py:162,183)
    out = oracle.sweeps(prob, s, float(g["T"]), int(g["n_sweeps"]), site_mode=oracle.SITE_REPLAY,
                        rule=rule, replay_site=g["site"], replay_u=g["u"], recompute_energy=True,
                        trace=True)
    assert np.array_equal(out["accept_trace"][0].astype(bool), g["accepted"])
    assert np.array_equal(s[0], g["s_final"])
    assert int(out["n_accepted"][0]) == int(g["n_accepted"])
    if exact:
        assert np.array_equal(out["dE_trace"][0], g["dE"])
        assert np.array_equal(out["energy_trace"][:, 0], g["sweep_energy"])
        s2 = g["s0"].copy()[None, :]  # incremental tracking uses the true dE for heat bath too
        inc = oracle.sweeps(prob, s2, float(g["T"]), int(g["n_sweeps"]), rule=rule,
                            site_mode=oracle.SITE_REPLAY, replay_site=g["site"], replay_u=g["u"])
        assert np.array_equal(inc["energy_trace"][:, 0], g["sweep_energy"])
    else:
        assert np.allclose(out["dE_trace"][0], g["dE"], rtol=1e-5, atol=1e-5)


def test_incremental_energy_equals_recompute_for_integer_couplings():
    g = load_golden("sweeps_field_n64")
    prob = oracle.Problem(J=g["J"], h=g["h"])
    s = g["s0"].copy()[None, :]
    u = np.nan_to_num(g["u"], nan=2.0)
    out = oracle.sweeps(prob, s, float(g["T"]), int(g["n_sweeps"]), site_mode=oracle.SITE_REPLAY,
                        replay_site=g["site"], replay_u=u, recompute_energy=False)
    assert np.array_equal(out["energy_trace"][:, 0], g["sweep_energy"])


def test_single_update_and_field_identities():
    # reference tests/unit/test_core_ising_model.py:95-107: dE == E_new - E_old
    g = load_golden("sweeps_field_n64")
    prob = oracle.Problem(J=g["J"], h=g["h"])
    s = g["s0"].copy()
    for site in range(0, 64, 7):
        e_old = oracle.energy(prob, s)
        f = oracle.local_field(prob, s, site)
        acc, dE = oracle.metropolis_update(prob, s, site, 1e9, 0.0)  # always accepted
        assert acc and dE == 2.0 * (-s[site]) * f
        assert oracle.energy(prob, s) - e_old == dE


# ----------------------------------------------------------------------------- SA driver
@pytest.mark.parametrize("name", ["sa_default_n64", "sa_linear_n20"])
def test_sa_replay_matches_reference(name):
    g = load_golden(name)
    prob = oracle.Problem(J=g["J"], h=g["h"])
    n, ns = prob.n, int(g["n_sweeps"])
    s = g["s0"].copy()[None, :]
    u = np.nan_to_num(g["u"], nan=2.0)
    out = oracle.sweeps(prob, s, g["T_per_sweep"][:, None], ns, site_mode=oracle.SITE_REPLAY,
                        replay_site=g["site"], replay_u=u, recompute_energy=True, trace=True)
    assert np.array_equal(out["accept_trace"][0].astype(bool), g["accepted"])
    assert float(out["best_energy"][0]) == float(g["best_energy"])
    assert np.array_equal(out["best_spins"][0], g["best_configuration"])
    ri = int(g["record_interval"])
    rec = [float(oracle.energy(prob, g["s0"]))] + [out["energy_trace"][k, 0]
                                                   for k in range(0, ns, ri)]
    assert np.array_equal(np.asarray(rec), g["energy_history"])
    assert np.array_equal(s[0], g["s_final"])


@pytest.mark.parametrize("name,exact", [("sweeps_wolff_n24", True), ("sweeps_wolff_gauss_n20", False)])
def test_wolff_cluster_moves_replay_matches_reference(name, exact):
    """UpdateRule.WOLFF (spin_dynamics.py:193-255): start sites and every candidate-bond uniform
    replayed from the reference's stream; cluster sizes, per-move dE, per-sweep energies, spins."""
    g = load_golden(name)
    prob = oracle.Problem(J=g["J"], h=g["h"])
    n, ns = prob.n, int(g["n_sweeps"])
    s = g["s0"].copy()[None, :]
    out = oracle.sweeps(prob, s, float(g["T"]), ns, site_mode=oracle.SITE_REPLAY, rule=oracle.RULE_WOLFF,
                        replay_site=g["site"][None, :], replay_u=g["all_u"][None, :], u_compact=True,
                        energy=np.asarray([float(g["e0"])]), recompute_energy=True, trace=True)
    assert out["accept_trace"].all() and int(out["n_accepted"][0]) == int(g["n_accepted"]) == g["cluster"].sum()
    assert np.array_equal(s[0], g["s_final"])
    if exact:
        assert np.array_equal(out["dE_trace"][0], g["dE"])
        assert np.array_equal(out["energy_trace"][:, 0], g["sweep_energy"])
    else:  # Gaussian J: the reference's fp32 MKL sums, stated tolerance
        assert np.allclose(out["dE_trace"][0], g["dE"], rtol=1e-5, atol=1e-4)
        assert np.allclose(out["energy_trace"][:, 0], g["sweep_energy"], rtol=1e-5, atol=1e-4)


# ----------------------------------------------------------------------------- PT driver
@pytest.mark.parametrize("name", ["pt_small_n16_r4", "pt_c1_n64_r8"])
def test_pt_replay_matches_reference(name):
    g = load_golden(name)
    prob = oracle.Problem(J=g["J"], h=g["h"])
    n, R, ns = prob.n, int(g["n_replicas"]), int(g["n_sweeps"])
    temps = g["temperatures"]
    site = g["site"].astype(np.int32).reshape(ns, R, n)     # [sweep][slot][t]
    u = np.nan_to_num(g["u"], nan=2.0).reshape(ns, R, n)
    acc_ref = g["accepted"].reshape(ns, R, n)
    spins = g["s0"].copy()                                   # storage replica == initial slot
    slot_to_rep = np.arange(R, dtype=np.int32)
    energy = oracle.energy(prob, spins)
    attempts, accepts = np.zeros(R - 1, np.int64), np.zeros(R - 1, np.int64)
    ei, ri = int(g["exchange_interval"]), int(g["record_interval"])
    hist = [[] for _ in range(R)]
    best_e, best_cfg, rnd, ucur = math.inf, None, 0, 0
    for k in range(ns):
        rep_of = slot_to_rep                                 # slot i -> storage replica
        inv = np.argsort(rep_of)                             # storage replica -> slot
        out = oracle.sweeps(prob, spins, temps[inv], 1, site_mode=oracle.SITE_REPLAY,
                            replay_site=site[k][inv], replay_u=u[k][inv], energy=energy,
                            recompute_energy=True, trace=True)
        assert np.array_equal(out["accept_trace"].astype(bool), acc_ref[k][inv])
        energy = out["energy"]
        if k % ei == 0 and k > 0:                            # parallel_tempering.py:113
            start = int(g["exch_start"][rnd])
            npairs = len(range(start, R - 1, 2))
            uu = g["exch_u"][ucur:ucur + npairs]
            for q, i in enumerate(range(start, R - 1, 2)):   # energies seen by the reference
                assert energy[slot_to_rep[i]] == g["exch_Ei"][ucur + q]
                assert energy[slot_to_rep[i + 1]] == g["exch_Ej"][ucur + q]
            before = accepts.copy()
            oracle.pt_exchange_round(temps, energy, slot_to_rep, start=start, u=uu,
                                     attempts=attempts, accepts=accepts)
            got = [(accepts[i] - before[i]) > 0 for i in range(start, R - 1, 2)]
            assert got == list(g["exch_accepted"][ucur:ucur + npairs])
            ucur += npairs
            rnd += 1
        if k % ri == 0:                                      # parallel_tempering.py:117-125
            for i in range(R):
                hist[i].append(energy[slot_to_rep[i]])
            i_best = int(np.argmin([energy[slot_to_rep[i]] for i in range(R)]))
            if energy[slot_to_rep[i_best]] < best_e:
                best_e = float(energy[slot_to_rep[i_best]])
                best_cfg = spins[slot_to_rep[i_best]].copy()
    assert ucur == len(g["exch_u"])
    assert np.array_equal(attempts, g["exchange_attempts"].astype(np.int64))
    assert np.array_equal(accepts, g["exchange_accepts"].astype(np.int64))
    assert np.array_equal(np.asarray(hist), g["energy_histories"])
    assert best_e == float(g["best_energy"])
    assert np.array_equal(best_cfg, g["best_configuration"])
    assert np.array_equal(spins[slot_to_rep], g["s_final"])


def test_pt_all_pairs_replay_matches_reference():
    """exchange_method="all_pairs" (parallel_tempering.py:222-232): per pair i < j a gate draw
    (`rand() < 0.1`) and, behind an open gate, _attempt_single_exchange(i, j)."""
    g = load_golden("pt_allpairs_n16_r5")
    prob = oracle.Problem(J=g["J"], h=g["h"])
    n, R, ns = prob.n, int(g["n_replicas"]), int(g["n_sweeps"])
    temps = g["temperatures"]
    site = g["site"].astype(np.int32).reshape(ns, R, n)
    u = np.nan_to_num(g["u"], nan=2.0).reshape(ns, R, n)
    spins = g["s0"].copy()
    slot_to_rep = np.arange(R, dtype=np.int32)
    energy = oracle.energy(prob, spins)
    attempts, accepts = np.zeros(R - 1, np.int64), np.zeros(R - 1, np.int64)
    ei, ri = int(g["exchange_interval"]), int(g["record_interval"])
    hist = [[] for _ in range(R)]
    stream, cur, n_att = g["np_rand_all"], 0, 0
    for k in range(ns):
        inv = np.argsort(slot_to_rep)
        out = oracle.sweeps(prob, spins, temps[inv], 1, site_mode=oracle.SITE_REPLAY,
                            replay_site=site[k][inv], replay_u=u[k][inv], energy=energy,
                            recompute_energy=True)
        energy = out["energy"]
        if k % ei == 0 and k > 0:
            pairs, uu = [], []
            for i in range(R - 1):
                for j in range(i + 1, R):
                    gate = stream[cur]
                    cur += 1
                    if gate < 0.1:
                        pairs.append((i, j))
                        uu.append(stream[cur])
                        cur += 1
            before = accepts.sum()
            got = oracle.pt_exchange_pairs(temps, energy, slot_to_rep, pairs, u=np.asarray(uu),
                                           attempts=attempts, accepts=accepts)
            assert [tuple(p) for p in pairs] == list(zip(g["exch_i"][n_att:n_att + len(pairs)],
                                                          g["exch_j"][n_att:n_att + len(pairs)]))
            assert got == accepts.sum() - before == g["exch_accepted"][n_att:n_att + len(pairs)].sum()
            n_att += len(pairs)
        if k % ri == 0:
            for i in range(R):
                hist[i].append(energy[slot_to_rep[i]])
    assert cur == len(stream) and n_att == len(g["exch_u"]) > 0
    assert np.array_equal(attempts, g["exchange_attempts"].astype(np.int64))
    assert np.array_equal(accepts, g["exchange_accepts"].astype(np.int64))
    assert np.array_equal(np.asarray(hist), g["energy_histories"])
    assert np.array_equal(spins[slot_to_rep], g["s_final"])


# ----------------------------------------------------------------------------- operator API
def test_operator_fallback_semantics():
    g = load_golden("operator_n48")
    prob = oracle.Problem(J=g["J"], h=g["h"])
    n = prob.n
    s = g["s0"].copy()[None, :]
    nu = int(g["n_updates"])
    # two sequential-order passes, fp32 arithmetic, uniforms consumed only when dE > 0
    ucap = len(g["u"]) + 4
    ulist = np.concatenate([g["u"], np.full(4, 2.0, np.float32)])[None, :]
    out = oracle.sweeps(prob, s, float(g["T"]), nu, site_mode=oracle.SITE_SEQUENTIAL,
                        arith=oracle.ARITH_F32, replay_u=ulist, u_compact=True, trace=True)
    assert np.array_equal(s[0], g["s_out"])
    assert int(out["n_accepted"][0]) == int(g["accepted"])
    ech = out["dE_trace"][0].reshape(nu, n).sum(0)
    assert np.array_equal(ech.astype(np.float32), g["energy_changes"])
    assert oracle.energy(prob, s[0]) == float(g["energy"])
    sp, en = g["pt_spins_in"].copy(), g["pt_energies_in"].astype(np.float32).copy()
    k = oracle.pt_exchange_operator(sp, en, g["pt_temps"], g["pt_u"])
    assert k == int(g["pt_exchanges"])
    assert np.array_equal(sp, g["pt_spins_out"]) and np.array_equal(en, g["pt_energies_out"])


def test_csr_matches_dense():
    # reference tests/unit/test_core_ising_model.py:202-231: dense == sparse energy
    rng = np.random.RandomState(3)
    n = 50
    J = np.triu((rng.rand(n, n) < 0.2) * rng.randint(-3, 4, (n, n)), 1).astype(np.float32)
    J = J + J.T
    h = rng.randint(-2, 3, n).astype(np.float32)
    rowptr = np.concatenate([[0], np.cumsum((J != 0).sum(1))]).astype(np.int32)
    colidx = np.concatenate([np.nonzero(J[i])[0] for i in range(n)]).astype(np.int32)
    val = np.concatenate([J[i][J[i] != 0] for i in range(n)]).astype(np.float32)
    pd, ps = oracle.Problem(J=J, h=h), oracle.Problem(csr=(rowptr, colidx, val), h=h)
    s = oracle.init_spins(n, 3, seed=9)
    assert np.array_equal(oracle.energy(pd, s), oracle.energy(ps, s))
    a, b = s.copy(), s.copy()
    oa = oracle.sweeps(pd, a, [3.0, 1.0, 0.3], 20, seed=5)
    ob = oracle.sweeps(ps, b, [3.0, 1.0, 0.3], 20, seed=5)
    assert np.array_equal(a, b) and np.array_equal(oa["energy_trace"], ob["energy_trace"])
