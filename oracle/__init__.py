"""ctypes loader for the CPU oracle (oracle/sg_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package.  Parity: pinned against tests/golden.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SG_ORACLE_LIBRARY: another build of the same source (oracle/Makefile `asan`: ASan + UBSan, tests/test_sanitizers.py)
_SO = os.environ.get("SG_ORACLE_LIBRARY") or os.path.join(_HERE, "libsg_oracle.so")

SITE_RANDOM, SITE_SEQUENTIAL, SITE_REPLAY = 0, 1, 2
ARITH_F64, ARITH_F32 = 0, 1
RULE_METROPOLIS, RULE_GLAUBER, RULE_HEAT_BATH, RULE_WOLFF = 0, 1, 2, 3


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("sg_oracle.c", "sg_oracle.h", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        target = ["asan"] if _SO.endswith("_asan.so") else []
        subprocess.run(["make", "-C", _HERE, "-B"] + target, check=True, capture_output=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        p = C.c_void_p
        L.sgo_philox4x32_10.argtypes = [p, p, p]
        L.sgo_philox4x32_10.restype = None
        L.sgo_expf.argtypes = [C.c_float]
        L.sgo_expf.restype = C.c_float
        L.sgo_exp.argtypes = [C.c_double]
        L.sgo_exp.restype = C.c_double
        L.sgo_local_field.argtypes = [C.c_int, p, C.c_int64, p, p, p, p, p, C.c_int]
        L.sgo_local_field.restype = C.c_double
        L.sgo_energy.argtypes = [C.c_int, p, C.c_int64, p, p, p, p, p]
        L.sgo_energy.restype = C.c_double
        L.sgo_metropolis_update.argtypes = [C.c_int, p, C.c_int64, p, p, p, p, p, C.c_int,
                                            C.c_double, C.c_float, C.c_int, C.c_int, p]
        L.sgo_metropolis_update.restype = C.c_int
        L.sgo_sweeps.argtypes = [C.c_int, p, C.c_int64, p, p, p, p, C.c_int, p, p, p, C.c_int64,
                                 C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32,
                                 C.c_uint32, p, p, C.c_int, C.c_int64, p, p, p, p, p, p, C.c_int,
                                 C.c_int]
        L.sgo_sweeps.restype = C.c_int
        L.sgo_pt_exchange_round.argtypes = [C.c_int, p, p, p, C.c_int, p, C.c_uint64, C.c_uint32,
                                            C.c_uint32, p, p]
        L.sgo_pt_exchange_round.restype = C.c_int
        L.sgo_tsp_to_csr.argtypes = [C.c_int, p, C.c_float, C.c_float, p, p, p]
        L.sgo_tsp_to_csr.restype = C.c_int
        L.sgo_pt_exchange_pairs.argtypes = [C.c_int, p, p, p, p, p, C.c_int, C.c_uint64, C.c_uint32, p, p]
        L.sgo_pt_exchange_pairs.restype = C.c_int
        L.sgo_pt_exchange_operator.argtypes = [C.c_int, C.c_int, p, p, p, p]
        L.sgo_pt_exchange_operator.restype = C.c_int
        L.sgo_init_spins.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint32, p]
        L.sgo_init_spins.restype = None
        L.sgo_set_exact_f32.argtypes = [C.c_int]
        L.sgo_set_exact_f32.restype = None
        L.sgo_stream_site.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.sgo_stream_site.restype = C.c_uint32
        L.sgo_stream_u.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.sgo_stream_u.restype = C.c_float
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


class Problem:
    """J (dense fp32 [n,n]) or CSR (rowptr, colidx, val) plus h."""

    def __init__(self, J=None, h=None, csr=None):
        if J is not None:
            self.J = _c(J, np.float32)
            self.n = self.J.shape[0]
            self.ld = self.J.shape[1]
            self.rowptr = self.colidx = self.val = None
        else:
            rp, ci, v = csr
            self.J = None
            self.ld = 0
            self.rowptr, self.colidx, self.val = _c(rp, np.int32), _c(ci, np.int32), _c(v, np.float32)
            self.n = len(self.rowptr) - 1
        self.h = _c(np.zeros(self.n) if h is None else h, np.float32)

    def args(self):
        return (self.n, _ptr(self.J), self.ld, _ptr(self.rowptr), _ptr(self.colidx),
                _ptr(self.val), _ptr(self.h))


def set_exact_f32(on):
    """Allow fp32 SIMD accumulation (only valid for integer-valued J with exact row sums)."""
    lib().sgo_set_exact_f32(1 if on else 0)


def philox(ctr, key):
    c, k = np.asarray(ctr, np.uint32), np.asarray(key, np.uint32)
    o = np.zeros(4, np.uint32)
    lib().sgo_philox4x32_10(_ptr(c), _ptr(k), _ptr(o))
    return o


def expf(x):
    return float(lib().sgo_expf(float(x)))


def exp(x):
    return float(lib().sgo_exp(float(x)))


def local_field(prob, s, i):
    s = _c(s, np.int8)
    return float(lib().sgo_local_field(*prob.args(), _ptr(s), int(i)))


def energy(prob, s):
    s = _c(s, np.int8)
    if s.ndim == 1:
        return float(lib().sgo_energy(*prob.args(), _ptr(s)))
    return np.asarray([float(lib().sgo_energy(*prob.args(), _ptr(np.ascontiguousarray(r))))
                       for r in s])


def metropolis_update(prob, s, site, T, u, arith=ARITH_F64, rule=RULE_METROPOLIS):
    """In-place on s (int8 contiguous). Returns (accepted, dE)."""
    assert s.dtype == np.int8 and s.flags.c_contiguous
    d = C.c_double(0.0)
    a = lib().sgo_metropolis_update(*prob.args(), _ptr(s), int(site), float(T), float(u),
                                    int(arith), int(rule), C.byref(d))
    return bool(a), d.value


def sweeps(prob, spins, temps, n_sweeps, site_mode=SITE_RANDOM, arith=ARITH_F64,
           rule=RULE_METROPOLIS, seed=0, sweep0=0,
           replica0=0, replay_site=None, replay_u=None, u_compact=False, energy=None,
           best_energy=None, recompute_energy=False, trace=False, n_threads=1):
    """Run R replicas x n_sweeps sweeps.  spins [R,n] int8 is updated in place.

    temps: scalar, [R], or [n_sweeps, R] temperatures.  Returns a dict.
    """
    n = prob.n
    spins2 = spins.reshape(-1, n)
    assert spins2.dtype == np.int8 and spins2.flags.c_contiguous
    R = spins2.shape[0]
    t = np.asarray(temps, np.float64)
    if t.ndim == 0:
        t = np.full(R, float(t))
    if t.ndim == 1:
        assert t.shape[0] == R
        ss, rs = 0, 1
    else:
        assert t.shape == (n_sweeps, R)
        ss, rs = R, 1
    t = np.ascontiguousarray(t)
    if energy is None:
        energy = np.asarray([lib().sgo_energy(*prob.args(), _ptr(np.ascontiguousarray(spins2[r])))
                             for r in range(R)], np.float64)
    energy = np.ascontiguousarray(energy, np.float64).copy()
    if best_energy is None:
        best_energy = energy.copy()
    best_energy = np.ascontiguousarray(best_energy, np.float64).copy()
    best_spins = spins2.copy()
    e_trace = np.zeros((n_sweeps, R), np.float64)
    n_acc = np.zeros(R, np.int64)
    per = n_sweeps * n
    acc_tr = np.zeros((R, per), np.uint8) if trace else None
    dE_tr = np.zeros((R, per), np.float64) if trace else None
    rs_ = _c(replay_site, np.int32)
    ru_ = _c(replay_u, np.float32)
    ucap = 0
    if rs_ is not None:
        assert rs_.size == R * per
    if ru_ is not None:
        if u_compact:
            ru_ = ru_.reshape(R, -1)
            ucap = ru_.shape[1]
        else:
            assert ru_.size == R * per
    rc = lib().sgo_sweeps(*prob.args(), R, _ptr(spins2), _ptr(energy), _ptr(t), ss, rs,
                          int(n_sweeps), int(site_mode), int(arith), int(rule), int(seed), int(sweep0),
                          int(replica0), _ptr(rs_), _ptr(ru_), int(bool(u_compact)), int(ucap),
                          _ptr(e_trace), _ptr(n_acc), _ptr(best_energy), _ptr(best_spins),
                          _ptr(acc_tr), _ptr(dE_tr), int(bool(recompute_energy)), int(n_threads))
    if rc != 0:
        raise RuntimeError(f"sgo_sweeps failed rc={rc}")
    return dict(energy=energy, energy_trace=e_trace, n_accepted=n_acc, best_energy=best_energy,
                best_spins=best_spins, accept_trace=acc_tr, dE_trace=dE_tr)


def pt_exchange_round(slot_temps, rep_energy, slot_to_rep, start=-1, u=None, seed=0, round_=0,
                      ladder=0, attempts=None, accepts=None):
    """In place on slot_to_rep (int32), attempts/accepts (int64)."""
    t = _c(slot_temps, np.float64)
    e = _c(rep_energy, np.float64)
    assert slot_to_rep.dtype == np.int32 and slot_to_rep.flags.c_contiguous
    uu = _c(u, np.float64)
    return int(lib().sgo_pt_exchange_round(len(t), _ptr(t), _ptr(e), _ptr(slot_to_rep), int(start),
                                           _ptr(uu), int(seed), int(round_), int(ladder), _ptr(attempts),
                                           _ptr(accepts)))


def tsp_to_csr(dist, city_visit, position_fill):
    """(rowptr int64, colidx int32, val float32) of the TSP-structured couplings, written out."""
    d = np.ascontiguousarray(dist, np.float32)
    n = d.shape[0]
    nnz = 4 * (n - 1) * n * n
    rowptr, col, val = np.zeros(n * n + 1, np.int64), np.zeros(nnz, np.int32), np.zeros(nnz, np.float32)
    if lib().sgo_tsp_to_csr(n, _ptr(d), float(city_visit), float(position_fill), _ptr(rowptr), _ptr(col),
                            _ptr(val)) != 0:
        raise RuntimeError("sgo_tsp_to_csr: bad arguments")
    return rowptr, col, val


def pt_exchange_pairs(slot_temps, rep_energy, slot_to_rep, pairs, u=None, seed=0, round_=0,
                      attempts=None, accepts=None):
    """In place on slot_to_rep (int32), attempts/accepts (int64); pairs [count, 2] slot indices."""
    t, e = _c(slot_temps, np.float64), _c(rep_energy, np.float64)
    assert slot_to_rep.dtype == np.int32 and slot_to_rep.flags.c_contiguous
    pr = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    uu = _c(u, np.float64)
    rc = int(lib().sgo_pt_exchange_pairs(len(t), _ptr(t), _ptr(e), _ptr(slot_to_rep), _ptr(pr), _ptr(uu),
                                         len(pr), int(seed), int(round_), _ptr(attempts), _ptr(accepts)))
    if rc < 0:
        raise RuntimeError("sgo_pt_exchange_pairs: slot index out of range")
    return rc


def pt_exchange_operator(spins, energies, temps, u):
    """In place on spins [R,n] int8 and energies [R] float32."""
    assert spins.dtype == np.int8 and spins.flags.c_contiguous
    assert energies.dtype == np.float32 and energies.flags.c_contiguous
    t, uu = _c(temps, np.float32), _c(u, np.float32)
    R, n = spins.shape
    return int(lib().sgo_pt_exchange_operator(R, n, _ptr(spins), _ptr(energies), _ptr(t), _ptr(uu)))


def init_spins(n, R, seed, replica0=0):
    s = np.zeros((R, n), np.int8)
    lib().sgo_init_spins(int(n), int(R), int(seed), int(replica0), _ptr(s))
    return s


def stream_site(seed, replica, sweep, t, n):
    return int(lib().sgo_stream_site(int(seed), int(replica), int(sweep), int(t), int(n)))


def stream_u(seed, replica, sweep, t):
    return float(lib().sgo_stream_u(int(seed), int(replica), int(sweep), int(t)))
