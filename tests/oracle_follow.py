"""Follow single replicas of a large GPU run with the CPU oracle (test infrastructure).

The Philox streams are keyed by the GLOBAL replica id (site and uniform of update t of sweep k of replica r
do not depend on how many replicas run beside it), so any replica of a 1024- or 8192-replica run is reproduced
by a one-replica oracle run started at `replica0 = r` -- the hot end, the middle and the COLD end of a ladder
cost the same.  The cold end is where the accept table's boundary, the underflow of exp, the look-ahead replay
and the multi-wave tail forms of the kernels are exercised (reference rule: core/spin_dynamics.py:131-152)."""
import numpy as np

import oracle


def ladder_ends(R, extra=(1, 2)):
    """Replica ids to follow in a ladder of R: hot end, middle, cold end (+ a few neighbours of the hot end)."""
    return sorted({0, R // 2, R - 1, *[x for x in extra if x < R]})


def follow(prob, n, seed, temps_global, replicas, n_sweeps, exact_f32=False, **kw):
    """{r: (energy trace [n_sweeps], final spins int8 [n], accepted count)} for the global replica ids given."""
    out = {}
    if exact_f32:
        oracle.set_exact_f32(True)
    try:
        for r in replicas:
            s = oracle.init_spins(n, 1, seed, replica0=r)
            ref = oracle.sweeps(prob, s, np.asarray(temps_global, np.float64)[r:r + 1], n_sweeps, seed=seed, replica0=r,
                                n_threads=1, **kw)
            out[r] = (ref["energy_trace"][:, 0].copy(), s[0].copy(), int(ref["n_accepted"][0]))
    finally:
        if exact_f32:
            oracle.set_exact_f32(False)
    return out
