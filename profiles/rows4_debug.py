"""Find where the four-updates-per-step CSR form leaves the one-at-a-time chain: first sweep and replica whose
spins differ, then the first update of that sweep (oracle trace) at a site that differs, with its step mates."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle  # noqa: E402
import spin_glass_anneal_rl_amd as sg  # noqa: E402

n, deg, R = int(os.environ.get("N", 200)), int(os.environ.get("DEG", 6)), int(os.environ.get("R", 9))
rng = np.random.RandomState(n)
J = np.zeros((n, n), np.float32)
for i in range(n):
    for j in rng.choice(n, deg // 2, replace=False):
        if i != j:
            J[i, j] = J[j, i] = rng.choice([-1.0, 1.0])
h = rng.randint(-1, 2, n).astype(np.float32)
import scipy.sparse as sp
A = sp.csr_matrix(J)
A.sort_indices()
csr = (A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float32))
prob = oracle.Problem(csr=csr, h=h)
seed = 5150 + n
temps = np.geomspace(4.0, 0.4, R)
s = oracle.init_spins(n, R, seed)
with sg.AnnealEngine(0) as e:
    e.set_csr(*csr, h)
    e.init_replicas(R, seed=seed)
    e.set_temperatures(temps)
    print(e.describe())
    energy = None
    for k in range(12):
        before = s.copy()
        ref = oracle.sweeps(prob, s, temps, 1, seed=seed, sweep0=k, energy=energy, trace=True)
        energy = ref["energy"]
        e.sweep(1)
        got = e.spins()
        bad = np.nonzero((got != s).any(1))[0]
        if len(bad):
            for r in bad.tolist():
                sites_bad = np.nonzero(got[r] != s[r])[0]
                print(f"sweep {k}: replica {r}: sites {sites_bad.tolist()} differ")
                acc = ref["accept_trace"][r][:n]
                site_of = np.array([oracle.stream_site(seed, r, k, t, n) for t in range(n)])
                for t in range(n):
                    if site_of[t] in sites_bad:
                        m = t // 4
                        mates = list(range(4 * m, min(4 * m + 4, n)))
                        print(f"  update {t} (step {m}, row {t % 4}) site {site_of[t]} accepted {acc[t]}; step: "
                              + ", ".join(f"t{u}: site {site_of[u]} acc {acc[u]} nbrs {A.indices[A.indptr[site_of[u]]:A.indptr[site_of[u]+1]].tolist()}"
                                          for u in mates))
                # steps of this sweep in which an accepted update touches a later one
                for m in range((n + 3) // 4):
                    ts = list(range(4 * m, min(4 * m + 4, n)))
                    kinds = []
                    for ia, ta in enumerate(ts):
                        if not acc[ta]:
                            continue
                        for tb in ts[ia + 1:]:
                            if site_of[tb] == site_of[ta]:
                                kinds.append(f"same-site t{ta}->t{tb}")
                            elif site_of[ta] in A.indices[A.indptr[site_of[tb]]:A.indptr[site_of[tb] + 1]]:
                                kinds.append(f"neighbour t{ta}->t{tb}")
                    if kinds:
                        print(f"    step {m}: {kinds}")
            break
    else:
        print("no difference in 12 sweeps")
