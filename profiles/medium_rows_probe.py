import os, sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, "/root/repo") if os.path.isdir("/root/repo") else None
sys.path.insert(0, os.getcwd())
import spin_glass_anneal_rl_amd as sg
from spin_glass_anneal_rl_amd.engine import last_kernel
def assignment(na, nt, seed):
    rng = np.random.RandomState(seed)
    n = na * nt
    idx = np.arange(n).reshape(na, nt)
    rows, cols = [], []
    for a in range(na):
        r = np.repeat(idx[a], nt); c = np.tile(idx[a], nt); k = r != c
        rows.append(r[k]); cols.append(c[k])
    for t in range(nt):
        r = np.repeat(idx[:, t], na); c = np.tile(idx[:, t], na); k = r != c
        rows.append(r[k]); cols.append(c[k])
    rows = np.concatenate(rows); cols = np.concatenate(cols)
    A = sp.coo_matrix((np.full(rows.size, -2.0), (rows, cols)), shape=(n, n)).tocsr()
    A.sort_indices()
    h = rng.randint(-3, 4, n).astype(np.float32)
    return A, h
for (na, nt, R) in [(100, 100, 1024), (100, 100, 4096), (50, 50, 4096), (32, 32, 4096)]:
    A, h = assignment(na, nt, 1)
    n = A.shape[0]
    with sg.AnnealEngine(0) as e:
        e.set_csr(A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float32), h)
        e.init_replicas(R, seed=3)
        e.set_ladder(np.geomspace(20.0, 0.5, R))
        e.sweep(3)
        e.enable_timing(True); e.kernel_time(reset=True)
        a0 = e.stats()[0].sum()
        e.sweep(10)
        launches, ms = e.kernel_time(reset=True)
        acc = (e.stats()[0].sum() - a0) / (10.0 * n * R)
        deg = A.nnz / n
        rate = R * n * 10 / (ms * 1e-3)
        print(f"assignment {na}x{nt} n={n} deg={deg:.0f} R={R}: {ms/10:.3f} ms/sweep {rate:.3e} attempts/s  {rate*(deg*8+8)/1e12:.2f} TB/s algorithmic  acc {acc:.3f}  {e.describe()} | {last_kernel()}", flush=True)
