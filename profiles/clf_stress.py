"""Stress of the cached-field sweep's multi-wave synchronisation (the race of round 3 showed only at scale):
random sizes, replica counts, storages, waves per replica and ladders (hot ones: millions of accepts);
after every block of sweeps the tracked energies must equal the energies recomputed from the spins, and the
fields rebuilt from the spins must continue the same chain as the resident ones (two engines side by side:
one keeps its fields across blocks, the other goes through export / import before every block).
usage: clf_stress.py <seconds> [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spin_glass_anneal_rl_amd as sg  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)
t_end, t_note = time.time() + budget, time.time()
cases = fails = 0
accepts = 0
while time.time() < t_end:
    if time.time() - t_note > 60.0:
        print(f"... {cases} cases, {fails} failures, {accepts:.3g} accepts so far", flush=True)
        t_note = time.time()
    n = int(rng.choice([600, 1024, 2500, 4097, 7000, 10000, 12000]))
    R = int(rng.choice([64, 200, 512, 1024]))
    amp = int(rng.choice([1, 1, 3, 100]))
    storage = str(rng.choice(["i8", "f32", "auto"]))
    waves = int(rng.choice([0, 1, 2, 3, 4, 8]))
    hot = float(rng.choice([0.3, 3.0, 30.0]))  # ladder top in units of sqrt(n) * amp
    g = torch.Generator(device=dev)
    g.manual_seed(int(rng.randint(1 << 30)))
    J = torch.randint(-amp, amp + 1, (n, n), generator=g, device=dev, dtype=torch.int32).float()
    J = torch.triu(J, 1)
    J = J + J.T
    h = torch.randint(-amp, amp + 1, (n,), generator=g, device=dev, dtype=torch.int32).float()
    if rng.rand() < 0.3:
        h = h + 0.5
    scale = np.sqrt(n) * amp
    temps = np.geomspace(hot * scale, 0.01 * scale, R)
    seed = int(rng.randint(1 << 30))
    if waves:
        os.environ["SGA_CLF_WAVES"] = str(waves)
    else:
        os.environ.pop("SGA_CLF_WAVES", None)
    desc = f"n={n} R={R} amp={amp} storage={storage} waves={waves} hot={hot} seed={seed}"
    try:
        engines = []
        for _ in range(2):
            e = sg.AnnealEngine(0)
            e.set_field_cache("on")
            e.set_dense(J, h, storage=storage)
            e.init_replicas(R, seed=seed)
            e.set_ladder(temps)
            engines.append(e)
        a, b = engines
        ok = True
        for block in (1, 3, 6, 10):
            b.import_state(b.export_state())    # b's fields are rebuilt from the spins before every block
            for e in (a, b):
                e.sweep(block)
                e.exchange(count=False)
            ta, tb = a.energies(), b.energies()
            blob = a.export_state()
            a.recompute_energies()
            ra = a.energies()
            a.import_state(blob)                # (the tracked, exact energies stay the ones in use)
            # from-scratch energies: -1/2 fp32(sum_i mv_i s_i) - fp32(h.s) -- the sum is 2 |E|: exact below 2^22 or so
            small = np.abs(ra) < 2.0 ** 22
            ok = ok and np.array_equal(a.spins(), b.spins()) and np.array_equal(ta, tb)
            ok = ok and np.array_equal(ta[small], ra[small]) and np.allclose(ta, ra, rtol=2e-7, atol=0)
            a.set_field_cache("on")
        accepts += float(a.stats()[0].sum())
        if not ok:
            fails += 1
            print("MISMATCH", desc, "| spins equal", np.array_equal(a.spins(), b.spins()), "tracked equal", np.array_equal(ta, tb),
                  "max |tracked - recomputed|", float(np.max(np.abs(ta - ra))), "at |E|", float(np.abs(ra[np.argmax(np.abs(ta - ra))])),
                  "|", a.describe(), flush=True)
        for e in engines:
            e.close()
    except Exception as ex:  # noqa: BLE001
        fails += 1
        print("ERROR", desc, "|", str(ex)[:300], flush=True)
    cases += 1
print(f"{cases} cases, {fails} failures, {accepts:.3g} accepts")
