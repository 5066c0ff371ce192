"""Where the chain-wave cached-field sweep spends a replica's time: run with build/libsga_clfcprof.so
(bash profiles/build_variant.sh clfcprof sweep_clfc "-DCLFC_PROFILE"; SGA_LIBRARY_PATH=build/libsga_clfcprof.so).
The instrumented kernel returns its counters through the first rows of the energy trace."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spin_glass_anneal_rl_amd as sg  # noqa: E402

n, R = 10000, 1024
dev = torch.device("cuda", 0)
J = bench.make_sk_instance(n, 2, dev)
names = ["windows", "accepts", "unpredicted", "listed", "cut@64", "budget stops", "gen ticks", "chain ticks", "drain ticks",
         "gather-wait ticks", "total ticks", "table ticks"]
for flips in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "12").split(",")]:
    with sg.AnnealEngine(0) as e:
        e.set_option("clf_chain", 1)
        e.set_option("clf_flips", flips)
        e.set_field_cache("on")
        e.set_dense(J, torch.zeros(n, device=dev), storage="auto")
        e.init_replicas(R, seed=42)
        e.set_ladder(bench.geometric_ladder(R))
        done = 0
        for lo, hi in ((5, 25), (100, 120)):
            while done < lo:
                k = min(10 - done % 10, lo - done)
                e.sweep(k)
                done += k
                if done % 10 == 0:
                    e.exchange(count=False)
            a0 = e.stats()[0].copy()
            out = e.sweep(hi - lo, energy_trace=True)["energy_trace"]   # ONE launch of 20 sweeps: counters in rows 0..11
            done = hi
            acc = e.stats()[0] - a0
            c = out[:12, :]
            hot = int(np.argmax(c[10]))
            sw = hi - lo
            print(f"K={flips} sweeps {lo}..{hi}: slowest replica {hot} (accepts/sweep {acc[hot] / sw:.1f}, mean over replicas {acc.mean() / sw:.1f})")
            for who, col in (("slowest", c[:, hot]), ("mean   ", c.mean(1))):
                us = lambda t: t / 100.0 / sw  # noqa: E731  (100 MHz ticks -> us per sweep)
                print(f"   {who}: per sweep: windows {col[0] / sw:.1f} accepts {col[1] / sw:.1f} (not gathered ahead {col[2] / sw:.1f}) "
                      f"listed/window {col[3] / max(col[0], 1):.1f} cut {col[4] / sw:.2f} budget stops {col[5] / sw:.2f} | us/sweep: total "
                      f"{us(col[10]):.1f} = table {us(col[11]):.1f} + gen {us(col[6]):.1f} + chain {us(col[7]):.1f} (gather waits "
                      f"{us(col[9]):.1f}) + drain {us(col[8]):.1f}", flush=True)
