"""Where the several-accepts-per-round cached-field sweep spends a replica's time: run with build/libsga_clfbprof.so
(bash profiles/build_variant.sh clfbprof sweep_clfb "-DCLFB_PROFILE"; SGA_LIBRARY_PATH=build/libsga_clfbprof.so).
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
for waves in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]:
    with sg.AnnealEngine(0) as e:
        e.set_options({"clf_batched": 1, "clf_tail_waves": 0, "clf_waves": waves})
        e.set_field_cache("on")
        e.set_dense(J, torch.zeros(n, device=dev), storage=os.environ.get("STORAGE", "auto"))
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
            c = out[:20, :]
            hot = int(np.argmax(c[10]))
            sw = hi - lo
            print(f"waves={waves} sweeps {lo}..{hi}: slowest replica {hot} (accepts/sweep {acc[hot] / sw:.1f}, mean over replicas {acc.mean() / sw:.1f})")
            for who, col in (("slowest", c[:, hot]), ("mean   ", c.mean(1))):
                us = lambda t: t / 100.0 / sw  # noqa: E731  (100 MHz ticks -> us per sweep)
                print(f"   {who}: per sweep: windows {col[0] / sw:.1f} rounds {col[1] / sw:.1f} rows {col[2] / sw:.1f} changed-decision rounds "
                      f"{col[3] / sw:.1f} listed/round {col[4] / max(col[1], 1):.1f} lookers/round (wave 0) {col[5] / max(col[1], 1):.1f} | us/sweep: "
                      f"total {us(col[10]):.1f} = table {us(col[11]):.1f} + draw {us(col[9]):.1f} + guess {us(col[6]):.1f} + check "
                      f"{us(col[7]):.1f} + apply {us(col[8]):.1f} || guess before barrier {us(col[12]):.1f}; check: list {us(col[13]):.1f}, "
                      f"through pairs {us(col[14]):.1f}, through publish {us(col[15]):.1f} (last wave: {us(col[19]):.1f}, lookers/round {col[18] / max(col[1], 1):.1f}); apply: commit {us(col[16]):.1f}, through rows {us(col[17]):.1f}", flush=True)
