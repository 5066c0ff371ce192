"""Round 4: the cached-field sweep that commits several accepts per round (csrc/sweep_clfb_impl.h) against the one
accept per round form (csrc/sweep_clf_impl.h) on bench.py's C2a variant: 10 000-spin +-1 SK instance, 1024 replicas,
ladder 10 -> 0.1, exchange every 10 sweeps.  Kernel ms per sweep (HIP events) over sweeps 5..25, 25..45 and
100..120, one sweep per launch (bench.py's step) and ten per launch (as the tempering classes drive the engine).

    [STORAGE=auto|f32|i8] [T_HOT=10] python profiles/r04_clfb_timing.py [waves,...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spin_glass_anneal_rl_amd as sg  # noqa: E402
from spin_glass_anneal_rl_amd.engine import last_kernel  # noqa: E402

n, R = int(os.environ.get("N", 10000)), int(os.environ.get("R", 1024))
dev = torch.device("cuda", 0)
J = bench.make_sk_instance(n, 2, dev)
h = torch.zeros(n, device=dev)
waves = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
t_hot, t_cold = float(os.environ.get("T_HOT", 10.0)), float(os.environ.get("T_COLD", 0.1))
storage = os.environ.get("STORAGE", "auto")
variants = [("one per round", {"clf_batched": 0, "clf_tail_waves": 0}), ("+ 8 waves in the tail", {"clf_batched": 0}), ("default", {})] + [(f"batched w={w}", {"clf_batched": 1, "clf_waves": w, "clf_tail_waves": 0}) for w in waves if 0 <= w <= 16] + \
           [(f"one/round w={-w}", {"clf_batched": 0, "clf_waves": -w}) for w in waves if -16 <= w < 0]
ref = None
for name, opts in variants:
    for per_launch in (1, 10):
        with sg.AnnealEngine(0) as e:
            e.set_options(opts)
            e.set_field_cache("on")
            e.set_dense(J, h, storage=storage)
            e.init_replicas(R, seed=42)
            e.set_ladder(bench.geometric_ladder(R, t_hot, t_cold))
            e.enable_timing(True)
            done, res = 0, []

            def run(upto):
                global done
                while done < upto:
                    k = min(per_launch, upto - done, 10 - done % 10)
                    e.sweep(k)
                    done += k
                    if done % 10 == 0:
                        e.exchange(count=False)

            for lo, hi in ((0, 5), (5, 25), (25, 45), (100, 120)):
                run(lo)
                a0 = e.stats()[0].copy()
                e.kernel_time(reset=True)
                run(hi)
                launches, ms = e.kernel_time(reset=True)
                per = (e.stats()[0] - a0) / (hi - lo)
                res.append(f"{lo}..{hi}: {ms / (hi - lo):7.4f} ms ({R * n * (hi - lo) / (ms * 1e-3):.2e}/s; acc "
                           f"{per.mean():6.1f}/{per.max():6.1f})")
            en = e.energies()
            if ref is None:
                ref = en
            print(f"{name:14s} {per_launch:2d}/launch  " + "  ".join(res) + f"  same chain: {np.array_equal(ref, en)}  "
                  f"[{last_kernel()[:60]}]", flush=True)
