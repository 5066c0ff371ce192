"""Round 4: the chain-wave form of the cached-field sweep (csrc/sweep_clfc_impl.h) against the windowed form
(csrc/sweep_clf_impl.h) on bench.py's C2a variant: 10 000-spin +-1 SK instance, 1024 replicas, ladder 10 -> 0.1,
exchange every 10 sweeps.  Kernel ms per sweep (HIP events) over sweeps 5..25, 25..45 and 100..120, one sweep per
launch (as bench.py's step) and ten per launch (as the tempering classes drive the engine).

    python profiles/r04_clfc_timing.py [flips=12,...] """
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
flips = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "12").split(",")]
t_hot, t_cold = float(os.environ.get("T_HOT", 10.0)), float(os.environ.get("T_COLD", 0.1))
variants = [("windowed", {"clf_chain": 0})] + [(f"chain K={k}", {"clf_chain": 1, "clf_flips": k}) for k in flips]
ref = None
for name, opts in variants:
    for per_launch in (1, 10):
        with sg.AnnealEngine(0) as e:
            e.set_options(opts)
            e.set_field_cache("on")
            e.set_dense(J, h, storage="auto")
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

            for lo, hi in ((5, 25), (25, 45), (100, 120)):
                run(lo)
                a0 = e.stats()[0].copy()
                e.kernel_time(reset=True)
                run(hi)
                launches, ms = e.kernel_time(reset=True)
                per = (e.stats()[0] - a0) / (hi - lo)
                res.append(f"sweeps {lo}..{hi}: {ms / (hi - lo):7.4f} ms/sweep ({R * n * (hi - lo) / (ms * 1e-3):.2e}/s; accepts "
                           f"mean {per.mean():6.1f} max {per.max():6.1f})")
            en = e.energies()
            if ref is None:
                ref = en
            print(f"{name:12s} {per_launch:2d} sweep(s)/launch  " + "  ".join(res) + f"  same chain: {np.array_equal(ref, en)}  "
                  f"[{last_kernel()[:40]}]", flush=True)
