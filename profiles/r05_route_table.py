"""Writes tests/golden/route_table.json (run on the GPU box): the form selection's inputs and answers for the five
BASELINE configs as bench.py builds them and for 30 shapes drawn the way profiles/fuzz_parity.py draws them.

For every case a real engine is set up; its own sga_route_query (what the set-time scans found, replicas, tuning,
options) is recorded with the answer of sga_explain_route, sga_describe and the kernel the first sweep launched -- and
checked for agreement (waves, spins, updates per step, kernel family) before anything is written.  The CPU test
tests/test_host_logic.py::test_route_table rebuilds every query from the recorded fields and pins the answer, so an edit of
csrc/sga_route.cpp that reroutes a BASELINE config fails without a GPU.

    python profiles/r05_route_table.py [--big]     (--big: BASELINE configs at full size; else only the fuzz shapes)"""
import json
import os
import re
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import spin_glass_anneal_rl_amd as sg  # noqa: E402
from spin_glass_anneal_rl_amd import _native as N  # noqa: E402

FIELDS = [f for f, _ in N.RouteQuery._fields_ if f not in ("opt", "reserved_")]


def query_dict(q):
    d = {f: int(getattr(q, f)) for f in FIELDS}
    names = N.option_names()
    base = N.route_query()
    d["options"] = {k: int(q.opt[i]) for i, k in enumerate(names) if int(q.opt[i]) != int(base.opt[i])}
    return d


def consistent(explain, describe, kernel):
    """The answer of the pure function against what the engine did."""
    kv = dict(re.findall(r"(\w+)=([^\s(]+)", explain.split(" cached=")[0]))
    dv = dict(re.findall(r"(\w+)=([^\s(]+)", describe.split(" sweep=")[0]))
    if explain.startswith("dense"):
        assert kv["waves"] == dv["waves_per_replica"] and kv["chunks_per_wave"].rstrip("(streaming)") == dv["chunks_per_wave"].rstrip("(streaming)"), (explain, describe)
        assert kv["storage"] == dv["storage"] and kv["look_ahead"] == dv["look_ahead"], (explain, describe)
        want = "sweep_dense"
        assert kernel.startswith(want) or kernel.startswith("sweep_clf"), (explain, kernel)
        if kernel.startswith("sweep_dense_kernel"):
            assert f"x {kv['waves']} wave(s)" in kernel, (explain, kernel)
    elif explain.startswith("csr"):
        assert kv["waves"] == dv["waves_per_replica"] and kv["replicas_per_block"] == dv["replicas_per_block"], (explain, describe)
        assert ("lds-bits" if kv["spins"] == "bits" else "lds-int8") == dv["spins"], (explain, describe)
        assert kv["sstride"] == dv["sstride"] and kv["table_m"] == dv["table_m"], (explain, describe)
        assert (kv["entries"] == "packed") == ("entries=packed-32bit" in describe), (explain, describe)
        assert (kv["slots"] == "1") == ("rows=64-entry-slots" in describe), (explain, describe)
        fam = kv["form"]
        if kernel.startswith("sweep_clf"):
            return
        if fam == "rows":
            assert kernel.startswith(f"sweep_csr_rows_kernel<{kv['updates_per_step']} rows"), (explain, kernel)
            assert f"updates_per_step={kv['updates_per_step']}" in describe, (explain, describe)
        else:
            assert kernel.startswith("sweep_csr_kernel<"), (explain, kernel)
            assert ("narrow" in kernel) == fam.startswith("narrow"), (explain, kernel)
            assert ("bit spins" in kernel) == fam.endswith("bits"), (explain, kernel)
    else:
        assert kv["waves"] == dv["waves_per_replica"] and kv["passes"] == dv["passes"], (explain, describe)
        assert kernel.startswith("sweep_tsp"), (explain, kernel)


def record(name, eng, out):
    q = eng.route_query()
    explain = N.explain_route(q)
    eng.sweep(1)
    kernel = eng.last_kernel()
    describe = eng.describe()
    consistent(explain, describe, kernel)
    # the query after the first sweep equals the one before it up to what the layout call settled
    out.append({"name": name, "query": query_dict(q), "explain": explain, "describe": describe, "kernel": kernel})
    print(name, "|", explain, "|", kernel, flush=True)


def fuzz_shapes(out, count=30, seed=20261005):
    rng = np.random.RandomState(seed)
    done = 0
    while done < count:
        kind = str(rng.choice(["dense", "csr", "csr"]))
        n = int(rng.choice([8, 17, 64, 65, 200, 257, 700, 1025, 2049, 3000, 4100, 5000]))
        R = int(rng.choice([1, 3, 16, 33, 70, 300]))
        integer = rng.rand() < 0.6
        fixed = (not integer) and rng.rand() < 0.5
        deg = int(rng.choice([4, 30, 100, 200, 700]))
        dens = float(rng.choice([0.05, 0.3, 1.0])) if kind == "dense" else min(1.0, deg / n)
        vals = rng.randint(-2, 3, (n, n)) if integer else (np.rint(rng.randn(n, n) * 1024.0) / 1024.0 if fixed else rng.randn(n, n))
        J = np.triu(vals * (rng.rand(n, n) < dens), 1).astype(np.float32)
        J = J + J.T
        h = (rng.randint(-2, 3, n) if integer else rng.randn(n)).astype(np.float32)
        if integer and rng.rand() < 0.3:
            h = h + np.float32(0.5)
        storage = "auto"
        if kind == "dense" and integer:
            storage = str(rng.choice(["auto", "f32", "i8"]))
        waves = int(rng.choice([0, 0, 0, 1, 2, 4, 8]))
        opts = {}
        if kind == "csr" and rng.rand() < 0.3:
            opts["force_csr_bits"] = 1
        if kind == "csr" and rng.rand() < 0.3:
            opts["csr_updates_per_step"] = int(rng.choice([0, 1, 2, 4, 8]))
        cache = str(rng.choice(["off", "off", "auto", "on"]))
        name = f"fuzz{done:02d} {kind} n={n} R={R} int={int(integer)} fixed={int(fixed)} dens={dens:.3g} storage={storage} waves={waves} cache={cache} opts={opts}"
        try:
            with sg.AnnealEngine(0) as e:
                e.set_options(opts)
                e.set_tuning(waves_per_replica=waves)
                e.set_field_cache(cache)
                if kind == "dense":
                    e.set_dense(J, h, storage=storage)
                else:
                    rowptr = np.concatenate([[0], np.cumsum((J != 0).sum(1))]).astype(np.int32)
                    col = np.concatenate([np.nonzero(J[i])[0] for i in range(n)]).astype(np.int32)
                    val = np.concatenate([J[i][J[i] != 0] for i in range(n)]).astype(np.float32)
                    if col.size == 0:
                        continue
                    e.set_csr(rowptr, col, val, h)
                e.init_replicas(R, seed=7)
                e.set_temperatures(np.geomspace(3.0 * max(1.0, np.sqrt(n)), 0.2, R) if R > 1 else np.asarray([1.5]))
                record(name, e, out)
                done += 1
        except sg.AnnealingError as exc:  # (a forced form the problem does not admit, cache "on" on a real-valued problem)
            print("skipped:", name, "|", str(exc)[:100], flush=True)


def baseline_configs(out):
    import argparse
    import bench
    dev = torch.device("cuda", 0)
    a = argparse.Namespace(spins=10000, replicas=0, cities=100, implicit=False, storage="f32", workload="c2a")
    for name, kw in (("c2a", {}), ("c3", {}), ("c4", {}), ("c5", {}), ("c5_1000_implicit", dict(R=256, cities=1000, implicit=True))):
        wl = bench.build_workload(name.split("_")[0], a, dev, 1, **kw)
        with sg.AnnealEngine(0) as e:
            e.set_tuning(waves_per_replica=0, sweeps_per_launch=1)
            wl["load"](e)
            e.set_field_cache("off")
            e.init_replicas(wl["R"], seed=42)
            e.set_temperatures(np.geomspace(wl["t_hot"], wl["t_cold"], wl["R"]))
            record("BASELINE " + name + ": " + (wl["label"] or "C2a / C3") + f", {wl['R']} replicas", e, out)
        del wl
        torch.cuda.empty_cache()
    # configs[4] at 1000 cities with the 32 GB of CSR written out: the query alone (traits as the scans of the full-size
    # instance report them; recorded by tests/test_baseline_configs_gpu.py when it runs that instance)


if __name__ == "__main__":
    cases = []
    if "--big" in sys.argv:
        baseline_configs(cases)
    fuzz_shapes(cases)
    path = os.path.join(ROOT, "gpurun_out", "route_table.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump({"note": "written by profiles/r05_route_table.py on an MI355X; pinned by tests/test_host_logic.py::test_route_table",
                   "cases": cases}, f, indent=1)
    print(f"wrote {path}: {len(cases)} cases")
