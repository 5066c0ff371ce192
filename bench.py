#!/usr/bin/env python3
"""Benchmark of the Ising spin-sweep hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): spin-flip attempts per second = replicas x spins x sweeps / wall time.
Workload at N=1 (BASELINE configs[1], SURVEY.md 8(d) "C2a"): 10 000-spin dense +-1 SK instance
(fp32 couplings, 400 MB), 1024 replicas on a geometric temperature ladder 10 -> 0.1, random
sites with replacement (per-replica Philox streams).  One *step* = one Metropolis sweep of
every replica (R x N single-spin updates); every `--exchange-interval` steps a replica
exchange round runs (default 10, the reference's default).  With N > 1 GPUs every rank holds
its own 1024 replicas of one global ladder (weak scaling, J replicated); the exchange step
all-gathers the R_global energies over RCCL and every rank applies the same decisions.

Set-up (untimed): couplings to HBM, replicas, ladder, then `sga_autotune` -- the engine times
its feasible launch geometries on the resident replicas and keeps the fastest; the chain and the
state are unaffected (`--no-autotune` keeps the heuristic geometry) -- then W warm-up steps.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (the dense sweep):
algorithmic bytes = attempts x N x sizeof(J element) (one coupling-row read per attempt,
SURVEY.md 8(d)) divided by its HIP-event-timed launch duration.  `cpu_baseline` times the CPU
oracle (oracle/, a C restatement of the reference's algorithm; OpenMP over replicas) on a
bounded sample of the same workload on the host cores, replays that sample on the GPU and
reports the energy gap between the two (`energy_gap_vs_gpu`; 0 = bit-identical).
`--workload c3 | c4 | c5 [--cities N]` run the CSR configurations of BASELINE.json.
"""
import argparse
import json
import gc
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def geometric_ladder(R, tmax=10.0, tmin=0.1):
    # reference annealing/parallel_tempering.py:148-155, index 0 = hottest
    ratio = tmin / tmax
    return np.asarray([tmax * ratio ** (i / max(R - 1, 1)) for i in range(R)], np.float64)


def make_sk_instance(n, seed, device):
    """Dense symmetric +-1 couplings, zero diagonal (SURVEY.md 8(c) recipe), built on device."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    J = (torch.randint(0, 2, (n, n), generator=g, device=device, dtype=torch.int8) * 2 - 1)
    J = torch.triu(J, 1)
    J = (J + J.T).to(torch.float32)
    return J


def make_sparse_instance(n, half_degree, seed):
    """C3: random ~2*half_degree-regular symmetric graph, +-1 couplings, CSR."""
    import scipy.sparse as sp
    rng = np.random.RandomState(seed)
    rows = np.repeat(np.arange(n), half_degree)
    cols = rng.randint(0, n, rows.size)
    keep = rows != cols
    rows, cols = rows[keep], cols[keep]
    lo, hi = np.minimum(rows, cols), np.maximum(rows, cols)
    up = sp.coo_matrix((np.ones(lo.size, np.float32), (lo, hi)), shape=(n, n)).tocsr()
    up.data[:] = rng.randint(0, 2, up.nnz).astype(np.float32) * 2 - 1  # duplicates merged first
    A = (up + up.T).tocsr()
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float32)


def host_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" | "max <period>"
            quota, period = f.read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0:
                cores = min(cores, max(1, q // per))
        except (OSError, ValueError):
            pass
    return int(os.environ.get("SGA_CPU_THREADS", cores))


def cpu_baseline(J_host, n, seed, budget_replicas_per_core=8, sweeps=32, csr=None, h=None, eng=None,
                 t_range=(10.0, 0.1), scale=1.0):
    """Time the CPU port on a bounded sample and, with `eng` (already loaded with the same
    couplings), replay the identical sample -- same seed, replica ids, temperatures -- on the GPU:
    the energy gap between the two is the metric's "best-energy gap vs ref" (0 = bit-identical)."""
    import oracle
    cores = host_cores()
    R = max(cores * budget_replicas_per_core, 1)
    if csr is not None:
        # sparse sweeps are cheap: scale the sample to seconds of CPU work
        sweeps = max(1, int(scale * sweeps * 100 * 32 * n / max(len(csr[1]), 1) * 10000 / n))
        prob = oracle.Problem(csr=csr, h=np.zeros(n, np.float32) if h is None else h)
    else:
        prob = oracle.Problem(J=J_host, h=np.zeros(n, np.float32))
    oracle.set_exact_f32(True)  # +-1 couplings: fp32 SIMD accumulation is exact
    s = oracle.init_spins(n, R, seed)
    e0 = np.zeros(R)
    temps = geometric_ladder(R, *t_range)
    t0 = time.perf_counter()
    res = oracle.sweeps(prob, s, temps, sweeps, seed=seed, energy=e0, n_threads=cores)
    dt = time.perf_counter() - t0
    # the faithful single-thread figure beside it, on a slice of the sample
    R1 = max(1, min(R, 8))
    sw1 = max(1, sweeps // 8)
    s1 = oracle.init_spins(n, R1, seed)
    t1 = time.perf_counter()
    oracle.sweeps(prob, s1, temps[:R1], sw1, seed=seed, energy=np.zeros(R1), n_threads=1)
    dt1 = time.perf_counter() - t1
    oracle.set_exact_f32(False)
    gap = None
    if eng is not None:  # outside every timed region
        eng.set_tuning(waves_per_replica=0, sweeps_per_launch=0)
        eng.init_replicas(R, seed=seed)
        eng.set_temperatures(temps)
        e_start = eng.energies()  # the CPU sample tracked the change from these
        eng.sweep(sweeps)
        e_gpu, e_cpu = eng.energies(), e_start + res["energy"]
        gap = {"max_abs_energy_gap": float(np.max(np.abs(e_gpu - e_cpu))),
               "best_energy_cpu": float(e_cpu.min()), "best_energy_gpu": float(e_gpu.min()),
               "spins_identical": bool(np.array_equal(eng.spins(), s))}
    return {"value": R * n * sweeps / dt, "unit": "spin-flip attempts/s", "cores": cores,
            "kind": "port", "energy_gap_vs_gpu": gap,
            "single_thread_value": R1 * n * sw1 / dt1,
            "build": "oracle/sg_oracle.c, gcc -O3 -fopenmp -ffp-contract=off -mavx2 -mfma (oracle/Makefile)",
            "sample": f"{R} replicas x {sweeps} sweep(s) of the same {n}-spin "
                      f"{'CSR' if csr is not None else 'dense'} instance, OpenMP over replicas"
                      f"{'' if csr is not None else ', fp32 SIMD row dot'}, {dt:.2f} s"}


def build_workload(name, a, dev, world=1, dist=None, backend="nccl", R=None, cities=None, implicit=None):
    """The problem of one BASELINE config, ready to load: {"n", "R", "n_ladders", "t_hot", "t_cold", "label",
    "J" (dense device matrix | None), "csr" (host arrays | (None, length-only, None) | None), "h", "load": f(engine)}.
    c2a: BASELINE configs[1] roofline instance; c3: configs[2]; c4: configs[3] (one rank's 1024 replicas);
    c5: configs[4] (TSP QUBO, n = cities^2; 100 cities unless --cities)."""
    from spin_glass_anneal_rl_amd import encoders as enc
    n = a.spins
    default_R = {"c2a": 1024, "c3": 4096, "c4": 1024, "c5": 2048}[name]
    R = R or (a.replicas if name == a.workload else 0) or default_R
    cities = cities or (a.cities if name == a.workload else 100)
    implicit = (bool(a.implicit) and name == a.workload) if implicit is None else implicit
    storage = a.storage if name == a.workload else "f32"
    out = {"n_ladders": 1, "t_hot": 10.0, "t_cold": 0.1, "label": "", "J": None, "csr": None, "implicit": implicit,
           "cities": cities, "storage": storage}
    bld = tsp = None
    if name == "c4":  # BASELINE configs[3]: 500 tasks x 100 slots, cardinality penalties
        bld = enc.scheduling_ising(np.full(500, 1.0), n_agents=1, time_horizon=100.0, time_discretization=100,
                                   objective="total_time", penalty_weights={"assignment": 100.0, "capacity": 50.0})
        out.update(t_hot=500.0, t_cold=5.0, label="C4: 50000-spin scheduling Ising (500 tasks x 100 slots)")
        n = bld.n
    elif name == "c5":  # BASELINE configs[4]: n = cities^2, degree 4(cities - 1)
        rs = np.random.RandomState(5)
        xy = rs.rand(cities, 2) * 100.0
        dmat = np.hypot(xy[:, None, 0] - xy[None, :, 0], xy[:, None, 1] - xy[None, :, 1])
        out["dmat"] = dmat
        tsp = None if implicit else enc.tsp_csr(dmat, city_visit=200.0, position_fill=200.0, device=dev)
        # BASELINE configs[4]: 32 ladders x 64 temperatures over 8 GPUs = 4 ladders per GPU; a smaller
        # replica count keeps the 64-temperature ladders (256 replicas on one GPU = one rank's share)
        out["n_ladders"] = (R * world) // 64 if (R * world) % 64 == 0 else 32
        out.update(t_hot=200.0, t_cold=2.0, label=f"C5: {cities}-city TSP QUBO ({cities ** 2} spins)")
        n = cities ** 2
    h = torch.zeros(n, device=dev)
    if name == "c2a":
        J = make_sk_instance(n, 2, dev)
        if dist is not None and backend == "nccl":
            # J is replicated: rank 0's matrix goes to everybody over RCCL (400 MB, once, untimed) instead
            # of trusting eight device generators to agree; the checksum below still verifies it
            dist.broadcast(J, src=0)
        out["J"] = J
        load = lambda eng: eng.set_dense(J, h, storage=storage)  # noqa: E731
    elif name == "c3":
        csr = make_sparse_instance(n, 16, 3)
        out["csr"] = csr
        load = lambda eng: eng.set_csr(*csr, h)  # noqa: E731
    elif bld is not None:
        csr = bld.to_csr()
        h = torch.from_numpy(bld.fields()).to(dev)
        out["csr"] = csr

        def load(eng):
            eng.set_csr_storage("f32")  # the graded figure: (column int32, value fp32) entries, B = deg * 8 + 8
            eng.set_csr(*csr, h)
    elif implicit:  # the couplings are never stored: distances + penalty weights + fields
        d32, w_city, w_pos, h_np, _ = enc.tsp_structure(dmat, 200.0, 200.0)
        h = torch.from_numpy(h_np).to(dev)
        d32_dev = torch.from_numpy(d32).to(dev)
        load = lambda eng: eng.set_tsp(d32_dev, w_city, w_pos, h)  # noqa: E731
    else:  # rows written on the device (int64 extents); host copy only while it is small
        h = tsp[3]
        nnz = int(tsp[1].numel())
        if nnz <= 200_000_000:
            out["csr"] = (tsp[0].cpu().numpy().astype(np.int32), tsp[1].cpu().numpy(), tsp[2].cpu().numpy())
        else:
            out["csr"] = (None, np.broadcast_to(np.int32(0), (nnz,)), None)  # length only
        holder = [tsp]

        def load(eng):
            t = holder[0]
            eng.set_csr(t[0], t[1], t[2], t[3])
            if nnz > 200_000_000:  # 32 GB: the caller's arrays are released once the engine holds its layout
                holder[0] = None
                del t
                torch.cuda.empty_cache()
    out.update(n=n, R=R, h=h, load=load)
    return out



def measured_copy_bandwidth(dev, nbytes=1 << 30, reps=5):
    """Device stream-copy rate (read + write bytes / s) on this box, GB/s -- the practical HBM
    ceiling next to the 8 TB/s spec figure (SURVEY.md 8d asks for both denominators)."""
    src = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        dst.copy_(src)
    ev1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (ev0.elapsed_time(ev1) * 1e-3) / 1e9


def pmc_traffic(tag, kernel):
    """HBM bytes per sweep-kernel launch from the committed rocprofv3 PMC passes
    (profiles/r*_<tag>_pmc.json, written by profiles/summarize_rocprof.py), or None."""
    import glob
    if tag is None:
        return None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}_pmc.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            for name, e in d["kernels"].items():
                if kernel in name and "hbm_bytes" in e:
                    return e["hbm_bytes"], os.path.relpath(path, ROOT)
        except (OSError, ValueError, KeyError):
            continue
    return None, None


SIMDS, CLOCK_HZ = 256 * 4, 2.4e9   # MI355X: 256 CUs x 4 SIMD-32, 2.4 GHz (MI355X_MICROARCH.md)
VALU_CYCLES = 2.0                  # a wave64 vector instruction holds its SIMD-32 for 2 cycles (same guide)


def pmc_counters(tag, kernel):
    """Per-launch counters of a kernel from the committed rocprofv3 passes (profiles/r*_<tag>_pmc.json)."""
    import glob
    if tag is None:
        return None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}_pmc.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            for name, e in d["kernels"].items():
                if kernel in name:
                    return e, os.path.relpath(path, ROOT)
        except (OSError, ValueError, KeyError):
            continue
    return None, None


IC_BYTES = 256 * 1024 * 1024   # Infinity Cache (MI355X_MICROARCH.md)
IC_GATHER_GBS = 8600.0          # that guide's measured chip-wide rate of random row gathers from an Infinity-Cache-resident
                                # table (38 MB, 1,152-B rows: 8.6 TB/s); it gives no spec figure for the cache


def roofline_block(wl, name, R, avg_launch_s, launches, kernel_inst, copy_gbs=None, read_gbs=None):
    """The `roofline` object of one bench line: algorithmic bytes per launch (one coupling row per
    attempt) over the HIP-event-timed launch duration, against the roof the streamed structure really has:
      * "hbm": the structure is larger than the 256 MiB Infinity Cache -- against the 8 TB/s HBM spec peak.  Between 256
        and 512 MiB part of every pass is still re-served on chip (`cache_served: true`): the ratio is printed raw, it
        can exceed 1, and the HBM-only figure is the line's `roofline_beyond_cache`;
      * "infinity-cache": the structure fits the Infinity Cache (C4: 239 MB, C5 at 100 cities: 32 MB) but not L2 -- against
        the guide's measured gather rate from an Infinity-Cache-resident table (it has no spec figure); the ratio to the
        HBM spec number is kept as `frac_of_hbm_spec`;
      * "valu-issue" / "latency": the committed PMC pass of this configuration shows that the bytes never leave L2
        (traffic < 0.2 x algorithmic): the line reports vector-instruction issue from the committed instruction counts
        and keeps the byte rate as `cache_served_GBs`.
    No ratio is capped."""
    n, csr, implicit, cities, storage = wl["n"], wl["csr"], wl["implicit"], wl["cities"], wl["storage"]
    elem = {"f32": 4, "i8": 1, "t2": 0.25}[storage]
    per_launch_attempts = float(R) * n  # one sweep per launch on this rank
    if implicit:
        bytes_per_attempt = 8.0 * cities + 8.0       # two fp32 distance rows + the field (its own byte model)
        structure_bytes = 8.0 * cities * cities      # the two distance tables
    elif csr is None:
        bytes_per_attempt = float(n * elem)          # one coupling row
        structure_bytes = float(n) * n * elem
    else:
        bytes_per_attempt = float(len(csr[1])) / n * 8.0 + 8.0   # deg*(val+idx) + row extent
        structure_bytes = float(len(csr[1])) * 8.0
    algo = per_launch_attempts * bytes_per_attempt
    achieved = algo / avg_launch_s / 1e9 if launches else 0.0
    # the committed PMC passes were taken on exactly these configurations
    pmc_tag = None
    if name == "c2a" and (n, R) == (10000, 1024):
        pmc_tag = f"c2a_{storage}"
    elif name == "c3" and (n, R) == (10000, 4096):
        pmc_tag = "c3_csr"
    elif name == "c4" and R == 1024:
        pmc_tag = "c4_csr"
    elif name == "c5" and (cities, R) in ((100, 2048), (1000, 256)):
        pmc_tag = "c5_csr" if cities == 100 else "c5_1000_csr"
    kernel_name = "sweep_tsp_kernel" if implicit else ("sweep_dense_kernel" if csr is None else "sweep_csr_kernel")
    if kernel_inst.startswith("sweep_csr_rows_kernel"):  # several updates per step (short integer rows: C3)
        kernel_name = "sweep_csr_rows_kernel"
    elif kernel_inst.startswith("sweep_tsp_par_kernel"):  # implicit TSP form, one update per wave
        kernel_name = "sweep_tsp_par_kernel"
    if implicit:
        pmc_tag = f"c5_{cities}_implicit"
    traffic, traffic_src = pmc_traffic(pmc_tag, kernel_name)
    counters, counters_src = pmc_counters(pmc_tag, kernel_name)
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
         "traffic_unit": "fabric-side bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes): HBM and "
                         "Infinity Cache together -- the counters sit at L2's far side",
         "traffic_source": traffic_src,
         "algorithmic_bytes_per_launch": algo, "structure_bytes": structure_bytes,
         "kernel": kernel_name, "kernel_instantiation": kernel_inst,
         "launches": launches, "avg_launch_ms": avg_launch_s * 1e3,
         "algorithmic_bytes_per_attempt": bytes_per_attempt}
    if traffic_src:  # which instantiation the committed profile holds (the autotuner's pick is deterministic: ties -> fewest waves)
        try:
            with open(os.path.join(ROOT, traffic_src)) as f:
                r["traffic_source_note"] = json.load(f).get("note")
        except (OSError, ValueError):
            pass
    if copy_gbs:
        r["measured_stream_copy_GBs"] = copy_gbs
        r["frac_of_measured_stream_copy"] = achieved / copy_gbs
    if read_gbs:
        r["measured_stream_read_GBs"] = read_gbs
        r["frac_of_measured_stream_read"] = achieved / read_gbs
    l2_resident = (traffic is not None and traffic < 0.2 * algo) or (traffic is None and structure_bytes < 3.0e7)
    if l2_resident or implicit:
        # not a bandwidth-bound kernel: the bytes are served by L2 (implicit TSP form: L2 and the Infinity Cache, half
        # each).  What paces it is how fast the SIMDs get through its (dependent) instruction stream: the vector-issue
        # fraction from the committed instruction counts of the same configuration -- wave-instructions x cycles per
        # instruction over SIMD-cycles of the launch.  profiles/r05_valu_issue_probe.txt: v_add / v_fma / v_and hold a
        # SIMD for 2 cycles, VOP3 / DPP / compare / convert / dot4 / readlane forms for 4 -- `frac` is the 2-cycle figure
        # (a lower bound), `frac_at_4_cycles_per_instruction` the upper one.
        r["cache_served_GBs"] = achieved
        r["cache_served"] = True
        r["bound"] = "valu-issue" if kernel_name in ("sweep_csr_rows_kernel", "sweep_tsp_par_kernel") else "latency"
        valu = (counters or {}).get("SQ_INSTS_VALU")
        issue_peak = SIMDS * CLOCK_HZ / VALU_CYCLES / 1e9   # G wave-instructions / s
        if valu and launches:
            rate = valu / avg_launch_s / 1e9
            r.update(achieved=rate, peak=issue_peak, unit="G wave-instr/s (VALU)", frac=rate / issue_peak,
                     issue_counters_source=counters_src, valu_wave_instructions_per_launch=valu,
                     salu_wave_instructions_per_launch=(counters or {}).get("SQ_INSTS_SALU"),
                     lds_wave_instructions_per_launch=(counters or {}).get("SQ_INSTS_LDS"),
                     frac_at_4_cycles_per_instruction=2.0 * rate / issue_peak)
            wc = (counters or {}).get("SQ_WAVE_CYCLES")
            if wc:  # where a wave's time goes (quad-cycles of its lifetime; MI355X_MICROARCH.md, rocprofv3 PMC slots)
                r["wave_time_shares"] = {
                    "waiting (s_waitcnt / barrier)": (counters.get("SQ_WAIT_ANY") or 0.0) / wc,
                    "issue stalled": (counters.get("SQ_WAIT_INST_ANY") or 0.0) / wc,
                    "instruction in flight": (counters.get("SQ_ACTIVE_INST_ANY") or 0.0) / wc,
                    "of which VALU": (counters.get("SQ_ACTIVE_INST_VALU") or 0.0) / wc,
                    "of which scalar": (counters.get("SQ_ACTIVE_INST_SCA") or 0.0) / wc,
                    "of which LDS": (counters.get("SQ_ACTIVE_INST_LDS") or 0.0) / wc}
        else:
            r.update(achieved=None, peak=issue_peak, unit="G wave-instr/s (VALU)", frac=None)
        r["note"] = ("cache resident (PMC traffic far below the algorithmic bytes): paced by " +
                     ("vector-instruction issue and the dependent chain of a step (sweep_csr_rows.hip / sweep_tsp.hip; "
                      "profiles/r05_experiments.md has the per-step cycle budget)"
                      if r["bound"] == "valu-issue" else
                      "the dependent chain of one update (reduction, barrier, decision)") +
                     ", not by HBM; `frac` = VALU wave-instructions x 2 cycles / SIMD-cycles of the launch, "
                     "`cache_served_GBs` = algorithmic bytes / launch time")
        if implicit:
            r["note"] += ("; different byte model from the graded CSR figure: the couplings are never stored, an attempt "
                          f"reads two {4 * cities}-byte rows of the scaled distance table "
                          f"({8 * cities ** 2 / 1e6:.0f} MB: L2 and Infinity Cache); the same chain as the CSR form, bit for bit")
    elif structure_bytes <= IC_BYTES:
        r.update(bound="infinity-cache", peak=IC_GATHER_GBS, frac=achieved / IC_GATHER_GBS, cache_served=True,
                 frac_of_hbm_spec=achieved / HBM_PEAK_GBS,
                 peak_source="MI355X_MICROARCH.md, 'Indexed rows': random rows of a 38 MB table (Infinity Cache) gathered at "
                             "8.6 TB/s chip-wide -- measured; the guide has no spec figure for the cache")
        r["note"] = (f"streamed structure = {structure_bytes / 1e6:.0f} MB <= 256 MiB: re-read from the Infinity Cache every sweep "
                     "(beyond L2: the fabric-side counters see every byte), not from HBM -- bandwidth bound against that "
                     "cache; the HBM-bound figures are roofline_beyond_cache (dense) and configs.c5_1000_csr (32 GB of CSR)")
    elif csr is not None:
        r["cache_served"] = False
        r["note"] = (f"CSR structure = {structure_bytes / 1e9:.1f} GB: streamed from HBM; bandwidth bound, DESIGN.md 4.2")
    elif name == "c2a":
        r["cache_served"] = bool(structure_bytes <= 2 * IC_BYTES)
        r["note"] = (
            f"{structure_bytes / 1e6:.0f} MB of couplings against a 256 MiB Infinity Cache: part of every pass is re-served "
            "on chip, so this algorithmic rate is a fabric figure and can touch or pass the HBM spec number (the ratio is "
            "printed raw); the HBM-bound figure on a matrix beyond every cache is roofline_beyond_cache")
    return r


def cached_csr_variant(eng, wl, R, n_ladders, ladder, comm_dev, exchange_interval, warm=20, steps=10):
    """The cached-local-field sweep over CSR couplings (sga_set_field_cache ON, csrc/sweep_clf_csr.hip) on the workload
    the engine holds: the same chain bit for bit, a row's entries read only when a proposal is ACCEPTED.  A variant with
    its own byte model -- B = acceptance rate x (deg x 8 + 8) bytes per attempt (last sentence) --
    reported beside the graded one-row-per-proposal figure, never instead of it."""
    from spin_glass_anneal_rl_amd.sharded import ShardedTempering
    n, csr = wl["n"], wl["csr"]
    eng.set_field_cache("on")
    eng.set_tuning(waves_per_replica=0, sweeps_per_launch=0)
    pt = ShardedTempering(eng, R_local=R, rank=0, world=1, seed=42, slot_temps=ladder, n_ladders=n_ladders, dist=None,
                          device=comm_dev)
    done = 0

    def run(k):  # sweeps between two exchange rounds go into one launch, as the tempering classes drive the engine
        nonlocal done
        while k > 0:
            chunk = min(k, exchange_interval - done % exchange_interval) if exchange_interval > 0 else k
            pt.sweep(chunk)
            done += chunk
            k -= chunk
            if exchange_interval > 0 and done % exchange_interval == 0:
                pt.exchange(count=False)

    try:
        run(warm)
    except Exception as exc:  # noqa: BLE001 - the problem does not qualify: the line says so
        eng.set_field_cache("off")
        return {"available": False, "reason": str(exc)}
    torch.cuda.synchronize()
    acc0 = int(eng.stats()[0].sum())
    eng.enable_timing(True)
    eng.kernel_time(reset=True)
    t1 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    launches, ms = eng.kernel_time(reset=True)
    eng.enable_timing(False)
    rate = (int(eng.stats()[0].sum()) - acc0) / (float(R) * n * steps)
    row_bytes = float(len(csr[1])) / n * 8.0 + 8.0
    val = float(R) * n * steps / dt
    out = {"available": True, "value": val, "unit": "attempts/s", "ms_per_step": dt / steps * 1e3,
           "kernel_ms_per_step": ms / steps, "sweeps": f"{warm}..{warm + steps}", "sweeps_per_launch": exchange_interval,
           "acceptance_rate": rate, "algorithmic_bytes_per_attempt": rate * row_bytes,
           "achieved_GBs": val * rate * row_bytes / 1e9, "kernel_instantiation": eng.last_kernel(), "geometry": eng.describe(),
           "roofline": {"bound": "latency (one serial chain per replica: evaluation rounds + one row fetch per accept)",
                        "byte_model": "B = acceptance rate x (deg x 8 + 8) bytes per attempt", "row_bytes": row_bytes,
                        "achieved": val * rate * row_bytes / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": val * rate * row_bytes / 1e9 / HBM_PEAK_GBS},
           "note": "bit-identical chain to the graded kernel (same Philox sites and uniforms, same accept rule; "
                   "tests/test_cached_fields_gpu.py): the dynamic part of the local fields (J s) resident in LDS as int16"}
    eng.set_field_cache("off")
    return out


def _all_gather_i64(dist, backend, value, world, comm_dev):
    mine = torch.tensor([value], dtype=torch.int64, device=comm_dev)
    every = torch.zeros(world, dtype=torch.int64, device=comm_dev)
    if backend == "nccl":
        dist.all_gather_into_tensor(every, mine)
    else:
        dist.all_gather(list(every.chunk(world)), mine)
    return every


def checksums_agree(eng, dist, backend, world, comm_dev):
    """Every rank must hold the same couplings (replicated, built per rank): (agree, this rank's checksum)."""
    checksum = eng.problem_checksum()
    if dist is None:
        return True, checksum
    every = _all_gather_i64(dist, backend, checksum - (1 << 64) if checksum >= (1 << 63) else checksum, world, comm_dev)
    if not bool((every == every[0]).all().item()):
        raise SystemExit(f"bench.py: ranks hold different couplings (checksums {every.tolist()})")
    return True, checksum


def config_line(name, a, dev, local_rank, comm_dev, rank=0, world=1, dist=None, warmup=5, steps=10,
                R=None, cities=None, implicit=None):
    """A short line of another BASELINE config, run after the headline in the default invocation: the same engine
    calls as `--workload <name>` (warm-up + timed sweeps, exchange every `--exchange-interval`), with a small CPU-oracle
    sample replayed on the GPU (energy gap) beside it at N = 1.  With N > 1 ranks the replicas are sharded as BASELINE
    states them: C4 = ONE ladder of R x N temperatures spanning the ranks (exchange = all-gather of the energies on the
    shared stream, identical decisions everywhere); C5 = whole 64-temperature ladders per rank (exchange rounds are
    local: no collective).  Timing: barrier + synchronize on both sides, MAX over ranks."""
    import spin_glass_anneal_rl_amd as sg
    from spin_glass_anneal_rl_amd.sharded import ShardedTempering
    from spin_glass_anneal_rl_amd import encoders as enc
    t_setup = time.perf_counter()
    wl = build_workload(name, a, dev, world, R=R, cities=cities, implicit=implicit)
    n, R, n_ladders = wl["n"], wl["R"], wl["n_ladders"]
    Rg = R * world
    t_built = time.perf_counter()
    eng = sg.AnnealEngine(local_rank)
    eng.use_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_tuning(waves_per_replica=0, sweeps_per_launch=1)
    wl["load"](eng)
    torch.cuda.synchronize()
    t_loaded = time.perf_counter()
    eng.set_field_cache("off")  # the graded form: one coupling-row read per proposal
    ladder = np.tile(geometric_ladder(Rg // n_ladders, wl["t_hot"], wl["t_cold"]), n_ladders)
    pt = ShardedTempering(eng, R_local=R, rank=rank, world=world, seed=42, slot_temps=ladder, n_ladders=n_ladders,
                          dist=dist, device=comm_dev, force_dist=a.force_dist)
    agree, checksum = checksums_agree(eng, dist, a.backend, world, comm_dev)
    done = 0

    def step():
        nonlocal done
        pt.sweep(1)
        done += 1
        if a.exchange_interval > 0 and done % a.exchange_interval == 0:
            pt.exchange(count=False)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # (as in the headline: no collector pause of the interpreter inside a timed region of a few milliseconds --
    #  profiles/r02_experiments.md 13; `kernel_ms_total` beside `ms_per_step` would show any other host gap)
    gc.collect()
    gc.disable()
    for _ in range(warmup):
        step()
    if a.exchange_interval > 0:
        pt.exchange()  # one untimed round: first-use allocations of the collective happen here (the same chain at any N)
    barrier()
    eng.enable_timing(True)
    eng.kernel_time(reset=True)
    pt.gather_calls, pt.gather_ms = 0, 0.0
    pt.time_collectives = True
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt_mine = time.perf_counter() - t0
    gc.enable()
    dt = dt_mine
    if dist is not None:
        tmax = torch.tensor([dt], device=comm_dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    launches, kernel_ms = eng.kernel_time(reset=True)
    eng.enable_timing(False)
    kernel_inst = eng.last_kernel()
    what = ("couplings implicit (TSP structure: 2 distance rows per attempt)" if wl["implicit"] else
            f"CSR mean degree {len(wl['csr'][1]) / n:.1f}")
    line = {"workload": (wl["label"] or f"C3: {n}-spin CSR +-1 Ising") +
                        f", {what}, {R} replicas/GPU, {n_ladders} geometric ladder(s) T "
                        f"{wl['t_hot']:g}->{wl['t_cold']:g}, exchange every {a.exchange_interval}",
            "value": float(Rg) * n * steps / dt, "unit": "attempts/s", "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "kernel_ms_total": kernel_ms, "geometry": eng.describe(),
            "kernel_instantiation": kernel_inst,
            "setup_ms": {"build_couplings": (t_built - t_setup) * 1e3, "engine_load": (t_loaded - t_built) * 1e3,
                         "note": "untimed set-up on this rank: writing the couplings, then sga_set_* (validation incl. the "
                                 "symmetry pass, layout, packing)"},
            "roofline": roofline_block(wl, name, R, (kernel_ms / max(launches, 1)) * 1e-3, launches, kernel_inst),
            "best_energy": eng.best(with_spins=False)[0]}
    if pt.dist is not None:   # N > 1 ranks, or the one-rank process group of --force-dist
        line.update({
            "n_gpus": world, "ranks_seen": dist.get_world_size(), "backend": dist.get_backend(), "scaling": "weak",
            "replicas_total": Rg, "couplings_checksum_agree": agree, "couplings_checksum": f"{checksum:016x}",
            "ms_per_step_this_rank": dt_mine / steps * 1e3,
            "placement": ("whole ladders per rank: exchange rounds are local, no collective (SURVEY.md 8e)"
                          if pt.ladders_local else
                          f"{n_ladders} ladder(s) of {Rg // n_ladders} temperatures over {world} rank(s), ladders may span ranks: "
                          "all-gather of the energies per round, identical Philox decisions on every rank, temperature "
                          "labels swap"),
            "exchange": {"rounds_timed": (done // a.exchange_interval - warmup // a.exchange_interval)
                                         if a.exchange_interval > 0 else 0,
                         "allgathers_timed": pt.gather_calls,
                         "allgather_ms_per_round": pt.gather_device_ms_per_round(),
                         "enqueue_ms_per_round": (pt.gather_ms / pt.gather_calls) if pt.gather_calls else None,
                         "bytes_per_rank": 0 if pt.ladders_local else 8 * R}})
    import hashlib
    e_all = pt.gather_energies().cpu().numpy() if pt.dist is not None else eng.energies()
    line["energies_sha256"] = hashlib.sha256(np.ascontiguousarray(e_all, np.float64).tobytes()).hexdigest()[:16]
    e_best, _, who = pt.global_best()
    line["best_energy_global"], line["best_replica_global"] = e_best, who
    if not a.no_cpu_baseline and world == 1 and rank == 0:
        if wl["csr"] is not None and wl["csr"][0] is not None:
            line["cpu_baseline"] = cpu_baseline(None, n, 42, csr=wl["csr"], h=wl["h"].cpu().numpy(), eng=eng,
                                                t_range=(wl["t_hot"], wl["t_cold"]), budget_replicas_per_core=2,
                                                scale=0.25)
        elif name == "c5":
            # no host copy of this instance (implicit couplings / 32 GB of CSR; the CPU port indexes entries with 32
            # bits): the CPU sample runs on the first 500 cities of the same point set, replayed on a second engine in
            # the same coupling form as this line
            sub = min(wl["cities"], 500)
            dsub = wl["dmat"][:sub, :sub]
            tsp2 = enc.tsp_csr(dsub, city_visit=200.0, position_fill=200.0, device=dev)
            csr2 = (tsp2[0].cpu().numpy().astype(np.int32), tsp2[1].cpu().numpy(), tsp2[2].cpu().numpy())
            h2 = tsp2[3].cpu().numpy()
            eng2 = sg.AnnealEngine(local_rank)
            if wl["implicit"]:
                d32, w_city, w_pos, h_np, _ = enc.tsp_structure(dsub, 200.0, 200.0)
                eng2.set_tsp(torch.from_numpy(d32).to(dev), w_city, w_pos, torch.from_numpy(h_np).to(dev))
            else:
                eng2.set_csr(tsp2[0], tsp2[1], tsp2[2], tsp2[3])
            del tsp2
            torch.cuda.empty_cache()
            line["cpu_baseline"] = cpu_baseline(None, sub * sub, 42, csr=csr2, h=h2, eng=eng2,
                                                t_range=(wl["t_hot"], wl["t_cold"]), budget_replicas_per_core=2,
                                                scale=0.25)
            line["cpu_baseline"]["substitute_instance"] = (
                f"the first {sub} of the {wl['cities']} cities ({sub * sub} spins, {len(csr2[1])} CSR entries) on the CPU; "
                f"replayed on the GPU in this line's coupling form "
                f"({'implicit' if wl['implicit'] else 'stored CSR'})")
            eng2.close()
            del csr2
    if name == "c4" and not a.no_variants and world == 1:
        line["variants"] = {"cached_local_fields": cached_csr_variant(eng, wl, R, n_ladders, ladder, comm_dev,
                                                                      a.exchange_interval)}
    eng.close()
    wl.clear()
    gc.collect()
    torch.cuda.empty_cache()
    line["wall_s_with_setup"] = time.perf_counter() - t_setup
    return line


def spawn_ranks(n_ranks):
    """`python bench.py --gpus N` without torchrun: this process has made no GPU call yet; it
    starts N child processes of the same command line, one rank per GPU, with the rendezvous in
    the environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), relays rank 0's
    JSON line and fails if any rank fails.  (Children are started, never exec'ed into; a rank that
    dies takes the others down with it -- exactly those PIDs -- instead of leaving them waiting in a
    collective.)"""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    with tempfile.TemporaryFile() as out0:
        for rank in range(n_ranks):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n_ranks),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out0 if rank == 0 else sys.stderr))
        codes = [None] * n_ranks
        while any(c is None for c in codes):
            for r, p in enumerate(procs):
                if codes[r] is None:
                    codes[r] = p.poll()
            if any(c not in (None, 0) for c in codes):  # one rank failed: stop the rest
                for r, p in enumerate(procs):
                    if codes[r] is None:
                        p.kill()
                        codes[r] = p.wait()
                break
            time.sleep(0.2)
        out0.seek(0)
        sys.stdout.write(out0.read().decode())
        sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        raise SystemExit(f"bench.py: ranks failed (rank, exit code): {bad}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2a", choices=["c2a", "c3", "c4", "c5"],
                    help="c2a: dense SK instance (headline); c3: CSR, degree ~32, 4096 replicas; "
                         "c4: 50k-spin scheduling penalties (CSR), 1024 replicas/GPU; "
                         "c5: 100-city TSP QUBO (CSR), 2048 replicas in 32 ladders")
    ap.add_argument("--spins", type=int, default=10000)
    ap.add_argument("--cities", type=int, default=100,
                    help="c5 only: TSP size (spins = cities^2; above 400 the spins are held as bits "
                         "in LDS; 1000 = BASELINE configs[4] at full size, 32 GB of CSR, use with "
                         "--replicas 256 = one GPU's share of the 2048)")
    ap.add_argument("--implicit", action="store_true",
                    help="c5 only: run the TSP-structured couplings without storing them (sga_set_tsp): a "
                         "different byte model (two distance rows per attempt), reported beside the CSR figure")
    ap.add_argument("--replicas", type=int, default=0, help="replicas per GPU (0 = workload default)")
    ap.add_argument("--storage", default="f32", choices=["f32", "i8", "t2"])
    ap.add_argument("--exchange-interval", type=int, default=10)
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--no-autotune", action="store_true",
                    help="keep the heuristic launch geometry instead of letting the engine time its "
                         "feasible geometries on the resident replicas before the warm-up "
                         "(sga_autotune: part of the set-up, results unaffected)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the short C3 / C4 / C5 lines that follow the headline in the default run")
    ap.add_argument("--configs", default=None,
                    help="comma list of the config lines to run after the headline (c3,c4,c5,c5_1000,c5_1000_csr); "
                         "given explicitly it overrides --no-variants / --no-configs")
    ap.add_argument("--config-replicas", type=int, default=0,
                    help="replicas per rank of the c4 / c5 config lines (tests: a small sharded run against the same "
                         "global replica set on one rank)")
    ap.add_argument("--no-c5-1000", action="store_true", help="skip the 1000-city lines of configs[4] in the default run")
    ap.add_argument("--no-c5-1000-csr", action="store_true",
                    help="skip only the stored-CSR 1000-city line (32 GB of couplings written out: ~15 s of set-up)")
    ap.add_argument("--no-beyond-cache", action="store_true",
                    help="skip the second roofline block (same kernel, 4.3 GB matrix beyond every cache)")
    ap.add_argument("--beyond-cache-spins", type=int, default=32768)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse "
                         "the multi-rank path on one GPU together with --share-device)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--force-dist", action="store_true",
                    help="keep torch.distributed in the path at --gpus 1: a one-rank process group (RCCL "
                         "with --backend nccl), started as a child process like the ranks of --gpus N; the "
                         "exchange step all-gathers through it.  Results equal the plain run")
    a = ap.parse_args()

    if (a.gpus > 1 or a.force_dist) and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(a.gpus)  # before anything here has touched the GPU
    # stdout carries ONE JSON line and nothing else: whatever libraries print while the process
    # group comes up (gloo's "[Gloo] Rank 0 is connected to ..." goes to fd 1) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (no CPU fallback)")
    if a.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # one side stream for everything: torch's work (probes, RCCL collectives) and the engine's kernels
    # are ordered by it, so an exchange round (energies -> all-gather -> exchange kernel) needs no
    # host synchronisation
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    comm_dev = dev if a.backend == "nccl" else torch.device("cpu")
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import spin_glass_anneal_rl_amd as sg
    from spin_glass_anneal_rl_amd.sharded import ShardedTempering

    wl = build_workload(a.workload, a, dev, world, dist=dist, backend=a.backend)
    n, R, n_ladders, t_hot, t_cold, label = wl["n"], wl["R"], wl["n_ladders"], wl["t_hot"], wl["t_cold"], wl["label"]
    J, csr, h = wl["J"], wl["csr"], wl["h"]
    dmat = wl.get("dmat")
    from spin_glass_anneal_rl_amd import encoders as enc
    Rg = R * world
    eng = sg.AnnealEngine(local_rank)
    eng.use_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_tuning(waves_per_replica=a.waves, sweeps_per_launch=1)
    wl["load"](eng)
    ladder = np.tile(geometric_ladder(Rg // n_ladders, t_hot, t_cold), n_ladders)
    eng.set_field_cache("off")  # the graded figure: one coupling-row read per proposal

    def tempering():
        return ShardedTempering(eng, R_local=R, rank=rank, world=world, seed=42, slot_temps=ladder,
                                n_ladders=n_ladders, dist=dist, device=comm_dev, force_dist=a.force_dist)

    pt = tempering()
    implicit = a.workload == "c5" and a.implicit
    autotuned = (not a.no_autotune) and csr is None and a.waves == 0 and not implicit
    autotune_ms = None
    if autotuned:
        if dist is None:
            eng.autotune()  # keeps its winner; sweeps per launch stay as set above
            autotune_ms = eng.autotune_table()
        else:
            # every rank times the SAME geometry: rank 0 measures, its winner is broadcast (results do
            # not depend on the geometry, the pace of the slowest rank would)
            w = torch.zeros(1, dtype=torch.int32, device=comm_dev)
            if rank == 0:
                eng.autotune()
                autotune_ms = eng.autotune_table()
                w[0] = eng.geometry()[0]
            dist.broadcast(w, src=0)
            eng.set_tuning(waves_per_replica=int(w.item()), sweeps_per_launch=1)
            pt = tempering()
    geometry = eng.describe()
    # every rank must hold the same couplings (J is replicated, built per rank from a seeded generator)
    checksum_agree, checksum = checksums_agree(eng, dist, a.backend, world, comm_dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    step_no = 0

    def step():
        nonlocal step_no
        pt.sweep(1)
        step_no += 1
        if a.exchange_interval > 0 and step_no % a.exchange_interval == 0:
            pt.exchange(count=False)  # no read-back: the round stays on the stream

    # Bandwidth probes (the two practical denominators) before the warm-up, on every rank, so that
    # nothing but the steps and the final barrier lies between the warm-up and the end of timing.
    from spin_glass_anneal_rl_amd.engine import probe_read_bandwidth
    copy_gbs = read_gbs = None
    if os.environ.get("SGA_BENCH_NOPROBE") is None:
        copy_gbs = measured_copy_bandwidth(dev)
        read_gbs = probe_read_bandwidth(local_rank)  # 4 GiB: beyond the caches
    torch.cuda.synchronize()
    # (round 1 paused 0.3 s here to step around a 30-40 ms gap that 5-15 % of fresh processes showed once
    # on the short-kernel workloads.  profiles/r02_experiments.md 13: it is a full collection of
    # CPython's garbage collector landing in the timed region -- the GPU idles while the host collects.
    # The collector is off from the warm-up to the end of timing, as in timeit; wall_ms_total vs
    # kernel_ms_total below would show any other gap.)
    if os.environ.get("SGA_BENCH_SETTLE"):
        time.sleep(float(os.environ["SGA_BENCH_SETTLE"]))
    if os.environ.get("SGA_BENCH_GC") is None:  # (SGA_BENCH_GC=1: leave the collector on, the A/B of experiments 13)
        gc.collect()
        gc.disable()  # as timeit does: no collector pause of the interpreter inside the timed region
    for _ in range(a.warmup):
        step()
    if a.exchange_interval > 0:
        pt.exchange()  # one untimed round: first-use allocations of the exchange path happen here
    barrier()
    eng.enable_timing(os.environ.get("SGA_BENCH_NOEVENTS") is None)
    eng.kernel_time(reset=True)
    pt.gather_calls, pt.gather_ms = 0, 0.0
    pt.time_collectives = True
    debug = os.environ.get("SGA_BENCH_DEBUG") is not None
    marks = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
        if debug:
            if os.environ.get("SGA_BENCH_DEBUG") == "sync":
                torch.cuda.synchronize()
            marks.append(time.perf_counter() - t0)
    barrier()
    dt = time.perf_counter() - t0
    if debug and rank == 0:
        print("step end marks (ms): " + " ".join(f"{m * 1e3:.1f}" for m in marks) + f" | total {dt * 1e3:.1f}",
              file=sys.stderr, flush=True)
    launches, kernel_ms = eng.kernel_time(reset=True)
    eng.enable_timing(False)
    gc.enable()
    if debug and rank == 0:
        print(f"kernel events: {launches} launches, {kernel_ms:.1f} ms", file=sys.stderr, flush=True)

    tmax = torch.tensor([dt], device=comm_dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    attempts = float(Rg) * n * a.steps
    value = attempts / dt
    elem = {"f32": 4, "i8": 1, "t2": 0.25}[a.storage]
    per_launch_attempts = float(R) * n  # one sweep per launch on this rank
    avg_launch_s = (kernel_ms / max(launches, 1)) * 1e-3
    best_e, _, _ = eng.best(with_spins=False)
    kernel_inst = eng.last_kernel()  # template arguments of what the timed steps launched (this engine's)
    roof = roofline_block(wl, a.workload, R, avg_launch_s, launches, kernel_inst, copy_gbs, read_gbs)

    out = {
        "metric": "spin-flip attempts/s (replicas x spins x sweeps / s)",
        "value": value,
        "unit": "attempts/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3,
        "wall_ms_total": dt * 1e3, "kernel_ms_total": kernel_ms,  # a gap between them = device idle
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if (a.storage == "f32" or csr is not None or implicit) else ("i8" if a.storage == "i8" else "b2"),
        "data": "synthetic",
        "ranks_seen": dist.get_world_size() if dist is not None else 1,
        "backend": (dist.get_backend() if dist is not None else None),
        "couplings_checksum_agree": checksum_agree, "couplings_checksum": f"{checksum:016x}",
        # all-gather of the energies: device time between two events on the shared stream (RCCL), not the host
        # time of the asynchronous enqueue; `enqueue_ms_per_round` is that host share
        "exchange": {"rounds_timed": pt.gather_calls, "allgather_ms_per_round": pt.gather_device_ms_per_round(),
                     "enqueue_ms_per_round": (pt.gather_ms / pt.gather_calls) if pt.gather_calls else None,
                     "bytes_per_rank": 8 * R},
        "config": {"workload": ((label + ", couplings implicit (TSP structure: 2 distance rows per attempt)")
                                if implicit else f"C2a: {n}-spin dense +-1 SK Ising" if csr is None else
                                (label or f"C3: {n}-spin CSR +-1 Ising") +
                                f", CSR mean degree {len(csr[1]) / n:.1f}") +
                               f", {R} replicas/GPU, {n_ladders} geometric ladder(s) T {t_hot:g}->"
                               f"{t_cold:g}, random-site Metropolis sweeps, exchange every "
                               f"{a.exchange_interval}",
                   "spins": n, "replicas_per_gpu": R, "replicas_total": Rg,
                   "coupling_storage": "implicit-tsp" if implicit else (a.storage if csr is None else "csr"),
                   "geometry": geometry,
                   "geometry_autotuned": autotuned,
                   # what the autotuner measured (ms per sweep by waves x chunks per wave); ties within 1 % go to the fewest
                   # waves, so the pick is the same on every box whose timings agree to that
                   "autotune_ms": autotune_ms,
                   "best_energy_rank0": best_e},
        "roofline": roof,
    }
    # the same workload with the couplings held as int8 / as two bit-planes (what
    # coupling_storage="auto" picks for integer / ternary J; exact arithmetic, identical
    # chain): reported beside the fp32 headline
    if a.workload == "c2a" and a.storage == "f32" and world == 1 and not a.no_variants:
        out["variants"] = {}
        for name, st, bytes_per in (("int8_couplings", "i8", float(n)),
                                    ("bit_plane_couplings", "t2", n / 4.0)):
            eng.set_tuning(waves_per_replica=0, sweeps_per_launch=1)  # not the fp32 winner
            eng.set_dense(J, h, storage=st)
            pt2 = ShardedTempering(eng, R_local=R, rank=0, world=1, seed=42, slot_temps=ladder,
                                   n_ladders=1, dist=None, device=comm_dev)
            if autotuned:
                eng.autotune()
            pt2.sweep(1)
            torch.cuda.synchronize()
            eng.enable_timing(True)
            eng.kernel_time(reset=True)
            t1 = time.perf_counter()
            pt2.sweep(4)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            l2, ms2 = eng.kernel_time(reset=True)
            eng.enable_timing(False)
            ach2 = per_launch_attempts * bytes_per / ((ms2 / max(l2, 1)) * 1e-3) / 1e9
            out["variants"][name] = {
                "value": float(R) * n * 4 / dt2, "unit": "attempts/s", "ms_per_step": dt2 / 4 * 1e3,
                "roofline": {"bound": "hbm", "achieved": ach2, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": ach2 / HBM_PEAK_GBS, "algorithmic_bytes_per_attempt": bytes_per},
                "geometry": eng.describe(),
                "note": "exact arithmetic, bit-identical chain to the fp32 layout" +
                        ("" if st == "i8" else "; 2 bits per coupling, popcount row sums "
                                               "(latency bound, not HBM bound)")}
    # The cached-local-field sweep (sga_set_field_cache): the same chain bit for bit, a coupling row read
    # only when a proposal is ACCEPTED.  A variant with its own byte model -- B = acceptance rate x row
    # bytes per attempt (last sentence) -- reported beside the graded one-row-per-proposal
    # figure, never instead of it.  Storage = what the engine picks by itself for these couplings.
    if a.workload == "c2a" and world == 1 and not a.no_variants:
        # (sweeps between two exchange rounds go into ONE launch, as ParallelTempering.run / SpinGlassScheduler.anneal
        #  drive the engine: a launch of this variant lasts as long as its hottest replica's chain, and ten sweeps
        #  average that chain's fluctuations; `one_sweep_per_launch` is the figure of rounds 3's line)
        eng.set_tuning(waves_per_replica=0, sweeps_per_launch=0)
        eng.set_dense(J, h, storage="auto")
        eng.set_field_cache("on")
        pt3 = ShardedTempering(eng, R_local=R, rank=0, world=1, seed=42, slot_temps=ladder, n_ladders=1,
                               dist=None, device=comm_dev)
        row_bytes = float(n if "storage=f32" not in eng.describe() else 4 * n)
        done = 0

        def clf_steps(k, per_launch=None):
            nonlocal done
            while k > 0:
                chunk = min(k, a.exchange_interval - done % a.exchange_interval) if a.exchange_interval > 0 else k
                chunk = min(chunk, per_launch or chunk)
                pt3.sweep(chunk)
                done += chunk
                k -= chunk
                if a.exchange_interval > 0 and done % a.exchange_interval == 0:
                    pt3.exchange(count=False)

        def clf_timed(k, per_launch=None):
            torch.cuda.synchronize()
            acc0 = int(eng.stats()[0].sum())
            eng.enable_timing(True)
            eng.kernel_time(reset=True)
            t1 = time.perf_counter()
            clf_steps(k, per_launch)
            torch.cuda.synchronize()
            dtv = time.perf_counter() - t1
            lv, msv = eng.kernel_time(reset=True)
            eng.enable_timing(False)
            rate = (int(eng.stats()[0].sum()) - acc0) / (float(R) * n * k)
            val = float(R) * n * k / dtv
            return {"value": val, "unit": "attempts/s", "ms_per_step": dtv / k * 1e3,
                    "kernel_ms_per_step": msv / k, "launches": lv, "acceptance_rate": rate,
                    "algorithmic_bytes_per_attempt": rate * row_bytes,
                    "achieved_GBs": val * rate * row_bytes / 1e9}

        clf_steps(a.warmup)
        first = clf_timed(a.steps)          # the same sweeps the headline times
        first["kernel_instantiation_these_sweeps"] = eng.last_kernel()   # (the engine picks the form by the hottest replica's acceptance)
        clf_steps(max(0, 100 - done))
        later = clf_timed(a.steps)          # after 100 sweeps of the same ladder
        later["kernel_instantiation_these_sweeps"] = eng.last_kernel()
        later["one_sweep_per_launch"] = clf_timed(a.steps, per_launch=1)
        tracked = eng.energies()
        eng.recompute_energies()
        exact = bool(np.array_equal(tracked, eng.energies()))
        out.setdefault("variants", {})["cached_local_fields"] = {
            **first, "sweeps": f"{a.warmup}..{a.warmup + a.steps} (the headline's)",
            "sweeps_per_launch": f"up to the next exchange round (interval {a.exchange_interval})",
            "after_100_sweeps": later,
            "tracked_energy_equals_recomputed": exact,
            "kernel_instantiation": eng.last_kernel(), "geometry": eng.describe(),
            "roofline": {"bound": "latency (one serial chain per replica: evaluation rounds + one row fetch per "
                                  "accept)", "byte_model": "B = acceptance rate x row bytes per attempt",
                         "row_bytes": row_bytes, "achieved": first["achieved_GBs"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": first["achieved_GBs"] / HBM_PEAK_GBS},
            "note": "bit-identical chain to the headline kernel (same Philox sites and uniforms, same accept rule; "
                    "tests/test_cached_fields_gpu.py): local fields resident in LDS, seeded by one MFMA pass over J; "
                    "speed-up over the headline = value / headline value"}
        eng.set_field_cache("off")
        eng.set_dense(J, h, storage=a.storage)
        pt = ShardedTempering(eng, R_local=R, rank=0, world=1, seed=42, slot_temps=ladder, n_ladders=1,
                              dist=None, device=comm_dev)
    # C4's couplings are small integers: what the engine picks by itself there is one dword per entry
    # (24-bit column | 8-bit value), half the bytes of a row, the same chain -- a storage variant
    # with its own byte model, reported beside the fp32-value figure
    if a.workload == "c4" and world == 1 and not a.no_variants:
        eng.set_csr_storage("auto")
        pt2 = ShardedTempering(eng, R_local=R, rank=0, world=1, seed=42, slot_temps=ladder,
                               n_ladders=n_ladders, dist=None, device=comm_dev)
        pt2.sweep(1)
        torch.cuda.synchronize()
        eng.enable_timing(True)
        eng.kernel_time(reset=True)
        t1 = time.perf_counter()
        pt2.sweep(4)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        l2, ms2 = eng.kernel_time(reset=True)
        eng.enable_timing(False)
        bytes_per = float(len(csr[1])) / n * 4.0 + 8.0
        ach2 = per_launch_attempts * bytes_per / ((ms2 / max(l2, 1)) * 1e-3) / 1e9
        out["variants"] = {"cached_local_fields": None, "packed_entries": {
            "value": float(R) * n * 4 / dt2, "unit": "attempts/s", "ms_per_step": dt2 / 4 * 1e3,
            "roofline": {"bound": "hbm", "achieved": ach2, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach2 / HBM_PEAK_GBS, "algorithmic_bytes_per_attempt": bytes_per},
            "geometry": eng.describe(),
            "note": "integer couplings as one dword per entry (24-bit column, 8-bit value), integer row sums: "
                    "bit-identical chain to the (column, fp32 value) layout; the default for such problems"}}
        out["variants"]["cached_local_fields"] = cached_csr_variant(eng, wl, R, n_ladders, ladder, comm_dev,
                                                                    a.exchange_interval)
        eng.set_csr_storage("f32")
        pt = tempering()  # back to the graded form (cpu_baseline replays on this engine)
    # The headline matrix (400 MB) is partly re-served by the 256 MB Infinity Cache, which the
    # fabric-side counters cannot tell from HBM.  The same kernel on a matrix far beyond every
    # cache (n = 32 768: 4.3 GB of fp32 couplings, same 1024 replicas, heuristic geometry) is the
    # genuinely HBM-bound figure: reported beside the headline, PMC passes under profiles/.
    if a.workload == "c2a" and a.storage == "f32" and world == 1 and not a.no_variants and not a.no_beyond_cache:
        nb = a.beyond_cache_spins
        Jb = make_sk_instance(nb, 7, dev)
        eng.set_tuning(waves_per_replica=0, sweeps_per_launch=1)
        eng.set_dense(Jb, torch.zeros(nb, device=dev), storage="f32")
        del Jb
        ptb = ShardedTempering(eng, R_local=R, rank=0, world=1, seed=42, slot_temps=ladder,
                               n_ladders=1, dist=None, device=comm_dev)
        ptb.sweep(1)
        torch.cuda.synchronize()
        eng.enable_timing(True)
        eng.kernel_time(reset=True)
        t1 = time.perf_counter()
        ptb.sweep(2)
        torch.cuda.synchronize()
        dtb = time.perf_counter() - t1
        lb, msb = eng.kernel_time(reset=True)
        eng.enable_timing(False)
        algo = float(R) * nb * nb * 4.0                       # one fp32 row per attempt
        achb = algo / ((msb / max(lb, 1)) * 1e-3) / 1e9
        trb, trb_src = pmc_traffic(f"dense_f32_n{nb}", "sweep_dense_kernel")
        out["roofline_beyond_cache"] = {
            "bound": "hbm", "achieved": achb, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achb / HBM_PEAK_GBS, "traffic": trb, "traffic_source": trb_src,
            "frac_of_measured_stream_read": (achb / read_gbs) if read_gbs else None,
            "spins": nb, "coupling_bytes": float(nb) * nb * 4.0, "replicas": R,
            "algorithmic_bytes_per_launch": algo, "launches": lb, "avg_launch_ms": msb / max(lb, 1),
            "value": float(R) * nb * 2 / dtb, "unit_value": "attempts/s", "geometry": eng.describe(),
            "kernel": "sweep_dense_kernel", "kernel_instantiation": eng.last_kernel(), "cache_served": False}
        eng.set_dense(J, h, storage=a.storage)  # back to the headline instance (cpu_baseline replays on it)
    substitute = None
    if csr is not None and csr[0] is None and not implicit and rank == 0 and world == 1 and not a.no_cpu_baseline:
        # An instance too large for a host copy (1000 cities: 32 GB of CSR; the CPU port indexes entries
        # with 32 bits).  The CPU baseline runs on the largest sub-instance it can hold -- the first 500
        # cities of the same point set (250 000 spins, 5e8 entries), same penalties and ladder range --
        # replayed on a second engine; the line says so.
        sub = min(a.cities, 500)
        tsp2 = enc.tsp_csr(dmat[:sub, :sub], city_visit=200.0, position_fill=200.0, device=dev)
        eng2 = sg.AnnealEngine(local_rank)
        eng2.set_csr(tsp2[0], tsp2[1], tsp2[2], tsp2[3])
        csr2 = (tsp2[0].cpu().numpy().astype(np.int32), tsp2[1].cpu().numpy(), tsp2[2].cpu().numpy())
        h2 = tsp2[3].cpu().numpy()
        del tsp2
        torch.cuda.empty_cache()
        out["cpu_baseline"] = cpu_baseline(None, sub * sub, 42, csr=csr2, h=h2, eng=eng2, t_range=(t_hot, t_cold))
        out["cpu_baseline"]["substitute_instance"] = (
            f"the first {sub} of the {a.cities} cities ({sub * sub} spins, {len(csr2[1])} CSR entries): the "
            f"{a.cities}-city instance ({len(csr[1])} entries) does not fit the CPU port's host copy")
        eng2.close()
        substitute = True
    if (csr is not None and csr[0] is None) or implicit:
        a.no_cpu_baseline = True  # no host copy of an instance this large / the CPU port runs on CSR
    if substitute:
        pass
    elif rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(None if J is None else J.cpu().numpy(), n, 42, csr=csr,
                                           h=None if csr is None else h.cpu().numpy(), eng=eng,
                                           t_range=(t_hot, t_cold))
    else:
        out["cpu_baseline"] = None
    # BASELINE configs[2], [3], [4] in the line the driver runs: short runs of the same engine calls as
    # `--workload c3 | c4 | c5` (5 warm-up + 10 timed sweeps each), each with its own roofline and, at N = 1, a small
    # CPU sample.  With N > 1 ranks: configs[3] and [4] sharded over the ranks as BASELINE states them (every rank
    # takes part; rank 0 prints).
    wanted = None if a.configs is None else [c for c in a.configs.split(",") if c]
    if a.workload == "c2a" and (wanted is not None or (not a.no_configs and not a.no_variants)):
        eng.close()
        J = None
        wl.clear()
        gc.collect()
        torch.cuda.empty_cache()
        if wanted is None:
            wanted = (["c3"] if world == 1 else []) + ["c4", "c5"]
            if not a.no_c5_1000:
                wanted += ["c5_1000"] + (["c5_1000_csr"] if world == 1 and not a.no_c5_1000_csr else [])
        kw = dict(rank=rank, world=world, dist=dist)
        cr = a.config_replicas
        cfgs = {}
        for c in wanted:
            if c == "c3":  # (1.4 ms per sweep: sweeps 5 .. 25, as the cached-field variant's first window)
                cfgs[c] = config_line("c3", a, dev, local_rank, comm_dev, steps=20, **kw)
            elif c == "c4":  # configs[3]: 1024 replicas per GPU of ONE ladder spanning the GPUs
                cfgs[c] = config_line("c4", a, dev, local_rank, comm_dev, R=cr or None, **kw)
            elif c == "c5":
                # configs[4]: 2048 replicas in 32 ladders of 64 over 8 GPUs = 256 replicas (4 whole ladders) per GPU; one
                # GPU alone holds all 32 ladders of the 100-city instance
                cfgs[c] = config_line("c5", a, dev, local_rank, comm_dev, R=cr or (2048 if world == 1 else 256), **kw)
            elif c == "c5_1000":
                # ... at its stated size (1000 cities, 10^6 spins), one rank's share, couplings never stored (sga_set_tsp)
                cfgs[c] = config_line("c5", a, dev, local_rank, comm_dev, R=cr or 256, cities=1000, implicit=True,
                                      warmup=2, steps=3, **kw)
            elif c == "c5_1000_csr":
                # ... and with the 32 GB of CSR written out (the graded byte model; set-up reported as setup_ms)
                cfgs[c] = config_line("c5", a, dev, local_rank, comm_dev, R=cr or 256, cities=1000, implicit=False,
                                      warmup=1, steps=2, **kw)
            else:
                raise SystemExit(f"bench.py: unknown config line {c!r}")
        out["configs"] = cfgs
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
