"""The N > 1 path on real engines: `python bench.py --gpus N` starts its own ranks (no torchrun),
and a 2-rank run over torch.distributed reproduces the 1-rank run bit for bit.  On a 1-GPU box
the two ranks share cuda:0 and talk over gloo (the rehearsal mode of bench.py); with >= 2 GPUs
the same check runs one rank per GPU over RCCL (backend "nccl")."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SPINS, R_GLOBAL, ROUNDS, SEED = 600, 12, 6, 4242


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_without_gpu_fails_loudly_in_every_rank():
    """CPU-side check of the launcher: without a GPU every rank refuses (no CPU fallback) and the
    parent reports the failed ranks with a non-zero exit code."""
    if torch.cuda.device_count() > 0:
        pytest.skip("checks the no-GPU failure mode")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0
    assert "ranks failed" in p.stderr and "no HIP device visible" in p.stderr


@pytest.mark.gpu
def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun around it: two ranks, one JSON line."""
    two_gpus = torch.cuda.device_count() >= 2
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "2",
           "--spins", "2000", "--replicas", "64", "--no-autotune", "--no-variants"]
    if not two_gpus:
        cmd += ["--backend", "gloo", "--share-device"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.splitlines()
    assert len(lines) == 1, p.stdout[:500]      # ONE JSON line and nothing else on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "weak"
    assert d["backend"] == ("nccl" if two_gpus else "gloo")
    assert d["config"]["replicas_total"] == 128 and d["config"]["replicas_per_gpu"] == 64
    assert d["value"] == pytest.approx(128 * 2000 * 12 / (d["ms_per_step"] * 12 * 1e-3))
    assert d["exchange"]["rounds_timed"] >= 1 and d["exchange"]["allgather_ms_per_round"] > 0


@pytest.mark.gpu
def test_bench_multirank_lines_of_configs_3_and_4_equal_the_one_rank_run():
    """BASELINE configs[3] (ONE ladder spanning the ranks, exchange through an all-gather of the energies) and configs[4]
    (whole 64-temperature ladders per rank, no collective) as `bench.py --gpus 2` runs them after the headline: the final
    energies of all replicas, the global best and its replica equal the same global replica set on one rank, bit for bit."""
    two_gpus = torch.cuda.device_count() >= 2
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--spins", "2000",
            "--replicas", "64", "--no-autotune", "--no-variants", "--no-cpu-baseline", "--configs", "c4,c5"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    runs = {}
    for world, per_rank in ((1, 256), (2, 128)):
        cmd = base + ["--gpus", str(world), "--config-replicas", str(per_rank)]
        if world > 1 and not two_gpus:
            cmd += ["--backend", "gloo", "--share-device"]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = p.stdout.splitlines()
        assert len(lines) == 1, p.stdout[:500]
        runs[world] = json.loads(lines[0])["configs"]
    one, two = runs[1], runs[2]
    assert set(two) == {"c4", "c5"}
    for c in ("c4", "c5"):
        assert two[c]["ranks_seen"] == 2 and two[c]["n_gpus"] == 2 and two[c]["couplings_checksum_agree"] is True
        assert two[c]["replicas_total"] == 256 and two[c]["scaling"] == "weak"
        assert two[c]["ms_per_step"] >= two[c]["ms_per_step_this_rank"] * 0.999  # MAX over the ranks
        assert two[c]["energies_sha256"] == one[c]["energies_sha256"], c
        assert two[c]["best_energy_global"] == one[c]["best_energy_global"], c
        assert two[c]["best_replica_global"] == one[c]["best_replica_global"], c
    # configs[3]: the ladder spans the ranks -> every timed round gathers the energies
    assert two["c4"]["exchange"]["rounds_timed"] >= 1
    assert two["c4"]["exchange"]["allgathers_timed"] == two["c4"]["exchange"]["rounds_timed"]
    assert two["c4"]["exchange"]["allgather_ms_per_round"] > 0 and two["c4"]["exchange"]["bytes_per_rank"] == 8 * 128
    assert "may span ranks" in two["c4"]["placement"] and "1 geometric ladder" in two["c4"]["workload"]
    # configs[4]: 2 whole ladders of 64 per rank -> rounds are local, nothing is gathered
    assert two["c5"]["exchange"]["rounds_timed"] >= 1 and two["c5"]["exchange"]["allgathers_timed"] == 0
    assert two["c5"]["exchange"]["bytes_per_rank"] == 0 and "whole ladders per rank" in two["c5"]["placement"]
    assert "4 geometric ladder" in two["c5"]["workload"]


def _rank_main(rank, world, port, backend, share, n_ladders, out_path, force=False, side_stream=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    import spin_glass_anneal_rl_amd as sg
    dev_index = 0 if share else rank
    torch.cuda.set_device(dev_index)
    if side_stream:  # engine and torch on one stream: the exchange round runs without host synchronisation
        torch.cuda.set_stream(torch.cuda.Stream(torch.device("cuda", dev_index)))
    if world > 1 or force:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.RandomState(5)
    J = np.triu(rng.randint(0, 2, (N_SPINS, N_SPINS)) * 2 - 1, 1).astype(np.float32)
    J, h = J + J.T, rng.randint(-1, 2, N_SPINS).astype(np.float32)
    L = R_GLOBAL // n_ladders
    ladder = np.asarray([6.0 * (0.3 / 6.0) ** (i / (L - 1)) for i in range(L)] * n_ladders)
    eng = sg.AnnealEngine(dev_index)
    if side_stream:
        eng.use_stream(torch.cuda.current_stream().cuda_stream)
        assert eng.shares_torch_stream()
    eng.set_dense(J, h)
    comm = torch.device("cuda", dev_index) if backend == "nccl" else torch.device("cpu")
    pt = sg.ShardedTempering(eng, R_GLOBAL // world, rank, world, SEED, ladder, n_ladders,
                             dist if (world > 1 or force) else None, comm, force_dist=force)
    assert (pt.dist is not None) == (world > 1 or force)
    swaps = []
    for k in range(ROUNDS):
        pt.sweep(2)
        if side_stream and k % 2 == 1:  # the asynchronous form: no read-back; the count comes from the statistics
            before = int(eng.exchange_stats()[1].sum())
            assert pt.exchange(count=False) is None
            swaps.append(int(eng.exchange_stats()[1].sum()) - before)
        else:
            swaps.append(pt.exchange())
    e, s, idx = pt.global_best()
    np.savez(out_path, swaps=np.asarray(swaps), energies=pt.gather_energies().cpu().numpy(),
             best_e=e, best_s=s, best_idx=idx, spins=eng.spins(), temps=eng.temperatures(),
             slot_map=eng.slot_map())
    eng.close()
    if world > 1 or force:
        dist.barrier()
        dist.destroy_process_group()


def _run_ranks(world, backend, share, n_ladders, tmp_path, force=False, side_stream=False):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    port = _free_port()
    tag = f"w{world}_{backend}_{int(force)}{int(side_stream)}"
    paths = [str(tmp_path / f"{tag}_r{r}.npz") for r in range(world)]
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, backend, share, n_ladders, paths[r], force,
                                                  side_stream))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    return [dict(np.load(p)) for p in paths]


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["gloo", "nccl"])
@pytest.mark.parametrize("n_ladders", [1, 2, 3])
def test_two_ranks_on_real_engines_equal_one_rank(backend, n_ladders, tmp_path):
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank; this box has one (the 8-GPU driver run covers it)")
    share = backend == "gloo"
    single = _run_ranks(1, backend, True, n_ladders, tmp_path)[0]
    assert single["swaps"].sum() > 0
    got = _run_ranks(2, backend, share, n_ladders, tmp_path)
    half = R_GLOBAL // 2
    for rank, o in enumerate(got):
        sl = slice(rank * half, (rank + 1) * half)
        assert np.array_equal(o["swaps"], single["swaps"])
        assert np.array_equal(o["energies"], single["energies"])
        if n_ladders % 2 == 0:  # whole ladders per rank: rounds decided locally, no gather (sga_exchange, sga.h)
            assert np.array_equal(o["slot_map"][sl], single["slot_map"][sl])
        else:
            assert np.array_equal(o["slot_map"], single["slot_map"])
        assert np.array_equal(o["spins"], single["spins"][sl])
        assert np.array_equal(o["temps"], single["temps"][sl])
        assert o["best_e"] == single["best_e"] and o["best_idx"] == single["best_idx"]
        assert np.array_equal(o["best_s"], single["best_s"])


@pytest.mark.gpu
@pytest.mark.parametrize("side_stream", [False, True])
@pytest.mark.parametrize("n_ladders", [1, 3])
def test_one_rank_process_group_over_rccl_equals_the_plain_run(n_ladders, side_stream, tmp_path):
    """The RCCL branch of ShardedTempering on ONE GPU: a one-rank process group (backend "nccl"), started
    in a fresh child process, with the collectives forced into the path -- sweeps, exchange rounds through
    all_gather_into_tensor on device tensors, global_best through all_gather + broadcast -- equals the
    dist-free run bit for bit; with the engine on torch's stream the round needs no host synchronisation."""
    plain = _run_ranks(1, "gloo", True, n_ladders, tmp_path)[0]
    assert plain["swaps"].sum() > 0
    got = _run_ranks(1, "nccl", True, n_ladders, tmp_path, force=True, side_stream=side_stream)[0]
    for key in ("swaps", "energies", "slot_map", "spins", "temps", "best_s"):
        assert np.array_equal(got[key], plain[key]), key
    assert got["best_e"] == plain["best_e"] and got["best_idx"] == plain["best_idx"]


@pytest.mark.gpu
def test_bench_force_dist_runs_the_rccl_path_on_one_gpu():
    """`python bench.py --gpus 1 --force-dist`: a one-rank RCCL group in a child process; the line says so
    and reports that the ranks' couplings agree."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "12",
           "--warmup", "2", "--spins", "2000", "--replicas", "64", "--no-variants", "--no-cpu-baseline",
           "--configs", "c4,c5", "--config-replicas", "128"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.splitlines()
    assert len(lines) == 1, p.stdout[:500]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["backend"] == "nccl"
    assert d["couplings_checksum_agree"] is True and len(d["couplings_checksum"]) == 16
    assert d["exchange"]["rounds_timed"] >= 1 and d["config"]["geometry_autotuned"] is True
    plain = subprocess.run([c for c in cmd if c != "--force-dist"], capture_output=True, text=True, timeout=900,
                           env=env)
    assert plain.returncode == 0, plain.stderr[-2000:]
    q = json.loads(plain.stdout.splitlines()[0])
    assert q["backend"] is None and q["couplings_checksum"] == d["couplings_checksum"]
    assert q["config"]["best_energy_rank0"] == d["config"]["best_energy_rank0"]
    # the config lines of BASELINE configs[3] / [4] through the same one-rank RCCL group: checksum all-gather, energies
    # all-gather on the shared stream (device tensors), MAX all-reduce of the time, best all-gather + broadcast
    for c in ("c4", "c5"):
        assert d["configs"][c]["backend"] == "nccl" and d["configs"][c]["ranks_seen"] == 1
        assert d["configs"][c]["exchange"]["allgathers_timed"] >= 1 and d["configs"][c]["exchange"]["allgather_ms_per_round"] > 0
        assert "backend" not in q["configs"][c]
        assert d["configs"][c]["energies_sha256"] == q["configs"][c]["energies_sha256"]
        assert d["configs"][c]["best_energy_global"] == q["configs"][c]["best_energy_global"]


def _class_rank_main(with_dist, port, out_path):
    """MultiGPUAnnealer.anneal (strategy replica_exchange) in a fresh process: under a one-rank RCCL group, or
    without torch.distributed (one process owning the GPU)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
    import torch.distributed as dist
    import spin_glass_anneal_rl_amd as sg
    torch.cuda.set_device(0)
    if with_dist:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    rng = np.random.RandomState(5)
    J = np.triu(rng.randint(0, 2, (N_SPINS, N_SPINS)) * 2 - 1, 1).astype(np.float32)
    J, h = J + J.T, rng.randint(-1, 2, N_SPINS).astype(np.float32)
    m = sg.IsingModel(sg.IsingModelConfig(n_spins=N_SPINS, use_sparse=False))
    m.set_couplings_from_matrix(torch.from_numpy(J))
    m.external_fields = torch.from_numpy(h)
    ann = sg.MultiGPUAnnealer(sg.MultiGPUConfig(gpu_ids=[0], strategy="replica_exchange", synchronization_interval=2,
                                                replicas_per_gpu=R_GLOBAL, n_ladders=2),
                              sg.GPUAnnealerConfig(n_sweeps=12, initial_temp=6.0, final_temp=0.3, random_seed=SEED))
    res = ann.anneal(m)
    util = ann.get_device_utilization()
    ann.cleanup()
    np.savez(out_path, best=res.best_energy, cfg=res.best_configuration.numpy(), hist=np.asarray(res.energy_history),
             exchanges=res.metadata["exchanges"], attempts=res.metadata["exchange_attempts"],
             mem=util["gpu_0"]["memory_allocated"], total=util["gpu_0"]["memory_total"])
    if with_dist:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.gpu
def test_multi_gpu_annealer_over_a_one_rank_rccl_group_equals_the_plain_call(tmp_path):
    """The product's class path (MultiGPUAnnealer.anneal -> anneal_replica_exchange) under torch.distributed with
    backend "nccl": couplings checksum all-gathered, engine on torch's side stream (exchange rounds without host
    synchronisation), all-reduced energy history -- the same result as the call without a process group."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = {}
    for with_dist in (False, True):
        path = str(tmp_path / f"class_{int(with_dist)}.npz")
        p = ctx.Process(target=_class_rank_main, args=(with_dist, _free_port(), path))
        p.start()
        p.join(timeout=600)
        assert p.exitcode == 0
        out[with_dist] = dict(np.load(path))
    a, b = out[False], out[True]
    assert a["best"] == b["best"] and np.array_equal(a["cfg"], b["cfg"]) and np.array_equal(a["hist"], b["hist"])
    assert a["exchanges"] == b["exchanges"] > 0 and a["attempts"] == b["attempts"] and len(a["hist"]) == 6
    assert 0 < a["mem"] <= a["total"]


def _class_mismatch_main(port, out_path):
    """A run that is refused after the engine and the side stream are up (the coupling check says no)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
    import torch.distributed as dist
    import spin_glass_anneal_rl_amd as sg
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    rng = np.random.RandomState(5)
    J = np.triu(rng.randint(0, 2, (N_SPINS, N_SPINS)) * 2 - 1, 1).astype(np.float32)
    m = sg.IsingModel(sg.IsingModelConfig(n_spins=N_SPINS, use_sparse=False))
    m.set_couplings_from_matrix(torch.from_numpy(J + J.T))
    made = []

    class Refusing(sg.MultiGPUAnnealer):
        def _make_engine(self, gpu, model):
            made.append(super()._make_engine(gpu, model))
            return made[-1]

        @staticmethod
        def _same_couplings_everywhere(engines, dist_, comm_dev):
            raise sg.AnnealingError("the ranks hold different couplings (injected)")

    ann = Refusing(sg.MultiGPUConfig(gpu_ids=[0], strategy="replica_exchange", synchronization_interval=2,
                                     replicas_per_gpu=R_GLOBAL),
                   sg.GPUAnnealerConfig(n_sweeps=4, initial_temp=6.0, final_temp=0.3, random_seed=SEED))
    err = ""
    try:
        ann.anneal_replica_exchange(m)
    except sg.AnnealingError as exc:
        err = str(exc)
    np.savez(out_path, err=err, made=len(made), closed=all(not e._h.value for e in made), left=len(ann._engines),
             default_stream=bool(torch.cuda.current_stream() == torch.cuda.default_stream()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_a_refused_run_closes_its_engine_and_hands_the_default_stream_back(tmp_path):
    """The coupling-mismatch error leaves after the engine is built and torch runs on the private side stream: the
    finally block closes the engine (its HBM) and restores torch's default stream on that path too."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    path = str(tmp_path / "refused.npz")
    p = ctx.Process(target=_class_mismatch_main, args=(_free_port(), path))
    p.start()
    p.join(timeout=600)
    assert p.exitcode == 0
    o = dict(np.load(path))
    assert "different couplings" in str(o["err"])
    assert int(o["made"]) == 1 and bool(o["closed"]) and int(o["left"]) == 0 and bool(o["default_stream"])


@pytest.mark.gpu
@pytest.mark.parametrize("world,n_ladders,L", [(2, 2, 5), (2, 6, 3), (3, 6, 4), (4, 8, 2), (8, 32, 2)])
def test_ladder_local_exchange_equals_the_unsharded_engine(world, n_ladders, L):
    """sga_exchange with energies_global = NULL on sharded replicas (every ladder whole on one shard): `world` engines
    in one process, each deciding only its own ladders from its own energies, walk the chain of ONE engine holding all
    replicas -- swaps per round, temperatures, spins, the slot map and the exchange statistics of the local ladders."""
    import spin_glass_anneal_rl_amd as sg
    n, seed, rounds = 300, 909, 7
    R = n_ladders * L
    Rl = R // world
    assert Rl % L == 0
    rng = np.random.RandomState(3)
    J = np.triu(rng.randint(0, 2, (n, n)) * 2 - 1, 1).astype(np.float32)
    J, h = J + J.T, rng.randint(-1, 2, n).astype(np.float32)
    ladder = np.tile(np.geomspace(8.0, 0.4, L), n_ladders)

    def engine(R_local, replica0):
        e = sg.AnnealEngine(0)
        e.set_dense(J, h)
        e.init_replicas(R_local, seed=seed, R_global=R, replica0=replica0)
        e.set_ladder(ladder, n_ladders)
        return e

    whole = engine(R, 0)
    shards = [engine(Rl, k * Rl) for k in range(world)]
    try:
        for _ in range(rounds):
            whole.sweep(2)
            want = whole.exchange()
            got = 0
            for e in shards:
                e.sweep(2)
                got += e.exchange()          # no energies from anybody else
            assert got == want
        assert want >= 0 and int(whole.exchange_stats()[1].sum()) > 0
        att, acc = whole.exchange_stats()
        for k, e in enumerate(shards):
            sl = slice(k * Rl, (k + 1) * Rl)
            assert np.array_equal(e.spins(), whole.spins()[sl])
            assert np.array_equal(e.temperatures(), whole.temperatures()[sl])
            assert np.array_equal(e.energies(), whole.energies()[sl])
            assert np.array_equal(e.slot_map()[sl], whole.slot_map()[sl])
            a2, c2 = e.exchange_stats()
            assert np.array_equal(a2[sl], att[sl]) and np.array_equal(c2[sl], acc[sl])
        # a ladder that straddles two shards still needs the gathered energies
        if L > 1 and Rl > 1:
            bad = sg.AnnealEngine(0)
            bad.set_dense(J, h)
            bad.init_replicas(Rl, seed=seed, R_global=R, replica0=1)
            bad.set_ladder(ladder, n_ladders)
            with pytest.raises(sg.AnnealingError, match="all-gathered energies"):
                bad.exchange()
            bad.close()
    finally:
        whole.close()
        for e in shards:
            e.close()
