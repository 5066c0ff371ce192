"""N > 1 path on CPU: two (and four) ranks over gloo, each holding its share of the replicas,
must reproduce the single-rank run bit for bit (decisions are a pure function of the gathered
energies and the shared Philox key; spins never move between ranks)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle_engine import OracleEngine

N_SPINS, R_GLOBAL, ROUNDS, SEED = 48, 12, 9, 777


def _instance():
    rng = np.random.RandomState(5)
    J = np.triu(rng.randint(0, 2, (N_SPINS, N_SPINS)) * 2 - 1, 1).astype(np.float32)
    return J + J.T, rng.randint(-1, 2, N_SPINS).astype(np.float32)


def _ladder(R, n_ladders):
    L = R // n_ladders
    one = [6.0 * (0.3 / 6.0) ** (i / (L - 1)) for i in range(L)]
    return np.asarray(one * n_ladders)


def _run(rank, world, n_ladders, sink, dist_mod, force=False):
    import spin_glass_anneal_rl_amd as sg
    J, h = _instance()
    pt = sg.ShardedTempering(OracleEngine(J=J, h=h), R_GLOBAL // world, rank, world, SEED,
                             _ladder(R_GLOBAL, n_ladders), n_ladders, dist_mod,
                             torch.device("cpu"), force_dist=force)
    assert (pt.dist is not None) == (world > 1 or force)
    swaps = []
    for k in range(ROUNDS):
        pt.sweep(2)
        if force and k % 2:  # the form without a count read-back
            before = pt.engine.rounds
            assert pt.exchange(count=False) in (None, pt.engine.last_accepted)
            swaps.append(pt.engine.last_accepted)
            assert pt.engine.rounds == before + 1
        else:
            swaps.append(pt.exchange())
    e, s, idx = pt.global_best()
    sink.update(swaps=swaps, energies=pt.gather_energies().numpy().copy(), best=(e, s, idx),
                spins=pt.engine.spins(), temps=pt.engine.temperatures(),
                slot_map=pt.engine.slot_map())


def _worker(rank, world, port, n_ladders, q, force=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    _run(rank, world, n_ladders, out, dist, force)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ranks(world, n_ladders, force=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_ladders, q, force)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def _check(n_ladders, world=2):
    single = {}
    _run(0, 1, n_ladders, single, None)
    assert sum(single["swaps"]) > 0
    got = _ranks(world, n_ladders)
    half = R_GLOBAL // world
    for rank in range(world):
        o = got[rank]
        assert o["swaps"] == single["swaps"]
        assert np.array_equal(o["energies"], single["energies"])
        assert np.array_equal(o["slot_map"], single["slot_map"])
        sl = slice(rank * half, (rank + 1) * half)
        assert np.array_equal(o["spins"], single["spins"][sl])
        assert np.array_equal(o["temps"], single["temps"][sl])
        assert o["best"][0] == single["best"][0] and o["best"][2] == single["best"][2]
        assert np.array_equal(o["best"][1], single["best"][1])


def test_two_rank_run_equals_single_rank_one_ladder():
    _check(1)


def test_two_rank_run_equals_single_rank_three_ladders():
    _check(3)


def test_four_rank_run_equals_single_rank_ladders_straddling_ranks():
    # 3 ladders of 4 replicas over 4 ranks of 3 replicas: every ladder spans two ranks
    _check(3, world=4)


def test_one_rank_group_with_the_collectives_forced_equals_the_plain_run():
    """force_dist keeps all_gather / broadcast in the path at world size 1 (what `bench.py --gpus 1
    --force-dist` and the one-GPU RCCL tests rely on): same swaps, energies, ladder and best."""
    for n_ladders in (1, 3):
        single = {}
        _run(0, 1, n_ladders, single, None)
        o = _ranks(1, n_ladders, force=True)[0]
        assert o["swaps"] == single["swaps"] and np.array_equal(o["energies"], single["energies"])
        assert np.array_equal(o["slot_map"], single["slot_map"]) and np.array_equal(o["spins"], single["spins"])
        assert o["best"][0] == single["best"][0] and np.array_equal(o["best"][1], single["best"][1])
