"""Replica sharding across the GPUs of one node (one process per GPU, torch.distributed).

The reference has no communication backend at all (`communication_backend`
is a validated string, annealing/multi_gpu.py:26,41-43; "multi-GPU" is a thread pool,
:148,201,264).  This is the MI355X design for its `replica_exchange` strategy
(multi_gpu.py:110-132, 233-308):

* replicas are independent between exchange steps, so rank k owns global replicas
  [k*R_local, (k+1)*R_local) and a full copy of the couplings -- no data-path collective
  inside a sweep;
* an exchange round needs only the R_global energies: one all-gather of R_local doubles per
  rank (RCCL over xGMI; 8 KiB per rank at 1024 replicas -- latency bound, link bandwidth is
  irrelevant).  Every rank then evaluates the same Philox-seeded decisions on the gathered
  vector and swaps temperature *labels*; spins never cross a link;
* several ladders (BASELINE configs[4]: 32 ladders of 64 temperatures): when the ladder count is a
  multiple of the rank count every ladder lies whole on one rank and an exchange round is purely
  local -- no collective at all ("zero exchange traffic"); the decisions are keyed by
  the GLOBAL ladder index, so the run equals the one-rank run bit for bit;
* at the end the global best is found with one all-gather of the per-rank best energies and
  one broadcast of the winner's configuration.

The engine argument only needs the small surface used below, which lets the coordination
logic be exercised on CPU with a stand-in engine (tests/test_sharded_gloo.py).
"""
import time

import numpy as np
import torch


class ShardedTempering:
    def __init__(self, engine, R_local: int, rank: int = 0, world: int = 1, seed: int = 0,
                 slot_temps=None, n_ladders: int = 1, dist=None, device=None, s0=None,
                 force_dist: bool = False):
        """force_dist keeps the collectives in the path at world == 1 (a one-rank process group): the
        RCCL branch can then be exercised on a single GPU; results equal the dist-free run."""
        self.engine = engine
        self.R_local, self.rank, self.world = int(R_local), int(rank), int(world)
        self.R_global = self.R_local * self.world
        self.replica0 = self.rank * self.R_local
        self.dist = dist if (world > 1 or force_dist) else None
        # every ladder whole on one rank: exchange rounds need no energies from anybody else
        self.ladders_local = self.dist is not None and self.world > 1 and int(n_ladders) % self.world == 0
        self.device = device if device is not None else torch.device("cpu")
        engine.init_replicas(self.R_local, seed=seed, s0=s0, R_global=self.R_global,
                             replica0=self.replica0)
        if slot_temps is not None:
            t = np.asarray(slot_temps, np.float64)
            if t.size != self.R_global:
                raise ValueError("slot_temps must cover the global replica set")
            engine.set_ladder(t, n_ladders)
        self.gather_calls, self.gather_ms = 0, 0.0  # all-gather rounds and the host time of their (asynchronous) enqueue
        self.time_collectives = False               # bracket every stream-ordered all-gather by two events
        self._gather_events = []
        self._local_E = torch.zeros(self.R_local, dtype=torch.float64, device=self.device)
        self._all_E = torch.zeros(self.R_global, dtype=torch.float64, device=self.device)

    # ------------------------------------------------------------------ hot path
    def sweep(self, n_sweeps: int = 1, **kw):
        return self.engine.sweep(n_sweeps, **kw)

    def _stream_ordered(self) -> bool:
        """RCCL on device tensors with the engine on torch's current stream: energies copy, all-gather
        and exchange kernel are ordered by that one stream -- no host synchronisation in a round."""
        return (self.dist is not None and self.dist.get_backend() == "nccl" and self._all_E.is_cuda and
                getattr(self.engine, "shares_torch_stream", lambda: False)())

    def gather_energies(self) -> torch.Tensor:
        """All ranks' energies, indexed by global replica id."""
        ordered = self._stream_ordered()
        if ordered:
            self.engine.energies_into(self._local_E, stream_ordered=True)
        else:
            self.engine.energies_into(self._local_E)
        if self.dist is None:
            self._all_E.copy_(self._local_E)
            return self._all_E
        timed = self.time_collectives and self._all_E.is_cuda
        if timed:  # device time of the collective: two events on the stream that carries it
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(torch.cuda.current_stream(self._all_E.device))
        t0 = time.perf_counter()
        self._all_gather(ordered)
        self.gather_calls += 1
        self.gather_ms += (time.perf_counter() - t0) * 1e3
        if timed:
            ev[1].record(torch.cuda.current_stream(self._all_E.device))
            self._gather_events.append(ev)
        return self._all_E

    def gather_device_ms_per_round(self):
        """Mean device time of the timed all-gathers (events on the collective's stream), None without any."""
        if not self._gather_events:  # host-side collectives (gloo on CPU tensors) are synchronous: their wall time
            return (self.gather_ms / self.gather_calls) if self.gather_calls else None
        self._gather_events[-1][1].synchronize()
        return sum(a.elapsed_time(b) for a, b in self._gather_events) / len(self._gather_events)

    def _all_gather(self, ordered: bool = False):
        if self.dist.get_backend() == "nccl":
            self.dist.all_gather_into_tensor(self._all_E, self._local_E)
            if not ordered:
                # the engine launches on its own HIP stream: the gathered vector must be complete
                # before sga_exchange reads it (8 KiB collective -- the wait is microseconds)
                torch.cuda.current_stream(self._all_E.device).synchronize()
        else:
            parts = list(self._all_E.chunk(self.world))
            self.dist.all_gather(parts, self._local_E)

    def exchange(self, count: bool = True):
        """One replica-exchange round over the global ladder(s); identical on every rank.  Returns the
        number of accepted swaps, or None with count=False (no read-back: the round stays asynchronous)."""
        if self.dist is None:
            return self.engine.exchange(count=count)
        if self.ladders_local:
            mine = self.engine.exchange(count=count)  # the local ladders only; nothing crosses a link
            if not count:
                return None
            total = torch.tensor([mine], dtype=torch.int64, device=self.device)
            self.dist.all_reduce(total)  # (only the caller's count: the round itself needed no collective)
            return int(total.item())
        return self.engine.exchange(energies_global=self.gather_energies(), count=count)

    # ------------------------------------------------------------------ results
    def exchange_totals(self):
        """(attempts, accepted swaps) of all rounds so far over ALL ladders.  A rank that decides only its own
        ladders (ladders_local) holds only their counters: the totals are summed over the ranks."""
        att, acc = self.engine.exchange_stats()
        if not self.ladders_local:
            return int(np.sum(att)), int(np.sum(acc))
        sl = slice(self.replica0, self.replica0 + self.R_local)
        t = torch.tensor([int(np.sum(att[sl])), int(np.sum(acc[sl]))], dtype=torch.int64, device=self.device)
        self.dist.all_reduce(t)
        return int(t[0].item()), int(t[1].item())

    def global_best(self):
        """(energy, spins int8 [n], global replica id) of the best configuration seen."""
        e, s, idx = self.engine.best()
        if self.dist is None:
            return e, s, self.replica0 + idx
        mine = torch.tensor([e], dtype=torch.float64, device=self.device)
        allb = torch.zeros(self.world, dtype=torch.float64, device=self.device)
        if self.dist.get_backend() == "nccl":
            self.dist.all_gather_into_tensor(allb, mine)
        else:
            self.dist.all_gather(list(allb.chunk(self.world)), mine)
        winner = int(torch.argmin(allb).item())  # first minimum: lowest rank wins ties
        payload = torch.zeros(s.size + 1, dtype=torch.int32, device=self.device)
        if self.rank == winner:
            payload[:-1] = torch.from_numpy(s.astype(np.int32)).to(self.device)
            payload[-1] = self.replica0 + idx
        self.dist.broadcast(payload, src=winner)
        out = payload.cpu().numpy()
        return float(allb[winner].item()), out[:-1].astype(np.int8), int(out[-1])


class LocalShardedTempering:
    """The same replica sharding driven from ONE process that owns several GPUs.

    Used when the caller is not running under torchrun (e.g. MultiGPUAnnealer constructed in
    a plain script): one engine per GPU, launches are asynchronous so all GPUs sweep
    concurrently, and the exchange step gathers the R_global energies through the host
    (a few KiB) instead of RCCL.  Decisions and results are identical to ShardedTempering.
    """

    def __init__(self, engines, R_local: int, seed: int = 0, slot_temps=None, n_ladders: int = 1,
                 s0=None):
        self.engines = list(engines)
        self.world = len(self.engines)
        self.R_local = int(R_local)
        self.R_global = self.R_local * self.world
        self._pool = None
        for k, e in enumerate(self.engines):
            part = None if s0 is None else s0[k * R_local:(k + 1) * R_local]
            e.init_replicas(self.R_local, seed=seed, s0=part, R_global=self.R_global,
                            replica0=k * self.R_local)
            if slot_temps is not None:
                e.set_ladder(np.asarray(slot_temps, np.float64), n_ladders)

    def sweep(self, n_sweeps: int = 1):
        """All engines sweep concurrently.  A plain sga_sweep only enqueues its launches, but under the field-cache
        modes a call may read the acceptance counters back (a host synchronisation every 4 ... 16 sweeps), so one
        thread per engine drives them: ctypes drops the GIL during the call, every call sets its own device, and each
        engine serialises its own calls (engine.py)."""
        if self.world == 1:
            self.engines[0].sweep(n_sweeps)
            return
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=self.world, thread_name_prefix="sga-shard")
        for f in [self._pool.submit(e.sweep, n_sweeps) for e in self.engines]:
            f.result()  # (an engine's error surfaces here)

    def gather_energies(self) -> np.ndarray:
        return np.concatenate([e.energies() for e in self.engines])

    def exchange(self) -> int:
        if self.world == 1:
            return self.engines[0].exchange()
        all_e = self.gather_energies()
        counts = [e.exchange(energies_global=all_e) for e in self.engines]
        assert len(set(counts)) == 1, "ranks disagree on the exchange decisions"
        return counts[0]

    def exchange_totals(self):
        att, acc = self.engines[0].exchange_stats()  # every engine decides every ladder: one holds the totals
        return int(np.sum(att)), int(np.sum(acc))

    def global_best(self):
        bests = [e.best() for e in self.engines]
        k = int(np.argmin([b[0] for b in bests]))
        return bests[k][0], bests[k][1], k * self.R_local + bests[k][2]
