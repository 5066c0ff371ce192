"""MultiGPUAnnealer: the reference's multi-GPU interface on real devices.

Config and entry points follow spin_glass_rl/annealing/multi_gpu.py:20-351 (gpu_ids, strategy,
communication_backend, synchronization_interval; anneal() dispatch; errors wrapped in
AnnealingError).  The reference's implementation is a thread pool with no communication
(SURVEY.md 0.2, 2.1); here:

* data_parallel    -- independent models, one per GPU at a time (no collective);
* replica_exchange -- one model, replicas sharded over the GPUs, temperatures exchanged every
                      `synchronization_interval` sweeps.  Under torchrun (one process per GPU)
                      the energies travel by RCCL all-gather (sharded.ShardedTempering); in a
                      single process that owns all `gpu_ids` they are gathered through the host
                      (sharded.LocalShardedTempering).  The criterion is the reference's
                      parallel-tempering one (parallel_tempering.py:234-258), not the
                      sign-inverted variant of multi_gpu.py:441-443;
* model_parallel   -- rejected: the reference's version drops all inter-block couplings
                      (multi_gpu.py:366-395); J for the supported sizes fits one GPU.
"""
import os
import threading
import time
from dataclasses import dataclass
from typing import Any, List, Optional

import numpy as np
import torch

from .engine import AnnealEngine
from .exceptions import AnnealingError, DeviceError
from .gpu_annealer import GPUAnnealer, GPUAnnealerConfig, fresh_seed
from .ising_model import IsingModel
from .result import AnnealingResult
from .sharded import LocalShardedTempering, ShardedTempering
from .temperature_scheduler import temperature_ladder


@dataclass
class MultiGPUConfig:
    gpu_ids: List[int]
    strategy: str = "data_parallel"
    communication_backend: str = "nccl"
    synchronization_interval: int = 10
    load_balancing: bool = True
    fault_tolerance: bool = True
    max_retries: int = 3
    replicas_per_gpu: int = 64  # build-specific: replica_exchange shard size

    def __post_init__(self):
        if not self.gpu_ids:
            raise ValueError("At least one GPU ID must be specified")
        if self.strategy not in ("data_parallel", "model_parallel", "replica_exchange"):
            raise ValueError("Strategy must be one of ['data_parallel', 'model_parallel', "
                             "'replica_exchange']")
        if self.communication_backend not in ("nccl", "gloo", "mpi"):
            raise ValueError("Backend must be one of ['nccl', 'gloo', 'mpi']")


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


class MultiGPUAnnealer:
    def __init__(self, config: MultiGPUConfig, annealer_config: GPUAnnealerConfig):
        self.config = config
        self.annealer_config = annealer_config
        if not torch.cuda.is_available():
            raise DeviceError("no MI355X visible for multi-GPU annealing")
        count = torch.cuda.device_count()
        for g in config.gpu_ids:
            if g >= count or g < 0:
                raise DeviceError(f"GPU {g} not available (only {count} GPUs found)")
        self.devices = [torch.device(f"cuda:{g}") for g in config.gpu_ids]
        self.master_device = self.devices[0]

    # ------------------------------------------------------------------ strategies
    def anneal_data_parallel(self, models: List[IsingModel]) -> List[AnnealingResult]:
        dist = _dist()
        if dist is not None:  # one process per GPU: rank r takes models r, r+world, ...
            rank, world = dist.get_rank(), dist.get_world_size()
            mine = {i: self._anneal_on(models[i], int(os.environ.get("LOCAL_RANK", rank)))
                    for i in range(rank, len(models), world)}
            gathered: List[Optional[dict]] = [None] * world
            dist.all_gather_object(gathered, mine)
            merged = {}
            for part in gathered:
                merged.update(part)
            return [merged[i] for i in range(len(models))]
        results: List[Optional[AnnealingResult]] = [None] * len(models)
        errors: List[BaseException] = []

        def worker(slot: int, gpu: int):
            try:
                for i in range(slot, len(models), len(self.config.gpu_ids)):
                    results[i] = self._anneal_on(models[i], gpu)
            except BaseException as exc:  # noqa: BLE001 - re-raised on the caller's thread
                errors.append(exc)

        threads = [threading.Thread(target=worker, args=(k, g))
                   for k, g in enumerate(self.config.gpu_ids)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return results  # type: ignore[return-value]

    def _anneal_on(self, model: IsingModel, gpu: int) -> AnnealingResult:
        cfg = GPUAnnealerConfig(**{**self.annealer_config.__dict__, "device_index": gpu})
        return GPUAnnealer(cfg).anneal(model)

    def anneal_replica_exchange(self, model: IsingModel) -> AnnealingResult:
        cfg, acfg = self.config, self.annealer_config
        dist = _dist()
        Rl = cfg.replicas_per_gpu
        world = dist.get_world_size() if dist is not None else len(cfg.gpu_ids)
        Rg = Rl * world
        temps = np.asarray(temperature_ladder(Rg, acfg.final_temp, acfg.initial_temp, "geometric")
                           if Rg > 1 else [acfg.final_temp])
        seed = fresh_seed(acfg.random_seed)
        if dist is not None:
            rank = dist.get_rank()
            gpu = int(os.environ.get("LOCAL_RANK", rank))
            seed_t = torch.tensor([seed], dtype=torch.int64,
                                  device=torch.device("cuda", gpu) if dist.get_backend() == "nccl"
                                  else torch.device("cpu"))
            dist.broadcast(seed_t, src=0)  # all ranks must share the Philox key
            eng = AnnealEngine(gpu)
            eng.set_field_cache(acfg.field_cache)
            model.load_into(eng, storage=acfg.coupling_storage)
            pt = ShardedTempering(eng, Rl, rank, world, int(seed_t.item()), temps, 1, dist,
                                  torch.device("cuda", gpu))
            engines = [eng]
        else:
            engines = [AnnealEngine(g) for g in cfg.gpu_ids]
            for e in engines:
                e.set_field_cache(acfg.field_cache)
                model.load_into(e, storage=acfg.coupling_storage)
            pt = LocalShardedTempering(engines, Rl, seed, temps, 1)
        done, history = 0, []
        while done < acfg.n_sweeps:
            step = min(cfg.synchronization_interval, acfg.n_sweeps - done)
            pt.sweep(step)
            done += step
            if Rg > 1 and done < acfg.n_sweeps:
                pt.exchange()
            history.append(float(min(e.best(with_spins=False)[0] for e in engines)))
        best_e, best_s, _ = pt.global_best()
        for e in engines:
            e.close()
        return AnnealingResult(
            best_configuration=torch.from_numpy(best_s.astype(np.float32)), best_energy=best_e,
            energy_history=history, temperature_history=[float(temps.min())] * len(history),
            acceptance_rate_history=[], total_time=0.0, n_sweeps=acfg.n_sweeps,
            algorithm="replica_exchange", device=str(self.master_device),
            random_seed=acfg.random_seed)

    # ------------------------------------------------------------------ dispatch (reference :309-351)
    def anneal(self, models: Any) -> Any:
        t0 = time.time()
        try:
            if self.config.strategy == "data_parallel":
                if not isinstance(models, list):
                    raise ValueError("Data parallel strategy requires list of models")
                results = self.anneal_data_parallel(models)
            elif self.config.strategy == "model_parallel":
                raise ValueError("model_parallel is not supported: the reference's version drops "
                                 "the couplings between device blocks; use replica_exchange")
            else:
                if isinstance(models, list):
                    raise ValueError("Replica exchange strategy requires single model")
                results = self.anneal_replica_exchange(models)
            total = time.time() - t0
            if isinstance(results, list):
                for r in results:
                    r.total_time = total / len(results)
            else:
                results.total_time = total
            return results
        except Exception as exc:  # noqa: BLE001 - mirrors the reference's wrapping
            raise AnnealingError(f"Multi-GPU annealing failed: {exc}") from exc
