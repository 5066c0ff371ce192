"""BatchProcessor: many independent Ising models through one engine.

Interface of the reference's spin_glass_rl/annealing/batch_processor.py:23-555
(`BatchConfig`, `BatchProcessor.process_models_batch / process_models_stream`): a list of
models goes in, one AnnealingResult per model comes out.  The reference loops a GPUAnnealer
over the models in a thread pool (:423-454).  Here models of equal size are stacked into ONE
engine (`sga_set_dense_batch`): every model gets `replicas_per_model` replicas, a single
kernel launch sweeps all of them (each replica reads its own model's coupling rows), and with
more than one replica per model each model is its own temperature ladder.  Simulated
annealing semantics per replica are those of GPUAnnealer (schedule, best at sweep ends).
"""
import time
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional

import numpy as np
import torch

from .engine import AnnealEngine
from .exceptions import AnnealingError
from .gpu_annealer import GPUAnnealerConfig, fresh_seed
from .ising_model import IsingModel
from .result import AnnealingResult
from .temperature_scheduler import TemperatureScheduler


@dataclass
class BatchConfig:
    batch_size: int = 32
    max_memory_usage: float = 0.8
    prefetch_batches: int = 2
    use_mixed_precision: bool = False
    enable_gradient_checkpointing: bool = True
    memory_optimization_level: int = 1
    streaming_mode: bool = False
    checkpoint_interval: int = 100
    replicas_per_model: int = 1  # build-specific: independent restarts per model, best is kept

    def __post_init__(self):
        if self.batch_size <= 0:
            raise ValueError("Batch size must be positive")
        if not 0 < self.max_memory_usage <= 1:
            raise ValueError("Max memory usage must be between 0 and 1")
        if self.memory_optimization_level not in (0, 1, 2):
            raise ValueError("Memory optimization level must be 0, 1, or 2")
        if self.replicas_per_model <= 0:
            raise ValueError("replicas_per_model must be positive")


class BatchProcessor:
    def __init__(self, annealer_config: GPUAnnealerConfig, batch_config: Optional[BatchConfig] = None,
                 device_index: int = 0):
        self.annealer_config = annealer_config
        self.batch_config = batch_config or BatchConfig()
        self.device_index = device_index
        self.processed_models = 0
        self.total_processing_time = 0.0
        self.batch_times: List[float] = []

    # ------------------------------------------------------------------ public API
    def process_models_batch(self, models: List[IsingModel]) -> List[AnnealingResult]:
        """Anneal every model; results come back in input order."""
        results: List[Optional[AnnealingResult]] = [None] * len(models)
        by_size: Dict[int, List[int]] = {}
        for i, m in enumerate(models):
            by_size.setdefault(m.n_spins, []).append(i)
        for _, idxs in sorted(by_size.items()):
            for lo in range(0, len(idxs), self.batch_config.batch_size):
                part = idxs[lo:lo + self.batch_config.batch_size]
                t0 = time.time()
                for i, r in zip(part, self._anneal_stack([models[i] for i in part])):
                    results[i] = r
                self.batch_times.append(time.time() - t0)
        self.processed_models += len(models)
        self.total_processing_time += sum(self.batch_times[-len(by_size):])
        return results  # type: ignore[return-value]

    def process_models_stream(self, models: Iterable[IsingModel]):
        """Generator form (reference :290-345): yields lists of results batch by batch."""
        chunk: List[IsingModel] = []
        for m in models:
            chunk.append(m)
            if len(chunk) == self.batch_config.batch_size:
                yield self.process_models_batch(chunk)
                chunk = []
        if chunk:
            yield self.process_models_batch(chunk)

    def get_processing_stats(self) -> Dict:
        n = max(len(self.batch_times), 1)
        return {"processed_models": self.processed_models,
                "total_processing_time": self.total_processing_time,
                "average_batch_time": float(np.mean(self.batch_times)) if self.batch_times else 0.0,
                "batches": len(self.batch_times),
                "models_per_second": self.processed_models / self.total_processing_time
                if self.total_processing_time > 0 else 0.0, "n": n}

    def reset(self) -> None:
        self.processed_models, self.total_processing_time, self.batch_times = 0, 0.0, []

    # ------------------------------------------------------------------ one stacked run
    def _anneal_stack(self, models: List[IsingModel]) -> List[AnnealingResult]:
        cfg, k = self.annealer_config, self.batch_config.replicas_per_model
        M, n = len(models), models[0].n_spins
        t0 = time.time()
        J = np.stack([m.dense_couplings().detach().cpu().numpy().astype(np.float32) for m in models])
        h = np.stack([m.external_fields.detach().cpu().numpy().astype(np.float32) for m in models])
        s0 = np.repeat(np.stack([m.spins_int8() for m in models]), k, axis=0)  # [M*k, n]
        schedule = TemperatureScheduler.create_schedule(
            cfg.schedule_type, cfg.initial_temp, cfg.final_temp, cfg.n_sweeps, **cfg.schedule_params)
        if cfg.schedule_type.value == "adaptive":
            raise AnnealingError("the adaptive schedule needs per-model feedback; use GPUAnnealer")
        temps = np.maximum(np.asarray([schedule.update(s) for s in range(cfg.n_sweeps)]), 1e-10)
        hist_e = [[] for _ in range(M)]
        with AnnealEngine(self.device_index) as eng:
            eng.set_dense_batch(J, h, storage=cfg.coupling_storage)
            eng.init_replicas(M * k, seed=fresh_seed(cfg.random_seed), s0=s0)
            e0 = eng.energies().reshape(M, k).min(1)
            for m in range(M):
                hist_e[m].append(float(e0[m]))
            ri = cfg.record_interval
            for lo in range(0, cfg.n_sweeps, ri):
                hi = min(lo + ri, cfg.n_sweeps)
                eng.sweep(hi - lo, sched=temps[lo:hi])
                en = eng.energies().reshape(M, k).min(1)
                for m in range(M):
                    hist_e[m].append(float(en[m]))
            acc, att = eng.stats()
            out = []
            for m in range(M):
                cand = [eng.best(m * k + j) for j in range(k)]
                j = int(np.argmin([c[0] for c in cand]))
                rate = float(acc[m * k:(m + 1) * k].sum()) / float(max(att[m * k:(m + 1) * k].sum(), 1))
                out.append(AnnealingResult(
                    best_configuration=torch.from_numpy(cand[j][1].astype(np.float32)),
                    best_energy=cand[j][0], energy_history=hist_e[m],
                    temperature_history=[cfg.initial_temp] + [float(t) for t in temps[ri - 1::ri]],
                    acceptance_rate_history=[rate], total_time=(time.time() - t0) / M,
                    n_sweeps=cfg.n_sweeps, algorithm="simulated_annealing",
                    device=f"cuda:{self.device_index}", random_seed=cfg.random_seed))
        return out
