"""GPUAnnealer: simulated annealing of one IsingModel on the MI355X engine.

Host loop of the reference's spin_glass_rl/annealing/gpu_annealer.py:30-269 (same config
fields, schedule handling, best tracking at sweep ends, histories every `record_interval`
sweeps, early stop on a flat energy history).  The sweeps between two record points are one
kernel call carrying the temperatures of those sweeps; nothing but control flow runs on the
host.  Sites are drawn uniformly with replacement from a per-run Philox stream (the
reference's CPU semantics, core/spin_dynamics.py:69); `site_order="sequential"` gives the
order of its GPU fallback (annealing/cuda_kernels.py:381).
"""
import time
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _native as N
from .engine import AnnealEngine
from .exceptions import ConfigurationError
from .ising_model import IsingModel, _device_index
from .result import AnnealingResult
from .spin_dynamics import UpdateRule, rule_code
from .temperature_scheduler import ScheduleType, TemperatureScheduler


@dataclass
class GPUAnnealerConfig:
    n_sweeps: int = 1000
    initial_temp: float = 10.0
    final_temp: float = 0.01
    schedule_type: ScheduleType = ScheduleType.GEOMETRIC
    schedule_params: Dict = None
    block_size: int = 256                 # accepted for compatibility; geometry is chosen
    shared_memory_size: int = 48 * 1024   # by the engine (sga_describe)
    record_interval: int = 10
    energy_tolerance: float = 1e-8
    random_seed: Optional[int] = None
    enable_adaptive_optimization: bool = True
    enable_caching: bool = True
    enable_performance_profiling: bool = True
    adaptive_config: Optional[object] = None
    compute_config: Optional[object] = None
    # build-specific
    site_order: str = "random"            # "random" | "sequential"
    coupling_storage: str = "auto"        # "auto" | "f32" | "i8"
    field_cache: str = "auto"             # "auto" | "on" | "off": resident local fields, a coupling row read
    #                                       only on accept where the problem allows (sga_set_field_cache) --
    #                                       the same chain bit for bit as "off" (one row read per proposal)
    device_index: Optional[int] = None

    def __post_init__(self):
        if self.schedule_params is None:
            self.schedule_params = {"alpha": 0.95}
        if self.n_sweeps <= 0 or self.record_interval <= 0:
            raise ConfigurationError("n_sweeps and record_interval must be positive")
        if self.site_order not in ("random", "sequential"):
            raise ConfigurationError("site_order must be 'random' or 'sequential'")


def check_convergence(energy_history: List[float], tolerance: float) -> bool:
    """Reference gpu_annealer.py:254-269: >= 50 records and rel. std of the last 20 < tol."""
    if len(energy_history) < 50:
        return False
    recent = np.asarray(energy_history[-20:], dtype=np.float64)
    std, mean = float(np.std(recent)), float(np.mean(recent))
    return (std / abs(mean) < tolerance) if abs(mean) > 0 else (std < tolerance)


def fresh_seed(seed: Optional[int]) -> int:
    if seed is not None:
        return int(seed)
    return int(np.random.SeedSequence().generate_state(2, np.uint32).view(np.uint64)[0] >> 1)


class GPUAnnealer:
    def __init__(self, config: GPUAnnealerConfig):
        self.config = config
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.use_cuda = self.device.type == "cuda"
        self.total_flips = 0
        self.total_time = 0.0

    def anneal(self, model: IsingModel, update_rule: UpdateRule = UpdateRule.METROPOLIS,
               _replay=None) -> AnnealingResult:
        """Anneal `model` (its spins are the start and receive the final configuration).

        `_replay=(sites, uniforms)`: parity-test hook -- per-update arrays recorded from the
        reference's RNG replace the Philox stream (tests/test_host_api_gpu.py).
        """
        rule = rule_code(update_rule)
        cfg = self.config
        t_start = time.time()
        dev_idx = cfg.device_index if cfg.device_index is not None else _device_index(model.device)
        schedule = TemperatureScheduler.create_schedule(
            cfg.schedule_type, cfg.initial_temp, cfg.final_temp, cfg.n_sweeps,
            **cfg.schedule_params)
        adaptive = cfg.schedule_type == ScheduleType.ADAPTIVE
        site_mode = N.SITE_RANDOM if cfg.site_order == "random" else N.SITE_SEQUENTIAL
        arith = N.ARITH_F64 if (cfg.site_order == "random" or rule != N.RULE_METROPOLIS) \
            else N.ARITH_F32
        n = model.n_spins
        with AnnealEngine(dev_idx) as eng:
            eng.set_field_cache(cfg.field_cache)  # (before the couplings: "on" keeps a sparse matrix dense)
            model.load_into(eng, storage=cfg.coupling_storage)
            eng.set_update_rule(rule)
            eng.init_replicas(1, seed=fresh_seed(cfg.random_seed), s0=model.spins_int8()[None, :])
            energy_history = [float(eng.energies()[0])]
            temperature_history = [cfg.initial_temp]
            acceptance_rate_history = [0.0]
            acc_rate = 0.0
            sweep, last = 0, -1
            converged = False
            while sweep < cfg.n_sweeps and not converged:
                # run up to and including the next record sweep (sweep % record_interval == 0)
                nxt = sweep if sweep % cfg.record_interval == 0 else \
                    (sweep // cfg.record_interval + 1) * cfg.record_interval
                stop = min(nxt, cfg.n_sweeps - 1)
                if adaptive:
                    stop = sweep  # the schedule needs the acceptance rate after every sweep
                count = stop - sweep + 1
                raw = [schedule.update(s, acceptance_rate=acc_rate) for s in range(sweep, stop + 1)]
                temps = np.maximum(np.asarray(raw, np.float64), 1e-10)  # spin_dynamics.py:59
                kw = {}
                if _replay is not None:
                    lo, hi = sweep * n, (stop + 1) * n
                    kw = dict(replay_site=np.asarray(_replay[0][lo:hi], np.int32)[None, :],
                              replay_u=np.asarray(_replay[1][lo:hi], np.float32)[None, :])
                out = eng.sweep(count, site_mode=N.SITE_REPLAY if _replay is not None else site_mode,
                                arith=arith, sched=temps, energy_trace=True, **kw)
                acc, att = eng.stats()
                acc_rate = float(acc[0]) / float(att[0]) if att[0] else 0.0
                last = stop
                if stop % cfg.record_interval == 0:
                    energy_history.append(float(out["energy_trace"][-1, 0]))
                    temperature_history.append(float(raw[-1]))
                    acceptance_rate_history.append(acc_rate)
                    if check_convergence(energy_history, cfg.energy_tolerance):
                        converged = True
                sweep = stop + 1
            best_energy, best_spins, _ = eng.best(0)
            final = eng.spins(0)
            self.total_flips += int(eng.stats()[0][0])
        model.set_spins(torch.from_numpy(final.astype(np.float32)))
        total_time = time.time() - t_start
        self.total_time += total_time
        return AnnealingResult(
            best_configuration=torch.from_numpy(best_spins.astype(np.float32)),
            best_energy=best_energy, energy_history=energy_history,
            temperature_history=temperature_history,
            acceptance_rate_history=acceptance_rate_history, total_time=total_time,
            n_sweeps=last + 1, algorithm="simulated_annealing", device=f"cuda:{dev_idx}",
            random_seed=cfg.random_seed)

    def benchmark(self, model_sizes: List[int], n_trials: int = 3) -> Dict:
        """Reference gpu_annealer.py:271-329: random sparse-ish instances per size."""
        from .ising_model import IsingModelConfig
        results = {}
        for size in model_sizes:
            times, energies, sps = [], [], []
            for _ in range(n_trials):
                m = IsingModel(IsingModelConfig(n_spins=size, use_sparse=False))
                for _ in range(size):
                    i, j = np.random.randint(0, size, 2)
                    if i != j:
                        m.set_coupling(int(i), int(j), float(np.random.uniform(-1, 1)))
                r = self.anneal(m)
                times.append(r.total_time)
                energies.append(r.best_energy)
                sps.append(r.n_sweeps / r.total_time)
            results[size] = {"mean_time": np.mean(times), "std_time": np.std(times),
                             "mean_energy": np.mean(energies), "std_energy": np.std(energies),
                             "mean_sps": np.mean(sps), "std_sps": np.std(sps)}
        return results

    def get_memory_usage(self) -> Dict:
        if not self.use_cuda:
            return {"device": "cpu", "memory_allocated": 0, "memory_reserved": 0}
        free, total = torch.cuda.mem_get_info()
        return {"device": str(self.device), "memory_allocated": total - free,
                "memory_reserved": total - free, "memory_allocated_mb": (total - free) / 2 ** 20,
                "memory_reserved_mb": (total - free) / 2 ** 20}

    def __repr__(self) -> str:
        return (f"GPUAnnealer(device={self.device}, n_sweeps={self.config.n_sweeps}, "
                f"schedule={self.config.schedule_type.value})")
