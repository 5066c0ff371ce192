"""SpinGlassScheduler.anneal: the README-level entry point (reference README.md:51-83).

The reference never defines this class; its documented call is

    solution = scheduler.anneal(ising_model, n_replicas=1000, n_sweeps=10000,
                                beta_schedule='geometric')

Here it runs `n_replicas` replicas of the model on one temperature ladder spaced by
`beta_schedule`, sweeps them together on the GPU and exchanges neighbours every
`exchange_interval` sweeps; the best configuration seen by any replica at any sweep end is
returned.  (Problem encoding / decoding -- problem_to_ising, spins_to_schedule -- are callers
of this path and out of scope)
"""
import time
from typing import Optional, Sequence, Union

import numpy as np
import torch

from .engine import AnnealEngine
from .exceptions import ConfigurationError
from .gpu_annealer import fresh_seed
from .ising_model import IsingModel, _device_index
from .result import AnnealingResult


def beta_ladder(n_replicas: int, beta_min: float, beta_max: float, schedule: str) -> np.ndarray:
    """Inverse temperatures from hot (index 0) to cold."""
    if n_replicas == 1:
        return np.asarray([beta_max])
    x = np.arange(n_replicas) / (n_replicas - 1)
    if schedule == "geometric":
        return beta_min * (beta_max / beta_min) ** x
    if schedule == "linear":
        return beta_min + (beta_max - beta_min) * x
    if schedule == "exponential":  # evenly spaced log10(beta), numpy.logspace form
        return np.logspace(np.log10(beta_min), np.log10(beta_max), n_replicas)
    raise ConfigurationError(f"unknown beta_schedule '{schedule}'")


class SpinGlassScheduler:
    def __init__(self, coupling_strength: float = 1.0, external_field_strength: float = 0.5,
                 device: str = "cuda", random_seed: Optional[int] = None):
        self.coupling_strength = coupling_strength
        self.external_field_strength = external_field_strength
        self.device = torch.device(device)
        self.random_seed = random_seed

    def anneal(self, ising_model: IsingModel, n_replicas: int = 1000, n_sweeps: int = 10000,
               beta_schedule: Union[str, Sequence[float]] = "geometric", beta_min: float = 0.1,
               beta_max: float = 10.0, exchange_interval: int = 10, n_ladders: int = 1,
               coupling_storage: str = "auto", record_interval: int = 10,
               autotune: Optional[bool] = None, field_cache: str = "auto") -> AnnealingResult:
        """field_cache: "auto" keeps every replica's local fields resident where the problem allows it
        (dense integer-valued symmetric couplings) so that a coupling row is read only when a proposal is
        accepted -- the chain, and with it the result, is the one "off" (a row per proposal) gives."""
        if n_replicas < 1 or n_sweeps < 1 or n_replicas % n_ladders:
            raise ConfigurationError("bad replica / sweep / ladder counts")
        t0 = time.time()
        L = n_replicas // n_ladders
        if isinstance(beta_schedule, str):
            betas = beta_ladder(L, beta_min, beta_max, beta_schedule)
        else:
            betas = np.asarray(beta_schedule, np.float64)
            if betas.size != L:
                raise ConfigurationError("explicit beta ladder must have n_replicas/n_ladders entries")
        temps = np.tile(1.0 / betas, n_ladders)
        dev = _device_index(self.device)
        e_hist, t_hist = [], []
        with AnnealEngine(dev) as eng:
            eng.set_field_cache(field_cache)  # (before the couplings: "on" keeps a sparse matrix dense)
            ising_model.load_into(eng, storage=coupling_storage)
            eng.init_replicas(n_replicas, seed=fresh_seed(self.random_seed))
            eng.set_ladder(temps, n_ladders)
            eng.maybe_autotune(n_sweeps, autotune)  # measured launch geometry for long runs
            done = 0
            while done < n_sweeps:
                step = min(exchange_interval, n_sweeps - done)
                eng.sweep(step)
                done += step
                if L > 1 and done < n_sweeps:
                    eng.exchange(count=False)  # enqueue only: no host round trip per round
                if (done // exchange_interval) % max(1, record_interval // exchange_interval) == 0:
                    e_hist.append(float(eng.best(with_spins=False)[0]))
                    t_hist.append(float(temps.min()))
            best_e, best_s, _ = eng.best()
            acc, att = eng.stats()
        return AnnealingResult(
            best_configuration=torch.from_numpy(best_s.astype(np.float32)), best_energy=best_e,
            energy_history=e_hist, temperature_history=t_hist,
            acceptance_rate_history=[float(acc.sum()) / float(max(att.sum(), 1))],
            total_time=time.time() - t0, n_sweeps=n_sweeps, algorithm="parallel_tempering",
            device=f"cuda:{dev}", random_seed=self.random_seed)
