"""AnnealingResult: the record every annealer returns.

Field-for-field the reference's spin_glass_rl/annealing/result.py:9-196 (same validation,
derived statistics and npz wire format), so callers and saved files are interchangeable.
"""
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch


@dataclass
class AnnealingResult:
    best_configuration: torch.Tensor
    best_energy: float
    energy_history: List[float]
    temperature_history: List[float]
    acceptance_rate_history: List[float]
    total_time: float
    n_sweeps: int
    convergence_sweep: Optional[int] = None
    final_temperature: float = 0.0
    final_acceptance_rate: float = 0.0
    energy_std: float = 0.0
    algorithm: str = "simulated_annealing"
    device: str = "cpu"
    random_seed: Optional[int] = None

    def __post_init__(self):
        if not isinstance(self.best_configuration, torch.Tensor):
            raise TypeError("best_configuration must be a torch.Tensor")
        if not isinstance(self.best_energy, (int, float)):
            raise TypeError("best_energy must be a numeric value")
        if np.isnan(self.best_energy) or np.isinf(self.best_energy):
            raise ValueError("best_energy contains invalid values (NaN or Inf)")
        if self.total_time < 0:
            raise ValueError("total_time must be non-negative")
        if self.n_sweeps <= 0:
            raise ValueError("n_sweeps must be positive")
        if self.energy_history:
            e = np.asarray(self.energy_history, dtype=np.float64)
            if not np.all(np.isfinite(e)):
                raise ValueError("energy_history contains invalid values (NaN or Inf)")
            self.energy_std = float(np.std(e))
            if len(e) > 10:  # first window whose spread is < 1 % of |best| (reference :62-72)
                window = min(50, len(e) // 4)
                for i in range(window, len(e)):
                    if np.std(e[i - window:i]) < 0.01 * abs(self.best_energy):
                        self.convergence_sweep = i - window
                        break
        if self.temperature_history:
            self.final_temperature = self.temperature_history[-1]
        if self.acceptance_rate_history:
            self.final_acceptance_rate = self.acceptance_rate_history[-1]

    _SUMMARY = ("best_energy", "total_time", "n_sweeps", "convergence_sweep", "final_temperature",
                "final_acceptance_rate", "energy_std", "algorithm", "device")
    _SAVED = _SUMMARY[:1] + ("energy_history", "temperature_history", "acceptance_rate_history") + \
        _SUMMARY[1:] + ("random_seed",)

    def get_summary(self) -> Dict:
        return {k: getattr(self, k) for k in self._SUMMARY}

    def save(self, filepath: str) -> None:
        data = {k: getattr(self, k) for k in self._SAVED}
        data["best_configuration"] = self.best_configuration.cpu().numpy()
        np.savez_compressed(filepath, **data)

    # how each saved entry comes back from the npz (the reference's reader, result.py:167-188):
    # 0-d arrays to python scalars, histories to lists; a falsy convergence_sweep / random_seed
    # (None is stored as an object array, and 0 reads as None too) means "not set"
    _opt_int = staticmethod(lambda a: int(a) if a else None)
    _READ = {"best_energy": float, "total_time": float, "n_sweeps": int, "final_temperature": float,
             "final_acceptance_rate": float, "energy_std": float, "algorithm": str, "device": str,
             "energy_history": np.ndarray.tolist, "temperature_history": np.ndarray.tolist,
             "acceptance_rate_history": np.ndarray.tolist}

    @classmethod
    def load(cls, filepath: str) -> "AnnealingResult":
        with np.load(filepath, allow_pickle=True) as z:
            fields = {k: cls._READ.get(k, cls._opt_int)(z[k]) for k in cls._SAVED}
            fields["best_configuration"] = torch.from_numpy(z["best_configuration"])
        return cls(**fields)

    def __repr__(self) -> str:
        return (f"AnnealingResult(best_energy={self.best_energy:.6f}, n_sweeps={self.n_sweeps}, "
                f"time={self.total_time:.3f}s, converged_at={self.convergence_sweep})")
