"""SpinDynamics: single-replica Monte-Carlo driver over the HIP engine.

API of the reference's spin_glass_rl/core/spin_dynamics.py:11-429 for the Metropolis rule
(`sweep`, `single_spin_update`, acceptance statistics, histories).  A sweep is one kernel
launch (n random-site updates, Philox stream).  Metropolis, Glauber and heat-bath are modes of
the sweep kernels; the Wolff cluster rule (reference :193-255) has its own kernel and runs through
`sweep()` (a sweep = n cluster moves), not through `single_spin_update`.
"""
from enum import Enum
from typing import Optional, Tuple

import numpy as np

from . import _native as N
from .exceptions import AnnealingError
from .ising_model import IsingModel


class UpdateRule(Enum):
    METROPOLIS = "metropolis"
    GLAUBER = "glauber"
    HEAT_BATH = "heat_bath"
    WOLFF = "wolff"


_RULE_CODE = {UpdateRule.METROPOLIS: N.RULE_METROPOLIS, UpdateRule.GLAUBER: N.RULE_GLAUBER,
              UpdateRule.HEAT_BATH: N.RULE_HEAT_BATH, UpdateRule.WOLFF: N.RULE_WOLFF}


def rule_code(rule: UpdateRule) -> int:
    """Engine code of an update rule (sga_set_update_rule)."""
    if rule not in _RULE_CODE:
        raise AnnealingError(f"update rule '{rule}' is not implemented in the HIP engine")
    return _RULE_CODE[rule]


class SpinDynamics:
    def __init__(self, model: IsingModel, temperature: float = 1.0,
                 update_rule: UpdateRule = UpdateRule.METROPOLIS,
                 random_seed: Optional[int] = None):
        rule_code(update_rule)  # raises for rules the engine does not implement
        self.model = model
        self.temperature = temperature
        self.update_rule = update_rule
        self._rng = np.random.default_rng(random_seed)
        self._seed = int(self._rng.integers(0, 2 ** 63))
        self._sweeps = 0
        self.n_accepted = 0
        self.n_rejected = 0
        self.energy_history = []
        self.magnetization_history = []

    def set_temperature(self, temperature: float) -> None:
        self.temperature = max(temperature, 1e-10)  # reference :57-59

    def _engine(self):
        e = self.model._sync()
        e.set_update_rule(rule_code(self.update_rule))
        return e

    def single_spin_update(self, site: Optional[int] = None) -> Tuple[bool, float]:
        """One single-site update; returns (accepted, dE) with dE = 0 on rejection."""
        if self.update_rule is UpdateRule.WOLFF:
            raise AnnealingError("Wolff cluster moves run through sweep() on the HIP engine, not one at a time")
        if site is None:
            site = int(self._rng.integers(0, self.model.n_spins))
        u = float(np.float32(self._rng.random(dtype=np.float32)))
        accepted, dE = self._engine().update(0, site, self.temperature, u)
        if accepted:
            self.model.spins[site] *= -1
            self.model._mark_spins_synced()
            self.model._invalidate_cache()
            self.n_accepted += 1
            return True, dE
        self.n_rejected += 1
        return False, 0.0

    def sweep(self) -> float:
        """n single-spin updates at random sites (with replacement); returns the energy."""
        e = self._engine()
        n = self.model.n_spins
        e.set_counters(self._sweeps, 0)
        before = int(e.stats()[0][0])
        e.set_seed(self._seed)  # the model's engine is shared: carry our own stream key
        e.set_temperatures([self.temperature])
        e.sweep(1, site_mode=N.SITE_RANDOM)
        self._sweeps += 1
        acc = int(e.stats()[0][0]) - before
        self.n_accepted += acc  # Wolff: the sum of the cluster sizes (reference :252)
        if self.update_rule is not UpdateRule.WOLFF:
            self.n_rejected += n - acc
        spins = e.spins(0)
        self.model.spins.copy_(self.model.spins.new_tensor(spins.astype(np.float32)))
        self.model._mark_spins_synced()
        energy = float(e.energies()[0])
        self.model._energy_cache, self.model._cache_valid = energy, True
        self.energy_history.append(energy)
        self.magnetization_history.append(float(spins.sum()))
        return energy

    def run_dynamics(self, n_sweeps: int, record_interval: int = 1) -> dict:
        initial = self.model.compute_energy()
        for k in range(n_sweeps):
            self.sweep()
        final = self.model.compute_energy()
        return {"initial_energy": initial, "final_energy": final,
                "energy_history": list(self.energy_history),
                "acceptance_rate": self.get_acceptance_rate(), "n_sweeps": n_sweeps,
                "temperature": self.temperature}

    def get_acceptance_rate(self) -> float:
        total = self.n_accepted + self.n_rejected
        return self.n_accepted / total if total else 0.0

    @property
    def accepted_flips(self) -> int:
        return self.n_accepted

    @accepted_flips.setter
    def accepted_flips(self, value: int) -> None:
        self.n_accepted = value

    @property
    def total_flips(self) -> int:
        return self.n_accepted + self.n_rejected

    @total_flips.setter
    def total_flips(self, value: int) -> None:
        if value < self.n_accepted:
            raise ValueError("Total flips cannot be less than accepted flips")
        self.n_rejected = value - self.n_accepted

    def reset_statistics(self) -> None:
        self.n_accepted = 0
        self.n_rejected = 0
        self.energy_history = []

    # ------------------------------------------------------------------ diagnostics (host side)
    def get_autocorrelation_time(self, observable: str = "energy") -> float:
        """Sweeps until the normalised autocorrelation of the recorded history first falls below 1/e
        (reference core/spin_dynamics.py:361-391): inf with fewer than 10 records, the history's length
        when it never does (a constant history included)."""
        histories = {"energy": self.energy_history, "magnetization": self.magnetization_history}
        if observable not in histories:
            raise ValueError(f"Unknown observable: {observable}")
        x = np.asarray(histories[observable], np.float64)
        if x.size < 10:
            return float("inf")
        d = x - x.mean()
        c = np.correlate(d, d, mode="full")[x.size - 1:]  # lags 0 .. len - 1
        with np.errstate(divide="ignore", invalid="ignore"):
            c = c / c[0]
        below = np.flatnonzero(c < 1.0 / np.e)  # (NaN for a constant history: never below)
        return float(below[0]) if below.size else float(c.size)

    def thermal_equilibrium_check(self, window_size: int = 100) -> bool:
        """Are the means of the last two windows of the energy history indistinguishable (two-sample t
        test, p > 0.05; reference core/spin_dynamics.py:393-421)?  False until 2 windows are recorded."""
        if len(self.energy_history) < 2 * window_size:
            return False
        recent = np.asarray(self.energy_history[-window_size:], np.float64)
        older = np.asarray(self.energy_history[-2 * window_size:-window_size], np.float64)
        try:
            from scipy import stats
        except ImportError:
            # the reference's fallback when scipy is missing sets p_value to 0.05 or 0.01 by the variances and
            # returns p_value > 0.05 -- False either way (core/spin_dynamics.py:414-421): kept bit for bit
            return False
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # (identical windows: the statistic is 0 / 0)
            p_value = stats.ttest_ind(recent, older).pvalue
        return bool(p_value > 0.05)

    def __repr__(self) -> str:
        return (f"SpinDynamics(temperature={self.temperature:.4f}, "
                f"update_rule={self.update_rule.value}, "
                f"acceptance_rate={self.get_acceptance_rate():.4f})")
