"""Temperature schedules (host side).

Same names, parameters and values as the reference's
spin_glass_rl/annealing/temperature_scheduler.py:11-331 (checked against tables generated from
it, tests/golden/schedules.npz).  Written here as closed-form `T(sweep)` functions with one
shared `update` bookkeeping path; the hot path only ever receives the resulting array of
temperatures (sga_sweep's `sched`).
"""
from dataclasses import dataclass
from enum import Enum
from typing import Callable, Dict, List, Optional

import numpy as np


class ScheduleType(Enum):
    LINEAR = "linear"
    EXPONENTIAL = "exponential"
    GEOMETRIC = "geometric"
    LOGARITHMIC = "logarithmic"
    POWER_LAW = "power_law"
    ADAPTIVE = "adaptive"
    FAST = "fast"
    BOLTZMANN = "boltzmann"
    CUSTOM = "custom"


@dataclass
class ScheduleConfig:
    schedule_type: ScheduleType
    initial_temp: float
    final_temp: float
    total_sweeps: int
    alpha: float = 0.95
    k: float = 1.0
    c: float = 1.0
    target_acceptance: float = 0.44
    adaptation_window: int = 100
    adaptation_rate: float = 0.1


class TemperatureSchedule:
    """Base: subclasses give `_raw(sweep)`; the floor at final_temp is applied here."""

    floors = True

    def __init__(self, config: ScheduleConfig):
        self.config = config
        self.reset()

    def reset(self) -> None:
        self.current_sweep = 0
        self.current_temp = self.config.initial_temp
        self.temperature_history: List[float] = [self.config.initial_temp]

    def _raw(self, sweep: int) -> float:  # pragma: no cover - abstract
        raise NotImplementedError

    def get_temperature(self, sweep: int) -> float:
        t = self._raw(sweep)
        return max(t, self.config.final_temp) if self.floors else t

    def update(self, sweep: int, **kwargs) -> float:
        self.current_sweep = sweep
        self.current_temp = self.get_temperature(sweep)
        self.temperature_history.append(self.current_temp)
        return self.current_temp

    def table(self, first: int, count: int) -> np.ndarray:
        """Temperatures of sweeps [first, first+count) without touching the history."""
        return np.asarray([self.get_temperature(s) for s in range(first, first + count)])


class LinearSchedule(TemperatureSchedule):
    # reference :68-90: T0 - (T0 - Tf) * sweep / total, Tf once sweep >= total
    def _raw(self, sweep):
        c = self.config
        if sweep >= c.total_sweeps:
            return c.final_temp
        return c.initial_temp - (c.initial_temp - c.final_temp) * (sweep / c.total_sweeps)


class ExponentialSchedule(TemperatureSchedule):
    # reference :93-113: T0 * exp(-lambda * sweep), lambda chosen to reach Tf at total_sweeps
    def __init__(self, config):
        super().__init__(config)
        if config.final_temp > 0:
            self.lambda_param = -np.log(config.final_temp / config.initial_temp) / config.total_sweeps
        else:
            self.lambda_param = 0.01

    def _raw(self, sweep):
        return self.config.initial_temp * np.exp(-self.lambda_param * sweep)


class GeometricSchedule(TemperatureSchedule):
    # reference :116-129: T0 * alpha**sweep (ignores total_sweeps)
    def _raw(self, sweep):
        return self.config.initial_temp * (self.config.alpha ** sweep)


class LogarithmicSchedule(TemperatureSchedule):
    # reference :132-149: (c / log(1+sweep)) * T0 / c, T0 at sweep 0
    def _raw(self, sweep):
        c = self.config
        if sweep == 0:
            return c.initial_temp
        t = c.c / np.log(1 + sweep)
        return t * c.initial_temp / c.c

    def get_temperature(self, sweep):
        if sweep == 0:
            return self.config.initial_temp  # not floored in the reference (:136-137)
        return super().get_temperature(sweep)


class PowerLawSchedule(TemperatureSchedule):
    # reference :152-165
    def _raw(self, sweep):
        return self.config.initial_temp / ((1 + sweep) ** self.config.k)


class FastSchedule(TemperatureSchedule):
    # reference :168-184
    def _raw(self, sweep):
        return self.config.initial_temp if sweep == 0 else self.config.initial_temp / sweep

    def get_temperature(self, sweep):
        return self.config.initial_temp if sweep == 0 else super().get_temperature(sweep)


class BoltzmannSchedule(TemperatureSchedule):
    # reference :187-203
    def _raw(self, sweep):
        return self.config.initial_temp if sweep == 0 else self.config.initial_temp / np.log(1 + sweep)

    def get_temperature(self, sweep):
        return self.config.initial_temp if sweep == 0 else super().get_temperature(sweep)


class AdaptiveSchedule(TemperatureSchedule):
    """Geometric base, nudged by the recent acceptance rate (reference :206-253)."""

    def __init__(self, config):
        super().__init__(config)
        self.acceptance_history: List[float] = []
        self._base = GeometricSchedule(config)

    def get_temperature(self, sweep):
        return self.current_temp

    def update(self, sweep: int, acceptance_rate: Optional[float] = None, **kwargs) -> float:
        c = self.config
        self.current_sweep = sweep
        if acceptance_rate is not None:
            self.acceptance_history.append(acceptance_rate)
        base = self._base.get_temperature(sweep)
        if len(self.acceptance_history) >= c.adaptation_window:
            recent = float(np.mean(self.acceptance_history[-c.adaptation_window:]))
            if recent > c.target_acceptance:
                factor = 1.0 - c.adaptation_rate
            elif recent < c.target_acceptance:
                factor = 1.0 + c.adaptation_rate
            else:
                factor = 1.0
            self.current_temp = max(base * factor, c.final_temp)
        else:
            self.current_temp = base
        self.temperature_history.append(self.current_temp)
        return self.current_temp


class CustomSchedule(TemperatureSchedule):
    def __init__(self, config, schedule_func: Callable[[int], float]):
        super().__init__(config)
        self.schedule_func = schedule_func

    def _raw(self, sweep):
        return self.schedule_func(sweep)


class TemperatureScheduler:
    """Factory, reference :256-331."""

    _classes: Dict[ScheduleType, type] = {
        ScheduleType.LINEAR: LinearSchedule,
        ScheduleType.EXPONENTIAL: ExponentialSchedule,
        ScheduleType.GEOMETRIC: GeometricSchedule,
        ScheduleType.LOGARITHMIC: LogarithmicSchedule,
        ScheduleType.POWER_LAW: PowerLawSchedule,
        ScheduleType.ADAPTIVE: AdaptiveSchedule,
        ScheduleType.FAST: FastSchedule,
        ScheduleType.BOLTZMANN: BoltzmannSchedule,
        ScheduleType.CUSTOM: CustomSchedule,
    }

    @classmethod
    def create_schedule(cls, schedule_type: ScheduleType, initial_temp: float, final_temp: float,
                        total_sweeps: int, custom_func: Optional[Callable[[int], float]] = None,
                        **kwargs) -> TemperatureSchedule:
        config = ScheduleConfig(schedule_type=schedule_type, initial_temp=initial_temp,
                                final_temp=final_temp, total_sweeps=total_sweeps, **kwargs)
        klass = cls._classes[schedule_type]
        if schedule_type == ScheduleType.CUSTOM:
            if custom_func is None:
                raise ValueError("custom_func required for CUSTOM schedule type")
            return klass(config, custom_func)
        return klass(config)

    @classmethod
    def get_available_schedules(cls) -> List[str]:
        return [s.value for s in ScheduleType]

    @classmethod
    def compare_schedules(cls, initial_temp: float, final_temp: float, total_sweeps: int,
                          schedule_types: Optional[List[ScheduleType]] = None) -> dict:
        kinds = schedule_types or [ScheduleType.LINEAR, ScheduleType.EXPONENTIAL,
                                   ScheduleType.GEOMETRIC, ScheduleType.LOGARITHMIC]
        step = max(1, total_sweeps // 100)
        out = {}
        for kind in kinds:
            if kind == ScheduleType.CUSTOM:
                continue
            sch = cls.create_schedule(kind, initial_temp, final_temp, total_sweeps)
            out[kind.value] = [sch.get_temperature(s) for s in range(0, total_sweeps, step)]
        return out

    @classmethod
    def recommend_schedule(cls, problem_size: int, time_budget: int,
                           convergence_preference: str = "balanced"):
        # reference :425-463
        if convergence_preference == "fast":
            if problem_size < 1000:
                return ScheduleType.FAST, {"k": 1.0}
            return ScheduleType.EXPONENTIAL, {"alpha": 0.99}
        if convergence_preference == "quality":
            if time_budget > 10000:
                return ScheduleType.LOGARITHMIC, {"c": 10.0}
            return ScheduleType.GEOMETRIC, {"alpha": 0.95}
        if problem_size < 1000:
            return ScheduleType.GEOMETRIC, {"alpha": 0.95}
        return ScheduleType.ADAPTIVE, {"alpha": 0.95, "target_acceptance": 0.44,
                                       "adaptation_window": 100}


def temperature_ladder(n_replicas: int, temp_min: float, temp_max: float,
                       distribution: str = "geometric") -> List[float]:
    """Replica ladder, index 0 = hottest (reference annealing/parallel_tempering.py:146-173)."""
    if distribution == "geometric":
        ratio = temp_min / temp_max
        return [temp_max * (ratio ** (i / (n_replicas - 1))) for i in range(n_replicas)]
    if distribution == "linear":
        return np.linspace(temp_max, temp_min, n_replicas).tolist()
    if distribution == "exponential":
        return np.logspace(np.log10(temp_max), np.log10(temp_min), n_replicas).tolist()
    raise ValueError(f"Unknown temperature distribution: {distribution}")
