"""MI355X-native digital-annealing engine for the Ising spin-sweep hot path.

Host-side mirror of the reference's annealer API (IsingModel, GPUAnnealer.anneal,
ParallelTempering.run, SpinGlassScheduler.anneal, CUDAKernelManager's three operators) over
hand-written HIP kernels reached through the C ABI in include/sga.h.
"""
from . import _native
from .exceptions import (AnnealingError, ConfigurationError, DeviceError, ModelError,
                         ResourceError, SpinGlassError, ValidationError)
from .engine import AnnealEngine, op_pt_exchange
from .temperature_scheduler import (ScheduleConfig, ScheduleType, TemperatureScheduler,
                                    temperature_ladder)
from .result import AnnealingResult
from .ising_model import IsingModel, IsingModelConfig
from .spin_dynamics import SpinDynamics, UpdateRule
from .energy_computer import ComputeMode, EnergyComputer, EnergyStats
from .gpu_annealer import GPUAnnealer, GPUAnnealerConfig
from .parallel_tempering import ParallelTempering, ParallelTemperingConfig
from .kernel_manager import CUDAKernelManager, GPUMemoryOptimizer, HIPKernelManager
from .scheduler import SpinGlassScheduler
from .sharded import LocalShardedTempering, ShardedTempering
from .multi_gpu import LoadBalancer, MultiGPUAnnealer, MultiGPUConfig
from .batch import BatchConfig, BatchProcessor
from . import encoders
from .encoders import IsingBuilder

__all__ = [
    "_native", "AnnealEngine", "op_pt_exchange", "SpinGlassError", "AnnealingError",
    "DeviceError", "ModelError", "ValidationError", "ConfigurationError", "ResourceError",
    "ScheduleType", "ScheduleConfig", "TemperatureScheduler", "temperature_ladder",
    "AnnealingResult", "IsingModel", "IsingModelConfig", "SpinDynamics", "UpdateRule",
    "GPUAnnealer", "GPUAnnealerConfig", "ParallelTempering", "ParallelTemperingConfig",
    "HIPKernelManager", "CUDAKernelManager", "GPUMemoryOptimizer", "SpinGlassScheduler",
    "ShardedTempering", "LocalShardedTempering", "MultiGPUAnnealer", "MultiGPUConfig", "LoadBalancer",
    "encoders", "IsingBuilder", "BatchConfig", "BatchProcessor", "EnergyComputer", "ComputeMode",
    "EnergyStats",
]
