"""MI355X-native digital-annealing engine for the Ising spin-sweep hot path.

Host-side mirror of the reference's annealer API (IsingModel, GPUAnnealer.anneal,
ParallelTempering.run, SpinGlassScheduler.anneal, CUDAKernelManager's three operators) over
hand-written HIP kernels reached through the C ABI in include/sga.h.
"""
from . import _native
from .exceptions import (AnnealingError, ConfigurationError, DeviceError, ModelError,
                         ResourceError, SpinGlassError, ValidationError)
from .engine import AnnealEngine, op_pt_exchange

__all__ = ["_native", "AnnealEngine", "op_pt_exchange", "SpinGlassError", "AnnealingError",
           "DeviceError", "ModelError", "ValidationError", "ConfigurationError", "ResourceError"]
