"""Exception types raised by the host layer.

Names follow the reference's spin_glass_rl/utils/exceptions.py:6-71 so that callers catching
`AnnealingError` / `DeviceError` keep working; the C ABI's negative status codes are mapped
onto them in `_native.check`.
"""
from typing import Any, Dict, Optional


class SpinGlassError(Exception):
    def __init__(self, message: str, details: Optional[Dict[str, Any]] = None):
        super().__init__(message)
        self.message = message
        self.details = details or {}

    def __str__(self) -> str:
        if not self.details:
            return self.message
        extra = ", ".join(f"{k}={v}" for k, v in self.details.items())
        return f"{self.message} (Details: {extra})"


class ModelError(SpinGlassError):
    pass


class AnnealingError(SpinGlassError):
    pass


class DeviceError(SpinGlassError):
    pass


class ValidationError(SpinGlassError):
    pass


class ConfigurationError(SpinGlassError):
    pass


class ResourceError(SpinGlassError):
    pass
