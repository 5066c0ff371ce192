"""EnergyComputer: energy evaluation helpers around an IsingModel.

API of the reference's spin_glass_rl/core/energy_computer.py:12-311 (`ComputeMode`,
`EnergyStats`, `EnergyComputer`).  The O(n^2) work -- total energies, all local fields, batches
of configurations (32 and more: one pass over J on the matrix cores, `fields_mfma_kernel`) -- runs in
the HIP engine (`energy_*_kernel`, `point_op_kernel`); what is left
on the host are O(n) recombinations of GPU-produced local fields for the diagnostic breakdowns.
The three compute modes of the reference return the same number; they are accepted and ignored.
"""
from dataclasses import dataclass
from enum import Enum
from typing import Optional

import numpy as np
import torch

from .engine import AnnealEngine
from .ising_model import IsingModel, _device_index


class ComputeMode(Enum):
    FULL = "full"
    INCREMENTAL = "incremental"
    VECTORIZED = "vectorized"


@dataclass
class EnergyStats:
    total_energy: float
    interaction_energy: float
    field_energy: float
    per_spin_energy: torch.Tensor


class EnergyComputer:
    def __init__(self, model: IsingModel, mode: ComputeMode = ComputeMode.FULL):
        self.model = model
        self.mode = mode

    def set_mode(self, mode: ComputeMode) -> None:
        self.mode = mode

    def invalidate_cache(self) -> None:
        self.model._invalidate_cache()

    def _with_spins(self, spins: Optional[torch.Tensor]) -> IsingModel:
        if spins is None:
            return self.model
        m = self.model.copy()  # shares nothing with the caller's model; engine is built lazily
        m.set_spins(spins.detach().float())
        return m

    def compute_total_energy(self, spins: Optional[torch.Tensor] = None) -> float:
        """-1/2 s.J.s - h.s (reference :51-69, :160-164)."""
        if spins is None:
            return self.model.compute_energy()
        return float(self.compute_batch_energies(spins.reshape(1, -1))[0].item())

    def compute_energy_change(self, flip_site: int) -> float:
        """dE = 2 s_i (sum_j J_ij s_j + h_i) (reference :71-87)."""
        return 2.0 * self.model.spins[flip_site].item() * self.model.get_local_field(flip_site)

    def local_fields(self, spins: Optional[torch.Tensor] = None) -> np.ndarray:
        m = self._with_spins(spins)
        return m._sync().local_fields(0, np.arange(m.n_spins, dtype=np.int32))

    def compute_energy_gradient(self, spins: Optional[torch.Tensor] = None) -> torch.Tensor:
        """dE/ds_i = -(sum_j J_ij s_j + h_i) (reference :120-140)."""
        return torch.from_numpy((-self.local_fields(spins)).astype(np.float32))

    def compute_energy_stats(self, spins: Optional[torch.Tensor] = None) -> EnergyStats:
        s = (self.model.spins if spins is None else spins).detach().cpu().numpy().astype(np.float64)
        h = self.model.external_fields.detach().cpu().numpy().astype(np.float64)
        f = self.local_fields(spins)                 # J s + h, from the GPU
        interaction = float(-0.5 * np.dot(s, f - h))
        field = float(-np.dot(h, s))
        per_spin = torch.from_numpy((-0.5 * s * (f - h) - h * s).astype(np.float32))
        return EnergyStats(total_energy=interaction + field, interaction_energy=interaction,
                           field_energy=field, per_spin_energy=per_spin)

    def compute_batch_energies(self, spin_configs: torch.Tensor) -> torch.Tensor:
        """Energies of B configurations [B, n] in one batched pass (reference :142-158 loops over them):
        B >= 32 configurations of a dense model read J once, on the matrix cores."""
        cfg = spin_configs.detach().reshape(-1, self.model.n_spins)
        s0 = cfg.cpu().numpy().astype(np.int8)
        with AnnealEngine(_device_index(self.model.device)) as e:
            self.model.load_into(e)
            e.init_replicas(s0.shape[0], seed=0, s0=s0)
            out = e.energies()
        return torch.from_numpy(out.astype(np.float32)).to(spin_configs.device)

    def __repr__(self) -> str:
        return f"EnergyComputer(mode={self.mode.value}, n_spins={self.model.n_spins})"
