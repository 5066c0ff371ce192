"""IsingModel: the state container callers hand to the annealers.

Mirrors the reference's spin_glass_rl/core/ising_model.py:36-258 (same constructor, attributes
and method names; H = -1/2 sum_ij J_ij s_i s_j - sum_i h_i s_i with symmetric J stored in both
triangles).  Container operations (set_coupling, copy, to_dict ...) are plain tensor edits on
the host; everything arithmetic -- local field, flip, energy -- runs in the HIP engine, which
is (re)fed lazily whenever the tensors changed.  There is no CPU arithmetic path: those
methods raise DeviceError without an MI355X.
"""
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch

from .engine import AnnealEngine
from .exceptions import ModelError


@dataclass
class IsingModelConfig:
    n_spins: int
    coupling_strength: float = 1.0
    external_field_strength: float = 0.5
    use_sparse: bool = True
    device: str = "cpu"


def _device_index(device: torch.device) -> int:
    return device.index if (device.type == "cuda" and device.index is not None) else 0


def coo_to_csr(couplings: torch.Tensor):
    """Sparse COO tensor -> (rowptr, colidx, val) numpy arrays (duplicates summed)."""
    c = couplings.coalesce()
    n = c.shape[0]
    idx = c.indices().cpu().numpy()
    val = c.values().cpu().numpy().astype(np.float32)
    order = np.lexsort((idx[1], idx[0]))
    rows, cols, val = idx[0][order], idx[1][order], val[order]
    rowptr = np.zeros(n + 1, np.int32)
    np.add.at(rowptr, rows + 1, 1)
    return np.cumsum(rowptr).astype(np.int32), cols.astype(np.int32), val


class IsingModel:
    def __init__(self, config: IsingModelConfig):
        if config.n_spins <= 0:
            raise ModelError("n_spins must be positive")
        self.config = config
        self.n_spins = config.n_spins
        self.device = torch.device(config.device)
        n = self.n_spins
        self.spins = (torch.randint(0, 2, (n,), device=self.device) * 2 - 1).float()
        if config.use_sparse:
            self.couplings = torch.sparse_coo_tensor(torch.empty((2, 0), dtype=torch.long),
                                                     torch.empty(0), (n, n), device=self.device)
        else:
            self.couplings = torch.zeros((n, n), device=self.device)
        self.external_fields = torch.zeros(n, device=self.device)
        self._energy_cache: Optional[float] = None
        self._cache_valid = False
        # engine mirror state
        self._engine: Optional[AnnealEngine] = None
        self._seen = {"J": None, "h": None, "s": None}

    # ------------------------------------------------------------------ container edits
    def _check_index(self, *idx):
        for i in idx:
            if not 0 <= i < self.n_spins:
                raise ValueError(f"Spin indices out of range: {idx}, n_spins={self.n_spins}")

    def set_coupling(self, i: int, j: int, strength: float) -> None:
        self._check_index(i, j)
        if self.config.use_sparse:
            # keep COO, replace the two symmetric entries (reference :94-99 densifies instead)
            c = self.couplings.coalesce()
            idx, val = c.indices(), c.values()
            keep = ~(((idx[0] == i) & (idx[1] == j)) | ((idx[0] == j) & (idx[1] == i)))
            add = torch.tensor([[i, j], [j, i]] if i != j else [[i], [i]], dtype=torch.long,
                               device=idx.device)
            new_idx = torch.cat([idx[:, keep], add], dim=1)
            new_val = torch.cat([val[keep], torch.full((add.shape[1],), float(strength),
                                                       dtype=val.dtype, device=val.device)])
            self.couplings = torch.sparse_coo_tensor(new_idx, new_val, c.shape).coalesce()
        else:
            self.couplings[i, j] = strength
            self.couplings[j, i] = strength
        self._invalidate_cache()

    def set_couplings_from_matrix(self, coupling_matrix: torch.Tensor) -> None:
        if self.config.use_sparse:
            self.couplings = coupling_matrix.to_sparse()
        else:
            self.couplings = coupling_matrix.clone()
        self._invalidate_cache()

    def set_external_field(self, i: int, strength: float) -> None:
        self.external_fields[i] = strength
        self._invalidate_cache()

    def set_external_fields(self, fields: torch.Tensor) -> None:
        self.external_fields = fields.clone().to(self.device)
        self._invalidate_cache()

    def set_spins(self, spins: torch.Tensor) -> None:
        self.spins = spins.clone().to(self.device)
        self._invalidate_cache()

    def get_spins(self) -> torch.Tensor:
        return self.spins.clone()

    def reset_to_random(self) -> None:
        self.spins = (torch.randint(0, 2, (self.n_spins,), device=self.device) * 2 - 1).float()
        self._invalidate_cache()

    def get_magnetization(self) -> float:
        return self.spins.sum().item() / self.n_spins

    def copy(self) -> "IsingModel":
        m = IsingModel(self.config)
        m.spins = self.spins.clone()
        m.couplings = self.couplings.clone()
        m.external_fields = self.external_fields.clone()
        return m

    def dense_couplings(self) -> torch.Tensor:
        return self.couplings.to_dense() if self.couplings.is_sparse else self.couplings

    def to_dict(self) -> Dict:
        c = self.config
        return {
            "config": {"n_spins": c.n_spins, "coupling_strength": c.coupling_strength,
                       "external_field_strength": c.external_field_strength,
                       "use_sparse": c.use_sparse, "device": c.device},
            "spins": self.spins.cpu().numpy(),
            "couplings": self.dense_couplings().cpu().numpy(),
            "external_fields": self.external_fields.cpu().numpy(),
        }

    @classmethod
    def from_dict(cls, data: Dict) -> "IsingModel":
        m = cls(IsingModelConfig(**data["config"]))
        m.spins = torch.from_numpy(data["spins"]).to(m.device)
        J = torch.from_numpy(data["couplings"]).to(m.device)
        m.couplings = J.to_sparse() if m.config.use_sparse else J
        m.external_fields = torch.from_numpy(data["external_fields"]).to(m.device)
        return m

    def _invalidate_cache(self) -> None:
        self._cache_valid = False

    # ------------------------------------------------------------------ engine plumbing
    @staticmethod
    def _stamp(t: torch.Tensor):
        return (id(t), t._version)

    def load_into(self, engine: AnnealEngine, storage: str = "auto") -> None:
        """Hand J and h to an engine (dense fp32 matrix, or CSR for sparse models)."""
        if self.couplings.is_sparse:
            engine.set_csr(*coo_to_csr(self.couplings), self.external_fields)
        else:
            engine.set_dense(self.couplings, self.external_fields, storage=storage)

    def spins_int8(self) -> np.ndarray:
        return self.spins.detach().cpu().numpy().astype(np.int8)

    def _sync(self) -> AnnealEngine:
        """Bring the private 1-replica engine up to date with the tensors."""
        if self._engine is None:
            self._engine = AnnealEngine(_device_index(self.device))
        e = self._engine
        j, h, s = self._stamp(self.couplings), self._stamp(self.external_fields), self._stamp(self.spins)
        if self._seen["J"] != j or self._seen["h"] != h:
            self.load_into(e)
            self._seen.update(J=j, h=h, s=None)
        if self._seen["s"] != s or e.R != 1:
            e.init_replicas(1, seed=0, s0=self.spins_int8()[None, :])
            self._seen["s"] = s
        return e

    def _mark_spins_synced(self):
        self._seen["s"] = self._stamp(self.spins)

    # ------------------------------------------------------------------ arithmetic (on the GPU)
    def get_local_field(self, i: int) -> float:
        """sum_j J_ij s_j + h_i  (reference :176-185)."""
        self._check_index(i)
        return float(self._sync().local_fields(0, [i])[0])

    def flip_spin(self, i: int) -> float:
        """Flip spin i, return dE = 2 s_i (sum_j J_ij s_j + h_i)  (reference :125-147)."""
        self._check_index(i)
        e = self._sync()
        dE = e.flip(0, i)
        self.spins[i] *= -1
        self._mark_spins_synced()
        self._invalidate_cache()
        return dE

    def compute_energy(self) -> float:
        """-1/2 s.(J s) - h.s, cached until the next mutation (reference :149-174)."""
        if self._cache_valid and self._energy_cache is not None:
            stamp_ok = self._seen["s"] == self._stamp(self.spins) and \
                self._seen["J"] == self._stamp(self.couplings) and \
                self._seen["h"] == self._stamp(self.external_fields)
            if stamp_ok:
                return self._energy_cache
        e = self._sync()
        e.recompute_energies()
        self._energy_cache = float(e.energies()[0])
        self._cache_valid = True
        return self._energy_cache

    def __repr__(self) -> str:
        return f"IsingModel(n_spins={self.n_spins}, sparse={self.couplings.is_sparse})"
