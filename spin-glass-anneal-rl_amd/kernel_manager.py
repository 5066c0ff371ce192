"""HIPKernelManager: the reference's operator API over the HIP engine.

Same three methods, arguments and return values as the reference's CUDAKernelManager
(spin_glass_rl/annealing/cuda_kernels.py:228-369) -- exported under that name too -- with the
semantics of what that class really executes, its `_fallback` methods (:371-443):

* metropolis_update_optimized: `n_updates` passes over the sites IN ORDER i = 0..n-1, fp32
  arithmetic, local field h_i + sum_j J_ij s_j - J_ii s_i, accept if dE <= 0 or
  rand < exp(-dE/T); returns (spins, accepted, per-site accumulated dE);
* compute_energy_optimized: -1/2 s.J.s - h.s;
* parallel_tempering_exchange_optimized: sequential adjacent pairs, p = exp((b2-b1)(E1-E2)),
  swaps spin rows and energies in place, returns the number of swaps.

Unlike the reference there is no silent fallback: without the GPU library the constructor
raises DeviceError.  The engine (packed J in HBM) is cached per couplings tensor.
"""
from typing import Optional, Tuple

import numpy as np
import torch

from . import _native as N
from .engine import AnnealEngine, op_pt_exchange
from .gpu_annealer import fresh_seed


class HIPKernelManager:
    def __init__(self, device=None, seed: Optional[int] = None):
        if isinstance(device, torch.device):
            self.device = device
        else:
            self.device = torch.device("cuda" if device is None else device)
        self.device_index = self.device.index or 0
        N.lib()  # fail loudly if libsga.so is missing
        self._engine: Optional[AnnealEngine] = None
        self._key = None
        self._seed = fresh_seed(seed)
        self._calls = 0
        self.compiled_kernels = {"metropolis_update": True, "compute_energy": True,
                                 "parallel_tempering": True}

    def _engine_for(self, couplings: torch.Tensor, external_fields: torch.Tensor) -> AnnealEngine:
        key = (couplings.data_ptr(), couplings._version, tuple(couplings.shape),
               external_fields.data_ptr(), external_fields._version)
        if self._engine is None:
            self._engine = AnnealEngine(self.device_index)
        if key != self._key:
            J = couplings.to_dense() if couplings.is_sparse else couplings
            self._engine.set_dense(J, external_fields, storage="f32")
            self._key = key
        return self._engine

    @staticmethod
    def _spins_i8(spins: torch.Tensor) -> np.ndarray:
        return spins.detach().cpu().numpy().astype(np.int8)

    def metropolis_update_optimized(self, spins: torch.Tensor, couplings: torch.Tensor,
                                    external_fields: torch.Tensor, temperature: float,
                                    n_updates: int = 1, _uniforms=None
                                    ) -> Tuple[torch.Tensor, int, torch.Tensor]:
        eng = self._engine_for(couplings, external_fields)
        n = spins.shape[0]
        eng.init_replicas(1, seed=self._seed, s0=self._spins_i8(spins)[None, :])
        eng.set_counters(self._calls, 0)
        self._calls += int(n_updates)
        eng.set_temperatures([float(temperature)])
        out = eng.sweep(int(n_updates), site_mode=N.SITE_SEQUENTIAL, arith=N.ARITH_F32,
                        replay_u=_uniforms, trace=True)
        new = torch.from_numpy(eng.spins(0).astype(np.float32)).to(spins.device)
        spins.copy_(new)  # the reference updates `spins` in place and returns it (:282)
        dE = out["dE_trace"][0].reshape(int(n_updates), n).astype(np.float32)
        changes = np.zeros(n, np.float32)
        for k in range(int(n_updates)):  # energy_changes[i] += dE per pass (:392)
            changes = changes + dE[k]
        accepted = int(out["accept_trace"].sum())
        return spins, accepted, torch.from_numpy(changes).to(spins.device)

    def compute_energy_optimized(self, spins: torch.Tensor, couplings: torch.Tensor,
                                 external_fields: torch.Tensor) -> float:
        eng = self._engine_for(couplings, external_fields)
        eng.init_replicas(1, seed=self._seed, s0=self._spins_i8(spins)[None, :])
        return float(eng.energies()[0])

    def parallel_tempering_exchange_optimized(self, spins_arrays: torch.Tensor,
                                              energies: torch.Tensor,
                                              temperatures: torch.Tensor, _uniforms=None) -> int:
        on_gpu = spins_arrays.is_cuda
        sp = spins_arrays if (on_gpu and spins_arrays.dtype == torch.float32 and
                              spins_arrays.is_contiguous()) else \
            spins_arrays.detach().float().contiguous().to(self.device)
        en = energies if (energies.is_cuda and energies.dtype == torch.float32 and
                          energies.is_contiguous()) else \
            energies.detach().float().contiguous().to(self.device)
        tt = temperatures.detach().float().contiguous().to(self.device)
        self._calls += 1
        k = op_pt_exchange(self.device_index, sp, en, tt, u=_uniforms, seed=self._seed,
                           round_=self._calls)
        if sp is not spins_arrays:
            spins_arrays.copy_(sp.to(spins_arrays.device).to(spins_arrays.dtype))
        if en is not energies:
            energies.copy_(en.to(energies.device).to(energies.dtype))
        return k


CUDAKernelManager = HIPKernelManager


class GPUMemoryOptimizer:
    """Host helper of the reference, same method names and argument meaning (annealing/cuda_kernels.py:446-569):
    batch sizing, scratch tensors, dense / sparse choice, memory statistics.  What differs is the sizing model: the
    reference counts one copy of J per configuration (and caps the batch at 64); in this engine the replicas of a
    batch share ONE copy of the couplings, so `get_optimal_batch_size` returns how many replicas fit beside it."""

    def __init__(self, device=None):
        self.device = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.memory_pool = {}

    @staticmethod
    def bytes_per_replica(n_spins: int) -> int:
        return n_spins * 2 + 64  # spins + best spins (int8) + scalars; J is shared

    def get_optimal_batch_size(self, n_spins: int, available_memory: Optional[int] = None) -> int:
        """Replicas that fit into 80 % of `available_memory` (bytes; default: what the device has free) beside the raw and
        the packed fp32 copy of the couplings (cuda_kernels.py:458-490, with this engine's footprint)."""
        if available_memory is None:
            if self.device.type == "cuda" and torch.cuda.is_available():
                available_memory, _ = torch.cuda.mem_get_info(self.device)
            else:
                available_memory = 4 * 1024 ** 3   # (the reference's assumption without a GPU)
        usable = int(available_memory * 0.8)
        j_bytes = 2 * n_spins * n_spins * 4
        return max(1, (usable - j_bytes) // self.bytes_per_replica(n_spins))

    optimal_batch_size = get_optimal_batch_size   # (round-1 name)

    def create_memory_efficient_tensors(self, n_spins: int, batch_size: int, use_half_precision: bool = False) -> dict:
        """The reference's scratch set, same keys and shapes (cuda_kernels.py:492-518)."""
        dtype = torch.float16 if use_half_precision else torch.float32
        shapes = {"spins_batch": (batch_size, n_spins), "energies_batch": (batch_size,), "temp_spins": (n_spins,),
                  "random_values": (batch_size * n_spins,), "local_fields": (batch_size, n_spins)}
        return {k: torch.zeros(*shape, dtype=dtype, device=self.device) for k, shape in shapes.items()}

    def optimize_coupling_matrix_storage(self, couplings: torch.Tensor, sparsity_threshold: float = 0.1) -> torch.Tensor:
        """Sparse COO when more than `sparsity_threshold` of the entries are zero, else the dense matrix as it is
        (cuda_kernels.py:520-540; IsingModel / sga_set_csr take either)."""
        if couplings.is_sparse:
            return couplings
        zeros = 1.0 - float(torch.count_nonzero(couplings).item()) / max(couplings.numel(), 1)
        return couplings.to_sparse_coo() if zeros > sparsity_threshold else couplings

    @staticmethod
    def prefers_sparse(couplings: torch.Tensor, threshold: float = 0.9) -> bool:
        """CSR pays in THIS engine when fewer than ~10 % of the entries are non-zero."""
        if couplings.is_sparse:
            return True
        nz = int((couplings != 0).sum().item())
        return 1.0 - nz / couplings.numel() > threshold

    def clear_memory_cache(self):
        if self.device.type == "cuda" and torch.cuda.is_available():
            torch.cuda.empty_cache()
        self.memory_pool.clear()

    def get_memory_stats(self) -> dict:
        """Keys of the reference (cuda_kernels.py:548-569); zeros without a GPU."""
        stats = {"device": str(self.device), "memory_allocated": 0, "memory_reserved": 0, "max_memory_allocated": 0,
                 "memory_stats": {}}
        if self.device.type == "cuda" and torch.cuda.is_available():
            stats.update(memory_allocated=torch.cuda.memory_allocated(self.device),
                         memory_reserved=torch.cuda.memory_reserved(self.device),
                         max_memory_allocated=torch.cuda.max_memory_allocated(self.device),
                         memory_stats=torch.cuda.memory_stats(self.device))
        return stats
