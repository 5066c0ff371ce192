"""ParallelTempering: replica exchange Monte Carlo with every replica resident on the GPU.

Host loop of the reference's spin_glass_rl/annealing/parallel_tempering.py:16-313: geometric /
linear / exponential ladder with index 0 the hottest, one Metropolis sweep of every replica per
step, nearest-neighbour exchanges every `exchange_interval` sweeps (skipping sweep 0) with a
random even/odd start, statistics and best-over-replicas on record sweeps only.  All replicas
advance in one kernel call between two events; an exchange swaps temperature labels on the
device instead of moving spin tensors (same Markov chain, no copies).

`exchange_method="all_pairs"` follows the reference's CPU branch (:222-232: every pair i < j behind the
gate `np.random.rand() < 0.1`, the criterion of `_attempt_single_exchange`, :234-258; pinned to the
fixture pt_allpairs_n16_r5).  On a CUDA device the reference routes the same setting through
`_cuda_optimized_exchange` instead (:222-226 -> :260-293: sequential ADJACENT pairs, fp32 probability with
the inverted sign, spin rows and energies swapped) -- a different exchange rule under the same name.  That
rule is available here as the stateless operator `CUDAKernelManager.parallel_tempering_exchange_optimized`
/ `sga_op_pt_exchange` (pinned to operator_n48), not through this class: `ParallelTempering` means the
CPU branch on every device.
"""
import time
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _native as N
from .engine import AnnealEngine
from .exceptions import AnnealingError, ConfigurationError
from .gpu_annealer import fresh_seed
from .ising_model import IsingModel, _device_index
from .result import AnnealingResult
from .spin_dynamics import UpdateRule, rule_code
from .temperature_scheduler import temperature_ladder


@dataclass
class ParallelTemperingConfig:
    n_replicas: int = 8
    n_sweeps: int = 1000
    temp_min: float = 0.1
    temp_max: float = 10.0
    temp_distribution: str = "geometric"
    exchange_interval: int = 10
    exchange_method: str = "nearest_neighbor"
    n_threads: Optional[int] = None  # accepted for compatibility; replicas run on the GPU
    record_interval: int = 10
    random_seed: Optional[int] = None
    # build-specific
    coupling_storage: str = "auto"
    field_cache: str = "auto"             # resident local fields where the problem allows (identical chain)
    device_index: Optional[int] = None
    autotune: Optional[bool] = None       # measured launch geometry (None: for long runs only)


class ParallelTempering:
    def __init__(self, config: ParallelTemperingConfig):
        if config.n_replicas < 2:
            raise ConfigurationError("parallel tempering needs at least 2 replicas")
        if config.exchange_interval <= 0 or config.record_interval <= 0 or config.n_sweeps <= 0:
            raise ConfigurationError("intervals and n_sweeps must be positive")
        self.config = config
        self.temperatures = temperature_ladder(config.n_replicas, config.temp_min,
                                               config.temp_max, config.temp_distribution)
        R = config.n_replicas
        self.exchange_attempts = np.zeros((R - 1,))
        self.exchange_accepts = np.zeros((R - 1,))
        self.energy_histories: List[List[float]] = [[] for _ in range(R)]
        self.temp_histories: List[List[float]] = [[] for _ in range(R)]
        self.slot_acceptance = np.zeros(R)
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.use_cuda = self.device.type == "cuda"

    def run(self, model: IsingModel, update_rule: UpdateRule = UpdateRule.METROPOLIS,
            _replay=None) -> AnnealingResult:
        """`_replay`: parity-test hook, dict(s0, site[ns,R,n], u[ns,R,n], exch_start, exch_u)
        recorded from the reference, replacing every Philox draw."""
        rule = rule_code(update_rule)
        cfg = self.config
        if cfg.exchange_method not in ("nearest_neighbor", "all_pairs"):
            raise ValueError(f"Unknown exchange method: {cfg.exchange_method}")  # reference :212
        all_pairs = cfg.exchange_method == "all_pairs"
        # all_pairs: the gate `np.random.rand() < 0.1` of the reference (:222-232) only selects which
        # pairs are attempted; it is drawn on the host (seeded), the attempts run on the engine
        gate_rng = np.random.RandomState(fresh_seed(cfg.random_seed) & 0x7FFFFFFF)
        pair_list = [(i, j) for i in range(cfg.n_replicas - 1) for j in range(i + 1, cfg.n_replicas)]
        t_start = time.time()
        R, n = cfg.n_replicas, model.n_spins
        temps = np.asarray(self.temperatures, np.float64)
        dev_idx = cfg.device_index if cfg.device_index is not None else _device_index(model.device)
        best_energy, best_configuration = float("inf"), None
        acc_slot, att_slot = np.zeros(R, np.int64), np.zeros(R, np.int64)
        with AnnealEngine(dev_idx) as eng:
            eng.set_field_cache(cfg.field_cache)  # (before the couplings: "on" keeps a sparse matrix dense)
            model.load_into(eng, storage=cfg.coupling_storage)
            eng.set_update_rule(rule)
            eng.init_replicas(R, seed=fresh_seed(cfg.random_seed),
                              s0=None if _replay is None else _replay["s0"])
            eng.set_ladder(temps, 1)
            eng.maybe_autotune(cfg.n_sweeps, cfg.autotune)
            slot_to_rep = np.arange(R, dtype=np.int32)
            acc_prev = np.zeros(R, np.int64)
            rnd = ucur = 0
            sweep = 0
            while sweep < cfg.n_sweeps:
                # advance to the next sweep index that carries an event
                stop = sweep
                while stop < cfg.n_sweeps - 1 and not self._event(stop):
                    stop += 1
                count = stop - sweep + 1
                if _replay is None:
                    eng.sweep(count)
                else:
                    inv = np.argsort(slot_to_rep)
                    site = _replay["site"][sweep:stop + 1][:, inv].transpose(1, 0, 2).reshape(R, -1)
                    u = _replay["u"][sweep:stop + 1][:, inv].transpose(1, 0, 2).reshape(R, -1)
                    eng.sweep(count, site_mode=N.SITE_REPLAY, replay_site=site, replay_u=u)
                exchange_now = stop % cfg.exchange_interval == 0 and stop > 0  # reference :113
                if exchange_now and all_pairs:
                    if _replay is None:
                        gates = gate_rng.rand(len(pair_list)) < 0.1
                        eng.exchange_pairs([p for p, g in zip(pair_list, gates) if g])
                    else:  # the recorded stream: a gate draw per pair, an exchange draw behind an open gate
                        stream, chosen, uu = _replay["np_rand_all"], [], []
                        for p in pair_list:
                            g = stream[ucur]
                            ucur += 1
                            if g < 0.1:
                                chosen.append(p)
                                uu.append(stream[ucur])
                                ucur += 1
                        eng.exchange_pairs(chosen, u=np.asarray(uu, np.float64))
                    rnd += 1
                elif exchange_now:
                    if _replay is None:
                        eng.exchange(count=False)  # enqueued behind the sweeps, no host round trip
                    else:
                        start = int(_replay["exch_start"][rnd])
                        npairs = len(range(start, R - 1, 2))
                        uu = np.zeros(R // 2)
                        uu[:npairs] = _replay["exch_u"][ucur:ucur + npairs]
                        eng.exchange(start=[start], u=uu, count=False)
                        ucur += npairs
                    rnd += 1
                # one synchronisation per event: energies (an exchange moves labels, not energies),
                # acceptance counters, ladder permutation
                en_rep, acc_now, new_map = eng.snapshot()
                # per-slot acceptance bookkeeping (the reference's SpinDynamics stay with slots):
                # these sweeps ran under the permutation in force BEFORE the exchange
                acc_slot += (acc_now - acc_prev)[slot_to_rep]
                att_slot += count * n
                acc_prev = acc_now
                slot_to_rep = new_map
                if stop % cfg.record_interval == 0:  # reference :117-125
                    en = en_rep[slot_to_rep]
                    for i in range(R):
                        self.energy_histories[i].append(float(en[i]))
                        self.temp_histories[i].append(float(temps[i]))
                    i_best = int(np.argmin(en))
                    if en[i_best] < best_energy:
                        best_energy = float(en[i_best])
                        best_configuration = eng.spins(int(slot_to_rep[i_best]))
                sweep = stop + 1
            att, acc = eng.exchange_stats()
            self.exchange_attempts = att[:R - 1].astype(np.float64)
            self.exchange_accepts = acc[:R - 1].astype(np.float64)
            self.final_spins = eng.spins()[slot_to_rep]
        self.slot_acceptance = acc_slot / np.maximum(att_slot, 1)
        total_time = time.time() - t_start
        return AnnealingResult(
            best_configuration=torch.from_numpy(best_configuration.astype(np.float32)),
            best_energy=best_energy, energy_history=self.energy_histories[0],
            temperature_history=self.temp_histories[0],
            acceptance_rate_history=[float(x) for x in self.slot_acceptance],
            total_time=total_time, n_sweeps=cfg.n_sweeps, algorithm="parallel_tempering",
            device=f"cuda:{dev_idx}", random_seed=cfg.random_seed)

    def _event(self, sweep: int) -> bool:
        c = self.config
        return (sweep % c.exchange_interval == 0 and sweep > 0) or sweep % c.record_interval == 0

    def get_exchange_rates(self) -> np.ndarray:
        with np.errstate(divide="ignore", invalid="ignore"):
            r = np.where(self.exchange_attempts > 0, self.exchange_accepts / self.exchange_attempts, 0.0)
        return r

    def get_statistics(self) -> Dict:
        rates = self.get_exchange_rates()
        finals = [h[-1] if h else float("inf") for h in self.energy_histories]
        return {"n_replicas": self.config.n_replicas, "temperatures": self.temperatures,
                "exchange_rates": rates.tolist(), "mean_exchange_rate": float(np.mean(rates)),
                "final_energies": finals, "best_energy": min(finals),
                "energy_range": max(finals) - min(finals),
                "total_exchanges_attempted": self.exchange_attempts.sum(),
                "total_exchanges_accepted": self.exchange_accepts.sum()}

    def __repr__(self) -> str:
        c = self.config
        return (f"ParallelTempering(n_replicas={c.n_replicas}, "
                f"temp_range=[{c.temp_min:.2f}, {c.temp_max:.2f}], n_sweeps={c.n_sweeps})")
