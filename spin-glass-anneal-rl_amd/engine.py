"""AnnealEngine: thin object wrapper over the C ABI handle (include/sga.h).

All arithmetic happens in the HIP kernels behind the handle; this class only marshals
buffers (numpy arrays, or torch tensors whose `data_ptr()` is handed over) and turns status
codes into exceptions.  PyTorch is used for device memory only.
"""
import ctypes as C
import threading
from typing import Optional

import numpy as np

from . import _native as N
from .exceptions import AnnealingError

try:  # torch is plumbing here: device buffers and streams
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_tensor(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


def _producer_done(t):
    """The engine reads device buffers on its own stream: whatever torch still has queued on its
    current stream to produce `t` must have finished first (free when use_stream() shares it)."""
    if t.is_cuda:
        torch.cuda.current_stream(t.device).synchronize()


def _buf(x, np_dtype, torch_dtype_name, ordered=False):
    """Return (pointer, keepalive) for a numpy array / torch tensor / None.  ordered: the consumer runs
    on the stream that produces the tensor (no wait for the producer)."""
    if x is None:
        return None, None
    if _is_tensor(x):
        want = getattr(torch, torch_dtype_name)
        t = x.detach()
        if t.dtype != want or not t.is_contiguous():
            t = t.to(want).contiguous()
        if not ordered:
            _producer_done(t)
        return C.c_void_p(t.data_ptr()), t
    a = np.ascontiguousarray(x, dtype=np_dtype)
    return a.ctypes.data_as(C.c_void_p), a


def probe_read_bandwidth(device_index: int = 0, nbytes: int = 1 << 32, reps: int = 4) -> float:
    """GB/s of a plain streaming read of a fresh device buffer (sga_probe_read_bandwidth)."""
    out = C.c_double(0.0)
    N.check(N.lib().sga_probe_read_bandwidth(int(device_index), int(nbytes), int(reps), C.byref(out)),
            "sga_probe_read_bandwidth")
    return float(out.value)


class _SerialisedLib:
    """Calls on one handle must not overlap (include/sga.h); ctypes releases the GIL during a
    call, so the reference's habit of driving sweeps from a thread pool
    (parallel_tempering.py:199, multi_gpu.py:148) is made safe by one lock per engine."""

    def __init__(self, lib):
        self._lib = lib
        self._lock = threading.RLock()

    def __getattr__(self, name):
        fn = getattr(self._lib, name)

        def call(*args):
            with self._lock:
                return fn(*args)

        return call


class AnnealEngine:
    """One engine per GPU: packed couplings + R replicas resident in HBM."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        self._lib = _SerialisedLib(N.lib())
        N.check(self._lib.sga_create(int(device), C.byref(self._h)), "sga_create")
        self.device = int(device)
        self.stream_handle = 0
        self.n = 0
        self.R = 0
        self.R_global = 0
        self.replica0 = 0
        self.n_ladders = 0
        self.n_models = 1

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.sga_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def use_stream(self, stream_handle: Optional[int]):
        N.check(self._lib.sga_set_stream(self._h, C.c_void_p(stream_handle or 0)), "sga_set_stream")
        self.stream_handle = int(stream_handle or 0)

    def shares_torch_stream(self) -> bool:
        """True when the engine launches on torch's current stream of its device: torch work (an RCCL
        collective) and engine kernels are then ordered by the stream, no host synchronisation needed."""
        return bool(self.stream_handle) and torch is not None and torch.cuda.is_available() and \
            self.stream_handle == torch.cuda.current_stream(self.device).cuda_stream

    def problem_checksum(self) -> int:
        out = C.c_uint64(0)
        N.check(self._lib.sga_problem_checksum(self._h, C.byref(out)), "sga_problem_checksum")
        return int(out.value)

    def geometry(self):
        """(waves per replica, chunks per wave) of the dense sweep kernels."""
        w, c = C.c_int(0), C.c_int(0)
        N.check(self._lib.sga_get_geometry(self._h, C.byref(w), C.byref(c)), "sga_get_geometry")
        return int(w.value), int(c.value)

    def set_tuning(self, waves_per_replica: int = 0, sweeps_per_launch: int = 0):
        N.check(self._lib.sga_set_tuning(self._h, int(waves_per_replica), int(sweeps_per_launch)),
                "sga_set_tuning")

    def set_option(self, key: str, value: int):
        """A form-selection option of this engine by name (sga_set_option; keys in include/sga.h).  No option
        changes a result: every form walks the same chain."""
        N.check(self._lib.sga_set_option(self._h, key.encode(), int(value)), "sga_set_option")

    def set_options(self, options=None, **kw):
        """Several options at once: a dict and / or keywords."""
        for k, v in {**(options or {}), **kw}.items():
            self.set_option(k, v)

    def get_option(self, key: str) -> int:
        out = C.c_int64(0)
        N.check(self._lib.sga_get_option(self._h, key.encode(), C.byref(out)), "sga_get_option")
        return int(out.value)

    def set_csr_storage(self, storage: str = "auto"):
        """Entry storage of the long-row CSR sweep forms: "auto" (one dword per entry where the problem
        allows: integer couplings, |J| <= 127, n < 2^24), "f32" (column + fp32 value), "packed" (required)."""
        code = {"auto": 0, "f32": 1, "packed": 2}[storage]
        N.check(self._lib.sga_set_csr_storage(self._h, code), "sga_set_csr_storage")

    def set_field_cache(self, mode="on"):
        """How sweeps evaluate a proposal (sga_set_field_cache): "off" = one coupling-row read per
        proposal (the reference's get_local_field), "on" / "auto" = resident local fields, a row read
        only on accept (the reference's incremental mode); identical chains."""
        code = {"off": 0, False: 0, "on": 1, True: 1, "auto": 2}[mode]
        N.check(self._lib.sga_set_field_cache(self._h, code), "sga_set_field_cache")

    def autotune(self) -> float:
        """Time every feasible waves-per-replica on the current replicas and keep the fastest
        (dense problems; results are unaffected).  Returns the best kernel ms per sweep."""
        ms = C.c_double(0.0)
        N.check(self._lib.sga_autotune(self._h, C.byref(ms)), "sga_autotune")
        return float(ms.value)

    def autotune_table(self) -> dict:
        """{candidate: kernel ms per sweep} of the last autotune() (dense: "<waves>x<chunks per wave>")."""
        buf = C.create_string_buffer(4096)
        N.check(self._lib.sga_get_autotune_table(self._h, buf, 4096), "sga_get_autotune_table")
        out = {}
        for item in buf.value.decode().split(";"):
            if "=" in item:
                k, v = item.rsplit("=", 1)
                out[k] = float(v)
        return out

    def maybe_autotune(self, n_sweeps: int, setting: Optional[bool] = None) -> bool:
        """Host-loop helper: autotune when asked to (`setting=True`), never when `False`, and by
        default (`None`) only when the run is long enough to pay for the ~30 trial sweeps."""
        if setting is False or self.R <= 0:
            return False
        if setting is None and float(self.n) ** 2 * self.R * n_sweeps < 2e13:
            return False
        if setting is None and "sweep=cached-local-fields" in self.describe():
            return False  # the measured geometry belongs to the row-per-proposal kernels
        self.autotune()
        return True

    # ------------------------------------------------------------------ problem
    def set_dense(self, J, h, storage: str = "auto"):
        sel = {"auto": N.J_AUTO, "f32": N.J_F32, "i8": N.J_I8, "t2": N.J_T2}[storage]
        if _is_tensor(J):
            if J.dim() != 2 or J.shape[0] != J.shape[1]:
                raise AnnealingError("couplings must be a square matrix")
            Jt = J.detach()
            if Jt.dtype != torch.float32:
                Jt = Jt.float()
            if Jt.stride(1) != 1:
                Jt = Jt.contiguous()
            _producer_done(Jt)
            n, ld, jp, keep = Jt.shape[0], Jt.stride(0), C.c_void_p(Jt.data_ptr()), Jt
        else:
            Ja = np.ascontiguousarray(J, dtype=np.float32)
            if Ja.ndim != 2 or Ja.shape[0] != Ja.shape[1]:
                raise AnnealingError("couplings must be a square matrix")
            n, ld, jp, keep = Ja.shape[0], Ja.shape[1], Ja.ctypes.data_as(C.c_void_p), Ja
        hp, hk = _buf(h, np.float32, "float32")
        if (hk.numel() if _is_tensor(hk) else hk.size) != n:
            raise AnnealingError("external fields must have n entries")
        N.check(self._lib.sga_set_dense(self._h, jp, int(ld), hp, int(n), sel), "sga_set_dense")
        del keep
        self.n, self.R, self.n_models = n, 0, 1

    def set_dense_batch(self, J, h, storage: str = "auto"):
        """Many independent models of one size: J [M, n, n], h [M, n] (numpy or torch)."""
        sel = {"auto": N.J_AUTO, "f32": N.J_F32, "i8": N.J_I8, "t2": N.J_T2}[storage]
        if _is_tensor(J):
            Jt = J.detach().float().contiguous()
            _producer_done(Jt)
            M, n = Jt.shape[0], Jt.shape[1]
            jp, keep = C.c_void_p(Jt.data_ptr()), Jt
        else:
            Ja = np.ascontiguousarray(J, dtype=np.float32)
            M, n = Ja.shape[0], Ja.shape[1]
            jp, keep = Ja.ctypes.data_as(C.c_void_p), Ja
        if keep.ndim != 3 or keep.shape[2] != n:
            raise AnnealingError("batched couplings must be [M, n, n]")
        hp, hk = _buf(h, np.float32, "float32")
        if (hk.numel() if _is_tensor(hk) else hk.size) != M * n:
            raise AnnealingError("batched fields must be [M, n]")
        N.check(self._lib.sga_set_dense_batch(self._h, jp, int(n), hp, int(n), int(M), sel),
                "sga_set_dense_batch")
        del keep
        self.n, self.R, self.n_models = n, 0, M

    def set_csr(self, rowptr, colidx, val, h):
        """CSR couplings (both triangles).  int64 `rowptr` (numpy / torch) is passed through as
        64-bit extents -- required once nnz >= 2^31 -- anything else is taken as int32."""
        wide = getattr(rowptr, "dtype", None) in ((np.dtype(np.int64),) + ((torch.int64,) if torch is not None else ()))
        rp, k1 = _buf(rowptr, np.int64, "int64") if wide else _buf(rowptr, np.int32, "int32")
        ci, k2 = _buf(colidx, np.int32, "int32")
        vp, k3 = _buf(val, np.float32, "float32")
        hp, k4 = _buf(h, np.float32, "float32")
        n = (k1.numel() if _is_tensor(k1) else k1.size) - 1
        nnz = k2.numel() if _is_tensor(k2) else k2.size
        fn, name = (self._lib.sga_set_csr64, "sga_set_csr64") if wide else (self._lib.sga_set_csr, "sga_set_csr")
        N.check(fn(self._h, rp, ci, vp, hp, int(n), int(nnz)), name)
        self.n, self.R = n, 0

    def set_tsp(self, dist, city_visit: float, position_fill: float, h):
        """TSP-structured couplings, never stored (see sga_set_tsp): dist [n, n] float32 (numpy or
        torch), the two penalty weights as the encoder scaled them, and the fields h [n * n]
        (`encoders.tsp_structure` returns all four)."""
        if _is_tensor(dist):
            dt = dist.detach().float()
            if dt.dim() != 2 or dt.shape[0] != dt.shape[1]:
                raise AnnealingError("the distance matrix must be square")
            if dt.stride(1) != 1:
                dt = dt.contiguous()
            _producer_done(dt)
            n, ld, dp, keep = dt.shape[0], dt.stride(0), C.c_void_p(dt.data_ptr()), dt
        else:
            da = np.ascontiguousarray(dist, dtype=np.float32)
            if da.ndim != 2 or da.shape[0] != da.shape[1]:
                raise AnnealingError("the distance matrix must be square")
            n, ld, dp, keep = da.shape[0], da.shape[1], da.ctypes.data_as(C.c_void_p), da
        hp, hk = _buf(h, np.float32, "float32")
        if (hk.numel() if _is_tensor(hk) else hk.size) != n * n:
            raise AnnealingError("the fields must have n_cities^2 entries")
        N.check(self._lib.sga_set_tsp(self._h, dp, int(ld), int(n), float(city_visit), float(position_fill),
                                      hp), "sga_set_tsp")
        del keep
        self.n, self.R, self.n_models = n * n, 0, 1

    # ------------------------------------------------------------------ replicas
    def init_replicas(self, R: int, seed: int = 0, s0=None, R_global: Optional[int] = None,
                      replica0: int = 0):
        Rg = R if R_global is None else int(R_global)
        sp, keep = _buf(s0, np.int8, "int8")
        if keep is not None:
            cnt = keep.numel() if _is_tensor(keep) else keep.size
            if cnt != R * self.n:
                raise AnnealingError("s0 must be [R, n] int8")
        self.R = 0  # a failed call leaves the engine without replicas (include/sga.h)
        N.check(self._lib.sga_init_replicas(self._h, int(R), Rg, int(replica0),
                                            int(seed) & 0xFFFFFFFFFFFFFFFF, sp),
                "sga_init_replicas")
        self.R, self.R_global, self.replica0, self.n_ladders = int(R), Rg, int(replica0), 0

    def set_temperatures(self, T):
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(T, np.float64), (self.R,)))
        N.check(self._lib.sga_set_temperatures(self._h, t.ctypes.data_as(C.c_void_p)),
                "sga_set_temperatures")

    def set_ladder(self, slot_temps, n_ladders: int = 1):
        t = np.ascontiguousarray(slot_temps, dtype=np.float64)
        if t.size != self.R_global:
            raise AnnealingError("slot_temps must have R_global entries")
        N.check(self._lib.sga_set_ladder(self._h, t.ctypes.data_as(C.c_void_p), int(n_ladders)),
                "sga_set_ladder")
        self.n_ladders = int(n_ladders)

    # ------------------------------------------------------------------ hot path
    def sweep(self, n_sweeps: int = 1, site_mode: int = N.SITE_RANDOM, arith: int = N.ARITH_F64,
              sched=None, replay_site=None, replay_u=None, energy_trace: bool = False,
              trace: bool = False):
        """n_sweeps Metropolis sweeps of every replica.

        sched: None | [n_sweeps] (shared by all replicas) | [n_sweeps, R] temperatures.
        Returns {"energy_trace": [n_sweeps, R] | None, "accept_trace", "dE_trace"}.
        """
        n_sweeps = int(n_sweeps)
        sp, ss, rs, keep_s = None, 0, 0, None
        if sched is not None:
            keep_s = np.ascontiguousarray(sched, dtype=np.float64)
            if keep_s.ndim == 1 and keep_s.shape[0] == n_sweeps:
                ss, rs = 1, 0
            elif keep_s.shape == (n_sweeps, self.R):
                ss, rs = self.R, 1
            else:
                raise AnnealingError("sched must be [n_sweeps] or [n_sweeps, R]")
            sp = keep_s.ctypes.data_as(C.c_void_p)
        per = n_sweeps * self.n
        rsp, k1 = _buf(replay_site, np.int32, "int32")
        rup, k2 = _buf(replay_u, np.float32, "float32")
        for k in (k1, k2):
            if k is not None and (k.numel() if _is_tensor(k) else k.size) != self.R * per:
                raise AnnealingError("replay arrays must be [R, n_sweeps*n]")
        et = np.zeros((n_sweeps, self.R), np.float64) if energy_trace else None
        at = np.zeros((self.R, per), np.uint8) if trace else None
        dt = np.zeros((self.R, per), np.float64) if trace else None
        ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
        N.check(self._lib.sga_sweep(self._h, n_sweeps, int(site_mode), int(arith), sp, ss, rs, rsp,
                                    rup, ptr(et), ptr(at), ptr(dt)), "sga_sweep")
        return {"energy_trace": et, "accept_trace": at, "dE_trace": dt}

    # single-site operators (the reference's per-spin API)
    def local_fields(self, r: int, sites) -> np.ndarray:
        idx = np.ascontiguousarray(np.atleast_1d(sites), dtype=np.int32)
        out = np.zeros(idx.size, np.float64)
        N.check(self._lib.sga_local_fields(self._h, int(r), idx.ctypes.data_as(C.c_void_p),
                                           int(idx.size), out.ctypes.data_as(C.c_void_p)),
                "sga_local_fields")
        return out

    def flip(self, r: int, site: int) -> float:
        d = C.c_double(0.0)
        N.check(self._lib.sga_flip(self._h, int(r), int(site), C.byref(d)), "sga_flip")
        return float(d.value)

    def update(self, r: int, site: int, T: float, u: float, arith: int = N.ARITH_F64):
        acc, d = C.c_int(0), C.c_double(0.0)
        N.check(self._lib.sga_update(self._h, int(r), int(site), float(T), float(u), int(arith),
                                     C.byref(acc), C.byref(d)), "sga_update")
        return bool(acc.value), float(d.value)

    def set_update_rule(self, rule: int):
        N.check(self._lib.sga_set_update_rule(self._h, int(rule)), "sga_set_update_rule")

    def set_wolff_replay(self, u):
        """Recorded uniforms for the Wolff rule's candidate bonds, [R, capacity] float32, consumed in
        draw order by the following Wolff sweeps (parity tests); None: back to Philox."""
        if u is None:
            N.check(self._lib.sga_set_wolff_replay(self._h, None, 0), "sga_set_wolff_replay")
            return
        a = np.ascontiguousarray(u, dtype=np.float32).reshape(self.R, -1)
        N.check(self._lib.sga_set_wolff_replay(self._h, a.ctypes.data_as(C.c_void_p), int(a.shape[1])),
                "sga_set_wolff_replay")

    def recompute_energies(self):
        N.check(self._lib.sga_recompute_energies(self._h), "sga_recompute_energies")

    def exchange(self, energies_global=None, start=None, u=None, count: bool = True):
        """One exchange round.  With count=False the call only enqueues the kernel (no host
        synchronisation) and returns None -- use it inside sweep / exchange loops."""
        ep, k1 = _buf(energies_global, np.float64, "float64",
                      ordered=_is_tensor(energies_global) and energies_global.is_cuda and self.shares_torch_stream())
        stp, k2 = _buf(None if start is None else np.atleast_1d(start), np.int32, "int32")
        up, k3 = _buf(u, np.float64, "float64")
        if not count:
            N.check(self._lib.sga_exchange(self._h, ep, stp, up, None), "sga_exchange")
            self._pending = (k1, k2, k3)  # converted device tensors stay alive while the kernel is queued
            return None
        out = C.c_int(0)
        N.check(self._lib.sga_exchange(self._h, ep, stp, up, C.byref(out)), "sga_exchange")
        return int(out.value)

    def exchange_pairs(self, pairs, u=None, energies_global=None) -> int:
        """Exchange attempts over an ordered list of slot pairs [(i, j), ...], each seeing the
        swaps before it (the reference's exchange_method="all_pairs").  Returns the accepted count."""
        pr = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        ep, k1 = _buf(energies_global, np.float64, "float64")
        up, k2 = _buf(u, np.float64, "float64")
        if k2 is not None and (k2.numel() if _is_tensor(k2) else k2.size) != len(pr):
            raise AnnealingError("u must have one entry per pair")
        out = C.c_int(0)
        N.check(self._lib.sga_exchange_pairs(self._h, ep, pr.ctypes.data_as(C.c_void_p), up, len(pr),
                                             C.byref(out)), "sga_exchange_pairs")
        return int(out.value)

    # ------------------------------------------------------------------ state access
    def energies(self) -> np.ndarray:
        out = np.zeros(self.R, np.float64)
        N.check(self._lib.sga_get_energies(self._h, out.ctypes.data_as(C.c_void_p)),
                "sga_get_energies")
        return out

    def energies_into(self, tensor, stream_ordered: bool = False):
        """Copy the local energies into a float64 torch tensor (device-to-device on the GPU).  With
        stream_ordered=True (device tensors only) the copy is only enqueued on the engine's stream."""
        if tensor.dtype != torch.float64 or tensor.numel() != self.R or not tensor.is_contiguous():
            raise AnnealingError("need a contiguous float64 tensor of R entries")
        if stream_ordered and tensor.is_cuda:
            N.check(self._lib.sga_get_energies_async(self._h, C.c_void_p(tensor.data_ptr())),
                    "sga_get_energies_async")
        else:
            N.check(self._lib.sga_get_energies(self._h, C.c_void_p(tensor.data_ptr())),
                    "sga_get_energies")
        return tensor

    def temperatures(self) -> np.ndarray:
        out = np.zeros(self.R, np.float64)
        N.check(self._lib.sga_get_temperatures(self._h, out.ctypes.data_as(C.c_void_p)),
                "sga_get_temperatures")
        return out

    def spins(self, r: Optional[int] = None) -> np.ndarray:
        if r is None:
            out = np.zeros((self.R, self.n), np.int8)
            N.check(self._lib.sga_get_spins(self._h, -1, out.ctypes.data_as(C.c_void_p)),
                    "sga_get_spins")
            return out
        out = np.zeros(self.n, np.int8)
        N.check(self._lib.sga_get_spins(self._h, int(r), out.ctypes.data_as(C.c_void_p)),
                "sga_get_spins")
        return out

    def set_spins(self, r: int, s):
        a = np.ascontiguousarray(s, dtype=np.int8)
        if a.size != self.n:
            raise AnnealingError("spins must have n entries")
        N.check(self._lib.sga_set_spins(self._h, int(r), a.ctypes.data_as(C.c_void_p)),
                "sga_set_spins")

    def best(self, r: Optional[int] = None, with_spins: bool = True):
        """(energy, spins int8 [n] | None, local replica index)."""
        e, idx = C.c_double(0.0), C.c_int(0)
        s = np.zeros(self.n, np.int8) if with_spins else None
        N.check(self._lib.sga_get_best(self._h, -1 if r is None else int(r), C.byref(e),
                                       None if s is None else s.ctypes.data_as(C.c_void_p),
                                       C.byref(idx)), "sga_get_best")
        return float(e.value), s, int(idx.value)

    def reset_best(self):
        N.check(self._lib.sga_reset_best(self._h), "sga_reset_best")

    def stats(self):
        acc, att = np.zeros(self.R, np.int64), np.zeros(self.R, np.int64)
        N.check(self._lib.sga_get_stats(self._h, acc.ctypes.data_as(C.c_void_p),
                                        att.ctypes.data_as(C.c_void_p)), "sga_get_stats")
        return acc, att

    def snapshot(self, slots: bool = True):
        """(energies, accepted counters, ladder permutation | None) with one synchronisation."""
        en, acc = np.zeros(self.R, np.float64), np.zeros(self.R, np.int64)
        sm = np.zeros(self.R_global, np.int32) if (slots and self.n_ladders > 0) else None
        N.check(self._lib.sga_snapshot(self._h, en.ctypes.data_as(C.c_void_p), acc.ctypes.data_as(C.c_void_p),
                                       None if sm is None else sm.ctypes.data_as(C.c_void_p)), "sga_snapshot")
        return en, acc, sm

    def slot_map(self) -> np.ndarray:
        out = np.zeros(self.R_global, np.int32)
        N.check(self._lib.sga_get_slot_map(self._h, out.ctypes.data_as(C.c_void_p)),
                "sga_get_slot_map")
        return out

    def route_query(self):
        """The query the engine itself poses to the form selection (sga_get_route_query)."""
        q = N.RouteQuery()
        N.check(self._lib.sga_get_route_query(self._h, C.byref(q)), "sga_get_route_query")
        return q

    def explain_route(self) -> str:
        """What csrc/sga_route.cpp answers for this engine's problem, replicas and options."""
        return N.explain_route(self.route_query())

    def last_kernel(self) -> str:
        """The kernel instantiation THIS engine's last sweep launched (sga_get_last_kernel; per engine, so that two
        engines on two threads do not see each other's)."""
        buf = C.create_string_buffer(512)
        N.check(self._lib.sga_get_last_kernel(self._h, buf, 512), "sga_get_last_kernel")
        return buf.value.decode()

    def exchange_stats(self):
        a, b = np.zeros(self.R_global, np.int64), np.zeros(self.R_global, np.int64)
        N.check(self._lib.sga_get_exchange_stats(self._h, a.ctypes.data_as(C.c_void_p),
                                                 b.ctypes.data_as(C.c_void_p)),
                "sga_get_exchange_stats")
        return a, b

    def counters(self):
        s, r = C.c_uint32(0), C.c_uint32(0)
        N.check(self._lib.sga_get_sweep_counter(self._h, C.byref(s), C.byref(r)))
        return int(s.value), int(r.value)

    def set_seed(self, seed: int):
        N.check(self._lib.sga_set_seed(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF))

    def set_counters(self, sweeps_done: int, exchange_rounds: int):
        N.check(self._lib.sga_set_sweep_counter(self._h, int(sweeps_done), int(exchange_rounds)))

    # ------------------------------------------------------------------ checkpoint / resume
    def export_state(self) -> bytes:
        """Everything needed to continue this run bit-exactly (see sga_export_state)."""
        need = C.c_uint64(0)
        N.check(self._lib.sga_export_state(self._h, None, 0, C.byref(need)), "sga_export_state")
        buf = (C.c_ubyte * need.value)()
        N.check(self._lib.sga_export_state(self._h, buf, need.value, None), "sga_export_state")
        return bytes(buf)

    def import_state(self, blob: bytes):
        buf = (C.c_ubyte * len(blob)).from_buffer_copy(blob)
        N.check(self._lib.sga_import_state(self._h, buf, len(blob)), "sga_import_state")

    # ------------------------------------------------------------------ measurement
    def enable_timing(self, on: bool = True):
        N.check(self._lib.sga_enable_timing(self._h, 1 if on else 0))

    def kernel_time(self, reset: bool = True):
        """(launches, total_ms) of the sweep-kernel launches since the last reset."""
        n, ms = C.c_int64(0), C.c_double(0.0)
        N.check(self._lib.sga_get_kernel_time(self._h, C.byref(n), C.byref(ms), 1 if reset else 0))
        return int(n.value), float(ms.value)

    def describe(self) -> str:
        buf = C.create_string_buffer(512)
        N.check(self._lib.sga_describe(self._h, buf, 512))
        return buf.value.decode()


def option_names():
    """Every key sga_set_option accepts."""
    names, buf = [], C.create_string_buffer(64)
    while N.lib().sga_option_name(len(names), buf, 64) == N.OK:
        names.append(buf.value.decode())
    return names


def last_kernel() -> str:
    """The kernel instantiation this thread's last sweep launched (sga_last_kernel)."""
    buf = C.create_string_buffer(256)
    N.check(N.lib().sga_last_kernel(buf, 256), "sga_last_kernel")
    return buf.value.decode()


def op_pt_exchange(device: int, spins, energies, temps, u=None, seed: int = 0, round_: int = 0):
    """Operator-form PT exchange on fp32 buffers, in place (reference cuda_kernels.py:415-443)."""
    lib = N.lib()
    sp, k1 = _buf(spins, np.float32, "float32")
    ep, k2 = _buf(energies, np.float32, "float32")
    tp, k3 = _buf(temps, np.float32, "float32")
    up, k4 = _buf(u, np.float32, "float32")
    if _is_tensor(spins):
        if k1.data_ptr() != spins.data_ptr() or k2.data_ptr() != energies.data_ptr():
            raise AnnealingError("spins/energies must be contiguous float32 (updated in place)")
        R, n = spins.shape
    else:
        if k1 is not spins or k2 is not energies:
            raise AnnealingError("spins/energies must be contiguous float32 (updated in place)")
        R, n = k1.shape
    out = C.c_int(0)
    N.check(lib.sga_op_pt_exchange(int(device), sp, ep, tp, up, int(seed) & 0xFFFFFFFFFFFFFFFF,
                                   int(round_), int(R), int(n), C.byref(out)), "sga_op_pt_exchange")
    return int(out.value)
