"""ctypes binding of the C ABI in include/sga.h (csrc/libsga.so).

There is no CPU fallback: if the library is missing or no MI355X is visible the call fails
with DeviceError.  `build()` compiles the library in-tree with hipcc (gfx950).
"""
import ctypes as C
import os
import subprocess

from .exceptions import AnnealingError, DeviceError, ResourceError

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
_SO = os.environ.get("SGA_LIBRARY_PATH") or os.path.join(_CSRC, "libsga.so")  # env: A/B builds

OK, ERR_INVALID, ERR_DEVICE, ERR_MEMORY, ERR_UNSUPPORTED = 0, -1, -2, -3, -4
J_AUTO, J_F32, J_I8, J_T2 = 0, 1, 2, 3
SITE_RANDOM, SITE_SEQUENTIAL, SITE_REPLAY = 0, 1, 2
ARITH_F64, ARITH_F32 = 0, 1
RULE_METROPOLIS, RULE_GLAUBER, RULE_HEAT_BATH, RULE_WOLFF = 0, 1, 2, 3

_p, _i, _i64, _u64, _u32, _d = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_uint32, C.c_double

ROUTE_DENSE, ROUTE_CSR, ROUTE_TSP = 0, 1, 2
ROUTE_MAX_OPTS = 32


class RouteQuery(C.Structure):
    """sga_route_query (include/sga.h): a problem's traits, replica count, tuning and options -- what the form
    selection (csrc/sga_route.cpp, no device call) is a pure function of."""
    _fields_ = [(k, C.c_int32) for k in ("kind", "n", "n_models", "R_local", "cus", "tune_waves", "field_cache", "storage",
                                         "acc", "table_m", "table_scale", "clf_ok", "clf_bits", "clf_scale", "from_dense")] + \
               [("nnz", C.c_int64), ("max_row_len", C.c_int64), ("layout_entries", C.c_int64)] + \
               [(k, C.c_int32) for k in ("slotted", "rowptr32", "packed_ok", "n_cities", "sstride", "reserved_")] + \
               [("ldj", C.c_int64), ("opt", C.c_int64 * ROUTE_MAX_OPTS)]


# every symbol include/sga.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("sga_route_query_init", _i, [C.POINTER(RouteQuery)]),
    ("sga_explain_route", _i, [C.POINTER(RouteQuery), C.c_char_p, _i]),
    ("sga_get_route_query", _i, [_p, C.POINTER(RouteQuery)]),
    ("sga_get_last_kernel", _i, [_p, C.c_char_p, _i]),
    ("sga_get_autotune_table", _i, [_p, C.c_char_p, _i]),
    ("sga_create", _i, [_i, C.POINTER(_p)]),
    ("sga_destroy", None, [_p]),
    ("sga_last_error", C.c_char_p, []),
    ("sga_version", _i, []),
    ("sga_set_stream", _i, [_p, _p]),
    ("sga_set_dense", _i, [_p, _p, _i64, _p, _i, _i]),
    ("sga_set_dense_batch", _i, [_p, _p, _i64, _p, _i, _i, _i]),
    ("sga_set_csr", _i, [_p, _p, _p, _p, _p, _i, _i64]),
    ("sga_set_csr64", _i, [_p, _p, _p, _p, _p, _i, _i64]),
    ("sga_set_tsp", _i, [_p, _p, _i64, _i, C.c_float, C.c_float, _p]),
    ("sga_init_replicas", _i, [_p, _i, _i, _i, _u64, _p]),
    ("sga_set_temperatures", _i, [_p, _p]),
    ("sga_set_ladder", _i, [_p, _p, _i]),
    ("sga_sweep", _i, [_p, _i, _i, _i, _p, _i64, _i64, _p, _p, _p, _p, _p]),
    ("sga_local_fields", _i, [_p, _i, _p, _i, _p]),
    ("sga_flip", _i, [_p, _i, _i, C.POINTER(_d)]),
    ("sga_update", _i, [_p, _i, _i, _d, C.c_float, _i, C.POINTER(_i), C.POINTER(_d)]),
    ("sga_set_update_rule", _i, [_p, _i]),
    ("sga_set_wolff_replay", _i, [_p, _p, _i64]),
    ("sga_recompute_energies", _i, [_p]),
    ("sga_exchange", _i, [_p, _p, _p, _p, C.POINTER(_i)]),
    ("sga_exchange_pairs", _i, [_p, _p, _p, _p, _i, C.POINTER(_i)]),
    ("sga_op_pt_exchange", _i, [_i, _p, _p, _p, _p, _u64, _u32, _i, _i, C.POINTER(_i)]),
    ("sga_get_energies", _i, [_p, _p]),
    ("sga_get_energies_async", _i, [_p, _p]),
    ("sga_get_temperatures", _i, [_p, _p]),
    ("sga_get_spins", _i, [_p, _i, _p]),
    ("sga_set_spins", _i, [_p, _i, _p]),
    ("sga_get_best", _i, [_p, _i, C.POINTER(_d), _p, C.POINTER(_i)]),
    ("sga_reset_best", _i, [_p]),
    ("sga_get_stats", _i, [_p, _p, _p]),
    ("sga_get_slot_map", _i, [_p, _p]),
    ("sga_get_exchange_stats", _i, [_p, _p, _p]),
    ("sga_snapshot", _i, [_p, _p, _p, _p]),
    ("sga_set_seed", _i, [_p, _u64]),
    ("sga_get_sweep_counter", _i, [_p, C.POINTER(_u32), C.POINTER(_u32)]),
    ("sga_set_sweep_counter", _i, [_p, _u32, _u32]),
    ("sga_export_state", _i, [_p, _p, _u64, C.POINTER(_u64)]),
    ("sga_import_state", _i, [_p, _p, _u64]),
    ("sga_enable_timing", _i, [_p, _i]),
    ("sga_get_kernel_time", _i, [_p, C.POINTER(_i64), C.POINTER(_d), _i]),
    ("sga_describe", _i, [_p, C.c_char_p, _i]),
    ("sga_problem_checksum", _i, [_p, C.POINTER(_u64)]),
    ("sga_get_geometry", _i, [_p, C.POINTER(_i), C.POINTER(_i)]),
    ("sga_last_kernel", _i, [C.c_char_p, _i]),
    ("sga_set_csr_storage", _i, [_p, _i]),
    ("sga_set_field_cache", _i, [_p, _i]),
    ("sga_set_option", _i, [_p, C.c_char_p, _i64]),
    ("sga_get_option", _i, [_p, C.c_char_p, C.POINTER(_i64)]),
    ("sga_option_name", _i, [_i, C.c_char_p, _i]),
    ("sga_set_tuning", _i, [_p, _i, _i]),
    ("sga_autotune", _i, [_p, C.POINTER(_d)]),
    ("sga_probe_read_bandwidth", _i, [_i, _i64, _i, C.POINTER(_d)]),
]

_lib = None


def library_path() -> str:
    return _SO


def build(force: bool = False, jobs: int = 8) -> str:
    """Compile csrc/*.hip into libsga.so for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", _CSRC, f"-j{jobs}"] + (["-B"] if force else [])
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise DeviceError("building libsga.so failed", {"stderr": proc.stderr[-2000:]})
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise DeviceError(
                "libsga.so is not built (run __graft_entry__.build() or `make -C "
                f"{_CSRC}`); the engine has no CPU fallback")
        try:
            L = C.CDLL(_SO)
        except OSError as exc:  # missing ROCm runtime etc.
            raise DeviceError(f"cannot load {_SO}: {exc}") from exc
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    msg = lib().sga_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str = "") -> None:
    if rc == OK:
        return
    msg = f"{what}: {last_error()}" if what else last_error()
    if rc == ERR_DEVICE:
        raise DeviceError(msg, {"code": rc})
    if rc == ERR_MEMORY:
        raise ResourceError(msg, {"code": rc})
    raise AnnealingError(msg, {"code": rc})


def option_names():
    """The option keys in sga_option_name order (= the index into sga_route_query.opt)."""
    names, buf, i = [], C.create_string_buffer(64), 0
    while lib().sga_option_name(i, buf, 64) == OK:
        names.append(buf.value.decode())
        i += 1
    return names


def route_query(**fields) -> RouteQuery:
    """A sga_route_query with the defaults of sga_route_query_init (no environment), then `fields`; `options` = {key: value}."""
    q = RouteQuery()
    check(lib().sga_route_query_init(C.byref(q)), "sga_route_query_init")
    opts = fields.pop("options", None) or {}
    names = option_names()
    for k, v in opts.items():
        q.opt[names.index(k)] = int(v)
    for k, v in fields.items():
        if k not in dict(RouteQuery._fields_):
            raise AnnealingError(f"sga_route_query has no field {k!r}")
        setattr(q, k, int(v))
    return q


def explain_route(q: RouteQuery) -> str:
    """The form selection's answer for q, one line (sga_explain_route): works without a GPU."""
    buf = C.create_string_buffer(1024)
    check(lib().sga_explain_route(C.byref(q), buf, 1024), "sga_explain_route")
    return buf.value.decode()
