"""CPU tests of the host layer: schedules, result record, configs, model container, and the
C-ABI contract (libsga.so loads and exports every symbol include/sga.h declares; without a GPU
the product fails loudly instead of falling back)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import spin_glass_anneal_rl_amd as sg
from spin_glass_anneal_rl_amd.gpu_annealer import check_convergence
from spin_glass_anneal_rl_amd.ising_model import coo_to_csr
from spin_glass_anneal_rl_amd.scheduler import beta_ladder
from conftest import ROOT, load_golden


# ----------------------------------------------------------------------------- C ABI
def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sga.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sga_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    declared = _declared_symbols()
    assert len(declared) >= 30
    lib = ctypes.CDLL(sg._native.library_path())
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/sga.h but not exported"
    bound = sorted(n for n, _, _ in sg._native.SYMBOLS)
    assert bound == declared, "ctypes binding table and header disagree"
    assert sg._native.lib().sga_version() >= 100


def test_engine_options_are_documented_and_the_environment_is_read_in_one_place():
    """sga_set_option: the keys the library accepts are exactly the keys include/sga.h documents, and the
    library consults the environment at ONE site (the defaults, in sga_create)."""
    import glob
    import re
    from spin_glass_anneal_rl_amd.engine import option_names
    names = option_names()          # sga_option_name needs neither an engine nor a GPU
    assert len(names) >= 14 and len(set(names)) == len(names)
    text = open(os.path.join(ROOT, "include", "sga.h")).read()
    doc = re.search(r"/\* Form-selection options of ONE engine.*?\*/", text, re.S).group(0)
    documented = re.findall(r'^ \*   "(\w+)"', doc, re.M)
    assert sorted(documented) == sorted(names), (sorted(documented), sorted(names))
    csrc = os.path.join(ROOT, "spin-glass-anneal-rl_amd", "csrc")
    sites = []
    for path in sorted(glob.glob(os.path.join(csrc, "*.cpp")) + glob.glob(os.path.join(csrc, "*.h")) +
                       glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.inc"))):
        for no, line in enumerate(open(path), 1):
            if "getenv(" in line.split("//")[0]:
                sites.append(f"{os.path.basename(path)}:{no}")
    assert len(sites) == 1 and sites[0].startswith("sga_engine.cpp"), sites


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_means_device_error_not_fallback():
    with pytest.raises(sg.DeviceError):
        sg.AnnealEngine(0)
    m = sg.IsingModel(sg.IsingModelConfig(n_spins=4, use_sparse=False))
    with pytest.raises(sg.DeviceError):
        m.compute_energy()
    with pytest.raises(sg.DeviceError):
        sg.GPUAnnealer(sg.GPUAnnealerConfig(n_sweeps=2)).anneal(m)
    with pytest.raises(sg.DeviceError):
        sg.ParallelTempering(sg.ParallelTemperingConfig(n_sweeps=2)).run(m)


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "spin-glass-anneal-rl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in src and "sg_oracle" not in src, f


# ----------------------------------------------------------------------------- schedules
PLAIN = ["linear", "exponential", "geometric", "logarithmic", "power_law", "fast", "boltzmann"]


@pytest.mark.parametrize("kind", PLAIN)
def test_schedule_tables_equal_reference(kind):
    g = load_golden("schedules")
    s = sg.TemperatureScheduler.create_schedule(sg.ScheduleType(kind), 10.0, 0.01, 1000)
    got = np.asarray([s.get_temperature(k) for k in range(1200)])
    assert np.array_equal(got, g[kind])
    assert np.array_equal(s.table(5, 7), g[kind][5:12])


def test_adaptive_schedule_equals_reference():
    g = load_golden("schedules")
    s = sg.TemperatureScheduler.create_schedule(sg.ScheduleType.ADAPTIVE, 10.0, 0.01, 1000)
    got = [s.update(k, acceptance_rate=float(g["adaptive_acc"][k])) for k in range(400)]
    assert np.array_equal(np.asarray(got), g["adaptive"])
    assert len(s.temperature_history) == 401


def test_custom_schedule_and_factory_errors():
    s = sg.TemperatureScheduler.create_schedule(sg.ScheduleType.CUSTOM, 5.0, 0.5, 10,
                                                custom_func=lambda k: 5.0 - k)
    assert [s.get_temperature(k) for k in (0, 4, 9)] == [5.0, 1.0, 0.5]
    with pytest.raises(ValueError):
        sg.TemperatureScheduler.create_schedule(sg.ScheduleType.CUSTOM, 5.0, 0.5, 10)
    assert "geometric" in sg.TemperatureScheduler.get_available_schedules()
    assert sg.TemperatureScheduler.recommend_schedule(100, 1000)[0] == sg.ScheduleType.GEOMETRIC


def test_ladders():
    g = load_golden("pt_c1_n64_r8")
    assert np.array_equal(np.asarray(sg.temperature_ladder(8, 0.1, 10.0)), g["temperatures"])
    g2 = load_golden("pt_small_n16_r4")
    assert np.array_equal(np.asarray(sg.temperature_ladder(4, 0.5, 5.0, "linear")),
                          g2["temperatures"])
    assert np.allclose(sg.temperature_ladder(5, 0.1, 10.0, "exponential"),
                       sg.temperature_ladder(5, 0.1, 10.0, "geometric"))
    with pytest.raises(ValueError):
        sg.temperature_ladder(4, 0.1, 1.0, "bogus")
    b = beta_ladder(6, 0.1, 10.0, "geometric")
    assert b[0] == pytest.approx(0.1) and b[-1] == pytest.approx(10.0) and np.all(np.diff(b) > 0)


# ----------------------------------------------------------------------------- result / configs
def _result(**kw):
    base = dict(best_configuration=torch.ones(4), best_energy=-3.0,
                energy_history=[1.0, -1.0, -3.0], temperature_history=[2.0, 1.0, 0.5],
                acceptance_rate_history=[0.0, 0.5, 0.25], total_time=0.1, n_sweeps=3)
    base.update(kw)
    return sg.AnnealingResult(**base)


def test_result_derived_fields_and_validation(tmp_path):
    r = _result()
    assert r.final_temperature == 0.5 and r.final_acceptance_rate == 0.25
    assert r.energy_std == pytest.approx(np.std([1.0, -1.0, -3.0]))
    assert r.get_summary()["best_energy"] == -3.0
    for bad in (dict(best_energy=float("nan")), dict(n_sweeps=0), dict(total_time=-1.0),
                dict(energy_history=[0.0, float("inf")])):
        with pytest.raises(ValueError):
            _result(**bad)
    with pytest.raises(TypeError):
        _result(best_configuration=[1, 2])
    flat = _result(energy_history=[-3.0] * 40, temperature_history=[1.0] * 40,
                   acceptance_rate_history=[0.1] * 40)
    assert flat.convergence_sweep == 0
    path = str(tmp_path / "res.npz")
    r2 = _result(random_seed=7)
    r2.save(path)
    back = sg.AnnealingResult.load(path)
    assert back.best_energy == r2.best_energy and back.energy_history == r2.energy_history
    assert torch.equal(back.best_configuration, r2.best_configuration) and back.random_seed == 7


def test_convergence_rule_matches_reference_run():
    g = load_golden("sa_default_n64")  # the reference stopped exactly at its 50th record
    hist = list(g["energy_history"])
    assert check_convergence(hist, 1e-8)
    assert not check_convergence(hist[:-1], 1e-8)
    assert not check_convergence([1.0] * 49, 1e-8)


def test_config_validation():
    assert sg.GPUAnnealerConfig().schedule_params == {"alpha": 0.95}
    with pytest.raises(sg.ConfigurationError):
        sg.GPUAnnealerConfig(n_sweeps=0)
    with pytest.raises(sg.ConfigurationError):
        sg.GPUAnnealerConfig(site_order="zigzag")
    with pytest.raises(sg.ConfigurationError):
        sg.ParallelTempering(sg.ParallelTemperingConfig(n_replicas=1))
    with pytest.raises(ValueError):
        sg.MultiGPUConfig(gpu_ids=[])
    with pytest.raises(ValueError):
        sg.MultiGPUConfig(gpu_ids=[0], strategy="pipeline")
    with pytest.raises(ValueError):
        sg.MultiGPUConfig(gpu_ids=[0], communication_backend="smoke-signals")
    with pytest.raises(sg.AnnealingError):   # Wolff moves are sweeps, not single-site updates
        sg.SpinDynamics(sg.IsingModel(sg.IsingModelConfig(n_spins=3)),
                        update_rule=sg.UpdateRule.WOLFF).single_spin_update(0)


# ----------------------------------------------------------------------------- model container
@pytest.mark.parametrize("sparse", [False, True])
def test_model_container_edits(sparse):
    torch.manual_seed(0)
    m = sg.IsingModel(sg.IsingModelConfig(n_spins=6, use_sparse=sparse))
    assert set(m.spins.tolist()) <= {-1.0, 1.0} and m.spins.dtype == torch.float32
    m.set_coupling(0, 3, 2.5)
    m.set_coupling(3, 5, -1.0)
    m.set_coupling(0, 3, 1.5)  # overwrite, both triangles
    J = m.dense_couplings()
    assert J[0, 3] == 1.5 and J[3, 0] == 1.5 and J[5, 3] == -1.0 and torch.equal(J, J.T)
    with pytest.raises(ValueError):
        m.set_coupling(0, 6, 1.0)
    m.set_external_field(2, 0.75)
    m.set_external_fields(torch.arange(6, dtype=torch.float32))
    assert m.external_fields[2] == 2.0
    c = m.copy()
    c.set_spins(-m.get_spins())
    assert torch.equal(c.spins, -m.spins) and torch.equal(c.dense_couplings(), J)
    back = sg.IsingModel.from_dict(m.to_dict())
    assert torch.equal(back.dense_couplings(), J) and torch.equal(back.spins, m.spins)
    assert -1.0 <= m.get_magnetization() <= 1.0
    m.reset_to_random()
    assert m.spins.shape == (6,)


def test_coo_to_csr_sums_duplicates_and_sorts():
    idx = torch.tensor([[2, 0, 2, 1, 2], [1, 2, 0, 2, 1]])
    val = torch.tensor([1.0, 4.0, 3.0, -2.0, 0.5])
    rowptr, col, v = coo_to_csr(torch.sparse_coo_tensor(idx, val, (3, 3)))
    assert rowptr.tolist() == [0, 1, 2, 4] and col.tolist() == [2, 2, 0, 1]
    assert v.tolist() == [4.0, -2.0, 3.0, 1.5]


def test_spin_dynamics_diagnostics_equal_reference():
    """get_autocorrelation_time / thermal_equilibrium_check (core/spin_dynamics.py:361-421) on histories
    recorded from the reference (tests/golden/diagnostics.npz, written by make_golden.py): a real run's
    energy and magnetisation series, AR(1) series, a drift, short and constant series."""
    import spin_glass_anneal_rl_amd as sg
    from conftest import load_golden
    g = load_golden("diagnostics")
    names = sorted({k.split("__")[0] for k in g if "__" in k})
    assert len(names) >= 8
    dyn = sg.SpinDynamics.__new__(sg.SpinDynamics)  # diagnostics need the histories only (no GPU)
    for name in names:
        x = g[f"{name}__data"]
        for obs in ("energy", "magnetization"):
            dyn.energy_history = list(x) if obs == "energy" else []
            dyn.magnetization_history = list(x) if obs == "magnetization" else []
            want = float(g[f"{name}__tau_{obs}"])
            got = dyn.get_autocorrelation_time(obs)
            assert got == want or (np.isinf(got) and np.isinf(want)), (name, obs, got, want)
        dyn.energy_history = list(x)
        for win in (100, 50, 5):
            assert dyn.thermal_equilibrium_check(win) == bool(g[f"{name}__equilibrium_w{win}"]), (name, win)
    with pytest.raises(ValueError):
        dyn.get_autocorrelation_time("susceptibility")
    # without scipy the reference assigns p_value = 0.05 | 0.01 and returns p_value > 0.05: False either way
    # (core/spin_dynamics.py:414-421), also for two windows of equal variance
    import builtins
    import sys
    real_import = builtins.__import__

    def no_scipy(name, *a, **k):
        if name == "scipy" or name.startswith("scipy."):
            raise ImportError(name)
        return real_import(name, *a, **k)

    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "scipy" or k.startswith("scipy.")}
    builtins.__import__ = no_scipy
    try:
        dyn.energy_history = [1.0, 2.0] * 100
        assert dyn.thermal_equilibrium_check(100) is False
    finally:
        builtins.__import__ = real_import
        sys.modules.update(saved)


def test_memory_optimizer_keeps_the_reference_method_names():
    """annealing/cuda_kernels.py:446-569: get_optimal_batch_size / create_memory_efficient_tensors /
    optimize_coupling_matrix_storage / clear_memory_cache / get_memory_stats, same argument meaning."""
    import torch
    from spin_glass_anneal_rl_amd.kernel_manager import GPUMemoryOptimizer
    opt = GPUMemoryOptimizer(torch.device("cpu"))
    b = opt.get_optimal_batch_size(1000, available_memory=1 << 30)
    assert b == (int((1 << 30) * 0.8) - 2 * 1000 * 1000 * 4) // (2 * 1000 + 64) and opt.get_optimal_batch_size(10 ** 5, 1 << 20) == 1
    t = opt.create_memory_efficient_tensors(10, 3, use_half_precision=True)
    assert set(t) == {"spins_batch", "energies_batch", "temp_spins", "random_values", "local_fields"}
    assert t["spins_batch"].shape == (3, 10) and t["random_values"].shape == (30,) and t["local_fields"].dtype == torch.float16
    dense = torch.ones(8, 8)
    assert opt.optimize_coupling_matrix_storage(dense) is dense
    sparse = torch.zeros(8, 8)
    sparse[0, 1] = sparse[1, 0] = 1.0
    out = opt.optimize_coupling_matrix_storage(sparse, sparsity_threshold=0.1)
    assert out.is_sparse and torch.equal(out.to_dense(), sparse)
    opt.memory_pool["x"] = 1
    opt.clear_memory_cache()
    assert opt.memory_pool == {}
    st = opt.get_memory_stats()
    assert set(st) == {"device", "memory_allocated", "memory_reserved", "max_memory_allocated", "memory_stats"} and st["memory_allocated"] == 0


# ----------------------------------------------------------------------------- form selection (csrc/sga_route.cpp)
def _route_cases():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "route_table.json")) as f:
        return json.load(f)["cases"]


def test_route_table():
    """WHICH kernel form sweeps a problem is a pure function (csrc/sga_route.cpp, no device call) of the traits the
    set-time scans report, the replica count, the tuning and the options.  tests/golden/route_table.json holds, for the
    five BASELINE configs as bench.py builds them and for 30 shapes drawn like the fuzz's, the query a real engine posed
    on an MI355X and the answer (profiles/r05_route_table.py, which also checked the answer against sga_describe and the
    launched kernel): the same queries must get the same answers here, without a GPU -- a threshold edit that reroutes
    one of them fails this test."""
    from spin_glass_anneal_rl_amd import _native as N
    cases = _route_cases()
    assert sum(c["name"].startswith("BASELINE") for c in cases) == 5 and sum(c["name"].startswith("fuzz") for c in cases) >= 20
    for c in cases:
        q = N.route_query(**c["query"])
        assert N.explain_route(q) == c["explain"], c["name"]
    # what the table says about the BASELINE configs, in words
    by = {c["name"].split(":")[0]: c["explain"] for c in cases if c["name"].startswith("BASELINE")}
    assert by["BASELINE c2a"].startswith("dense storage=f32 acc=f32 waves=8 chunks_per_wave=5 ")       # (the heuristic; bench.py autotunes)
    assert "form=rows spins=int8 waves=1 replicas_per_block=4 updates_per_step=4" in by["BASELINE c3"]
    assert "form=wide-bits spins=bits waves=2" in by["BASELINE c4"] and "slots=1" in by["BASELINE c4"]
    assert "form=wide-bits spins=bits waves=1" in by["BASELINE c5"]
    assert by["BASELINE c5_1000_implicit"].startswith("tsp n_cities=1000 waves=2 passes=2")
    # configs[4] at 1000 cities with its 32 GB of CSR written out (traits as tests/test_baseline_configs_gpu.py sees them)
    q = N.route_query(kind=N.ROUTE_CSR, n=10 ** 6, R_local=256, nnz=3996 * 10 ** 6, max_row_len=3996, layout_entries=4032 * 10 ** 6,
                      slotted=1, rowptr32=0, acc=2, table_m=0)
    assert N.explain_route(q).startswith("csr form=wide-bits spins=bits waves=8 replicas_per_block=1 ")
    # ... and C4 as the engine takes it by itself (entries packed to one dword; bench.py's graded line asks for fp32 values)
    c4 = next(c for c in cases if c["name"].startswith("BASELINE c4"))
    q = N.route_query(**{**c4["query"], "storage": 0, "packed_ok": 1})
    assert "entries=packed" in N.explain_route(q)


def test_route_answers_move_with_their_inputs():
    """Sanity of the pure function itself: tuning, replica count, options and the field-cache request each change the answer
    the way include/sga.h says."""
    from spin_glass_anneal_rl_amd import _native as N
    c3 = next(c for c in _route_cases() if c["name"].startswith("BASELINE c3"))["query"]
    base = N.explain_route(N.route_query(**c3))
    assert "updates_per_step=4" in base
    assert "form=narrow " in N.explain_route(N.route_query(**{**c3, "options": {"csr_updates_per_step": 0}}))
    # (a row dealt to four waves: one replica per workgroup; 4096 of them are LDS resident only with the spins as bits)
    assert "form=wide-bits spins=bits waves=4 replicas_per_block=1" in N.explain_route(N.route_query(**{**c3, "tune_waves": 4}))
    assert "form=wide-bytes spins=int8 waves=4" in N.explain_route(N.route_query(**{**c3, "tune_waves": 4, "R_local": 512}))
    assert "spins=bits" in N.explain_route(N.route_query(**{**c3, "options": {"force_csr_bits": 1}}))
    assert " cached=on(waves=4)" in N.explain_route(N.route_query(**{**c3, "field_cache": 1}))
    assert " cached=auto(start=rows" in N.explain_route(N.route_query(**{**c3, "field_cache": 2}))
    dense = dict(kind=N.ROUTE_DENSE, n=10000, R_local=1024, storage=2, acc=0, table_m=2048, clf_ok=1)
    assert " cached=auto(start=cached" in N.explain_route(N.route_query(**dense, field_cache=2))
    assert "waves=13 chunks_per_wave=4" in N.explain_route(N.route_query(**{**dense, "storage": 1, "tune_waves": 13}))
    assert N.explain_route(N.route_query(kind=N.ROUTE_TSP, n=250000, n_cities=500)).startswith("tsp n_cities=500 waves=2 passes=1")
    with pytest.raises(Exception):
        N.explain_route(N.route_query(kind=7, n=10))


def test_integration_md_route_snippet_runs_as_written_without_a_gpu(capsys):
    """INTEGRATION.md shows how to ask the form selection about a problem from plain ctypes; executed here as it stands."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    section = text[text.index("### Which kernel form will run"):]
    code = re.search(r"```python\n(.*?)```", section, re.S).group(1)
    assert "sga_explain_route" in code
    cwd = os.getcwd()
    os.chdir(root)
    try:
        exec(compile(code, "INTEGRATION.md#route", "exec"), {})
    finally:
        os.chdir(cwd)
    assert capsys.readouterr().out.startswith("csr form=rows spins=int8 waves=1 replicas_per_block=4 updates_per_step=4 ")
