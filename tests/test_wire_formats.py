"""Wire formats pinned to files WRITTEN BY THE REFERENCE (tests/golden/wire_*.npz, produced by
tests/golden/make_golden.py::case_wire_formats through the reference's own
AnnealingResult.save, annealing/result.py:147-165, and IsingModel.to_dict,
core/ising_model.py:213-229): the build reads them, and what the build writes reads back with
the same keys, dtypes and values the reference's files hold."""
import json
import os

import numpy as np
import torch

import spin_glass_anneal_rl_amd as sg
from conftest import GOLDEN, load_golden

REF_RESULT = os.path.join(GOLDEN, "wire_result_reference.npz")


def test_result_written_by_the_reference_loads():
    exp = load_golden("wire_expected")
    r = sg.AnnealingResult.load(REF_RESULT)
    assert r.best_energy == float(exp["best_energy"]) and r.n_sweeps == int(exp["n_sweeps"])
    assert np.array_equal(r.best_configuration.numpy().astype(np.int8), exp["best_configuration"])
    assert r.best_configuration.dtype == torch.float32
    assert r.energy_history == exp["energy_history"].tolist()
    assert r.temperature_history == exp["temperature_history"].tolist()
    assert r.acceptance_rate_history == exp["acceptance_rate_history"].tolist()
    assert r.total_time == float(exp["total_time"])
    assert r.final_temperature == float(exp["final_temperature"])
    assert r.final_acceptance_rate == float(exp["final_acceptance_rate"])
    assert r.energy_std == float(exp["energy_std"])
    cs = int(exp["convergence_sweep"])
    assert r.convergence_sweep == (cs if cs > 0 else None)
    assert r.random_seed == int(exp["random_seed"])
    assert r.algorithm == "simulated_annealing" and r.device == "cpu"


def test_result_written_by_the_build_has_the_references_layout(tmp_path):
    r = sg.AnnealingResult.load(REF_RESULT)
    out = str(tmp_path / "mine.npz")
    r.save(out)
    with np.load(REF_RESULT, allow_pickle=True) as ref, np.load(out, allow_pickle=True) as mine:
        assert sorted(ref.files) == sorted(mine.files)
        for k in ref.files:
            assert ref[k].dtype == mine[k].dtype and ref[k].shape == mine[k].shape, k
            assert np.array_equal(ref[k], mine[k]) or (ref[k].dtype == object and ref[k].item() == mine[k].item()), k
    again = sg.AnnealingResult.load(out)
    assert again.get_summary() == r.get_summary() and again.energy_history == r.energy_history


def test_unset_optionals_round_trip_as_the_reference_stores_them(tmp_path):
    r = sg.AnnealingResult(best_configuration=torch.ones(4), best_energy=-1.0, energy_history=[-1.0],
                           temperature_history=[1.0], acceptance_rate_history=[0.5], total_time=0.1,
                           n_sweeps=3)
    out = str(tmp_path / "r.npz")
    r.save(out)
    with np.load(out, allow_pickle=True) as z:   # None is pickled into an object array, as np.savez does
        assert z["convergence_sweep"].dtype == object and z["random_seed"].dtype == object
    back = sg.AnnealingResult.load(out)
    assert back.convergence_sweep is None and back.random_seed is None


def test_model_dict_written_by_the_reference_loads_and_matches():
    with np.load(os.path.join(GOLDEN, "wire_model_reference.npz"), allow_pickle=False) as z:
        ref = {"config": json.loads(str(z["config_json"])), "spins": z["spins"],
               "couplings": z["couplings"], "external_fields": z["external_fields"]}
    exp = load_golden("wire_expected")
    m = sg.IsingModel.from_dict(ref)
    assert m.n_spins == 12 and not m.couplings.is_sparse
    assert np.array_equal(m.couplings.numpy(), exp["J"]) and np.array_equal(m.external_fields.numpy(), exp["h"])
    assert np.array_equal(m.spins.numpy().astype(np.int8), exp["spins"])
    mine = m.to_dict()
    assert sorted(mine) == sorted(ref) and mine["config"] == ref["config"]
    for k in ("spins", "couplings", "external_fields"):
        assert mine[k].dtype == ref[k].dtype and np.array_equal(mine[k], ref[k]), k
