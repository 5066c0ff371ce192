"""The bench.py output contract, checked on the committed lines of the last GPU runs
(profiles/r0N_bench_*.json): the keys and types the driver and the judge read."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[0-9]_bench_*.json")))


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_committed_bench_line_follows_the_contract(path):
    d = json.load(open(path))
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int),
                     ("warmup", int), ("ms_per_step", float), ("higher_is_better", bool),
                     ("scaling", str), ("dtype", str), ("data", str), ("config", dict),
                     ("roofline", dict)):
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None            # BASELINE.md publishes no number for this metric
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    if r["bound"] in ("hbm", "mfma", "infinity-cache"):
        assert r["unit"] in ("GB/s", "TFLOP/s")
        if "r04" <= os.path.basename(path) < "r05":   # (round 4 capped a byte rate above the spec figure, and said so)
            assert r["frac"] == pytest.approx(min(r["achieved"] / r["peak"], 1.0)) and r["frac"] <= 1.0
        else:                                         # (round 5: raw ratios, `cache_served` says what may lift them)
            assert r["frac"] == pytest.approx(r["achieved"] / r["peak"])
        if r["bound"] == "infinity-cache":       # round 5: a structure that fits the 256 MiB cache is not called HBM bound
            assert r["structure_bytes"] <= 256 * 2 ** 20 and r["cache_served"] is True and r["peak"] == 8600.0
            assert r["frac_of_hbm_spec"] == pytest.approx(r["achieved"] / 8000.0)
    else:
        # round 4: a cache-resident kernel reports the bound it really has -- vector-instruction issue / the
        # per-update dependent chain -- with the issue fraction from the committed instruction counts (or None
        # where no counter pass exists), and keeps the byte rate as cache_served_GBs
        assert os.path.basename(path) >= "r04" and r["bound"] in ("valu-issue", "latency")
        assert r["cache_served_GBs"] > 0 and (r["frac"] is None or 0.0 < r["frac"] <= 1.0)
        if r["frac"] is not None:
            assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and r["issue_counters_source"].startswith("profiles/r0")
    assert r["traffic"] is None or r["traffic"] > 0
    # value is whole-job throughput over exactly `steps` timed steps
    spins, total = d["config"]["spins"], d["config"]["replicas_total"]
    assert d["value"] == pytest.approx(total * spins * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3))
    c = d["cpu_baseline"]
    if c is not None:                          # (the 1000-city line has no host copy to time)
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
        assert isinstance(c["sample"], str) and isinstance(c["unit"], str)
        assert c["energy_gap_vs_gpu"]["max_abs_energy_gap"] == 0.0


def test_there_is_a_headline_line():
    assert any(p.endswith("r01_bench_c2a_f32.json") for p in LINES)
    assert any(p.endswith("r02_bench_c2a_f32.json") for p in LINES)


def test_round2_headline_carries_the_beyond_cache_roofline():
    d = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_c2a_f32.json")))
    b = d["roofline_beyond_cache"]
    assert b["bound"] == "hbm" and b["coupling_bytes"] >= 16 * 256 * 2 ** 20     # 16 x the 256 MB Infinity Cache
    assert b["frac"] == pytest.approx(b["achieved"] / b["peak"]) and b["frac"] >= 0.6
    assert abs(b["traffic"] / b["algorithmic_bytes_per_launch"] - 1.0) < 0.05      # PMC bytes within 5 %
    assert d["kernel_ms_total"] <= d["wall_ms_total"] and d["ranks_seen"] == 1


def test_round3_headline_names_its_kernel_and_carries_the_cached_field_variant():
    d = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_c2a_f32.json")))
    r = d["roofline"]
    # the graded figure is the one-row-per-proposal kernel with B = n x sizeof(J element)
    assert r["kernel"] == "sweep_dense_kernel" and r["kernel_instantiation"].startswith("sweep_dense_kernel<float")
    assert r["algorithmic_bytes_per_attempt"] == 40000.0 and "sweep=cached" not in d["config"]["geometry"]
    assert d["couplings_checksum_agree"] is True
    v = d["variants"]["cached_local_fields"]
    assert v["kernel_instantiation"].startswith("sweep_clf_kernel") and v["tracked_energy_equals_recomputed"] is True
    assert v["value"] > 50 * d["value"] and v["after_100_sweeps"]["value"] > v["value"]
    assert v["algorithmic_bytes_per_attempt"] == pytest.approx(v["acceptance_rate"] * v["roofline"]["row_bytes"])
    # the committed profile is of the instantiation the line names (CPW and waves from the geometry)
    import re
    m = re.match(r"sweep_dense_kernel<float, CPW=(\d+), ACC64=0, LEAN=1, BATCH=(\d), SINGLE=0, CANON=0> x (\d+) wave", r["kernel_instantiation"])
    assert m, r["kernel_instantiation"]
    stats = open(os.path.join(ROOT, "profiles", "r03_c2a_f32_kernel_stats.csv")).read()
    assert f"sweep_dense_kernel<float, {m.group(1)}, false, true, {'true' if m.group(2) == '1' else 'false'}, false, false>" in stats
    note = json.load(open(os.path.join(ROOT, "profiles", "r03_c2a_f32_pmc.json")))["note"]
    assert f"--waves {m.group(3)} " in note                      # ... at the same waves per replica
    avg_ms = [float(l.split(",")[-5]) / 1e6 for l in stats.splitlines() if "sweep_dense_kernel" in l][0]
    assert abs(avg_ms / r["avg_launch_ms"] - 1.0) < 0.02           # the trace's average agrees with the line's
    f = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_c2a_f32_force_dist.json")))
    assert f["backend"] == "nccl" and f["ranks_seen"] == 1 and f["couplings_checksum_agree"] is True
    assert f["couplings_checksum"] == d["couplings_checksum"]
    c5 = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_c5_1000_csr.json")))
    assert "4 geometric ladder(s)" in c5["config"]["workload"] and "substitute_instance" in c5["cpu_baseline"]


def test_round4_headline_carries_the_other_configs_and_honest_roofs():
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_c2a_f32.json")))
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] <= 1.0
    # BASELINE configs[2], [3], [4] in the line the driver runs
    cfg = d["configs"]
    assert sorted(cfg) == ["c3", "c4", "c5"]
    for name, c in cfg.items():
        assert c["value"] > 0 and c["steps"] == 10 and c["warmup"] == 5 and c["kernel_instantiation"].startswith("sweep_")
        assert c["cpu_baseline"]["energy_gap_vs_gpu"]["max_abs_energy_gap"] == 0.0
        assert c["cpu_baseline"]["energy_gap_vs_gpu"]["spins_identical"] is True
        r = c["roofline"]
        assert r["frac"] is None or r["frac"] <= 1.0
    assert cfg["c3"]["roofline"]["bound"] == "valu-issue" and cfg["c3"]["kernel_instantiation"].startswith("sweep_csr_rows_kernel")
    assert cfg["c4"]["roofline"]["bound"] == "hbm" and cfg["c4"]["roofline"]["frac"] >= 0.6
    v = cfg["c4"]["variants"]["cached_local_fields"]        # the row-on-accept form for sparse couplings
    assert v["available"] and v["kernel_instantiation"].startswith("sweep_clf_csr_kernel")
    assert v["value"] > 3.0 * cfg["c4"]["value"]
    assert v["algorithmic_bytes_per_attempt"] == pytest.approx(v["acceptance_rate"] * v["roofline"]["row_bytes"])
    # the practical-bandwidth probe is a ceiling again: the beyond-cache block sits below it
    b = d["roofline_beyond_cache"]
    assert b["frac_of_measured_stream_read"] <= 1.02 and b["frac"] >= 0.6
    # the cached-field variant of the headline instance: the same chain, several sweeps per launch
    cl = d["variants"]["cached_local_fields"]
    assert cl["tracked_energy_equals_recomputed"] is True and cl["value"] > 100 * d["value"]
    assert cl["after_100_sweeps"]["value"] > cl["after_100_sweeps"]["one_sweep_per_launch"]["value"] > cl["value"]
    # ... and once the launch is its hottest replica's chain every replica runs at eight waves (option clf_tail_waves)
    assert cl["after_100_sweeps"]["value"] >= 1.0e11 and "x 8 wave" in cl["after_100_sweeps"]["kernel_instantiation_these_sweeps"]
    # ... and several accepts per round while the hottest replica accepts more than ~1 % (option clf_batched = 2)
    assert cl["kernel_instantiation_these_sweeps"].startswith("sweep_clfb_kernel") and cl["value"] >= 3.9e10
    # the all-gather of the one-rank RCCL line is timed on the device (events), the host share beside it
    f = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_c2a_f32_force_dist.json")))
    assert f["backend"] == "nccl" and f["exchange"]["rounds_timed"] >= 1
    assert 0 < f["exchange"]["allgather_ms_per_round"] < 5.0 and f["exchange"]["enqueue_ms_per_round"] > 0
    c3 = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_c3_csr.json")))
    assert c3["roofline"]["bound"] == "valu-issue" and 0.0 < c3["roofline"]["frac"] <= 1.0
    assert c3["roofline"]["traffic"] < 0.2 * c3["roofline"]["algorithmic_bytes_per_launch"]
    for tag in ("c5_implicit", "c5_1000_implicit"):
        line = json.load(open(os.path.join(ROOT, "profiles", f"r04_bench_{tag}.json")))
        assert line["roofline"]["bound"] == "latency" and (line["roofline"]["frac"] or 0.0) <= 1.0
    # the committed headline profile is of the instantiation the line names, and its average agrees
    import re
    r = d["roofline"]
    m = re.match(r"sweep_dense_kernel<float, CPW=(\d+), ACC64=0, LEAN=1, BATCH=(\d), SINGLE=0, CANON=0> x (\d+) wave", r["kernel_instantiation"])
    stats = open(os.path.join(ROOT, "profiles", "r04_c2a_f32_kernel_stats.csv")).read()
    assert f"sweep_dense_kernel<float, {m.group(1)}, false, true, {'true' if m.group(2) == '1' else 'false'}, false, false>" in stats
    avg_ms = [float(l.split(",")[-5]) / 1e6 for l in stats.splitlines() if "sweep_dense_kernel" in l][0]
    assert abs(avg_ms / r["avg_launch_ms"] - 1.0) < 0.02


def test_round5_line_evidence_lines_up_and_no_ratio_is_capped():
    import re
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_c2a_f32.json")))
    r = d["roofline"]
    # the profiled instantiation IS the line's: the committed profile ran `bench.py --waves <the line's pick>` ...
    m = re.match(r"sweep_dense_kernel<float, CPW=(\d+), ACC64=0, LEAN=1, BATCH=(\d), SINGLE=0, CANON=0> x (\d+) wave", r["kernel_instantiation"])
    assert m and r["traffic_source"] == "profiles/r05_c2a_f32_pmc.json"
    assert f"--waves {m.group(3)} " in r["traffic_source_note"]
    stats = open(os.path.join(ROOT, "profiles", "r05_c2a_f32_kernel_stats.csv")).read()
    assert f"sweep_dense_kernel<float, {m.group(1)}, false, true, {'true' if m.group(2) == '1' else 'false'}, false, false>" in stats
    avg_ms = [float(l.split(",")[-5]) / 1e6 for l in stats.splitlines() if "sweep_dense_kernel" in l][0]
    assert abs(avg_ms / r["avg_launch_ms"] - 1.0) < 0.02
    # ... and the pick is deterministic: the fewest waves among the candidates within 1 % of the fastest, table in the line
    table = {k: v for k, v in d["config"]["autotune_ms"].items() if not k.startswith("heuristic:")}
    tied = [int(k.split("x")[0]) for k, v in table.items() if v <= 1.01 * min(table.values())]
    assert int(m.group(3)) == min(tied) and len(table) >= 8
    # no cap anywhere: every ratio is achieved / peak as measured
    blocks = [r, d["roofline_beyond_cache"]] + [c["roofline"] for c in d["configs"].values()]
    for b in blocks:
        if b["frac"] is not None:
            assert b["frac"] == pytest.approx(b["achieved"] / b["peak"], rel=1e-12), b["kernel"]
        assert "frac_note" not in b
    assert r["bound"] == "hbm" and r["cache_served"] is True and 0.9 < r["frac"] < 1.05
    # the beyond-cache block: this round's PMC pass of the final build, the HBM-only figure
    b = d["roofline_beyond_cache"]
    assert b["traffic_source"] == "profiles/r05_dense_f32_n32768_pmc.json" and b["cache_served"] is False
    assert 0.6 <= b["frac"] < 0.9 and abs(b["traffic"] / b["algorithmic_bytes_per_launch"] - 1.0) < 0.01
    # structures that fit the 256 MiB Infinity Cache are not called HBM bound
    cfg = d["configs"]
    assert sorted(cfg) == ["c3", "c4", "c5", "c5_1000", "c5_1000_csr"]
    for name in ("c4", "c5"):
        rr = cfg[name]["roofline"]
        assert rr["bound"] == "infinity-cache" and rr["structure_bytes"] <= 256 * 2 ** 20 and rr["cache_served"] is True
        assert rr["frac_of_hbm_spec"] == pytest.approx(rr["achieved"] / 8000.0)
    assert cfg["c5_1000_csr"]["roofline"]["bound"] == "hbm" and cfg["c5_1000_csr"]["roofline"]["structure_bytes"] > 3e10
    assert cfg["c5_1000_csr"]["roofline"]["frac"] >= 0.6
    # configs[4] at its stated size: a number, with its own profile
    k = cfg["c5_1000"]
    assert "1000-city" in k["workload"] and "couplings implicit" in k["workload"] and "256 replicas/GPU, 4 geometric ladder" in k["workload"]
    assert k["roofline"]["bound"] == "valu-issue" and 0.0 < k["roofline"]["frac"] < 1.0
    assert k["roofline"]["issue_counters_source"] == "profiles/r05_c5_1000_implicit_pmc.json"
    assert k["kernel_instantiation"].startswith("sweep_tsp_par_kernel")
    for name, c in cfg.items():
        assert c["cpu_baseline"]["energy_gap_vs_gpu"]["max_abs_energy_gap"] == 0.0, name
        assert c["cpu_baseline"]["energy_gap_vs_gpu"]["spins_identical"] is True, name
        assert c["setup_ms"]["engine_load"] > 0 and c["energies_sha256"]
    assert "500 of the 1000 cities" in k["cpu_baseline"]["substitute_instance"]
    assert cfg["c5_1000_csr"]["setup_ms"]["engine_load"] > 500.0          # (the symmetry pass over 4e9 entries is in there)
    # C3: where a wave's time goes, from this round's counters
    c3 = cfg["c3"]["roofline"]
    assert c3["bound"] == "valu-issue" and c3["issue_counters_source"] == "profiles/r05_c3_csr_pmc.json"
    shares = c3["wave_time_shares"]
    assert 0.9 < shares["waiting (s_waitcnt / barrier)"] + shares["issue stalled"] + shares["instruction in flight"] < 1.1


def test_round5_two_rank_rehearsal_line_carries_configs_3_and_4_sharded():
    """`bench.py --gpus 2 --backend gloo --share-device` on one GPU (what the driver's `--gpus 8` run does over RCCL):
    after the headline, configs[3] as ONE ladder spanning the ranks and configs[4] as whole ladders per rank."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_2rank_gloo.json")))
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "weak"
    cfg = d["configs"]
    assert sorted(cfg) == ["c4", "c5", "c5_1000"]
    c4, c5, k = cfg["c4"], cfg["c5"], cfg["c5_1000"]
    for c in (c4, c5, k):
        assert c["ranks_seen"] == 2 and c["n_gpus"] == 2 and c["couplings_checksum_agree"] is True and c["value"] > 0
    assert c4["replicas_total"] == 2048 and "1024 replicas/GPU, 1 geometric ladder" in c4["workload"]
    assert c4["exchange"]["allgathers_timed"] == c4["exchange"]["rounds_timed"] >= 1 and c4["exchange"]["bytes_per_rank"] == 8192
    assert c5["replicas_total"] == 512 and "256 replicas/GPU, 8 geometric ladder" in c5["workload"]
    assert c5["exchange"]["allgathers_timed"] == 0 and c5["exchange"]["bytes_per_rank"] == 0 and c5["exchange"]["rounds_timed"] >= 1
    assert "whole ladders per rank" in c5["placement"] and "whole ladders per rank" in k["placement"]


def test_round5_one_rank_rccl_line_carries_the_config_lines():
    """`bench.py --force-dist --configs c4,c5,c5_1000`: headline and config lines through a one-rank RCCL group on the GPU --
    the collectives of the multi-rank path (checksum and energies all-gather on device tensors, time all-reduce, best
    broadcast) executed by RCCL, timed with events on the shared stream."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_c2a_f32_force_dist.json")))
    assert d["backend"] == "nccl" and d["ranks_seen"] == 1 and d["couplings_checksum_agree"] is True
    assert sorted(d["configs"]) == ["c4", "c5", "c5_1000"]
    for name, c in d["configs"].items():
        assert c["backend"] == "nccl" and c["ranks_seen"] == 1 and c["couplings_checksum_agree"] is True, name
    for name in ("c4", "c5"):
        ex = d["configs"][name]["exchange"]
        assert ex["allgathers_timed"] == ex["rounds_timed"] >= 1 and 0 < ex["allgather_ms_per_round"] < 1.0
