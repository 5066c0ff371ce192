#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/, scratch) into the small summaries kept here.

    python profiles/summarize_rocprof.py --stats gpurun_out/prof_stats --fetch gpurun_out/prof_fetch \
        --write gpurun_out/prof_write --tag r01_c2a_f32 [--note "..."]

Writes profiles/<tag>_kernel_stats.csv (the `--kernel-trace --stats` table, engine kernels
only) and profiles/<tag>_pmc.json (per-launch HBM-side traffic of each engine kernel).
Counter handling follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB-ish
units of 1024 B, collected in separate --pmc passes; on gfx950 FETCH_SIZE reports exactly half
of the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled here.
"""
import argparse
import collections
import csv
import glob
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def one(pattern):
    hits = glob.glob(pattern, recursive=True)
    if not hits:
        raise SystemExit(f"no file matches {pattern}")
    return hits[0]


def short(name):
    name = name.strip('"')
    return name if len(name) < 120 else name[:117] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--extra", action="append", default=[],
                    help="further --pmc output dirs; every counter is averaged per launch")
    ap.add_argument("--tag", required=True)
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    if a.stats:
        src = one(os.path.join(a.stats, "**", "*_kernel_stats.csv"))
        rows = list(csv.reader(open(src)))
        keep = [rows[0]] + [r for r in rows[1:] if "sga::" in r[0]]
        with open(os.path.join(HERE, f"{a.tag}_kernel_stats.csv"), "w", newline="") as f:
            csv.writer(f).writerows(keep)
        print(f"wrote {a.tag}_kernel_stats.csv ({len(keep) - 1} engine kernels)")
    pmc = {"note": a.note, "unit": "bytes per launch (average over launches)", "kernels": {}}
    for label, d, scale in (("FETCH_SIZE", a.fetch, 2.0), ("WRITE_SIZE", a.write, 1.0)):
        if not d:
            continue
        src = one(os.path.join(d, "**", "*_counter_collection.csv"))
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(src)):
            if r["Counter_Name"] == label and "sga::" in r["Kernel_Name"]:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            e = pmc["kernels"].setdefault(k, {})
            raw = sum(v) / len(v)
            e[label + "_raw_KiB"] = raw
            e[label.split("_")[0].lower() + "_bytes"] = raw * 1024.0 * scale
            e["launches_" + label] = len(v)
    for d in a.extra:
        src = one(os.path.join(d, "**", "*_counter_collection.csv"))
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(src)):
            if "sga::" in r["Kernel_Name"]:
                agg[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            pmc["kernels"].setdefault(k, {})[c] = sum(v) / len(v)
    for e in pmc["kernels"].values():
        if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e and e["TCC_HIT_sum"] + e["TCC_MISS_sum"] > 0:
            e["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
    if a.fetch or a.write:
        pmc["corrections"] = ("fetch_bytes = FETCH_SIZE x 1024 x 2 (gfx950 wide-read under-count), "
                              "write_bytes = WRITE_SIZE x 1024")
        for e in pmc["kernels"].values():
            e["hbm_bytes"] = e.get("fetch_bytes", 0.0) + e.get("write_bytes", 0.0)
    if a.fetch or a.write or a.extra:
        with open(os.path.join(HERE, f"{a.tag}_pmc.json"), "w") as f:
            json.dump(pmc, f, indent=1)
        print(f"wrote {a.tag}_pmc.json")


if __name__ == "__main__":
    main()
