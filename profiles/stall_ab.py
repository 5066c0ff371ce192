"""A/B of the intermittent 45-75 ms device-side gap on the short-kernel workloads (r01 log):
fresh bench.py processes per variant, no settle pause unless the variant says so; a run is
'stalled' when its wall time exceeds the sum of its kernel events by more than 20 ms.
usage: stall_ab.py [runs per variant] [workload]"""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
workload = sys.argv[2] if len(sys.argv) > 2 else "c3"
VARIANTS = [
    ("baseline (no settle)", {"SGA_BENCH_SETTLE": "0"}),
    ("no bandwidth probes", {"SGA_BENCH_SETTLE": "0", "SGA_BENCH_NOPROBE": "1"}),
    ("keep colidx/val (no hipFree after packing)", {"SGA_BENCH_SETTLE": "0", "SGA_KEEP_CSR_ARRAYS": "1"}),
    ("no probes + keep arrays", {"SGA_BENCH_SETTLE": "0", "SGA_BENCH_NOPROBE": "1", "SGA_KEEP_CSR_ARRAYS": "1"}),
    ("settle 0.3 s (r01 workaround)", {"SGA_BENCH_SETTLE": "0.3"}),
]
if len(sys.argv) > 3:
    VARIANTS = [v for i, v in enumerate(VARIANTS) if str(i) in sys.argv[3].split(",")]
for name, env in VARIANTS:
    gaps = []
    for _ in range(runs):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "10",
                            "--warmup", "3", "--no-cpu-baseline"], capture_output=True, text=True,
                           env=dict(os.environ, **env))
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not line:
            print("run failed:", p.stderr[-300:], flush=True)
            continue
        d = json.loads(line[0])
        gaps.append(d["wall_ms_total"] - d["kernel_ms_total"])
    stalled = sum(g > 20.0 for g in gaps)
    print(f"{name:48s} stalled {stalled}/{len(gaps)}  wall-kernel gaps (ms): " + " ".join(f"{g:.1f}" for g in gaps),
          flush=True)
