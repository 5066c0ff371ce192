"""The CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer ("race detection /
sanitizers": a sanitizer build of the CPU restatement; GPU sanitizers are not available on this pool).

`make -C oracle asan` builds oracle/libsg_oracle_asan.so; the golden-vector suite (every fixture the reference
wrote, bit for bit) and the world-2 gloo suite (OpenMP sweeps from several processes) then run against it in a child
interpreter with libasan preloaded.  Any heap overflow, use after free or undefined operation aborts the child."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    path = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_oracle_suites_pass_under_asan_and_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("this gcc ships no libasan")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, capture_output=True)
    lib = os.path.join(ROOT, "oracle", "libsg_oracle_asan.so")
    assert os.path.exists(lib)
    env = dict(os.environ, SG_ORACLE_LIBRARY=lib, LD_PRELOAD=asan,
               # (the interpreter itself is not instrumented: its arena "leaks" are not the oracle's)
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="4")
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"),
                        os.path.join(ROOT, "tests", "test_sharded_gloo.py")],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert "passed" in p.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
