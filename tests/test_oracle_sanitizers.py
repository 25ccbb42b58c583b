"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers run on the CPU build only;
GPU ASan is not available on the pool -- the HIP side has the guard-zone test).  oracle/asan_driver.c drives every entry
point: vector env with every opponent kind / RNG kind / shaped reward / illegal actions / seeds that wrap past 2^32, the
stateless queries, the searches (all five heuristics), flat Monte-Carlo, board sizes 4x4 .. 11x11, cube layers 1 .. 5."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_and_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan-run"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "sanitizer run complete" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
