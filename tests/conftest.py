import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get



# ---------------------------------------------------------------- child processes that use the GPU
# A process that has initialised HIP must not fork/exec other programs on this pool, and by the time a test runs the pytest
# process has.  So the GPU runs that need FRESH processes (bench.py starting its own ranks) are started here, at session
# start, before anything has touched the GPU; the tests only read their output.
PRELAUNCH = {}


def _run(cmd, timeout=900):
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    return {"rc": p.returncode, "stdout": p.stdout.decode(errors="replace"), "stderr": p.stderr.decode(errors="replace")[-4000:]}


def pytest_sessionstart(session):
    expr = session.config.getoption("-m") or ""
    if "gpu" not in expr or "not gpu" in expr or not os.path.exists("/dev/kfd"):
        return
    common = ["--steps", "24", "--warmup", "6", "--no-cpu-baseline", "--no-extras", "--steps-per-launch", "8"]
    try:
        # two ranks sharing the one GPU of the box (gloo rehearsal of the N>1 path WITH the HIP engine), started by bench.py itself
        PRELAUNCH["bench_2ranks"] = _run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--lanes", "8192"] + common)
        PRELAUNCH["bench_1rank"] = _run([sys.executable, "bench.py", "--gpus", "1", "--lanes", "16384"] + common)
        PRELAUNCH["bench_2ranks_step"] = _run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--lanes", "8192", "--mode", "step"] + common)
        PRELAUNCH["bench_1rank_step"] = _run([sys.executable, "bench.py", "--gpus", "1", "--lanes", "16384", "--mode", "step"] + common)
        PRELAUNCH["bench_short"] = _run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-extras"])
        PRELAUNCH["bench_short_graph"] = _run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-extras", "--graph-rollout"])
        # the RCCL branch at world_size 1: bench.py's N>1 code path under backend nccl, and the trainers' gradient all-reduce
        PRELAUNCH["bench_nccl_w1"] = _run([sys.executable, "bench.py", "--gpus", "1", "--force-collectives", "--backend", "nccl", "--lanes", "16384"] + common)
        PRELAUNCH["nccl_selftest"] = _run([sys.executable, "tools/nccl_selftest.py"])
    except Exception as exc:   # reported by the tests that need the output
        PRELAUNCH["error"] = repr(exc)


@pytest.fixture(scope="session")
def prelaunched():
    return PRELAUNCH
