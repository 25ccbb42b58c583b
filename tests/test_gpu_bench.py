"""bench.py as the driver runs it: `--gpus N` must produce N ranks by itself, the shards' union must equal the single-process
run, and the line must be self-consistent (the timed region is what `config.launch` says; kernel_ms <= ms_per_step).
The child processes were started by conftest.pytest_sessionstart, before this process touched the GPU."""
import json

import pytest

pytestmark = pytest.mark.gpu


def _line(prelaunched, key):
    if "error" in prelaunched:
        pytest.fail("prelaunch failed: " + prelaunched["error"])
    if key not in prelaunched:
        pytest.skip("bench runs are started at session start on a GPU box only")
    r = prelaunched[key]
    assert r["rc"] == 0, r["stderr"]
    lines = [l for l in r["stdout"].splitlines() if l.startswith("{")]
    assert len(lines) == 1, r["stdout"][-2000:]      # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.mark.parametrize("suffix", ["", "_step"])
def test_bench_gpus2_starts_two_ranks_and_shards_are_invariant(prelaunched, suffix):
    two, one = _line(prelaunched, "bench_2ranks" + suffix), _line(prelaunched, "bench_1rank" + suffix)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["collective"]["ranks"] == 2 and two["collective"]["backend"] == "gloo"
    assert two["config"]["lanes_per_gpu"] * 2 == one["config"]["lanes_per_gpu"]
    # 2 x 8 192 lanes with global lane ids / seeds == 1 x 16 384 lanes: same final boards
    assert two["state_digest"] == one["state_digest"]
    assert two["config"]["mode"] == one["config"]["mode"] == ("step" if suffix else two["config"]["mode"])
    assert two["cpu_baseline"] is None                 # rank 0 at N=1 only
    assert "shared memory" in two["config"]["timing_barrier"] and one["config"]["timing_barrier"] is None


def test_short_bench_line_is_self_consistent(prelaunched):
    """the driver's own invocation shape (--steps 20): ONE 20-step launch in the timed region (a direct C-ABI call; --graph-rollout
    replays it as a hipGraph and must leave the same state), event time inside wall time"""
    b = _line(prelaunched, "bench_short")
    g = _line(prelaunched, "bench_short_graph")
    assert "direct C-ABI" in b["config"]["launch"] and "hipGraph" in g["config"]["launch"] and "replay" in g["config"]["launch"]
    assert b["state_digest"] == g["state_digest"] and b["config"]["kernel_launches_in_timed_region"] == g["config"]["kernel_launches_in_timed_region"] == 1
    assert b["steps"] == 20 and b["metric"].startswith("env steps/sec") and b["unit"] == "env steps/sec"
    r = b["roofline"]
    assert r["kernel_ms"] * b["config"]["kernel_launches_in_timed_region"] <= b["ms_per_step"] * b["steps"] * 1.0001
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    if r["valu_issue"]:
        assert 0 < r["valu_issue"]["frac"] <= 1.0
    assert abs(b["value"] - b["config"]["lanes_per_gpu"] * b["steps"] / (b["ms_per_step"] * 1e-3 * b["steps"])) / b["value"] < 1e-9


def test_rccl_branch_runs_at_world_size_one(prelaunched):
    """backend nccl (= RCCL) with one rank on the one-GPU box: bench.py's N>1 path (process group with device_id, barrier, the MAX /
    ones / digest all-reduces on device tensors) and the trainers' flat-gradient all-reduce"""
    b = _line(prelaunched, "bench_nccl_w1")
    one = _line(prelaunched, "bench_1rank")
    assert b["rccl_ranks"] == 1 and b["collective"] == {"backend": "nccl", "ranks": 1}
    assert b["state_digest"] == one["state_digest"]
    t = _line(prelaunched, "nccl_selftest")
    assert t["backend"] == "nccl" and t["world"] == 1 and t["rccl_ranks"] == 1
    assert t["bucket_ok"] and t["fused_updated"] and t["grad_norm"] > 0
