"""N>1 path on CPU: two gloo ranks shard the lanes exactly as bench.py does across GPUs
(contiguous global lane ranges, seeds and Philox streams following the GLOBAL lane id,
counters all-reduced); stepping uses the CPU oracle here (tests may).  The union of the
shards must equal the single-process run bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ewn_gym_amd.sharding import all_reduce_counters, lane_range, lane_seeds
from oracle import pyoracle as po

N_TOTAL, STEPS = 512, 12


def _run_shard(lo, hi, n_total):
    env = po.OracleVecEnv(hi - lo, opponent="minimax", max_depth=3, rng="philox", philox_key=2024, autoreset=True,
                          lane_offset=lo, seed_stride=n_total)
    env.reset(seeds=lane_seeds(lo, hi).numpy().view(np.uint32))
    wins = steps = 0
    for t in range(STEPS):
        out = env.step(env.sample_legal_actions(t))
        wins += int((out[5] == 2).sum())
        steps += hi - lo
    b, d = env.obs()
    return b, d, wins, steps


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = lane_range(N_TOTAL, world, rank)
    b, d, wins, steps = _run_shard(lo, hi, N_TOTAL)
    counters = all_reduce_counters(torch.tensor([wins, steps], dtype=torch.int64))
    boards = [torch.zeros((hi - lo, 5, 5), dtype=torch.int8) for _ in range(world)]
    dist.all_gather(boards, torch.from_numpy(b))
    if rank == 0:
        q.put((torch.cat(boards).numpy(), counters.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_lane_range_and_seeds():
    assert lane_range(65536 * 8, 8, 3) == (196608, 262144)
    with pytest.raises(ValueError):
        lane_range(10, 3, 0)
    s = lane_seeds(2**32 - 9487 - 2, 2**32 - 9487 + 2).numpy().view(np.uint32)
    assert s.tolist() == [2**32 - 2, 2**32 - 1, 0, 1]   # uint32 wrap-around, like np.random.seed's range


def test_two_gloo_ranks_equal_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    boards, counters = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    b1, _, wins1, steps1 = _run_shard(0, N_TOTAL, N_TOTAL)
    assert np.array_equal(boards, b1)
    assert counters == [wins1, steps1] and steps1 == N_TOTAL * STEPS


def _barrier_worker(rank, world, port, q):
    """bench.ShmBarrier under two gloo ranks: nobody leaves round i before everybody has entered it"""
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    b = bench.ShmBarrier(world, rank, dist)
    seen = []
    for i in range(50):
        if rank == i % world:
            time.sleep(0.002)                 # the late rank of this round
        b.wait()
        seen.append(int(b.slots.min()))       # after leaving round i + 1: every slot carries at least i + 1
    ok = b.ok and all(v >= i + 1 for i, v in enumerate(seen))
    b.close()
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_shared_memory_barrier_two_ranks():
    """the node-local barrier bench.py brackets its timed region with at N > 1 (instead of an RCCL all-reduce inside a 0.2 ms region)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_barrier_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]
