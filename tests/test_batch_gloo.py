"""CPU-only: the N > 1 sharding + gather path with torch.distributed gloo, world_size 2 (and the shard arithmetic)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from matchinglib_poselib_amd import batch


def test_pair_shard_partitions():
    for num_pairs in (0, 1, 7, 512, 513):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                b, e = batch.pair_shard(num_pairs, r, world)
                assert 0 <= b <= e <= num_pairs and e - b <= batch.shard_capacity(max(num_pairs, 1), world)
                cover += list(range(b, e))
            assert cover == list(range(num_pairs))


def _worker(rank, world, port, num_pairs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = batch.pair_shard(num_pairs, rank, world)
    local = np.zeros(e - b, batch.RECORD_DTYPE)
    for i, pid in enumerate(range(b, e)):
        rng = np.random.default_rng(1000 + pid)      # record content is a function of the pair id only
        local[i]["pair_id"] = pid
        local[i]["n_matches"] = int(rng.integers(0, 8192))
        local[i]["n_inliers"] = int(rng.integers(0, 4096))
        local[i]["E"] = rng.normal(size=9)
        local[i]["R"] = rng.normal(size=9)
        local[i]["t"] = rng.normal(size=3)
    rec = batch.gather_records(local, num_pairs, rank, world, device=torch.device("cpu"))
    # max-over-ranks timing reduction as bench.py does it
    tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    q.put((rank, rec.tobytes(), float(tt.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("num_pairs", [7, 16])
def test_gather_records_world2(num_pairs):
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, num_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = np.zeros(num_pairs, batch.RECORD_DTYPE)
    for pid in range(num_pairs):
        rng = np.random.default_rng(1000 + pid)
        expect[pid]["pair_id"] = pid
        expect[pid]["n_matches"] = int(rng.integers(0, 8192))
        expect[pid]["n_inliers"] = int(rng.integers(0, 4096))
        expect[pid]["E"] = rng.normal(size=9)
        expect[pid]["R"] = rng.normal(size=9)
        expect[pid]["t"] = rng.normal(size=3)
    for rank, blob, tmax in got:
        assert blob == expect.tobytes()       # every rank holds all records, in pair order
        assert tmax == float(world)


def _match_worker(rank, world, port, num_pairs, nq, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = batch.pair_shard(num_pairs, rank, world)
    cap = batch.shard_capacity(num_pairs, world)
    local = torch.zeros((cap, nq, 4), dtype=torch.int32)          # padded to the shard capacity, as the bench allocates it
    for i, pid in enumerate(range(b, e)):
        local[i] = torch.from_numpy(np.random.default_rng(500 + pid).integers(0, 1 << 20, (nq, 4)).astype(np.int32))
    out = batch.gather_match_lists(local, num_pairs, rank, world, root=0)
    q.put((rank, None if out is None else out.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("num_pairs", [5, 8])
def test_gather_match_lists_world2(num_pairs):
    """The padded match lists travel to the root by grouped send / recv (RCCL has no gather): root holds them in pair order."""
    world, nq = 2, 37
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_match_worker, args=(r, world, port, num_pairs, nq, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = np.stack([np.random.default_rng(500 + pid).integers(0, 1 << 20, (nq, 4)).astype(np.int32) for pid in range(num_pairs)])
    assert got[0] == expect.tobytes() and got[1] is None
