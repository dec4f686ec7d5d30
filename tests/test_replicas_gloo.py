"""N > 1 path on CPU: two gloo ranks run the replica sharding used by bench.py --gpus N
(partition of streams, barrier, max-over-ranks timing, token gather).  No GPU, no data-path collective."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "web-rwkv-gguf_amd", "wrk"))
    import torch.distributed as dist
    import replicas
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = replicas.ReplicaGroup(dist)
    mine = g.my_streams(5)
    g.barrier()
    ms = g.max_over_ranks(10.0 + rank)
    total = g.sum_over_ranks(len(mine))
    toks = g.gather_tokens([[100 * b + i for i in range(3)] for b in mine], 5)
    q.put((rank, mine, ms, total, toks))
    dist.destroy_process_group()


def test_two_rank_replica_group():
    # torch is imported HERE, not at module level: `pytest -m gpu` collects this module too, and a process that has loaded PyTorch's
    # bundled HIP runtime before libwrk_hip.so runs the GPU tests on a mixed pair of runtimes
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, ms0, tot0, toks0), (r1, s1, ms1, tot1, toks1) = res
    assert s0 == [0, 2, 4] and s1 == [1, 3]                       # disjoint cover, b % world
    assert ms0 == ms1 == 11.0                                     # MAX over ranks
    assert tot0 == tot1 == 5.0
    assert toks1 is None and toks0 == [[100 * b + i for i in range(3)] for b in range(5)]


def test_partition_streams():
    sys.path.insert(0, os.path.join(ROOT, "web-rwkv-gguf_amd", "wrk"))
    import replicas
    assert replicas.partition_streams(8, 8) == [[i] for i in range(8)]
    assert replicas.partition_streams(3, 4) == [[0], [1], [2], []]
    flat = sorted(b for part in replicas.partition_streams(128, 8) for b in part)
    assert flat == list(range(128)) and all(len(p) == 16 for p in replicas.partition_streams(128, 8))
    with pytest.raises(ValueError):
        replicas.partition_streams(4, 0)
