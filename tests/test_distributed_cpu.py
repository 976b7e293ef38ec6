"""world_size-2 gloo test of the multi-GPU path's host logic (sharding + the single all-gather)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from fastdm_amd import distributed as D
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    mine = D.shard_indices(n_items, r, w)
    # "denoise" each owned image: a deterministic function of its global index
    local = torch.stack([torch.full((4, 3), float(i)).to(torch.bfloat16) + torch.arange(3).to(torch.bfloat16) for i in mine]) if mine else torch.zeros((0, 4, 3), dtype=torch.bfloat16)
    D.barrier()
    full = D.gather_latents(local, n_items, r, w)
    t = D.max_over_ranks(float(rank + 1), "cpu")
    q.put((rank, full.float().tolist(), t))
    torch.distributed.destroy_process_group()


def _run(world, n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = torch.stack([torch.full((4, 3), float(i)) + torch.arange(3).float() for i in range(n_items)]).tolist()
    for rank, full, t in res:
        assert full == want, f"rank {rank} gathered a wrong batch"
        assert t == float(world)


def test_shard_and_gather_even():
    _run(2, 4)


def test_shard_and_gather_ragged():
    _run(2, 3)


def test_single_process_is_identity():
    from fastdm_amd import distributed as D
    x = torch.randn(2, 5, 3)
    assert D.gather_latents(x, 2, 0, 1) is x and D.shard_indices(5, 1, 2) == [1, 3] and D.max_over_ranks(3.0, "cpu") == 3.0
