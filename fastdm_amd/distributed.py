"""Multi-GPU driver for the denoise loop: one process per GPU, images sharded across ranks, ONE
collective per job (an all-gather of the final packed latents, 1 MiB per 1024x2048 image).

The reference has no distributed runtime; it only ever parallelises by forking a process per GPU
over a prompt shard (examples/evaluation/clip_score.py:346-365).  Each image's trajectory is
independent and FLUX is guidance-distilled (one forward per step), so there is no per-step
exchange: weights are replicated (12 GB fp8 + 6.5 GB bf16 << 288 GB) and ranks only meet at the end.
`backend='nccl'` is RCCL on ROCm (xGMI inside a node); 'gloo' is used by the CPU tests.
"""
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's env; single-process when WORLD_SIZE is unset."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # FDM_FORCE_DIST=1 initialises the process group for a single rank too (exercises the RCCL path on a 1-GPU box)
    if (world > 1 or os.environ.get("FDM_FORCE_DIST") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin image ownership: rank r takes images r, r+world, ...."""
    return list(range(rank, n_items, world))


def gather_latents(local: torch.Tensor, n_items: int, rank: int, world: int) -> torch.Tensor:
    """All-gather per-rank latents [n_local, S, C] into the global batch order [n_items, S, C].

    Ranks may own different counts (n_items not divisible by world): shorter shards are padded to the
    longest so a single fixed-size all_gather suffices."""
    if world == 1 and not dist.is_initialized():
        return local
    per = -(-n_items // world)
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    full = torch.empty((n_items,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = shard_indices(n_items, r, world)
        full[idx] = out[r][: len(idx)]
    return full


def barrier():
    if dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


def max_over_ranks(x: float, device) -> float:
    if not dist.is_initialized():
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
