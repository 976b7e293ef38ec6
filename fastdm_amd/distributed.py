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


def _all_gather(out: List[torch.Tensor], t: torch.Tensor, group=None):
    """dist.all_gather; the gloo backend (CPU tests, one-device protocol tests) has no GPU all-gather, so device tensors
    are staged through host memory there.  RCCL gathers device to device."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = [torch.empty(o.shape, dtype=o.dtype) for o in out]
        dist.all_gather(host, t.cpu(), group=group)
        for o, h in zip(out, host):
            o.copy_(h)
    else:
        dist.all_gather(out, t, group=group)


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin image ownership: rank r takes images r, r+world, ...."""
    return list(range(rank, n_items, world))


def images_of_rank(rank: int, world: int, per_rank: int = 1, cfg_pair: bool = False) -> List[int]:
    """Global image ids a rank denoises when every rank (CFG pair) holds `per_rank` images and works through them in turn, as the
    reference's per-process prompt loop does (/root/reference/examples/evaluation/clip_score.py:346-365).  Image sharding: rank r owns
    r, r + world, ... (`shard_indices`, the order `gather_latents` restores).  CFG pairs: ranks 2i and 2i+1 both own
    i * per_rank ... i * per_rank + per_rank - 1 (the order `gather_pair_latents` returns)."""
    if per_rank < 1:
        raise ValueError("per_rank must be >= 1")
    if cfg_pair:
        if world % 2:
            raise ValueError(f"CFG-pair sharding needs an even number of ranks, got {world}")
        return [(rank // 2) * per_rank + j for j in range(per_rank)]
    return shard_indices(world * per_rank, rank, world)


def gather_latents(local: torch.Tensor, n_items: int, rank: int, world: int) -> torch.Tensor:
    """All-gather per-rank latents [n_local, S, C] into the global batch order [n_items, S, C].

    Ranks may own different counts (n_items not divisible by world): shorter shards are padded to the
    longest so a single fixed-size all_gather suffices."""
    if world == 1 and not dist.is_initialized():
        return local
    per = -(-n_items // world)
    if local.shape[0] > per:
        raise ValueError(f"gather_latents: this rank holds {local.shape[0]} items but ceil({n_items}/{world}) = {per}; pass only the "
                         "items this rank owns (one latent per image, not the CFG batch)")
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    _all_gather(out, pad)
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


def group_size() -> int:
    """Ranks the collective backend (RCCL / gloo) actually sees: 1 without a process group."""
    return dist.get_world_size() if dist.is_initialized() else 1


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


def max_over_ranks(x: float, device) -> float:
    if not dist.is_initialized():
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(xs, device):
    """Every rank's list of floats, as [world][len(xs)] on every rank (per-rank step time / collective time of a benchmark run:
    the max over ranks alone hides which rank was slow)."""
    if not dist.is_initialized():
        return [list(map(float, xs))]
    t = torch.tensor(list(xs), dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(v) for v in o.tolist()] for o in out]


# ---------------------------------------------------------------------------------------------- CFG-pair split
# Models that run a conditional AND an unconditional forward per step (the reference flags them in
# caching/config.py:16 and loops over both in model_entry.py:782-789) leave GPUs idle when there are fewer images
# than GPUs.  A CFG pair is a 2-rank group that owns ONE image: role 0 runs the conditional forward, role 1 the
# unconditional one, each step ends with ONE all-gather of the two noise predictions (1 MiB per 1024x2048 image) inside
# the pair, and both ranks then apply the same guidance combine + Euler update, so their latents stay bit-identical
# without any further exchange.  Step caches need no coordination: each rank only ever sees its own branch, i.e.
# exactly the sequence of forwards the single-GPU loop feeds to that branch's cache context.

class CFGPair:
    """role 0 = conditional forward, role 1 = unconditional forward; `group` = the pair's process group (None: both on
    this rank, no exchange)."""

    def __init__(self, role: int, group=None, ranks=None):
        if role not in (0, 1):
            raise ValueError("CFGPair role must be 0 (cond) or 1 (uncond)")
        self.role, self.group, self.ranks = role, group, ranks

    @property
    def image_index(self) -> int:
        return 0 if self.ranks is None else self.ranks[0] // 2

    def pin_cache(self, transformer):
        """Step caches alternate positive / negative entries call by call (caching/xcaching.py get_cache_key); a pair
        member only ever makes one kind of call, so its entry is pinned."""
        cache = getattr(getattr(transformer, "core_model", None), "cache", None)
        if cache is not None and hasattr(cache, "pin_key"):
            cache.pin_key("positive" if self.role == 0 else "negative")

    def exchange(self, mine: torch.Tensor):
        """all-gather this rank's noise prediction inside the pair -> (cond, uncond), identical on both ranks."""
        mine = mine.contiguous()
        out = [torch.empty_like(mine), torch.empty_like(mine)]
        _all_gather(out, mine, group=self.group)
        return out[0], out[1]


def make_cfg_pairs(rank: int, world: int) -> Optional[CFGPair]:
    """Ranks (2i, 2i+1) form pair i.  Every rank must call this (new_group is collective over the default group).
    Returns None for world == 1 (both forwards run locally)."""
    if world == 1:
        return None
    if world % 2:
        raise ValueError(f"CFG-pair sharding needs an even number of ranks, got {world}")
    mine = None
    for i in range(world // 2):
        ranks = [2 * i, 2 * i + 1]
        g = dist.new_group(ranks=ranks)
        if rank in ranks:
            mine = CFGPair(rank - 2 * i, g, ranks)
    return mine


def gather_pair_latents(local: torch.Tensor, pair: Optional[CFGPair], rank: int, world: int) -> torch.Tensor:
    """The job's final collective under CFG-pair sharding: image i lives (identically) on ranks 2i and 2i+1; returns
    [world // 2, ...] in image order on every rank."""
    if pair is None or world == 1:
        return local
    out = [torch.empty_like(local) for _ in range(world)]
    _all_gather(out, local.contiguous())
    return torch.cat([out[2 * i] for i in range(world // 2)], dim=0)
