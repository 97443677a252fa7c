"""Data-parallel driver: independent prompts are sharded over one process per GPU; the sampler has
no cross-sequence coupling, so the only communication is ONE all_gather of the finished
completions (+ a small metrics vector) per generate call - RCCL over xGMI on a GPU node, gloo in
CPU tests.  Mirrors the reference's accelerate DP caller (evaluation/lm_eval/lm_eval/models/
diffllm.py:138-165; gather at lm_eval/evaluator.py:523-526,620-642)."""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch


def shard_bounds(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank `rank`; the first n_rows % world ranks get one extra row."""
    q, r = divmod(n_rows, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_completions(local: torch.Tensor, dist=None, n_total: Optional[int] = None) -> torch.Tensor:
    """all_gather of int64 [B_local, L] completions -> [sum B_local, L] on every rank (rank order).
    Ragged shards are padded to the largest shard for the collective and trimmed afterwards."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    home = local.device
    if local.is_cuda and dist.get_backend() == "gloo":
        local = local.cpu()                      # gloo has no device all_gather (CPU rehearsals of the GPU path)
    sizes = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    counts = [int(s.item()) for s in all_sizes]
    bmax = max(counts)
    padded = local
    if local.shape[0] < bmax:
        padded = torch.cat([local, local.new_zeros(bmax - local.shape[0], local.shape[1])], dim=0)
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded.contiguous())
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0).to(home)


def gather_metrics(vec: torch.Tensor, dist=None) -> torch.Tensor:
    """all_gather of one small fp32 metrics vector per rank -> [world, len(vec)] on every rank (rank order): the
    counterpart of the reference's per-rank counts / metrics gather (evaluator.py:523-526,620-642)."""
    vec = vec.reshape(-1).to(torch.float32)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return vec[None, :].clone()
    home = vec.device
    if vec.is_cuda and dist.get_backend() == "gloo":
        vec = vec.cpu()
    out = [torch.empty_like(vec) for _ in range(dist.get_world_size())]
    dist.all_gather(out, vec.contiguous())
    return torch.stack(out, dim=0).to(home)


def generate_data_parallel(generate: Callable[[torch.Tensor], torch.Tensor], input_ids: torch.Tensor, dist=None):
    """Every rank holds the same [B_total, P] prompts, generates its shard with `generate`
    (e.g. lambda ids: model.diffusion_generate(ids, ...)) and receives all completions in prompt order."""
    if dist is None or not dist.is_initialized():
        return generate(input_ids)
    lo, hi = shard_bounds(input_ids.shape[0], dist.get_rank(), dist.get_world_size())
    if hi > lo:
        local = generate(input_ids[lo:hi])
    else:
        probe = generate(input_ids[:1])
        local = probe[:0]
    return gather_completions(local, dist)
