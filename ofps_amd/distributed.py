"""Multi-GPU layout for the flow hot path: one process per GPU, frame pairs sharded across ranks.

Frame pairs are independent units (vectors, field, island and rotation of a pair depend
on nothing else; the Almeida estimator is stateless, almeida-estimator/src/lib.rs:100-121), so the data
path needs NO collective.  torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in the CPU tests) is used only for
  * broadcast_reference: the shared key frame of ref_mode 1 goes from the ingest rank to all ranks
    (one 2-8 MB broadcast: direct xGMI links, no ring needed at this size);
  * gather_results: per-pair quaternions / islands back to rank 0 in pair order;
  * max_over_ranks: the bench's timing reduction.
"""
from __future__ import annotations


def pair_range(n_pairs: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous block of pairs for `rank`: (first_pair, count); the first n_pairs % world ranks get one more."""
    assert world >= 1 and 0 <= rank < world and n_pairs >= 0
    base, extra = divmod(n_pairs, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def frame_range(n_pairs: int, world: int, rank: int, ref_mode: int = 0) -> tuple[int, int]:
    """Frames a rank must hold for its pairs: (first_frame, count).

    ref_mode 0 (pair k = frames k, k+1): count pairs need count+1 frames (one halo frame shared with
    the next rank).  ref_mode 1 (pair k = frames 0, k+1): the key frame 0 arrives by broadcast, the rank
    holds frames first+1 .. first+count."""
    first, count = pair_range(n_pairs, world, rank)
    if count == 0:
        return first, 0
    if ref_mode == 0:
        return first, count + 1
    return first + 1, count


def broadcast_reference(frame, src: int = 0):
    """In-place broadcast of the shared reference frame (torch tensor, device or host)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(frame, src=src)
    return frame


def gather_results(local, n_pairs: int):
    """All ranks contribute [count_r, ...] tensors; returns the [n_pairs, ...] tensor in pair order
    (on every rank).  Ranks may hold different counts, so tensors are padded to the largest shard."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    counts = [pair_range(n_pairs, world, r)[1] for r in range(world)]
    cap = max(counts) if counts else 0
    pad = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


def max_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
