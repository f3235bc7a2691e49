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


def _host_collectives() -> bool:
    """True when the process group cannot take device tensors (gloo: the CPU tests, and the one-GPU rehearsal of an N-rank
    run -- bench.py --backend gloo --device-map 0,0 -- where RCCL refuses two ranks on one device)."""
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo"


def _coll_device(device):
    import torch
    return torch.device("cpu") if _host_collectives() else device


def broadcast_reference(frame, src: int = 0):
    """In-place broadcast of the shared reference frame (torch tensor, device or host).  Under gloo a device tensor travels
    through a host copy (the source rank's bytes out, everybody's bytes in)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if frame.is_cuda and _host_collectives():
            host = frame.cpu()
            dist.broadcast(host, src=src)
            if dist.get_rank() != src:
                frame.copy_(host)
        else:
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
    if local.is_cuda and _host_collectives():
        local = local.cpu()                                   # gloo: the table is gathered (and returned) on the host
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
    t = torch.tensor([value], dtype=torch.float64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---------------------------------------------------------------------------------------------
# process layout: launching one rank per GPU and the bench's synchronisation primitives
# ---------------------------------------------------------------------------------------------

def free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launched_by_torchrun() -> bool:
    import os
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def launch_ranks(script: str, argv: list[str], n: int, timeout: float | None = None) -> int:
    """Re-executes `script argv` as n ranks of one node under torch.distributed.run (what the driver does for N > 1);
    rendezvous on 127.0.0.1 (the container hostname may not resolve).  Returns the launcher's exit code."""
    import os
    import subprocess
    import sys
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this host driver (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    return subprocess.call(cmd, env=env, timeout=timeout)


def init_from_env(backend: str = "nccl", device_index: int | None = None):
    """-> (rank, world, local_rank).  Creates the process group whenever a launcher set RANK/MASTER_PORT (also at
    world size 1, so the RCCL init / barrier / reduce path of an N-GPU run can be exercised on a 1-GPU box)."""
    import os
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank if device_index is None else device_index)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def active() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


class StreamBarrier:
    """Barrier as an all-reduce of one int on the launch stream (dist.barrier() builds a fresh tensor and
    device-synchronises by itself: ~1 ms per call under RCCL, which would land inside a timed region)."""

    def __init__(self, device):
        import torch
        self._token = torch.zeros(1, dtype=torch.int32, device=_coll_device(device)) if active() else None

    def __call__(self):
        import torch.distributed as dist
        if self._token is not None:
            dist.all_reduce(self._token)


def ranks_seen(device=None) -> int:
    """How many ranks took part (sum of ones over the group): the bench prints it so a run that silently measured one
    GPU cannot pass for an N-GPU run."""
    import torch
    import torch.distributed as dist
    if not active():
        return 1
    t = torch.ones(1, dtype=torch.int32, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def min_over_ranks(value: int, device=None) -> int:
    import torch
    import torch.distributed as dist
    if not active() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.int32, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def sum_over_ranks(value: int, device=None) -> int:
    import torch
    import torch.distributed as dist
    if not active() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.int64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def gather_counts(count: int, device=None) -> list[int]:
    """Every rank's pair count, in rank order."""
    import torch
    import torch.distributed as dist
    if not active() or dist.get_world_size() == 1:
        return [count]
    t = torch.tensor([count], dtype=torch.int64, device=_coll_device(device))
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [int(p.item()) for p in parts]
