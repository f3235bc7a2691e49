"""world_size-2 gloo test of the multi-GPU layout (CPU, runs in the build container): pair/frame
sharding, the key-frame broadcast and the pair-ordered result gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ofps_amd import distributed as D


def test_pair_and_frame_ranges_cover_everything():
    for n_pairs in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                first, count = D.pair_range(n_pairs, world, r)
                seen += list(range(first, first + count))
                f0, fc = D.frame_range(n_pairs, world, r, 0)
                if count:
                    assert (f0, fc) == (first, count + 1)         # consecutive pairs: one halo frame
                g0, gc = D.frame_range(n_pairs, world, r, 1)
                if count:
                    assert (g0, gc) == (first + 1, count)         # key-frame mode: frame 0 comes by broadcast
            assert seen == list(range(n_pairs))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_pairs, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # key frame: rank 0 owns it, everyone must end up with the same bytes
        key = torch.arange(64, dtype=torch.uint8).reshape(8, 8) if rank == 0 else torch.zeros((8, 8), dtype=torch.uint8)
        D.broadcast_reference(key, src=0)
        assert key.sum().item() == sum(range(64))
        # each rank "processes" its pairs: result row k = (pair index, rank)
        first, count = D.pair_range(n_pairs, world, rank)
        local = torch.tensor([[float(first + i), float(rank), 0.0, 0.0] for i in range(count)], dtype=torch.float32).reshape(count, 4)
        full = D.gather_results(local, n_pairs)
        assert full.shape == (n_pairs, 4)
        assert full[:, 0].tolist() == [float(i) for i in range(n_pairs)]          # pair order preserved
        t = D.max_over_ranks(1.0 + rank)
        assert t == float(world)
        np.save(os.path.join(out_dir, f"r{rank}.npy"), full.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [5, 8])
def test_gloo_world2_broadcast_and_gather(tmp_path, n_pairs):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_pairs, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "r0.npy"); b = np.load(tmp_path / "r1.npy")
    np.testing.assert_array_equal(a, b)
    owners = a[:, 1].tolist()
    assert owners == sorted(owners) and set(owners) == {0.0, 1.0}
