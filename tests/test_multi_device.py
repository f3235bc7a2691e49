"""In-process multi-device dispatcher (ofps_hip_multi_*): partition logic on CPU with fake device lists;
on the GPU box, one and several workers (repeated entries of the one device a gpurun box has) against the single-context
call and the oracle, pair mode and key mode."""
import numpy as np
import pytest

from ofps_amd import distributed as D
from ofps_amd.runtime import MultiDevice


@pytest.mark.parametrize("n_pairs", [0, 1, 5, 8, 63, 64, 65, 257])
@pytest.mark.parametrize("workers", [1, 2, 3, 8])
def test_partition_is_the_distributed_module_s_and_covers_the_batch_in_order(n_pairs, workers):
    """The C-ABI partition == ofps_amd.distributed.pair_range / frame_range (what bench.py's torchrun path uses): contiguous
    ranges in worker order, every pair exactly once, one halo frame in pair mode, the key frame excluded in key mode."""
    seen = []
    for k in range(workers):
        first, count = MultiDevice.pair_range(n_pairs, workers, k)
        assert (first, count) == D.pair_range(n_pairs, workers, k)
        seen += list(range(first, first + count))
        for ref_mode in (0, 1):
            assert MultiDevice.frame_range(n_pairs, workers, k, ref_mode) == D.frame_range(n_pairs, workers, k, ref_mode)
        ff, fc = MultiDevice.frame_range(n_pairs, workers, k, 0)
        assert fc == (count + 1 if count else 0) and ff == first
        ff, fc = MultiDevice.frame_range(n_pairs, workers, k, 1)
        assert fc == count and (count == 0 or ff == first + 1)
    assert seen == list(range(n_pairs))
    counts = [MultiDevice.pair_range(n_pairs, workers, k)[1] for k in range(workers)]
    assert max(counts) - min(counts) <= 1 and counts == sorted(counts, reverse=True)


def test_cfg4_batch_over_a_fake_node_of_eight():
    """BASELINE configs[3]: 64 pairs over 8 devices -> 8 pairs and 9 resident frames each (8 + 1 halo)."""
    for k in range(8):
        assert MultiDevice.pair_range(64, 8, k) == (8 * k, 8)
        assert MultiDevice.frame_range(64, 8, k, 0) == (8 * k, 9)
        assert MultiDevice.frame_range(64, 8, k, 1) == (8 * k + 1, 8)


def test_bad_device_list_fails_loudly():
    import torch
    from ofps_amd._lib import OfpsHipError
    with pytest.raises(OfpsHipError):
        MultiDevice([])
    if not torch.cuda.is_available():
        with pytest.raises(OfpsHipError) as ei:
            MultiDevice([0])
        assert "no CPU fallback" in str(ei.value) or "device" in str(ei.value)


@pytest.mark.gpu
@pytest.mark.parametrize("workers", [1, 2, 3])
@pytest.mark.parametrize("ref_mode", [0, 1])
def test_multi_sad_flow_equals_the_single_context_call_and_the_oracle(workers, ref_mode):
    """Results in pair order, bit for bit what ofps_hip_sad_flow_dev returns for the whole batch on one context; sampled
    pairs against the oracle.  Several workers = repeated entries of device 0 (independent contexts, one GPU)."""
    import torch
    import oracle
    from ofps_amd import synth
    from ofps_amd.runtime import HipContext
    W, H, B, R, F = 640, 360, 16, 8, 8
    fr = synth.luma_sequence(F, W, H, max_step=R, seed=synth.SEED0 + 70)
    md = MultiDevice([0] * workers)
    try:
        got = md.sad_flow(fr, B, R, ref_mode)
        again = md.sad_flow(fr[::-1].copy(), B, R, ref_mode)            # a second batch through the same dispatcher
    finally:
        md.close()
    ctx = HipContext(0)
    try:
        nb = (W // B) * (H // B)
        d = torch.from_numpy(fr).cuda()
        out = torch.zeros((F - 1, nb, 4), dtype=torch.float32, device="cuda")
        ctx.use_torch_stream()
        ctx.sad_flow_dev(d.data_ptr(), F, W, H, W, W * H, ref_mode, B, R, out.data_ptr(), None)
        torch.cuda.synchronize()
        ctx.use_own_stream()
        want = out.cpu().numpy()
    finally:
        ctx.close()
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    for k in (0, F - 2):
        prev = fr[0] if ref_mode else fr[k]
        ent_o, _ = oracle.sad_flow(prev, fr[k + 1], B, R)
        np.testing.assert_array_equal(got[k].view(np.uint32), ent_o.view(np.uint32))
    rev = fr[::-1]
    ent_o, _ = oracle.sad_flow(rev[0] if ref_mode else rev[3], rev[4], B, R)
    np.testing.assert_array_equal(again[3].view(np.uint32), ent_o.view(np.uint32))


@pytest.mark.gpu
def test_multi_resident_batch_more_workers_than_pairs_and_repeated_runs():
    import oracle
    from ofps_amd import synth
    W, H, B, R, F = 320, 192, 8, 16, 3                              # 2 pairs, 4 workers: two of them idle
    fr = synth.luma_sequence(F, W, H, max_step=R, seed=9)
    md = MultiDevice([0, 0, 0, 0])
    try:
        md.stage_frames(fr, 0)
        md.run_resident(B, R, steps=3)
        got = md.fetch(B)
    finally:
        md.close()
    for k in range(F - 1):
        ent_o, _ = oracle.sad_flow(fr[k], fr[k + 1], B, R)
        np.testing.assert_array_equal(got[k].view(np.uint32), ent_o.view(np.uint32))


@pytest.mark.gpu
def test_checksum_dev_is_the_wrapping_u64_sum_per_item():
    """ofps_hip_checksum_dev: what bench.py's strong-scaling step gathers across ranks instead of the records."""
    import torch
    from ofps_amd.runtime import HipContext
    rng = np.random.default_rng(5)
    for batch, words in ((1, 1), (3, 1000), (8, 129600 * 2), (5, 257)):
        a = rng.integers(0, 2 ** 63, size=(batch, words), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(batch, words), dtype=np.uint64)
        d = torch.from_numpy(a.view(np.int64)).cuda()
        out = torch.full((batch,), -1, dtype=torch.int64, device="cuda")
        ctx = HipContext(0)
        try:
            ctx.use_torch_stream()
            ctx.checksum_dev(d.data_ptr(), words * 8, batch, out.data_ptr())
            torch.cuda.synchronize()
            ctx.use_own_stream()
        finally:
            ctx.close()
        with np.errstate(over="ignore"):
            want = a.sum(axis=1, dtype=np.uint64)
        np.testing.assert_array_equal(out.cpu().numpy().view(np.uint64), want)
