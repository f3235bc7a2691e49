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


@pytest.mark.parametrize("workers", [1, 2, 3, 8])
def test_stream_batches_are_dealt_round_robin_and_no_slot_is_reused_while_it_may_be_in_flight(workers):
    """The stream pipeline's dealing (ofps_hip_multi_stream_plan), for a fake list of up to 8 devices: batch g goes to worker
    g % n; with at most two batches in flight per worker (2 n tickets) neither a ticket slot nor the halo buffer of a batch that
    may still be in flight is handed out again -- batch g + 1's halo (the copy of batch g's last frame) is written while
    batches g - 2 n + 1 .. g may be in flight, so it must differ from all of THEIR halo slots."""
    plan = [MultiDevice.stream_plan(g, workers) for g in range(6 * workers + 5)]
    for g, (w, h, t) in enumerate(plan):
        assert w == g % workers
        in_flight = range(max(0, g - 2 * workers + 1), g)            # pushed, possibly uncollected, when batch g is pushed
        assert all(plan[k][2] != t for k in in_flight), "ticket slot handed out twice"
        if g + 1 < len(plan):
            nh = plan[g + 1][1]                                        # written during push g
            assert all(plan[k][1] != nh for k in list(in_flight) + [g]), "halo buffer overwritten while its batch may be in flight"
    # every worker sees every n-th batch, in order
    for k in range(workers):
        mine = [g for g, (w, _, _) in enumerate(plan) if w == k]
        assert mine == list(range(k, len(plan), workers))


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


@pytest.mark.gpu
@pytest.mark.parametrize("workers", [1, 2, 3])
@pytest.mark.parametrize("pinned", [False, True])
def test_multi_stream_pipeline_equals_the_single_context_stream(workers, pinned):
    """ofps_hip_multi_push_frames_async / _frames_wait with 1-3 workers on one GPU (repeated entries of device 0): vectors,
    detector results and quaternions of every frame equal ofps_hip_push_frames_async on ONE context fed the same batches, bit
    for bit (the quaternions too: a batch is one solver launch either way).  Frames in pageable memory (staged by the workers)
    and in page-locked memory; RANSAC on odd batches, per-batch seeds."""
    from ofps_amd import synth
    from ofps_amd.runtime import HipContext
    W, H, B, R, nb, per = 320, 192, 16, 8, 7, 3                     # 7 batches of 3 frames
    fr = synth.luma_sequence(nb * per, W, H, max_step=6, seed=synth.SEED0 + 31)
    nblk = (W // B) * (H // B)
    ctx = HipContext(0)
    ref, ref_ent = [], np.zeros((nb, per, nblk, 4), np.float32)
    ctx.reset_frames()
    for b in range(nb):
        t = ctx.push_frames_async(np.ascontiguousarray(fr[b * per:(b + 1) * per]), block=B, search_range=R, use_ransac=bool(b & 1), num_iters=40,
                                  seed=100 * b, out_entries=ref_ent[b])
        ref += ctx.frames_wait(t)
    md = MultiDevice([0] * workers)
    try:
        got, got_ent = [], np.zeros((nb, per, nblk, 4), np.float32)
        bufs = [ctx.pinned_frame(per * H, W).reshape(per, H, W) if pinned else np.empty((per, H, W), np.uint8) for _ in range(2 * workers)]
        tickets = []
        for b in range(nb):                                          # keep 2 * workers batches in flight
            if len(tickets) == 2 * workers:
                got += md.frames_wait(tickets.pop(0))
            buf = bufs[b % len(bufs)]
            np.copyto(buf, fr[b * per:(b + 1) * per])
            tickets.append(md.push_frames_async(buf, block=B, search_range=R, use_ransac=bool(b & 1), num_iters=40, seed=100 * b,
                                                out_entries=got_ent[b]))
        while tickets:
            got += md.frames_wait(tickets.pop(0))
        assert len(got) == len(ref) == nb * per
        assert not got[0]["have_vectors"] and all(g["have_vectors"] for g in got[1:])
        for k, (g, r) in enumerate(zip(got, ref)):
            assert g["have_vectors"] == r["have_vectors"] and g["motion"] == r["motion"], k
            np.testing.assert_array_equal(g["quat"].view(np.uint32), r["quat"].view(np.uint32))
        np.testing.assert_array_equal(got_ent.reshape(-1, nblk, 4)[1:].view(np.uint32), ref_ent.reshape(-1, nblk, 4)[1:].view(np.uint32))
        # a third batch per worker without collecting is refused; a reset with tickets in flight too
        md.reset_frames()
        ts = [md.push_frames_async(bufs[k % len(bufs)], block=B, search_range=R) for k in range(2 * workers)]
        with pytest.raises(Exception):
            md.push_frames_async(bufs[0], block=B, search_range=R)
        with pytest.raises(Exception):
            md.reset_frames()
        for t in ts:
            md.frames_wait(t)
        with pytest.raises(Exception):
            md.frames_wait(ts[0])
    finally:
        md.close()
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("workers", [1, 2])
def test_multi_stream_row_stride_may_change_between_batches(workers):
    """the frame in front of a batch (its halo) was saved with the PRODUCING batch's row stride and uploaded with the
    consuming batch's; with another stride the first pair of the next batch was searched against a sheared frame, and a larger
    stride silently restarted the stream (first frame without vectors, no error).  The halo is kept dense now: batches with row
    strides W, W + 64, W + 32, W give the bits of the all-dense stream."""
    from ofps_amd import synth
    from ofps_amd.runtime import HipContext
    W, H, B, R, nb, per = 320, 192, 16, 8, 6, 2
    fr = synth.luma_sequence(nb * per, W, H, max_step=6, seed=synth.SEED0 + 77)
    nblk = (W // B) * (H // B)
    ctx = HipContext(0)
    ref, ref_ent = [], np.zeros((nb, per, nblk, 4), np.float32)
    for b in range(nb):
        ref += ctx.frames_wait(ctx.push_frames_async(np.ascontiguousarray(fr[b * per:(b + 1) * per]), block=B, search_range=R, seed=b,
                                                     out_entries=ref_ent[b]))
    ctx.close()
    md = MultiDevice([0] * workers)
    try:
        got, got_ent = [], np.zeros((nb, per, nblk, 4), np.float32)
        keep = []
        for b in range(nb):
            pad = (0, 64, 32, 0, 64, 0)[b]
            buf = np.full((per, H, W + pad), 0xA5, np.uint8)                      # the padding holds junk
            buf[:, :, :W] = fr[b * per:(b + 1) * per]
            keep.append(buf)
            got += md.frames_wait(md.push_frames_async(buf[:, :, :W], block=B, search_range=R, seed=b, out_entries=got_ent[b]))
        assert not got[0]["have_vectors"] and all(g["have_vectors"] for g in got[1:])
        for k, (g, r) in enumerate(zip(got, ref)):
            assert g["motion"] == r["motion"], k
            np.testing.assert_array_equal(g["quat"].view(np.uint32), r["quat"].view(np.uint32))
        np.testing.assert_array_equal(got_ent.reshape(-1, nblk, 4)[1:].view(np.uint32), ref_ent.reshape(-1, nblk, 4)[1:].view(np.uint32))
    finally:
        md.close()


@pytest.mark.gpu
def test_multi_fetch_refuses_results_of_another_batch_or_block_size():
    """ofps_hip_multi_fetch used to check only the buffer's capacity."""
    from ofps_amd import synth
    from ofps_amd._lib import OfpsHipError
    fr = synth.luma_sequence(5, 256, 128, max_step=4, seed=3)
    md = MultiDevice([0, 0])
    try:
        md.stage_frames(fr)
        with pytest.raises(OfpsHipError):
            md.fetch(16)                                             # nothing searched yet
        md.run_resident(16, 8)
        a = md.fetch(16)
        with pytest.raises(OfpsHipError):
            md.fetch(8)                                              # searched with another block size
        md.stage_frames(fr[:4])                                      # a new (smaller) batch: the old results are not its results
        with pytest.raises(OfpsHipError):
            md.fetch(16)
        md.run_resident(16, 8)
        np.testing.assert_array_equal(md.fetch(16).view(np.uint32), a[:3].view(np.uint32))
    finally:
        md.close()
