"""The reference's own test (almeida-estimator/src/lib.rs:308-372), written against the Python mirror of the plugin
interface the way the Rust test is written against the Rust types, plus the detection-loop semantics."""
import numpy as np
import pytest

import oracle
from ofps_amd import synth

import almeida_cases as ac

pytestmark = pytest.mark.gpu


def _test_rot(estimator):
    from ofps_amd.plugins import StandardCamera
    camera = StandardCamera(1.0, 90.0)                                  # lib.rs:309
    for rot, angles, q, field in ac.cases():
        r, tr = estimator.estimate(field, camera, None)                # lib.rs:341
        delta = ac.error_deg(q, r)                                     # q.angle_to(&r).to_degrees()
        assert delta < 0.1 * rot or delta == 0.0, (angles, delta, 0.1 * rot)     # lib.rs:347-348
        assert (tr == 0).all()


def test_rotation_default():
    from ofps_amd.plugins import HipAlmeidaEstimator
    estimator = HipAlmeidaEstimator()
    estimator.use_ransac = False                                       # lib.rs:361-362
    _test_rot(estimator)


def test_rotation_ransac():
    from ofps_amd.plugins import HipAlmeidaEstimator
    estimator = HipAlmeidaEstimator()
    estimator.use_ransac = True
    estimator.num_iters = 100                                          # lib.rs:368-370
    _test_rot(estimator)


def test_properties_surface_matches_reference_names():
    from ofps_amd.plugins import HipAlmeidaEstimator, HipBlockMotionDetection
    est, det = HipAlmeidaEstimator(), HipBlockMotionDetection()
    assert [p[0] for p in est.props()] == ["Use ransac", "Ransac iters", "Inlier threshold", "Ransac samples"]
    assert [(p[0], p[3], p[4]) for p in det.props()] == [("Min size", 0.01, 1.0), ("Subdivisions", 1, 16), ("Target motion", 0.0001, 0.1)]
    assert det.set_prop("Subdivisions", 5) and det.subdivide == 5 and not det.set_prop("nope", 1)


def test_decoder_detector_loop_semantics():
    """process_frame appends (callers clear), first frame -> False, end of stream -> error (decoder.rs:45-60)."""
    from ofps_amd.plugins import HipBlockMotionDetection, HipSadDecoder
    fr = synth.luma_sequence(4, 320, 192, max_step=8)
    dec, det = HipSadDecoder(list(fr), framerate=30.0), HipBlockMotionDetection()
    dec.range = 8
    field = []
    assert dec.process_frame(field) is False and field == []
    assert dec.get_aspect() == (320, 192) and dec.get_framerate() == 30.0
    for k in (1, 2, 3):
        field.clear()
        assert dec.process_frame(field) is True
        ent_o, _ = oracle.sad_flow(fr[k - 1], fr[k], 16, 8)
        np.testing.assert_array_equal(np.array(field, np.float32).view(np.uint32), ent_o.view(np.uint32))
        r, ro = det.detect_motion(field), oracle.detect_motion(ent_o)
        assert (r is None) == (ro is None)
        if r is not None:
            assert r[0] == ro[0] and r[1].dim() == (14, 14)
            np.testing.assert_array_equal(r[1].vf.view(np.uint32), ro[1].view(np.uint32))
    n = len(field)
    with pytest.raises(EOFError):
        dec.process_frame(field)
    assert len(field) == n


def test_decoder_skip_frames_and_geometry_change():
    """skip_frames = n reads n+1 frames; the vectors relate the last two read (decoder.rs:47-60).  Only the new frame
    is uploaded per call (pinned buffer + device-side previous frame), which must not change any result."""
    from ofps_amd.plugins import HipSadDecoder
    fr = synth.luma_sequence(8, 256, 144, max_step=4, seed=21)
    small = synth.luma_sequence(2, 128, 96, max_step=4, seed=22)
    dec = HipSadDecoder(list(fr) + list(small))
    dec.range = 8
    field = []
    assert dec.process_frame(field, skip_frames=2) is True             # reads 0,1,2 -> pair (1,2)
    np.testing.assert_array_equal(np.array(field, np.float32).view(np.uint32), oracle.sad_flow(fr[1], fr[2], 16, 8)[0].view(np.uint32))
    field.clear()
    out = []
    assert dec.process_frame(field, out_frame=out, skip_frames=1) is True   # reads 3,4 -> pair (3,4)
    np.testing.assert_array_equal(out[0], fr[4])
    np.testing.assert_array_equal(np.array(field, np.float32).view(np.uint32), oracle.sad_flow(fr[3], fr[4], 16, 8)[0].view(np.uint32))
    field.clear()
    assert dec.process_frame(field) is True                             # reads 5 -> pair (4,5)
    np.testing.assert_array_equal(np.array(field, np.float32).view(np.uint32), oracle.sad_flow(fr[4], fr[5], 16, 8)[0].view(np.uint32))
    field.clear()
    assert dec.process_frame(field, skip_frames=1) is True              # reads 6,7 -> pair (6,7)
    np.testing.assert_array_equal(np.array(field, np.float32).view(np.uint32), oracle.sad_flow(fr[6], fr[7], 16, 8)[0].view(np.uint32))
    field.clear()
    assert dec.process_frame(field) is False and field == []            # geometry change restarts the stream
    assert dec.get_aspect() == (128, 96)
    assert dec.process_frame(field) is True
    np.testing.assert_array_equal(np.array(field, np.float32).view(np.uint32), oracle.sad_flow(small[0], small[1], 16, 8)[0].view(np.uint32))


def test_decoder_exact_pruning_property_returns_the_same_vectors():
    from ofps_amd.plugins import HipSadDecoder
    fr = synth.luma_sequence(4, 640, 368, max_step=10, seed=5, region=4096, noise=1)      # a camera pan
    a, b = HipSadDecoder(list(fr)), HipSadDecoder(list(fr))
    assert b.set_prop("Exact pruning", True) and b.pruned is True
    for k in range(4):
        fa, fb = [], []
        assert a.process_frame(fa) == b.process_frame(fb)
        assert len(fa) == len(fb)
        if fa:
            np.testing.assert_array_equal(np.array(fa, np.float32).view(np.uint32), np.array(fb, np.float32).view(np.uint32))
            np.testing.assert_array_equal(np.array(fb, np.float32).view(np.uint32), oracle.sad_flow(fr[k - 1], fr[k], 16, 16)[0].view(np.uint32))
