"""C++ host layer (ofps_amd/host): .mvec round trip on CPU; decoder/detector/estimator loops on the GPU."""
import io
import json
import os
import subprocess

import numpy as np
import pytest

from ofps_amd import build as hip_build, mvec, synth

TOOL = hip_build.TOOL


def _tool(*args):
    if not os.path.exists(TOOL):
        hip_build.build_host()
    return subprocess.run([TOOL, *map(str, args)], check=True, capture_output=True, text=True).stdout


def test_mvec_python_round_trip():
    rng = np.random.default_rng(0)
    frames = [rng.normal(size=(n, 4)).astype(np.float32) for n in (5, 0, 880)]
    buf = io.BytesIO()
    for fr in frames:
        mvec.write_frame(buf, fr)
    assert len(buf.getvalue()) == sum(4 + 16 * len(fr) for fr in frames)      # no header, frames back to back
    buf.seek(0)
    back = list(mvec.read_frames(buf))
    assert len(back) == 3
    for a, b in zip(frames, back):
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    with pytest.raises(EOFError):
        list(mvec.read_frames(io.BytesIO(buf.getvalue()[:-3])))


def test_mvec_cpp_reader_writer_round_trip(tmp_path):
    """MvecFile reader (motion-loader/src/lib.rs:46-66) + writer (motion-extract/src/main.rs:23-35) in C++."""
    rng = np.random.default_rng(1)
    src, dst = tmp_path / "a.mvec", tmp_path / "b.mvec"
    with open(src, "wb") as f:
        for n in (3, 0, 17, 8040):
            mvec.write_frame(f, rng.normal(size=(n, 4)).astype(np.float32))
    out = json.loads(_tool("mvec-copy", src, dst))
    assert out == {"frames": 4, "vectors": 3 + 17 + 8040}
    assert open(src, "rb").read() == open(dst, "rb").read()


@pytest.mark.gpu
def test_extract_detect_track_through_cpp_host(tmp_path):
    import oracle
    W, H, B, R, F = 320, 192, 16, 16, 5
    fr = synth.luma_sequence(F, W, H, max_step=R)
    raw = tmp_path / "clip.y"
    raw.write_bytes(fr.tobytes())
    out = tmp_path / "clip.mvec"
    info = json.loads(_tool("extract", "hip_sad", f"{raw}?w={W}&h={H}&fps=30", out))
    nblk = (W // B) * (H // B)
    assert info == {"frames": F, "vectors": (F - 1) * nblk}
    frames = list(mvec.read_frames(open(out, "rb")))
    assert len(frames[0]) == 0                                                  # first frame: Ok(false), count 0
    for k in range(1, F):
        ent_o, _ = oracle.sad_flow(fr[k - 1], fr[k], B, R)
        np.testing.assert_array_equal(frames[k].view(np.uint32), ent_o.view(np.uint32))

    # detection loop over the .mvec file (MvecFile decoder -> hip_block_motion) vs the oracle's detector
    det = json.loads(_tool("detect", "mvec", out))
    ranges = []
    for k, e in enumerate(frames, start=1):                                     # frames counter is 1-based after ++
        if oracle.detect_motion(e) is not None:
            if ranges and ranges[-1][1] == k:
                ranges[-1][1] += 1
            else:
                ranges.append([k, k + 1])
    merged = []
    for s, e in ranges:
        if merged and s - merged[-1][1] <= 2:
            merged[-1][1] = e
        else:
            merged.append([s, e])
    assert det["frames"] == F and det["motion_ranges"] == [r for r in merged if r[1] - r[0] >= 2]

    # tracking loop (LSQ) vs pose accumulation with the oracle's estimator
    csv = _tool("track", "mvec", out, 16 / 9, 22.275, "lsq").strip().splitlines()
    assert csv[0].startswith("frame,w,i,j,k") and len(csv) == F + 1
    cam = oracle.camera(16 / 9, 22.275)
    rot = np.array([1, 0, 0, 0], np.float64)

    def qmul(a, b):
        aw, ai, aj, ak = a; bw, bi, bj, bk = b
        return np.array([aw * bw - ai * bi - aj * bj - ak * bk, aw * bi + ai * bw + aj * bk - ak * bj,
                         aw * bj - ai * bk + aj * bw + ak * bi, aw * bk + ai * bj - aj * bi + ak * bw])
    for k, e in enumerate(frames):
        if len(e):
            rot = qmul(oracle.solve_ypr_given(e, cam).astype(np.float64), rot)  # rot = r * rot
        got = np.array([float(x) for x in csv[k + 1].split(",")[1:5]])
        np.testing.assert_allclose(got, rot, atol=2e-5)


@pytest.mark.gpu
def test_hip_lk_decoder_through_cpp_host(tmp_path):
    import oracle
    W, H, F = 320, 180, 3
    fr = synth.flatten_regions(synth.luma_sequence(F, W, H, max_step=2, seed=4), region=40, seed=1)
    raw = tmp_path / "clip.y"
    raw.write_bytes(fr.tobytes())
    out = tmp_path / "clip_lk.mvec"
    info = json.loads(_tool("extract", "hip_lk", f"{raw}?w={W}&h={H}", out))
    frames = list(mvec.read_frames(open(out, "rb")))
    assert info["frames"] == F and len(frames[0]) == 0
    for k in range(1, F):
        rec = oracle.masked_flow_to_entries(oracle.lk_flow(fr[k - 1], fr[k], 3, 4, 3), oracle.contrast_mask(fr[k]))
        e_o = oracle.densify_to_entries(rec, 150, 84)             # the plugin masks by default, like cv-decoder
        assert 0 < len(e_o) < 150 * 84
        np.testing.assert_array_equal(frames[k].view(np.uint32), e_o.view(np.uint32))
