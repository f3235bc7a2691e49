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


@pytest.mark.gpu
def test_hip_flow_decoder_through_cpp_host(tmp_path):
    """create_decoder("hip_flow"): the Farneback flow with cv-decoder's arguments (cv-decoder/src/lib.rs:188-199) behind the
    same Decoder surface; records equal the oracle chain farneback_flow -> masked records -> densifier down-sampling."""
    import oracle
    W, H, F = 320, 180, 3
    fr = synth.flatten_regions(synth.luma_sequence(F, W, H, max_step=2, seed=5), region=40, seed=2)
    raw = tmp_path / "clip.y"
    raw.write_bytes(fr.tobytes())
    out = tmp_path / "clip_flow.mvec"
    info = json.loads(_tool("extract", "hip_flow", f"{raw}?w={W}&h={H}", out))
    frames = list(mvec.read_frames(open(out, "rb")))
    assert info["frames"] == F and len(frames[0]) == 0
    flow = None
    for k in range(1, F):
        flow = oracle.farneback_flow(fr[k - 1], fr[k], 5, 13, 3, 7, 1.5, init=flow)      # cv-decoder/src/lib.rs:161-165: the previous flow is the initial flow
        rec = oracle.masked_flow_to_entries(flow, oracle.contrast_mask(fr[k]))
        e_o = oracle.densify_to_entries(rec, 150, 84)
        assert 0 < len(e_o) < 150 * 84
        np.testing.assert_array_equal(frames[k].view(np.uint32), e_o.view(np.uint32))


# ---- (f)2 surface added in round 2: tcp:// inputs, saved configurations, perf CSV export ------------------------------
def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _mvec_bytes(frames):
    buf = io.BytesIO()
    for fr in frames:
        mvec.write_frame(buf, fr)
    return buf.getvalue()


def test_open_file_tcp_connect(tmp_path):
    """create_decoder("mvec", "tcp://host:port"): the tool connects to a listening peer (ofps/src/utils.rs:107-110)."""
    import socket, threading
    rng = np.random.default_rng(2)
    payload = _mvec_bytes([rng.normal(size=(n, 4)).astype(np.float32) for n in (7, 0, 8040, 33)])
    srv = socket.socket(); srv.bind(("127.0.0.1", 0)); srv.listen(1)
    port = srv.getsockname()[1]

    def feed():
        c, _ = srv.accept()
        c.sendall(payload); c.close(); srv.close()
    th = threading.Thread(target=feed); th.start()
    dst = tmp_path / "out.mvec"
    out = _tool("extract", "mvec", f"tcp://127.0.0.1:{port}", dst)
    th.join()
    assert "Connecting to 127.0.0.1" in out and "Got stream!" in out
    assert json.loads(out.strip().splitlines()[-1]) == {"frames": 4, "vectors": 7 + 8040 + 33}
    assert dst.read_bytes() == payload


def test_open_file_tcp_listen(tmp_path):
    """"tcp://@:port": the tool listens on 0.0.0.0:port and accepts one connection (utils.rs:100-106)."""
    import socket, time
    rng = np.random.default_rng(3)
    payload = _mvec_bytes([rng.normal(size=(n, 4)).astype(np.float32) for n in (880, 880)])
    port = _free_port()
    dst = tmp_path / "out.mvec"
    if not os.path.exists(TOOL):
        hip_build.build_host()
    proc = subprocess.Popen([TOOL, "extract", "mvec", f"tcp://@:{port}", str(dst)], stdout=subprocess.PIPE, text=True)
    for _ in range(100):
        try:
            c = socket.create_connection(("127.0.0.1", port), timeout=1.0)
            break
        except OSError:
            time.sleep(0.05)
    else:
        proc.kill(); pytest.fail("the tool never listened")
    c.sendall(payload); c.close()
    out, _ = proc.communicate(timeout=30)
    assert proc.returncode == 0 and "Accept 127.0.0.1" in out
    assert dst.read_bytes() == payload


def test_open_file_bad_tcp_spec_fails_loudly(tmp_path):
    p = subprocess.run([TOOL, "extract", "mvec", "tcp://nocolon", str(tmp_path / "x")], capture_output=True, text=True)
    assert p.returncode != 0 and "Invalid format" in p.stderr


SAVED_CONFIG = {     # MotionDetectionConfig as serde writes it (tuples = arrays, () = null)
    "decoder": [{"selected_plugin": "mvec", "arg": "@ARG@", "extra": False}, True],
    "detector": [{"selected_plugin": "hip_block_motion", "arg": "", "extra": None}, True],
    "settings": {"worker": {"decoder_properties": {}, "realtime_processing": False,
                            "detector_properties": {"Min size": {"Float": {"val": 0.02, "min": 0.01, "max": 1.0}},
                                                    "Subdivisions": {"Usize": {"val": 4, "min": 1, "max": 16}},
                                                    "Target motion": {"Float": {"val": 0.004, "min": 0.0001, "max": 0.1}},
                                                    "Not a property of this plugin": {"Bool": True}}},
                 "overlay_mf": False, "max_frame_gap": 3, "min_frames": 1,
                 "draw_perf_stats": {"summary_window": False, "graph_window": False}},
}


def test_saved_config_is_parsed_like_serde_would(tmp_path):
    cfg = tmp_path / "basic_detect.json"
    cfg.write_text(json.dumps(SAVED_CONFIG, indent=2).replace("@ARG@", "clip with \\\"quotes\\\".mvec"))
    got = json.loads(_tool("parse-config", cfg).replace('with "quotes"', "with 'quotes'"))
    assert got["decoder"] == ["mvec", "clip with 'quotes'.mvec", True] and got["detector"] == ["hip_block_motion", "", True]
    assert (got["max_frame_gap"], got["min_frames"], got["overlay_mf"], got["realtime_processing"]) == (3, 1, False, False)
    dp = got["detector_properties"]
    assert dp["Subdivisions"] == ["Usize", 4, 1, 16] and dp["Not a property of this plugin"] == ["Bool", True]
    assert dp["Min size"][0] == "Float" and abs(dp["Min size"][1] - 0.02) < 1e-7
    # malformed input is an error, not a default
    bad = tmp_path / "bad.json"
    bad.write_text('{"decoder": [{"selected_plugin": "mvec"}, true]}')
    p = subprocess.run([TOOL, "parse-config", str(bad)], capture_output=True, text=True)
    assert p.returncode != 0 and "config" in p.stderr


@pytest.mark.gpu
def test_detect_from_saved_config_with_perf_csv(tmp_path):
    """detect --config: plugins by name, saved properties transferred (detection.rs:100-109), ranges filtered with the
    saved max_frame_gap / min_frames, per-frame times exported as perf_<name>_<decoder>.csv (perf_stats.rs:86-121)."""
    import oracle
    W, H, B, R, F = 320, 192, 16, 16, 6
    fr = synth.luma_sequence(F, W, H, max_step=R)
    frames = [np.zeros((0, 4), np.float32)] + [oracle.sad_flow(fr[k - 1], fr[k], B, R)[0] for k in range(1, F)]
    clip = tmp_path / "clip.mvec"
    clip.write_bytes(_mvec_bytes(frames))
    cfg = tmp_path / "cfg.json"
    cfg.write_text(json.dumps(SAVED_CONFIG).replace("@ARG@", str(clip)))
    perf = tmp_path / "perf"; perf.mkdir()
    det = json.loads(_tool("detect", "--config", cfg, "--perf-csv", perf).strip().splitlines()[-1])
    assert det["frames"] == F and det["detector_properties"]["Subdivisions"] == 4
    assert abs(det["detector_properties"]["Min size"] - 0.02) < 1e-7 and abs(det["detector_properties"]["Target motion"] - 0.004) < 1e-7
    ranges = []
    for k, e in enumerate(frames, start=1):
        if oracle.detect_motion(e, 0.02, 4, 0.004) is not None:
            if ranges and ranges[-1][1] == k:
                ranges[-1][1] += 1
            else:
                ranges.append([k, k + 1])
    merged = []
    for s, e in ranges:
        if merged and s - merged[-1][1] <= 3:
            merged[-1][1] = e
        else:
            merged.append([s, e])
    assert det["motion_ranges"] == [r for r in merged if r[1] - r[0] >= 1]
    for name in ("perf_decoder_mvec.csv", "perf_hip_block_motion_mvec.csv"):
        rows = (perf / name).read_text().strip().splitlines()
        assert len(rows) == F and all(float(r) >= 0 for r in rows)


@pytest.mark.gpu
def test_hip_sad_decoder_over_tcp_and_native_stream_bench(tmp_path):
    """cfg5's input shape through the C++ host: raw luma frames over a socket into create_decoder("hip_sad", "tcp://...")."""
    import socket, threading
    import oracle
    W, H, B, R, F = 320, 192, 16, 16, 4
    fr = synth.luma_sequence(F, W, H, max_step=R)
    srv = socket.socket(); srv.bind(("127.0.0.1", 0)); srv.listen(1)
    port = srv.getsockname()[1]

    def feed():
        c, _ = srv.accept()
        c.sendall(fr.tobytes()); c.close(); srv.close()
    th = threading.Thread(target=feed); th.start()
    out = tmp_path / "tcp.mvec"
    info = json.loads(_tool("extract", "hip_sad", f"tcp://127.0.0.1:{port}?w={W}&h={H}&fps=60", out).strip().splitlines()[-1])
    th.join()
    assert info["frames"] == F
    frames = list(mvec.read_frames(open(out, "rb")))
    for k in range(1, F):
        np.testing.assert_array_equal(frames[k].view(np.uint32), oracle.sad_flow(fr[k - 1], fr[k], B, R)[0].view(np.uint32))
    r = json.loads(_tool("stream-bench", 640, 360, 50, "ahead"))
    assert r["mode"] == "read_ahead" and r["ms_per_frame"] > 0


# ---- two host processes on one GPU ---------------------------------------------------------------------------------
# The cluster Almeida solver is a persistent launch of co-resident workgroups; launches of different contexts are chained
# inside a process, but two PROCESSES can each hold part of the CUs.  Its spins are bounded and a launch that gave up
# finishes the estimate by itself (almeida.hip: almeida_solo_solve), so neither process may ever see a NaN or a wrong
# rotation -- the reference's estimator cannot fail either (almeida-estimator/src/lib.rs:181-185,246-250).
@pytest.mark.gpu
def test_two_tracking_processes_share_one_gpu(tmp_path):
    """Two `ofps_hip_tool track hip_sad` processes at 1080p (8,040 vectors per frame -> the 8-workgroup cluster solver)
    running at the same time: both CSVs equal the pose accumulation of the oracle's estimator on the oracle's vectors."""
    import oracle
    W, H, B, R, F = 1920, 1080, 16, 16, 6
    clips = [synth.luma_sequence(F, W, H, max_step=R, seed=synth.SEED0 + 300 + p) for p in range(2)]
    procs = []
    for p, fr in enumerate(clips):
        raw = tmp_path / f"clip{p}.y"
        raw.write_bytes(fr.tobytes())
        procs.append(subprocess.Popen([TOOL, "track", "hip_sad", f"{raw}?w={W}&h={H}&fps=60", str(16 / 9), "22.275", "lsq"],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    cam = oracle.camera(16 / 9, 22.275)

    def qmul(a, b):
        aw, ai, aj, ak = a; bw, bi, bj, bk = b
        return np.array([aw * bw - ai * bi - aj * bj - ak * bk, aw * bi + ai * bw + aj * bk - ak * bj,
                         aw * bj - ai * bk + aj * bw + ak * bi, aw * bk + ai * bj - aj * bi + ak * bw])
    for p, (fr, (so, se)) in enumerate(zip(clips, outs)):
        assert procs[p].returncode == 0, se[-1000:]
        csv = so.strip().splitlines()
        assert len(csv) == F + 1
        rot = np.array([1, 0, 0, 0], np.float64)
        for k in range(F):
            if k:
                ent_o, _ = oracle.sad_flow(fr[k - 1], fr[k], B, R, threads=4)
                rot = qmul(oracle.solve_ypr_given(ent_o, cam).astype(np.float64), rot)
            got = np.array([float(x) for x in csv[k + 1].split(",")[1:5]])
            assert np.isfinite(got).all()
            np.testing.assert_allclose(got, rot, atol=2e-5)


_CONTEND_WORKER = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from ofps_amd.runtime import HipContext
e = np.load(sys.argv[2]); q_o = np.load(sys.argv[3]); iters = int(sys.argv[4])
ctx = HipContext(0)
worst, finite = 0.0, True
for _ in range(iters):
    q, _ = ctx.almeida(e, 16 / 9, 22.275, use_ransac=False)
    finite = finite and bool(np.isfinite(q).all())
    worst = max(worst, float(np.abs(q - q_o).max())) if finite else float("inf")
print(json.dumps({"worst": worst, "finite": finite, "recoveries": ctx.almeida_recoveries()}))
"""


@pytest.mark.gpu
def test_two_processes_contending_for_every_cu_never_see_a_nan(tmp_path):
    """Each process solves 2,073,600-record fields (254 co-resident workgroups of 1024 threads: one per CU) back to back,
    so the two persistent launches really do compete for CUs.  Whatever the interleaving -- including launches whose
    workgroups never all became resident, which finish through the in-kernel recovery -- every estimate is finite and
    within 2e-6 of the oracle's."""
    import sys
    import oracle
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = synth.rotation_field(1920, 1080)
    q_o = oracle.solve_ypr_given(e, oracle.camera(16 / 9, 22.275), threads=min(16, oracle.num_threads()))
    np.save(tmp_path / "e.npy", e); np.save(tmp_path / "q.npy", q_o)
    script = tmp_path / "worker.py"
    script.write_text(_CONTEND_WORKER)
    procs = [subprocess.Popen([sys.executable, str(script), root, str(tmp_path / "e.npy"), str(tmp_path / "q.npy"), "12"],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    total_recoveries = 0
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-1500:]
        r = json.loads(so.strip().splitlines()[-1])
        assert r["finite"] and r["worst"] <= 2e-6, r
        total_recoveries += r["recoveries"]
    print(f"[two-process contention] in-kernel recoveries: {total_recoveries}")


_LK_CONTEND_WORKER = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from ofps_amd.runtime import HipContext
fr = np.load(sys.argv[2]); want = np.load(sys.argv[3])
ctx = HipContext(0)
bad = 0
for k in range(int(sys.argv[4])):
    f = ctx.lk_flow(fr[0], fr[1], 3, 4, 3)
    bad += int((f.view(np.uint32) != want.view(np.uint32)).any())
print(json.dumps({"runs": int(sys.argv[4]), "mismatching_runs": bad, "helped_tiles": ctx.lk_helped_tiles()}))
"""


@pytest.mark.gpu
def test_two_processes_running_the_one_launch_pyramid_flow_get_the_oracles_bits(tmp_path):
    """lk_levels_kernel runs the whole pyramid in one launch: a tile waits until its parent tile of the coarser level -- a workgroup
    of the same launch -- has published its flows, and computes it itself when that takes too long.  Two PROCESSES running 1080p flows
    back to back on one GPU share its CUs, so each launch's workgroups become resident in whatever order the two launches interleave;
    whatever happens, every run of either process returns the oracle's flow bit for bit (helped tiles cost time, never bits)."""
    import sys
    import oracle
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fr = synth.luma_sequence(2, 1920, 1080, max_step=16, seed=synth.SEED0 + 91)       # region jumps: unfit tiles at level 0 too
    np.save(tmp_path / "fr.npy", fr); np.save(tmp_path / "flow.npy", oracle.lk_flow(fr[0], fr[1], 3, 4, 3))
    script = tmp_path / "worker.py"
    script.write_text(_LK_CONTEND_WORKER)
    procs = [subprocess.Popen([sys.executable, str(script), root, str(tmp_path / "fr.npy"), str(tmp_path / "flow.npy"), "40"],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-1500:]
        r = json.loads(so.strip().splitlines()[-1])
        assert r["runs"] == 40 and r["mismatching_runs"] == 0, r


@pytest.mark.gpu
def test_multi_extract_writes_the_same_mvec_as_the_single_decoder(tmp_path):
    """ofps_hip_multi_* behind the C++ host layer (MultiDeviceSad): three workers on the one GPU of the box split a clip's
    pairs; the .mvec file equals the one `extract hip_sad` writes frame by frame, byte for byte."""
    W, H, F = 320, 192, 7
    fr = synth.luma_sequence(F, W, H, max_step=16, seed=33)
    raw = tmp_path / "clip.y"
    raw.write_bytes(fr.tobytes())
    one = tmp_path / "one.mvec"; many = tmp_path / "many.mvec"
    json.loads(_tool("extract", "hip_sad", f"{raw}?w={W}&h={H}&fps=30", one))
    info = json.loads(_tool("multi-extract", raw, W, H, many, 0, 0, 0))
    assert info["workers"] == 3 and info["frames"] == F
    assert one.read_bytes() == many.read_bytes()
