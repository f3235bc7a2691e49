"""tools/external_parity: the kit a maintainer with a Rust toolchain runs against the real OFPS to pin A1-A5 (and SURVEY.md
Appendix A.6).  Here: the committed expectations are what the oracle says today, the comparer accepts them and rejects a
flipped bit, and (on the GPU box) the HIP path produces the same numbers."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIT = os.path.join(ROOT, "tools", "external_parity")
DATA = os.path.join(KIT, "data")


def _expected(clip):
    return json.load(open(os.path.join(DATA, f"expected_clip{clip}.json")))


def _frames(clip):
    from ofps_amd import mvec
    return list(mvec.read_frames(open(os.path.join(DATA, f"clip{clip}.mvec"), "rb")))


def test_committed_expectations_are_the_oracles():
    import oracle
    cam = oracle.camera(16 / 9, 39.6 * 9 / 16)
    for clip in range(3):
        exp = _expected(clip)
        frames = _frames(clip)
        assert len(frames) == len(exp["frames"]) == 6 and len(frames[0]) == 0
        for k in (1, 3, 5):
            e, x = frames[k], exp["frames"][k]
            f14, cells = oracle.densify(e, 14, 14, want_cells=True)
            assert [[int(c[0]), int(c[1])] for c in cells] == x["cells"]
            assert [int(v) for v in f14.view(np.uint32).ravel()] == x["field_14x14"]
            assert [int(v) for v in oracle.densify(e, 60, 34).view(np.uint32).ravel()] == x["field_60x34"]
            d = oracle.detect_motion(e)
            assert (None if d is None else int(d[0])) == x["detect_area"]
            q = oracle.solve_ypr_given(e, cam)
            if np.isfinite(q).all():
                np.testing.assert_array_equal(q, np.array(x["quat"], np.float32))
    # the kit exercises both detector outcomes and the clamp-collapsed positions
    assert any(f["detect_area"] is not None for f in _expected(0)["frames"]) and all(f["detect_area"] is None for f in _expected(1)["frames"])
    assert [0, 0] in _expected(2)["frames"][3]["cells"][:7] and [13, 13] in _expected(2)["frames"][3]["cells"][:7]


def test_comparer_accepts_the_expectations_and_rejects_a_flipped_bit(tmp_path):
    lines = [json.dumps(fr) for c in range(3) for fr in _expected(c)["frames"]]
    ok = tmp_path / "ok.jsonl"; ok.write_text("\n".join(lines))
    r = subprocess.run([sys.executable, os.path.join(KIT, "compare.py"), DATA, str(ok)], capture_output=True, text=True)
    assert r.returncode == 0 and "all frames agree" in r.stdout
    bad = json.loads(lines[8]); bad["cells"][5][0] ^= 1; lines[8] = json.dumps(bad)
    badf = tmp_path / "bad.jsonl"; badf.write_text("\n".join(lines))
    r = subprocess.run([sys.executable, os.path.join(KIT, "compare.py"), DATA, str(badf)], capture_output=True, text=True)
    assert r.returncode == 1 and "MISMATCH clip 1 frame 2: cells" in r.stdout


@pytest.mark.gpu
def test_hip_path_produces_the_kit_s_expectations():
    from ofps_amd.runtime import HipContext
    ctx = HipContext(0)
    try:
        for clip in range(3):
            exp = _expected(clip)
            for k, e in enumerate(_frames(clip)):
                if not len(e):
                    continue
                x = exp["frames"][k]
                f14, cells = ctx.densify(e, 14, 14, want_cells=True)
                assert [[int(c[0]), int(c[1])] for c in cells] == x["cells"]
                assert [int(v) for v in f14.view(np.uint32).ravel()] == x["field_14x14"]
                assert [int(v) for v in ctx.densify(e, 60, 34).view(np.uint32).ravel()] == x["field_60x34"]
                d = ctx.detect(e)
                assert (None if d is None else int(d[0])) == x["detect_area"]
                if d is not None:
                    assert [int(v) for v in np.asarray(d[1], np.float32).view(np.uint32).ravel()] == x["detect_field"]
                q, _ = ctx.almeida(e, 16 / 9, 39.6 * 9 / 16, use_ransac=False)
                if np.isfinite(np.array(x["quat"])).all():
                    np.testing.assert_allclose(q, np.array(x["quat"], np.float32), atol=2e-6, rtol=0)
    finally:
        ctx.close()
