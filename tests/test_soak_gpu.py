"""Short runs of the two stream soaks (tools/lk_soak.py, tools/pipeline_soak.py): streams with tickets in flight while other entry
points use the same context; every frame compared with an undisturbed context.  The long runs are in profiles/r04/*_soak.txt."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, *args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *args], capture_output=True, text=True, timeout=600, env=e)


def test_lk_read_ahead_stream_soak_with_other_stages_in_between():
    r = _run("lk_soak.py", "400")
    assert r.returncode == 0 and "mismatching frames 0, tiles computed by a waiting child 0" in r.stdout, r.stdout + r.stderr


def test_fused_per_frame_stream_soak_with_other_entry_points_in_between():
    r = _run("pipeline_soak.py", "240")
    assert r.returncode == 0 and "mismatching frames 0" in r.stdout, r.stdout + r.stderr
    r = _run("pipeline_soak.py", "24", env={"SOAK_NEGATIVE_CONTROL": "1"})          # the comparison does see a wrong frame
    assert r.returncode == 1 and "mismatching frames 23" in r.stdout, r.stdout + r.stderr


def test_hip_flow_read_ahead_stream_soak_with_other_stages_in_between():
    """hip_flow with cv-decoder's flags (previous flow as initial flow) and the kept expansion, restarts, LK / Farneback / SAD / densify calls
    in between, against the same stream run synchronously on an idle context.  4,000 frames: profiles/r05/flow_soak.txt."""
    r = _run("flow_soak.py", "700")
    assert r.returncode == 0 and "mismatching frames 0" in r.stdout, r.stdout + r.stderr
    r = _run("flow_soak.py", "40", env={"SOAK_NEGATIVE_CONTROL": "1"})                 # the comparison does see a stream that forgets its flow
    assert r.returncode == 1 and "mismatching frames 38" in r.stdout, r.stdout + r.stderr


def test_stateful_api_fuzz_on_one_context():
    """1,500 random entry-point calls on one context, each compared with the same call on an otherwise idle context (state leaking
    from one call into another: scratch slots, flags, tickets, rings).  100,000 calls: profiles/r04/api_fuzz.txt."""
    r = _run("api_fuzz.py", "1500", "11")
    assert r.returncode == 0 and "mismatches 0 []" in r.stdout, r.stdout + r.stderr


def test_multi_device_stream_fuzz_on_one_gpu():
    """Random batch sizes, batches in flight, restarts and source kinds through 1, 2 and 3 workers on device 0 against one plain
    context replaying the stream.  63,000 frames: profiles/r04/api_fuzz.txt."""
    r = _run("multi_fuzz.py", "240", "5")
    assert r.returncode == 0 and "mismatching frames 0" in r.stdout, r.stdout + r.stderr


def test_hostile_arguments_get_error_codes_and_leave_the_context_intact():
    r = _run("badarg_fuzz.py")
    assert r.returncode == 0 and "gives a fresh context's bits: True" in r.stdout, r.stdout + r.stderr

