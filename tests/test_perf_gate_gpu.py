"""The performance gate on the GPU box: one full `python bench.py` (N = 1, every leg), checked against the committed line of the
PREVIOUS round (profiles/r05/bench_n1.json: never a line of the round being gated; slower-only, 5 % on the device-timed medians -- 10 % / 8 % on the two LK rows (the LK kernel's time moves from box to box: 9 % / 3.4 % over ten boxes, +-1 % within one), 8 % on the native read-ahead, whose two DMA-engine modes are 4 % apart --, 15 % on
the best of three processes' cfg5 p50) and against the relations that must hold inside one run (tools/perf_gate.py; VERDICT r4
item 2, r5 item 6).  The line that was gated is kept under gpurun_out/."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_line_passes_the_perf_gate():
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "perf_gate.py"), "--run", "--save", os.path.join(out_dir, "perf_gate_line.json")],
                       capture_output=True, text=True, timeout=1700)
    with open(os.path.join(out_dir, "perf_gate.txt"), "w") as f:
        f.write(p.stdout + p.stderr)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1500:]
