"""Real-compute multi-rank rehearsal on ONE GPU: `bench.py --gpus 2 --backend gloo --device-map 0,0` starts two
ranks under torch.distributed.run, each with its own HIP context on GPU 0 running the real search (and tail), with gloo
carrying the barrier, ranks_seen, gather_results and the key-frame broadcast through host tensors -- everything an 8-GPU
RCCL run does except the transport.  The reference's counterpart is its worker-thread model
(ofps-suite/src/app/tracking/worker.rs:251-260,347-352)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(*argv, timeout=900):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "LOCAL_WORLD_SIZE"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv, "--no-cpu-baseline", "--no-end-to-end", "--no-legs"],
                       capture_output=True, text=True, timeout=timeout, env=e)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_two_ranks_on_one_gpu_cfg4_strong_scaling_with_the_records_on_the_host():
    """BASELINE configs[3]'s geometry (4K, 8x8, +-32), one global batch split over two ranks, per-pair checksums gathered in
    pair order, and the D2H-inclusive step time beside the device-resident one."""
    p, d = _run("--gpus", "2", "--backend", "gloo", "--device-map", "0,0", "--config", "cfg4", "--pairs", "6", "--gen-pairs", "3",
                "--steps", "3", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-3000:]
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "strong" and d["data"] == "synthetic"
    assert d["config"]["pairs_per_rank"] == [3, 3] and d["config"]["workload"].startswith("cfg4")
    pc = d["parity_check"]
    assert pc["ok"] is True and pc["ranks_checked"] == 2 and pc["gathered_checksum_of_last_pair_matches_oracle"] is True
    assert d["ms_per_step_with_d2h"] >= 0.9 * d["ms_per_step"]
    assert d["with_d2h"]["host_copy_equals_device_records"] is True
    assert d["with_d2h"]["bytes_d2h_per_rank_and_step"] == 3 * 129600 * 16
    assert "rehearsal" in d and "gloo" in d["launcher"]


def test_two_ranks_on_one_gpu_key_frame_broadcast_and_the_fused_tail():
    """--ref-mode key: rank 1's key-frame slot starts zeroed, only the in-step broadcast from rank 0 can fill it (its pairs would
    not match the oracle otherwise); --pipeline adds detector + estimator to the step and their per-pair tables to the gather."""
    p, d = _run("--gpus", "2", "--backend", "gloo", "--device-map", "0,0", "--scaling", "strong", "--pairs", "7", "--gen-pairs", "4",
                "--ref-mode", "key", "--pipeline", "--steps", "3", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-3000:]
    assert d["ranks_seen"] == 2 and d["config"]["pairs_per_rank"] == [4, 3] and d["config"]["ref_mode"] == "key"
    assert "broadcast" in d["config"]["parallelism"] and "almeida" in d["config"]["step"]
    pc = d["parity_check"]
    assert pc["ok"] is True and pc["ranks_checked"] == 2 and pc["gathered_checksum_of_last_pair_matches_oracle"] is True
    assert d["pipeline"]["ms_per_step"] > 0


def test_one_rank_cfg4_line_carries_both_times():
    p, d = _run("--gpus", "1", "--config", "cfg4", "--pairs", "4", "--gen-pairs", "2", "--steps", "3", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-3000:]
    assert d["n_gpus"] == 1 and "ms_per_step_with_d2h" in d and d["with_d2h"]["host_copy_equals_device_records"] is True
    assert d["parity_check"]["ok"] is True
