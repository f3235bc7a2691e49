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


# ---- the exact BASELINE configs[3] shapes, 8 ways, on one GPU (the reference's threading model:
# ofps-suite/src/app/tracking/worker.rs:251-260,347-352).  Every line is kept under gpurun_out/r05/ for profiles/r05/.

def _keep(name, d):
    out = os.path.join(ROOT, "gpurun_out", "r05")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, name), "w") as f:
        json.dump(d, f)
        f.write("\n")


def test_eight_ranks_cfg4_64_pairs_8_per_rank_with_the_records_on_the_host():
    """64 pairs of 4K, 8x8 blocks, +-32 -> 8 pairs (9 frames) per rank, per-rank oracle parity, checksum of pair 63 (searched by rank 7)
    through the pair-ordered gather, and the D2H-inclusive step."""
    p, d = _run("--gpus", "8", "--backend", "gloo", "--device-map", "0,0,0,0,0,0,0,0", "--config", "cfg4", "--steps", "2", "--warmup", "1",
                timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    _keep("rehearsal_8ranks_cfg4_strong.json", d)
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["scaling"] == "strong"
    assert d["config"]["pairs_per_rank"] == [8] * 8 and d["config"]["pairs_per_step"] == 64 and d["config"]["vectors_per_pair"] == 129600
    pc = d["parity_check"]
    assert pc["ok"] is True and pc["ranks_checked"] == 8 and pc["ranks_skipped"] == 0
    assert pc["gathered_checksum_of_last_pair_matches_oracle"] is True
    assert d["ms_per_step_with_d2h"] > 0 and d["with_d2h"]["host_copy_equals_device_records"] is True
    assert d["with_d2h"]["bytes_d2h_per_rank_and_step"] == 8 * 129600 * 16
    assert "rehearsal" in d and "gloo" in d["launcher"]


def test_eight_ranks_cfg4_key_frame_broadcast_and_the_fused_tail():
    """The same batch against ONE shared key frame that only rank 0 holds before the step (ranks 1-7 start with a zeroed slot: their
    pairs match the oracle only if the in-step broadcast delivered it), detector + estimator in the step."""
    p, d = _run("--gpus", "8", "--backend", "gloo", "--device-map", "0,0,0,0,0,0,0,0", "--config", "cfg4", "--ref-mode", "key", "--pipeline",
                "--steps", "2", "--warmup", "1", timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    _keep("rehearsal_8ranks_cfg4_key_pipeline.json", d)
    assert d["ranks_seen"] == 8 and d["config"]["pairs_per_rank"] == [8] * 8 and d["config"]["ref_mode"] == "key"
    assert "broadcast" in d["config"]["parallelism"] and "almeida" in d["config"]["step"]
    pc = d["parity_check"]
    assert pc["ok"] is True and pc["ranks_checked"] == 8 and pc["gathered_checksum_of_last_pair_matches_oracle"] is True
    assert d["pipeline"]["ms_per_step"] > 0


def test_eight_worker_threads_on_one_gpu_resident_batch_and_the_stream_dispatcher():
    """`--launcher threads`: ONE process, eight worker threads + contexts (all on device 0 here) through ofps_hip_multi_*: the cfg4
    batch split into eight contiguous ranges (one oracle-checked pair per worker), then a 4K stream dealt to the same eight
    workers through ofps_hip_multi_push_frames_async (vectors + island + quaternion per frame, frame order)."""
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "LOCAL_WORLD_SIZE"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--launcher", "threads", "--device-map", "0,0,0,0,0,0,0,0",
                        "--config", "cfg4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-legs"],
                       capture_output=True, text=True, timeout=1500, env=e)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    _keep("rehearsal_8threads_cfg4.json", d)
    assert d["n_gpus"] == 8 and d["config"]["pairs_per_rank"] == [8] * 8 and "threads" in d["launcher"] and "rehearsal" in d
    assert d["parity_check"]["ok"] is True and d["parity_check"]["ranks"] == 8
    s = d["end_to_end"]
    assert "error" not in s, s
    assert s["workers"] == 8 and s["frames"] >= 8 * 32 and s["Mvectors_per_s"] > 0


def test_rccl_process_group_of_one_rank_initialises_and_runs_the_collectives():
    """backend nccl (= RCCL) under torch.distributed.run with ONE rank: the init, the barrier, the all-reduce behind ranks_seen, the
    device-tensor all_gather behind gather_results and the key-frame broadcast all execute on the GPU -- the code path of the
    8-GPU run, world size 1."""
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        e.pop(k, None)
    e["MASTER_ADDR"] = "127.0.0.1"
    sys.path.insert(0, ROOT)
    from ofps_amd.distributed import free_port
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "nccl", "--config", "cfg4",
                        "--pairs", "4", "--gen-pairs", "2", "--ref-mode", "key", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-end-to-end", "--no-legs"], capture_output=True, text=True, timeout=900, env=e)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    _keep("torchrun_world1_rccl_cfg4_key.json", d)
    assert d["ranks_seen"] == 1 and "RCCL" in d["launcher"] and d["config"]["ref_mode"] == "key"
    assert d["parity_check"]["ok"] is True and d["parity_check"]["gathered_checksum_of_last_pair_matches_oracle"] is True
