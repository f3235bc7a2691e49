#!/usr/bin/env python3
"""bench.py -- Mvectors/s of the flow hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the full-search SAD block matcher (N1, the dominant kernel of the hot path) over one batch of
frame pairs already resident in HBM: one launch of sad_strip_kernel<B,R> through the C ABI (ofps_hip_sad_flow_dev).
Default workload = BASELINE.json configs[1] (1080p synthetic, 16x16 blocks, +-16 full search), P = 256 pairs per step.

Multi-GPU (one process per GPU, RCCL).  `--gpus N` IS the number of ranks: under a launcher (WORLD_SIZE set, what the
driver does) WORLD_SIZE must equal N; without one, N > 1 re-executes this script as N ranks under
torch.distributed.run on 127.0.0.1.  Two sharding modes (ofps_amd/distributed.py):
  --scaling weak    (default) every rank searches its own P pairs; value = N * P pairs / time.
  --scaling strong  ONE global batch of P pairs (BASELINE configs[3]: `--config cfg4` = 4K, 8x8, +-32, 64 pairs) split
                    into contiguous ranges with distributed.pair_range; every step ends with distributed.gather_results
                    returning the per-pair results to all ranks in pair order (a 64-bit checksum of each pair's records;
                    with --pipeline also the detector record and the quaternion).
`--ref-mode key`: every pair is searched against one shared key frame that rank 0 broadcasts to all ranks over RCCL
inside the step (north_star's "RCCL broadcast of shared reference frames"); there is no other data-path collective.

Prints ONE JSON line on rank 0 (contract in the task description) with extra objects:
  roofline     -- algorithmic bytes per launch / average launch duration (HIP events on the launch stream) against
                  the 8 TB/s HBM peak, plus the packed-SAD VALU view of the same launch (the kernel is VALU-bound);
  cpu_baseline -- the oracle's full search on the host cores, timed on a bounded sample of the same frames (rank 0,
                  N=1 only);
  end_to_end   -- PCIe-inclusive rate of the Decoder::process_frame shape (N=1 only; never `value`).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
# packed-SAD VALU peak: tools/ubench_valu (profiles/ubench_valu_r02.txt) measures v_qsad_pk_u16_u8 at ~16.5 and v_sad_u8 at ~4.25 cycles per wave64
# instruction per SIMD, i.e. the SAD unit retires 64 |a-b| per clock per SIMD either way:
# 256 CU x 4 SIMD x 64 x 2.4 GHz = 157 T|a-b|/s.
SAD_ABSDIFF_PER_CLK_PER_SIMD = 64.0

PRESETS = {
    # BASELINE.json configs[1] / configs[3] / configs[0]'s geometry
    "cfg2": dict(width=1920, height=1080, block=16, search_range=16, pairs=256, gen_pairs=64, scaling="weak"),
    "cfg4": dict(width=3840, height=2160, block=8, search_range=32, pairs=64, gen_pairs=8, scaling="strong"),
    "cfg1": dict(width=640, height=360, block=16, search_range=8, pairs=256, gen_pairs=64, scaling="weak"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(PRESETS), default=None,
                    help="preset geometry/batch/scaling of a BASELINE.json config (explicit flags still override)")
    ap.add_argument("--pairs", type=int, default=None,
                    help="frame pairs per step (default 256): per rank with --scaling weak, in total with --scaling strong")
    ap.add_argument("--gen-pairs", type=int, default=None, help="distinct frame pairs generated on the host (default 64)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--block", type=int, default=16)
    ap.add_argument("--range", dest="search_range", type=int, default=16)
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-legs", action="store_true",
                    help="skip the bounded cfg3 / cfg4 / cfg5 legs (bench_legs.py) that a default N=1 run adds beside the line")
    ap.add_argument("--end-to-end-only", action="store_true",
                    help="internal: print only the end_to_end object (the main run starts this in a fresh process, whose HIP runtime "
                         "state is a decoder plugin's, not a training framework's)")
    ap.add_argument("--sad-mode", choices=["exhaustive", "pruned"], default="exhaustive",
                    help="search strategy of the SAD kernel (both return the same bits; pruned is content-dependent)")
    ap.add_argument("--content", choices=["regions", "camera"], default="regions",
                    help="regions (default, SURVEY.md 8d): an independent integer displacement per 64x64 region and frame.  "
                         "camera: one global translation per frame + sensor noise +-1 (smooth camera motion)")
    ap.add_argument("--ref-mode", choices=["pairs", "key"], default="pairs",
                    help="pairs: frame k vs k+1 (default, no collective).  key: every frame vs one shared key frame that rank 0 "
                         "broadcasts to all ranks over RCCL inside the timed step")
    ap.add_argument("--pipeline", action="store_true",
                    help="the step also runs the fused tail (detect + Almeida LSQ) on the device-resident vectors")
    ap.add_argument("--launcher", choices=["torchrun", "threads"], default="torchrun",
                    help="torchrun (default): one process per GPU over RCCL (what the driver starts).  threads: ONE process, one "
                         "worker thread + context per GPU through the C ABI's in-process dispatcher (ofps_hip_multi_*); the batch is "
                         "split into contiguous pair ranges, no collective at all")
    ap.add_argument("--prewarm-seconds", type=float, default=0.25,
                    help="untimed steps run before the W warm-ups until this much wall time has passed: a short timed region right "
                         "after an idle GPU reads up to 10 %% low while the clocks ramp (DESIGN.md 3)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl (default) = RCCL, one rank per GPU.  gloo: host-side collectives -- with --stub (CPU tests), or with real "
                         "compute for the one-GPU REHEARSAL of an N-rank run (--device-map 0,0,...): every rank runs the HIP step on its "
                         "mapped device, barrier / ranks_seen / gather_results / key-frame broadcast go through host tensors")
    ap.add_argument("--device-map", default=None,
                    help="comma-separated device index per rank (default: rank r uses GPU LOCAL_RANK).  Repeats are allowed only with "
                         "--backend gloo (RCCL refuses two ranks on one device): `--gpus 2 --backend gloo --device-map 0,0` runs both "
                         "ranks' real steps on GPU 0 -- plumbing rehearsal, not a scaling measurement (the line says so)")
    ap.add_argument("--stub", action="store_true",
                    help="TEST ONLY: no GPU, the step is a no-op that fills a deterministic result table; exercises the launch / "
                         "sharding / gather / reduce logic under gloo.  The line it prints is marked data='stub' and is not a measurement")
    ap.add_argument("--stub-parity", default=None,
                    help="TEST ONLY (with --stub): comma-separated per-rank parity verdicts 1|0|skip standing in for the oracle check")
    args = ap.parse_args(argv)
    given = {a.split("=")[0] for a in (sys.argv[1:] if argv is None else argv) if a.startswith("--")}
    if args.config:
        for k, v in PRESETS[args.config].items():
            flag = "--" + {"search_range": "range", "gen_pairs": "gen-pairs"}.get(k, k)
            if flag not in given:
                setattr(args, k, v)
    if args.pairs is None:
        args.pairs = 256
    if args.gen_pairs is None:
        args.gen_pairs = 64
    if args.scaling is None:
        args.scaling = "weak"
    if args.stub:
        args.backend = "gloo" if "--backend" not in given else args.backend
    if args.device_map is not None:
        try:
            args.device_map = [int(v) for v in args.device_map.split(",")]
        except ValueError:
            ap.error("--device-map: comma-separated integers")
        if len(args.device_map) != args.gpus or min(args.device_map) < 0:
            ap.error(f"--device-map needs one non-negative device index per rank ({args.gpus})")
        if len(set(args.device_map)) != len(args.device_map) and args.backend != "gloo" and args.launcher != "threads":
            ap.error("--device-map with a repeated device needs --backend gloo (RCCL refuses two ranks on one device) or --launcher threads")
    return args


# ---------------------------------------------------------------------------------------------------------------------
# sharding plan (pure function: tests/test_bench_logic.py)
# ---------------------------------------------------------------------------------------------------------------------

def shard_plan(scaling: str, pairs: int, gen_pairs: int, world: int, rank: int, key_mode: bool) -> dict:
    """Which frames of the walk a rank keeps resident and which pairs it searches.

    The resident sequence of a weak-scaling rank (or of the whole job with strong scaling) has pairs+1 frames walking
    G+1 generated frames forward and backward, so every consecutive pair is a generated pair.  Returns
      first, count      -- the rank's pairs [first, first+count) of the global batch (weak: 0, pairs)
      frame_ids         -- indices into the generated frames, in resident order; in key mode slot 0 is the key frame
      seed_rank         -- which rank's generator seed the frames come from (weak: own; strong: 0 -- one sequence)
    """
    from ofps_amd import distributed as D
    G = max(1, min(pairs, gen_pairs))
    walk = np.abs(((np.arange(pairs + 1) + G) % (2 * G)) - G) if G > 1 else np.arange(pairs + 1) % 2
    if scaling == "weak":
        first, count, seed_rank = 0, pairs, rank
    else:
        first, count = D.pair_range(pairs, world, rank)
        seed_rank = 0
    f0, fc = D.frame_range(pairs, 1, 0, 1 if key_mode else 0) if scaling == "weak" else D.frame_range(pairs, world, rank, 1 if key_mode else 0)
    ids = list(walk[f0:f0 + fc])
    if key_mode and count:
        ids = [int(walk[0])] + ids                     # slot 0 = the key frame (arrives by broadcast on ranks != 0)
    return {"first": int(first), "count": int(count), "frame_ids": [int(i) for i in ids], "seed_rank": seed_rank,
            "generated_pairs": int(G), "walk": walk}


def pair_checksums(d_out, ctx=None, scratch=None):
    """Per-pair 64-bit checksum of the records on the device: the wrapping sum of each pair's bytes read as int64 words --
    through the library's one-launch kernel (ofps_hip_checksum_dev) when a context is given: torch's generic reduction took
    0.11 ms of a 2.4 ms strong-scaling step."""
    import torch
    if ctx is None or d_out.shape[0] == 0:
        return torch.sum(d_out.view(torch.int64).reshape(d_out.shape[0], -1), dim=1)
    out = scratch if scratch is not None else torch.empty((d_out.shape[0],), dtype=torch.int64, device=d_out.device)
    ctx.checksum_dev(d_out.data_ptr(), d_out[0].numel() * d_out.element_size(), d_out.shape[0], out.data_ptr())
    return out


def time_steps(step, steps: int, warmup: int, sync, barrier) -> float:
    """W untimed warm-ups, then exactly K steps between (sync, barrier, sync) brackets -> seconds on this rank.
    The interpreter's objects are frozen out of CPython's cyclic collector for the duration (bench_legs.QuietGC: a generation-2
    pass after `import torch` stops the host thread for ~40 ms, ten steps' worth, at an arbitrary iteration); collections that
    still run inside the bracket are kept in time_steps.gc_inside (the line reports them)."""
    from bench_legs import QuietGC
    with QuietGC() as quiet:
        for _ in range(warmup):
            step()
        sync(); barrier(); sync()
        quiet.events.clear()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync(); barrier(); sync()
        el = time.perf_counter() - t0
    time_steps.gc_inside = quiet.summary()
    return el


time_steps.gc_inside = None


# What each row of the hot path is pinned to: a reader of the line alone must not mistake "bit-exact vs the
# oracle" on the headline kernel for parity with reference output -- the reference has no SAD search and no per-pixel flow.
PARITY_PIN = {"A6-A12 (camera + Almeida)": "reference-held mfield tables (docs/report/mfield/*.csv) + the reference's own known-answer test",
              "A1-A5 (densifier, detector)": "hand-derived literals from the Rust text (no reference-held vectors exist)",
              "N1 (full-search SAD, this line's kernel)": "build-defined spec: bit-exact vs the build's own CPU restatement only (reference has no SAD)",
              "N2 (dense LK flow)": "build-defined spec: bit-exact vs the build's own CPU restatement only (reference calls OpenCV Farneback)",
              "N2b (Farneback flow, hip_flow)": "the published algorithm in the form of OpenCV's calcOpticalFlowFarneback with cv-decoder's arguments: "
                                                "bit-identical to the build's own CPU restatement; OpenCV itself is not part of the reference tree "
                                                "(unpinned here; tools/external_parity/opencv_compare.py is the check for anyone who has cv2)",
              "cv-decoder front-end (resize INTER_LINEAR, BGR->gray; 'Process Fullres' = false)": "OpenCV's published 8-bit code paths restated "
                                                "integer for integer: bit-exact vs the build's C restatement and an independent NumPy one; unpinned (no OpenCV here)"}


def build_line(args, world: int, el: float, launch_ms: float, launch_pairs: int, counts: list, nblk: int, ranks_seen: int,
               traffic_per_pair=None, traffic_stale=None) -> dict:
    """The JSON line from measured times (pure function: tests/test_bench_contract.py feeds it synthetic timings)."""
    W, H, B, R, P = args.width, args.height, args.block, args.search_range, args.pairs
    total_pairs = sum(counts)
    ms_per_step = el / args.steps * 1e3
    algo_bytes = launch_pairs * (2 * W * H + 16 * nblk)                 # SURVEY.md 8d, per pair x pairs in rank 0's launch
    achieved = algo_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    abs_diffs = launch_pairs * nblk * B * B * (2 * R + 1) ** 2
    valu_peak = 256 * 4 * 2.4e9 * SAD_ABSDIFF_PER_CLK_PER_SIMD
    key = args.ref_mode == "key"
    geom = (W, H, B, R)
    out = {
        "metric": ("Mvectors/s, 1080p 16x16 blocks +-16 full-search SAD" if geom == (1920, 1080, 16, 16)
                   else f"Mvectors/s, {W}x{H} {B}x{B} blocks +-{R} full-search SAD"),
        "value": round(total_pairs * nblk * args.steps / el / 1e6, 3),
        "unit": "Mvectors/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "ms_per_frame_pair": round(ms_per_step / max(total_pairs, 1), 5),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "u8", "data": "stub" if args.stub else "synthetic",
        "ranks_seen": ranks_seen,
        "config": {"workload": ({(1920, 1080, 16, 16): "cfg2 (BASELINE.json configs[1]): ",
                                 (3840, 2160, 8, 32): "cfg4 (BASELINE.json configs[3]): ",
                                 (640, 360, 16, 8): "cfg1 geometry (BASELINE.json configs[0]): "}.get(geom, "")
                                + f"{W}x{H} synthetic luma, {B}x{B} blocks, +-{R} full-search SAD"),
                   "pairs_per_step": total_pairs, "pairs_per_rank": counts, "generated_pairs": max(1, min(P, args.gen_pairs)),
                   "content": args.content, "vectors_per_pair": nblk,
                   "parallelism": (f"frame-pair sharding x{world}, "
                                   + ("one global batch split into contiguous pair ranges, per-pair results gathered in pair order"
                                      if args.scaling == "strong" else "independent batch per rank")
                                   + (", key frame broadcast from rank 0 per step (RCCL)" if key else ", no data-path collective")),
                   "ref_mode": args.ref_mode,
                   "step": ("sad" + (" -> block-motion detect -> almeida LSQ" if args.pipeline else "")
                            + (" -> gather_results" if args.scaling == "strong" else "")),
                   "kernel": (f"sad_strip_kernel<{B},{R}>" if args.sad_mode == "exhaustive"
                              else "sad_pde_kernel + sad_strip_kernel<16,16> on overflow strips"),
                   "sad_mode": args.sad_mode},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5),
                     "traffic": None if traffic_per_pair is None else traffic_per_pair * launch_pairs,
                     "traffic_source": None if traffic_per_pair is None else "profiles/hbm_traffic.json (rocprofv3 --pmc passes of this command; not re-measured in this run)",
                     "traffic_stale": None if traffic_per_pair is None else traffic_stale,
                     "algorithmic_bytes_per_launch": algo_bytes, "launch_ms": round(launch_ms, 5), "pairs_per_launch": launch_pairs,
                     "note": "full search is VALU-bound (SURVEY.md 8d): see 'valu'",
                     "valu": {"abs_diffs_per_launch": abs_diffs,
                              "achieved_Tops": round(abs_diffs / (launch_ms * 1e-3) / 1e12, 3) if launch_ms > 0 else 0.0,
                              "peak_Tops": round(valu_peak / 1e12, 3),
                              "frac": round(abs_diffs / (launch_ms * 1e-3) / valu_peak, 4) if launch_ms > 0 else 0.0,
                              "peak_basis": "SAD unit: 64 |a-b| per clock per SIMD (v_qsad_pk_u16_u8 16 cyc, measured)"}},
    }
    out["parity_pin"] = PARITY_PIN
    if getattr(args, "device_map", None) and len(set(args.device_map)) != len(args.device_map):
        out["rehearsal"] = (f"{world} ranks on devices {args.device_map} over gloo: real HIP steps, host-side collectives -- a plumbing "
                            "rehearsal of the N-GPU run on fewer GPUs, NOT a scaling measurement")
    if args.sad_mode == "pruned":
        out["roofline"]["valu"]["note"] = ("exhaustive-equivalent rate: the pruned search returns the same winners but "
                                           "evaluates fewer |a-b|, so this fraction can exceed 1")
        out["roofline"]["traffic"] = None             # the committed PMC traffic figure belongs to the exhaustive kernel
    return out


def kernel_source_sha16(name: str = "sad.hip") -> str:
    import hashlib
    return hashlib.sha256(open(os.path.join(ROOT, "ofps_amd", "csrc", name), "rb").read()).hexdigest()[:16]


def committed_traffic_per_pair(W, H, B, R):
    """-> (HBM bytes per pair from the committed PMC passes or None, stale): stale = the kernel source the passes were taken
    on (sha stored by tools/make_hbm_traffic.py) is not the sad.hip in the tree -- the figure may no longer describe it."""
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        d = json.load(open(tpath)).get(f"sad_{W}x{H}_b{B}_r{R}", {})
        v = d.get("hbm_bytes_per_pair")
        return v, (None if v is None else d.get("kernel_source_sha16") != kernel_source_sha16())
    except Exception:
        return None, None


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle is the thing timed HERE and only here; it is never on the product path)
# ---------------------------------------------------------------------------------------------------------------------

def cpu_quota_cores():
    """CPUs the container may use on average (cgroup v2 cpu.max / v1 cfs quota), or None: beyond it threads get throttled
    in 100 ms periods, so short bursts with more threads look faster than the host can sustain."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except Exception:
        return None


def cpu_baseline(frames: np.ndarray, block: int, rng: int, budget_s: float):
    """The oracle (kind 'port': C restatement, run-time dispatched AVX2 vmpsadbw inner loop -- SSE2 psadbw on hosts without
    AVX2 --, OpenMP over block runs) on the host cores.  Thread counts up to what the host lets this process use
    (nproc, affinity mask, cgroup quota) are each measured SUSTAINED for 1.5 s; the bounded sample then runs at the best."""
    import oracle
    nblk = (frames.shape[2] // block) * (frames.shape[1] // block)
    nproc = oracle.num_threads()
    quota = cpu_quota_cores()
    usable = min(nproc, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else nproc)
    cap = usable if quota is None else max(1, min(usable, int(round(quota * 2))))     # try up to 2x the quota, no more
    cands = sorted({t for t in (cap, 256, 192, 128, 96, 64, 48, 32, 24, 16, 8, 4) if 1 < t <= cap} | {1})
    npair = len(frames) - 1

    def sustained(t, seconds):
        oracle.sad_flow(frames[0], frames[1], block, rng, threads=t)        # thread pool warm
        n = 0
        t0 = time.perf_counter()
        while True:
            oracle.sad_flow(frames[n % npair], frames[n % npair + 1], block, rng, threads=t)
            n += 1
            el = time.perf_counter() - t0
            if el >= seconds:
                return el / n, n, el
    sweep = {t: sustained(t, 1.5 if t > 1 else 0.5)[0] for t in cands}
    threads = min(sweep, key=sweep.get)
    single_dt = sweep[1]
    per_pair, done, el = sustained(threads, budget_s)
    # the tail of the path on the vectors of one pair, the way the reference runs it: one thread per plugin call
    ent, _ = oracle.sad_flow(frames[0], frames[1], block, rng, threads=threads)
    cam = oracle.camera(16 / 9, 22.275)

    def ms(fn, reps=3):
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        return round((time.perf_counter() - t) / reps * 1e3, 3)
    tail = {"n_vectors": int(len(ent)),
            "almeida_lsq_ms": ms(lambda: oracle.solve_ypr_given(ent, cam)),
            "almeida_ransac_ms": ms(lambda: oracle.solve_ypr_ransac(ent, cam, 200, 0.05, 1000, seed=1)),
            "block_motion_detect_ms": ms(lambda: oracle.detect_motion(ent)), "threads": 1}
    # the same estimates on all the cores this process may use (OpenMP over the per-vector loop and
    # the twelve dot products / over the RANSAC hypotheses; same bits as one thread), and the per-pixel solve of
    # BASELINE configs[2] (2,073,600 records), which one thread takes ~20 s for
    tail_all = {"n_vectors": int(len(ent)), "threads": threads,
                "almeida_lsq_ms": ms(lambda: oracle.solve_ypr_given(ent, cam, threads=threads), reps=5),
                "almeida_ransac_ms": ms(lambda: oracle.solve_ypr_ransac(ent, cam, 200, 0.05, 1000, seed=1, threads=threads), reps=5),
                "how": "OpenMP: per-vector delta loop + the 12 sequential dot sums side by side (LSQ); hypotheses (RANSAC); same bits as 1 thread"}
    from ofps_amd import synth as _synth
    dense = _synth.rotation_field(1920, 1080)
    tail_all["almeida_lsq_2073600_records_ms"] = ms(lambda: oracle.solve_ypr_given(dense, cam, threads=threads), reps=1)
    absd = nblk * block * block * (2 * rng + 1) ** 2
    return {"value": round(done * nblk / el / 1e6, 4), "unit": "Mvectors/s", "cores": threads, "kind": "port",
            "nproc": nproc, "cpu_quota_cores": quota, "simd": oracle.sad_simd_level(),
            "abs_diffs_per_s": round(done * absd / el, 0),
            "thread_sweep_ms_per_pair_sustained": {str(t): round(v * 1e3, 2) for t, v in sweep.items()},
            "tail_single_thread": tail, "tail_all_cores": tail_all,
            "sample": f"{done} frame-pair searches cycling over the bench sequence, {frames.shape[2]}x{frames.shape[1]}, "
                      f"{block}x{block} blocks, +-{rng}, {el:.1f} s wall",
            "ms_per_pair": round(el / done * 1e3, 2),
            # how the reference runs an estimator / detector: one thread per plugin call (tracking/worker.rs:347-352)
            "single_thread": {"value": round(nblk / single_dt / 1e6, 4), "ms_per_pair": round(single_dt * 1e3, 1), "sample": "1 pair"}}


def end_to_end_leg(frames, W, H, B, R, device, n_frames=300):
    """PCIe-inclusive rate of the Decoder::process_frame shape (SURVEY.md 8d "reported separately"; never `value`): every
    frame crosses PCIe from a page-locked buffer (2.07 MB at 1080p), the previous frame stays on the device, the vectors
    come back to page-locked host memory (16 B each).  Forms: the synchronous call; the read-ahead form
    (ofps_hip_push_frame_async, the previous ticket collected after the next push, so the H2D of frame k+1 overlaps the
    search of pair k-1, k) with the frames already in page-locked memory (a decoder that writes there directly); and
    the same with the host filling the next page-locked buffer while the GPU works (a decoder whose output must be
    copied)."""
    from ofps_amd.runtime import HipContext
    ctx = HipContext(device)                   # its own context / stream, like a decoder plugin instance
    nblk = (W // B) * (H // B)
    src = [np.ascontiguousarray(f[:, :W]).copy() for f in frames[:4]]
    pins = [ctx.pinned_frame(H, W) for _ in range(3)]
    ents = [ctx.pinned_array((nblk, 4)) for _ in range(2)]
    kw = dict(block=B, search_range=R, detector=False, estimator=False)
    for k in range(3):
        np.copyto(pins[k], src[k])

    def run_sync(n):
        ctx.reset_frames()
        for k in range(n):
            ctx.frame_wait(ctx.push_frame_async(pins[k % 3], out_entries=ents[0], **kw))

    def run_read_ahead(n, fill):
        ctx.reset_frames()
        prev = None
        for k in range(n):
            t = ctx.push_frame_async(pins[k % 3], out_entries=ents[k % 2], **kw)
            if fill:
                np.copyto(pins[(k + 1) % 3], src[(k + 1) % 4])        # the decoder produces frame k+1 while the GPU works
            if prev is not None:
                ctx.frame_wait(prev)
            prev = t
        ctx.frame_wait(prev)

    # Every number of this leg is the MEDIAN of `repeats` runs of n_frames frames, with the fastest and the slowest beside it
    # (one-shot timings of a host loop made round 4's read_ahead look 3x slower than round 3's -- a 40 ms
    # generation-2 pass of CPython's garbage collector landed in that loop; bench_legs.QuietGC, profiles/r05/read_ahead_bisect.txt).
    from bench_legs import QuietGC, median_min_max
    repeats = 7

    def timed(fn, *a):
        fn(20, *a)
        v = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            fn(n_frames, *a)
            v.append((time.perf_counter() - t0) / n_frames * 1e3)
        return median_min_max(v)

    def row(mm, **extra):
        sec = mm["median"] * 1e-3
        return dict({"ms_per_frame": mm["median"], "ms_per_frame_min": mm["min"], "ms_per_frame_max": mm["max"], "repeats": mm["repeats"],
                     "Mvectors_per_s": round(nblk / sec / 1e6, 2)}, **extra)
    with QuietGC() as quiet:
        s = timed(run_sync)
        a = timed(run_read_ahead, False)
        c = timed(run_read_ahead, True)
    gc_inside = quiet.summary()
    # the link itself, measured in this run: 16 frames' worth of page-locked bytes host -> device in one copy, and one frame
    # alone (what a per-frame upload pays); the ceiling below is derived from the first, not from a constant
    big = ctx.pinned_frame(16 * H, W)
    d_big = ctx.malloc(big.nbytes)

    def h2d_rate(arr, reps):
        ctx.memcpy_h2d(d_big, arr)
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.memcpy_h2d(d_big, arr)
        return arr.nbytes * reps / (time.perf_counter() - t0) / 1e9
    h2d_bulk = median_min_max([h2d_rate(big, 20) for _ in range(5)], 1)["median"]
    h2d_frame = median_min_max([h2d_rate(pins[0], 200) for _ in range(5)], 1)["median"]
    ctx.free(d_big)
    ceiling = nblk / (W * H / (h2d_bulk * 1e9)) / 1e6
    out = {"what": "Decoder::process_frame shape: one luma frame H2D from page-locked memory per call, previous frame resident, "
                   "vectors D2H to page-locked memory",
           "frames": n_frames, "bytes_h2d_per_frame": W * H, "bytes_d2h_per_frame": 16 * nblk,
           "timing": f"median of {repeats} runs of {n_frames} frames per form (min / max beside it); objects alive at the start are frozen out of "
                     "CPython's collector for the timed loops, collections that still ran inside them are counted here",
           "python_gc_inside_timed_loops": gc_inside,
           "h2d_GBs_measured": {"bulk_16_frames_per_copy": round(h2d_bulk, 1), "one_frame_per_blocking_copy": round(h2d_frame, 1)},
           "pcie_ceiling_Mvectors_per_s": round(ceiling, 1),
           "pcie_ceiling_basis": "every frame crosses the link once: vectors per frame / (frame bytes / the bulk H2D rate measured in this run)",
           "sync": row(s, entry_points="ofps_hip_push_frame"),
           "read_ahead": row(a, entry_points="ofps_hip_push_frame_async + ofps_hip_frame_wait, 2 tickets in flight"),
           "read_ahead_with_host_copy": row(c, includes="a host memcpy of every frame into the next page-locked buffer while the GPU works")}
    ctx.close()
    # the same two loops in the C++ host layer (ofps_amd/host/ofps_hip_tool stream-bench): no interpreter between the calls
    try:
        import subprocess
        from ofps_amd.build import TOOL
        if (W, H, B, R) == (1920, 1080, 16, 16) and os.path.exists(TOOL):
            env = dict(os.environ, HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", str(device)))
            def native(*mode_args, runs=5):
                rs = [json.loads(subprocess.run([TOOL, "stream-bench", str(W), str(H)] + list(mode_args), capture_output=True, text=True,
                                                timeout=120, env=env, check=True).stdout.strip().splitlines()[-1]) for _ in range(runs)]
                mm = median_min_max([r["ms_per_frame"] for r in rs])
                return rs[0], {"ms_per_frame": mm["median"], "ms_per_frame_min": mm["min"], "ms_per_frame_max": mm["max"], "repeats": runs,
                               "Mvectors_per_s": round(nblk / (mm["median"] * 1e-3) / 1e6, 2)}
            for mode, key in (("sync", "sync_native_host"), ("ahead", "read_ahead_native_host")):
                _, r = native("1000", mode)
                out[key] = dict(r, entry_points="C++ host layer, frames already in page-locked memory; each repeat a fresh process")
            r0, r = native("1024", "batch", "16")
            out["read_ahead_batched_native_host"] = dict(r, batch=r0["batch"],
                                                         entry_points="ofps_hip_push_frames_async + ofps_hip_frames_wait: 16 frames per ticket, 2 tickets "
                                                                      "in flight, one H2D + one search launch + one read-back per batch",
                                                         frac_of_measured_pcie_ceiling=round(r["Mvectors_per_s"] / ceiling, 3))
    except Exception as e:                                    # the tool is optional evidence, never the bench line
        out["native_host_error"] = repr(e)[:200]
    return out


# ---------------------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------------------

def run_rank(args) -> int:
    import torch
    from ofps_amd import distributed as D

    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {env_world} rank(s): --gpus is the number of "
                         f"ranks, one per GPU")
    env_rank, env_local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = args.device_map[env_rank] if args.device_map else env_local
    if not args.stub:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
        need = (max(args.device_map) + 1) if args.device_map else int(os.environ.get("LOCAL_WORLD_SIZE", env_world))
        if torch.cuda.device_count() < need:
            raise SystemExit(f"bench.py: {env_world} ranks need {need} GPUs"
                             + (f" (--device-map {args.device_map})" if args.device_map else "") + f", this node has {torch.cuda.device_count()}")
    rank, world, local_rank = D.init_from_env(args.backend, device_index=dev_index)
    dev = torch.device("cpu") if args.stub else torch.device("cuda", dev_index)
    if not args.stub:
        torch.cuda.set_device(dev_index)

    W, H, B, R, P = args.width, args.height, args.block, args.search_range, args.pairs
    key_mode = args.ref_mode == "key"
    plan = shard_plan(args.scaling, P, args.gen_pairs, world, rank, key_mode)
    first, count, walk = plan["first"], plan["count"], plan["walk"]
    nbx, nby = W // B, H // B
    nblk = nbx * nby
    stride = (W + 63) // 64 * 64
    n_res = len(plan["frame_ids"])                          # resident frames of this rank

    if args.stub:
        frames = None
        d_frames = torch.zeros((max(n_res, 1), 8), dtype=torch.uint8)
        if key_mode and rank == 0 and n_res:
            d_frames[0] = torch.arange(8, dtype=torch.uint8) + 1
        d_sum = torch.zeros((count,), dtype=torch.int64)

        def kernel_only():
            # stands in for the search: pair k's "checksum" = f(global pair index, key frame bytes)
            base = int(d_frames[0].sum().item()) if key_mode and n_res else 0
            d_sum.copy_(torch.arange(first, first + count, dtype=torch.int64) * 1000 + base)
        ctx = None
    else:
        from ofps_amd import synth
        from ofps_amd.runtime import HipContext
        gen = dict(max_step=R) if args.content == "regions" else dict(max_step=min(R, 12), region=1 << 14, noise=1)
        # weak: an independent sequence per rank (same generator, different seed); strong: ONE sequence, every rank
        # generates it and keeps only the frames of its pair range (+1 halo frame) resident
        frames = synth.luma_sequence(plan["generated_pairs"] + 1, W, H, seed=synth.SEED0 + 1000 * plan["seed_rank"], stride=stride, **gen)
        if n_res:
            d_frames = torch.from_numpy(frames[plan["frame_ids"]]).to(dev).contiguous()
            if key_mode and rank != 0:
                d_frames[0].zero_()                          # only the broadcast can put the key frame here
        else:
            d_frames = torch.zeros((1, H, stride), dtype=torch.uint8, device=dev)
        d_out = torch.empty((max(count, 1), nblk, 4), dtype=torch.float32, device=dev)
        d_chk = torch.zeros((count,), dtype=torch.int64, device=dev)
        ctx = HipContext(dev_index)
        ctx.use_torch_stream()            # launches go to torch's current stream: torch events see them
        ctx.set_sad_mode(ctx.SAD_PRUNED if args.sad_mode == "pruned" else ctx.SAD_EXHAUSTIVE)

        def kernel_only():
            if count:
                ctx.sad_flow_dev(d_frames.data_ptr(), n_res, W, H, stride, stride * H, 1 if key_mode else 0, B, R,
                                 d_out.data_ptr(), None)

    if args.pipeline and not args.stub:
        dim = ctx.block_dim(0.05, 3)
        d_res = torch.zeros((max(count, 1), 4), dtype=torch.int32, device=dev)
        d_field = torch.empty((max(count, 1), dim * dim, 2), dtype=torch.float32, device=dev)
        d_quat = torch.zeros((max(count, 1), 4), dtype=torch.float32, device=dev)

        def tail():
            if count:
                ctx.detect_dev(d_out.data_ptr(), nblk, count, 0.05, 3, 0.003, d_res.data_ptr(), d_field.data_ptr())
                ctx.almeida_dev(d_out.data_ptr(), nblk, count, W / H, 39.6 * H / W, False, 0, 0.05, 0, 0, d_quat.data_ptr())
    else:
        def tail():
            pass

    gathered = {}

    def step():
        if key_mode:
            # the shared reference frame travels rank 0 -> all ranks (on torch's current stream, like the search after it)
            D.broadcast_reference(d_frames[0], src=0)
        kernel_only()
        tail()
        if args.scaling == "strong":
            # per-pair results back to every rank in pair order: the only other collective, a few bytes per pair
            if not args.stub:
                local = pair_checksums(d_out[:count], ctx, d_chk)
            else:
                local = d_sum
            gathered["checksum"] = D.gather_results(local, P)
            if args.pipeline and not args.stub:
                gathered["quat"] = D.gather_results(d_quat[:count], P)
                gathered["detect"] = D.gather_results(d_res[:count], P)

    sync = (lambda: None) if args.stub else torch.cuda.synchronize
    barrier = D.StreamBarrier(dev)
    if not args.stub and args.prewarm_seconds > 0:
        # clocks: a rank's first launches after an idle period run slow; with strong scaling a step is a few milliseconds
        # (8 pairs of 4K frames at N = 8), so the W warm-ups alone do not cover the ramp
        t_pw = time.perf_counter()
        while time.perf_counter() - t_pw < args.prewarm_seconds:
            for _ in range(4):
                kernel_only()
            sync()
    el_local = time_steps(step, args.steps, args.warmup, sync, barrier)
    gc_main = time_steps.gc_inside
    el = D.max_over_ranks(el_local, device=dev)
    el_min = -D.max_over_ranks(-el_local, device=dev)

    # ---- per-launch duration of the dominant kernel with HIP events on the launch stream (roofline leg)
    if args.stub:
        launch_ms = el / args.steps * 1e3
    else:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for a, b in evs:
            a.record(); kernel_only(); b.record()
        torch.cuda.synchronize()
        launch_ms = float(np.mean([a.elapsed_time(b) for a, b in evs])) if count else 0.0

    # ---- the gather alone (strong scaling): HIP events around K gathers, nothing else on the stream
    gather_ms = None
    if args.scaling == "strong" and not args.stub and D.active() and world > 1:
        local = pair_checksums(d_out[:count], ctx, d_chk)
        ga, gb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        D.gather_results(local, P)
        torch.cuda.synchronize()
        ga.record()
        for _ in range(args.steps):
            D.gather_results(local, P)
        gb.record()
        torch.cuda.synchronize()
        gather_ms = D.max_over_ranks(ga.elapsed_time(gb) / args.steps, device=dev)

    # ---- strong scaling, time to solution WITH the vectors on the host (the results come home by hipMemcpyAsync
    # D2H): every rank copies its pairs' records to page-locked memory inside the step.  The search writes into two device
    # buffers in turn and the copy runs on a side stream behind an event, so the D2H of step k overlaps the search of step k+1;
    # a buffer is searched into again only after its previous copy finished.  Never `value` (the metric is device-resident).
    d2h_ms = None
    if args.scaling == "strong" and not args.stub and count:
        side = torch.cuda.Stream(device=dev)
        host = [torch.empty((count, nblk, 4), dtype=torch.float32).pin_memory() for _ in range(2)]
        outs = [d_out, torch.empty_like(d_out)]
        searched = [torch.cuda.Event() for _ in range(2)]
        copied = [None, None]
        kstep = [0]

        def step_d2h():
            b = kstep[0] & 1
            kstep[0] += 1
            if copied[b] is not None:
                torch.cuda.current_stream().wait_event(copied[b])       # the copy that still reads this buffer
            if key_mode:
                D.broadcast_reference(d_frames[0], src=0)
            ctx.sad_flow_dev(d_frames.data_ptr(), n_res, W, H, stride, stride * H, 1 if key_mode else 0, B, R, outs[b].data_ptr(), None)
            searched[b].record()
            with torch.cuda.stream(side):
                side.wait_event(searched[b])
                host[b].copy_(outs[b][:count], non_blocking=True)
                ev = torch.cuda.Event(); ev.record(side); copied[b] = ev
        el_d2h = D.max_over_ranks(time_steps(step_d2h, args.steps, args.warmup, sync, barrier), device=dev)
        d2h_ms = el_d2h / args.steps * 1e3
        d2h_same = bool((host[(kstep[0] - 1) & 1].numpy().view(np.uint32) == outs[(kstep[0] - 1) & 1][:count].cpu().numpy().view(np.uint32)).all())

    seen = D.ranks_seen(dev)
    counts = D.gather_counts(count, dev)
    if seen != args.gpus:
        raise SystemExit(f"bench.py: {seen} ranks answered the all-reduce, expected {args.gpus}")

    out = None
    if rank == 0:
        out = build_line(args, world, el, launch_ms, count, counts, nblk, seen,
                         *((None, None) if args.stub else committed_traffic_per_pair(W, H, B, R)))
        out["launcher"] = (("torchrun (one process per GPU, RCCL)" if args.backend == "nccl" else
                            "torchrun (one process per rank, gloo: host-side collectives)") if D.active() else "single process")
        out["per_rank_ms_per_step"] = {"min": round(el_min / args.steps * 1e3, 4), "max": round(el / args.steps * 1e3, 4)}
        out["python_gc_inside_timed_region"] = gc_main
        if gather_ms is not None:
            out["gather_ms_per_step"] = round(gather_ms, 4)
        if d2h_ms is not None:
            out["ms_per_step_with_d2h"] = round(d2h_ms, 4)
            out["with_d2h"] = {"what": "the same step with every rank's records copied to page-locked host memory inside it (D2H on a side "
                                       "stream behind an event, two device buffers in turn; no gather collective): time to solution with "
                                       "the vectors on the host",
                               "bytes_d2h_per_rank_and_step": int(count) * nblk * 16,
                               "Mvectors_per_s": round(sum(counts) * nblk * args.steps / (d2h_ms * 1e-3 * args.steps) / 1e6, 3),
                               "host_copy_equals_device_records": d2h_same}

    if args.pipeline and not args.stub:
        def full():
            kernel_only(); tail()
        for _ in range(2):
            full()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            full()
        torch.cuda.synchronize()
        pel = time.perf_counter() - t1
        if out is not None:
            out["pipeline"] = {"stages": "sad -> block-motion detect -> almeida LSQ (device resident)",
                               "ms_per_step": round(pel / args.steps * 1e3, 4),
                               "Mvectors_per_s_per_gpu": round(count * nblk * args.steps / pel / 1e6, 3)}

    # ---- outside the timed region: parity.  Every rank checks one pair it searched against the CPU oracle (the oracle
    # is the checker here, never the thing measured); with strong scaling rank 0 also checks the gathered table: the
    # checksum of a pair searched by the LAST rank must equal the checksum of the oracle's records for that pair
    ok = None
    gather_ok = None
    if args.stub:
        ok = 1
        if args.stub_parity:
            v = args.stub_parity.split(",")[rank]
            ok = None if v == "skip" else int(v)
        if args.scaling == "strong":
            base = 36 if key_mode else 0                     # sum(1..8): the key frame reached every rank
            gather_ok = int(gathered["checksum"].tolist() == [k * 1000 + base for k in range(P)])
    elif count:
        try:
            import oracle

            def host_pair(k):                                # global pair k -> (prev, cur) as the spec defines them
                a = frames[walk[0]] if key_mode else frames[walk[k]]
                return np.ascontiguousarray(a[:, :W]), np.ascontiguousarray(frames[walk[k + 1]][:, :W])
            if key_mode and args.scaling == "weak" and rank != 0:
                from ofps_amd import synth
                key = synth.luma_sequence(1, W, H, seed=synth.SEED0, stride=stride,
                                          **(dict(max_step=R) if args.content == "regions" else dict(max_step=min(R, 12), region=1 << 14, noise=1)))[0]
            else:
                key = None
            kl = (3 * rank + 1) % count                          # a different local pair on every rank
            prev, cur = host_pair(first + kl)
            if key is not None:
                prev = np.ascontiguousarray(key[:, :W])
            ent_o, _ = oracle.sad_flow(prev, cur, B, R, threads=4)
            ok = int((d_out[kl].cpu().numpy().view(np.uint32) == ent_o.view(np.uint32)).all())
            if args.scaling == "strong" and rank == 0:
                kg = P - 1                                       # searched by the last rank that holds pairs
                ent_g, _ = oracle.sad_flow(*host_pair(kg), B, R, threads=4)
                with np.errstate(over="ignore"):
                    want = int(np.ascontiguousarray(ent_g).view(np.int64).sum(dtype=np.int64))   # wrapping, like the device sum
                gather_ok = int(int(gathered["checksum"][kg].item()) == want)
        except ImportError as e:                                 # no oracle on this machine: report "not checked"
            print(f"[bench] parity check skipped on rank {rank}: {e}", file=sys.stderr)
    # mismatches and skips are reduced separately: a rank that had nothing to check (no pairs, no oracle) must not hide
    # another rank's mismatch
    ok_min = D.min_over_ranks(1 if ok is None else ok, dev)                 # 0 = at least one rank saw a mismatch
    skipped = world - int(round(D.sum_over_ranks(0 if ok is None else 1, dev)))
    mismatch = ok_min == 0 or gather_ok == 0
    if out is not None:
        out["parity_check"] = {"what": "one searched pair per rank vs the CPU oracle, bit for bit", "ranks": world,
                               "ranks_checked": world - skipped, "ranks_skipped": skipped,
                               "ok": False if ok_min == 0 else (True if skipped < world else None)}
        if args.scaling == "strong":
            out["parity_check"]["gathered_checksum_of_last_pair_matches_oracle"] = None if gather_ok is None else bool(gather_ok)

    if rank == 0 and world == 1 and not args.stub:
        if not args.no_end_to_end:
            import subprocess
            try:
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "--end-to-end-only", "--width", str(W), "--height", str(H),
                                    "--block", str(B), "--range", str(R)], capture_output=True, text=True, timeout=300, check=True)
                out["end_to_end"] = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
            except Exception as e:
                out["end_to_end"] = {"error": repr(e)[:300]}
        if not args.no_legs:
            # BASELINE configs[2], [3], [4] beside the line (bench_legs.py; fresh process, own oracle spot-checks)
            import subprocess
            try:
                p = subprocess.run([sys.executable, os.path.join(ROOT, "bench_legs.py")], capture_output=True, text=True, timeout=600,
                                   check=True)
                legs = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
            except Exception as e:
                legs = {k: {"error": repr(e)[:300]} for k in ("cfg3_chain", "cfg4", "cfg5_stream")}
            out.update(legs)
            for k, v in legs.items():
                if isinstance(v, dict) and isinstance(v.get("parity_check"), dict) and v["parity_check"].get("ok") is False:
                    mismatch = True
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(np.ascontiguousarray(frames[:, :, :W]) if stride != W else frames, B, R, args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if ctx is not None:
        ctx.close()
    if D.active():
        import torch.distributed as dist
        dist.destroy_process_group()
    if mismatch:
        print(f"[bench] rank {rank}: PARITY MISMATCH against the oracle -- the line above is not a valid measurement", file=sys.stderr)
        return 3
    return 0


def run_threads(args) -> int:
    """`--launcher threads`: one process drives N GPUs through ofps_hip_multi_* (one worker thread + context per device).  The
    batch is ONE sequence split into contiguous pair ranges (strong: P pairs in total; weak: N * P pairs, P per worker); a
    step is one search of every worker's resident pairs; the timed region is K steps per worker, bracketed by the
    dispatcher's join on both sides (every worker synchronises its stream before it reports back)."""
    import torch
    from ofps_amd import synth
    from ofps_amd.runtime import MultiDevice
    if args.stub or args.pipeline or args.ref_mode == "key" and args.scaling == "weak":
        raise SystemExit("bench.py --launcher threads: supports the SAD step, pairs or key mode (key mode with --scaling strong)")
    devices = list(args.device_map) if args.device_map else list(range(args.gpus))      # repeats: N workers on fewer GPUs (rehearsal)
    if not torch.cuda.is_available() or torch.cuda.device_count() < max(devices) + 1:
        raise SystemExit(f"bench.py: --gpus {args.gpus} on devices {devices} needs {max(devices) + 1} GPU(s); this node has "
                         f"{torch.cuda.device_count() if torch.cuda.is_available() else 0} (no CPU fallback)")
    N, W, H, B, R, P = args.gpus, args.width, args.height, args.block, args.search_range, args.pairs
    total = P if args.scaling == "strong" else N * P
    key_mode = args.ref_mode == "key"
    nblk = (W // B) * (H // B)
    G = max(1, min(total, args.gen_pairs))
    gen = dict(max_step=R) if args.content == "regions" else dict(max_step=min(R, 12), region=1 << 14, noise=1)
    frames = synth.luma_sequence(G + 1, W, H, seed=synth.SEED0, **gen)
    walk = np.abs(((np.arange(total + 1) + G) % (2 * G)) - G) if G > 1 else np.arange(total + 1) % 2
    host = np.ascontiguousarray(frames[walk])
    md = MultiDevice(devices)
    try:
        md.stage_frames(host, 1 if key_mode else 0)
        t_pw = time.perf_counter()
        while time.perf_counter() - t_pw < args.prewarm_seconds:
            md.run_resident(B, R, 4)
        if args.warmup:
            md.run_resident(B, R, args.warmup)
        t0 = time.perf_counter()
        md.run_resident(B, R, args.steps)
        el = time.perf_counter() - t0
        ms = md.run_resident(B, R, args.steps, timed=True)              # per-worker HIP events (roofline leg)
        got = md.fetch(B)
    finally:
        md.close()
    counts = [MultiDevice.pair_range(total, N, k)[1] for k in range(N)]
    args_line = argparse.Namespace(**vars(args))
    args_line.pairs = total if args.scaling == "strong" else P
    out = build_line(args_line, N, el, float(ms[0]) / args.steps, counts[0], counts, nblk, N, *committed_traffic_per_pair(W, H, B, R))
    out["launcher"] = "threads (one process, ofps_hip_multi_*: one worker thread + context per GPU, no collective)"
    if len(set(devices)) != len(devices):
        out["rehearsal"] = (f"{N} workers on devices {devices}: real HIP steps through the in-process dispatcher, several workers per GPU -- "
                            "a plumbing rehearsal of the N-GPU run on fewer GPUs, NOT a scaling measurement")
    out["per_rank_ms_per_step"] = {"min": round(float(ms[ms > 0].min()) / args.steps, 4) if (ms > 0).any() else 0.0,
                                   "max": round(float(ms.max()) / args.steps, 4), "what": "HIP events per worker"}
    # ---- PCIe-inclusive stream form through the same dispatcher (ofps_hip_multi_push_frames_async: batches of 16 frames dealt to
    # the workers, vectors + island + quaternion per frame back in frame order), C++ host, frames in page-locked memory
    if not args.no_end_to_end and (B, R) in ((16, 16), (8, 32)):
        try:
            import subprocess
            from ofps_amd.build import TOOL
            sb, sf = (16, 1024 * N) if W * H <= 1920 * 1080 else (4, 32 * N)          # 4K: 4 frames per batch (33 MB), 32 frames per worker
            r = json.loads(subprocess.run([TOOL, "stream-bench", str(W), str(H), str(sf), "multi", str(sb)] + [str(d) for d in devices] +
                                          ["--block", str(B), "--range", str(R)],
                                          capture_output=True, text=True, timeout=600, check=True).stdout.strip().splitlines()[-1])
            out["end_to_end"] = {"what": f"one stream of {W}x{H} frames over the workers: every frame crosses PCIe once, {sb} frames per batch, 2 batches in "
                                         "flight per worker; per frame the vectors (16 B each), the block-motion island and the Almeida quaternion come back",
                                 "entry_points": "ofps_hip_multi_push_frames_async + ofps_hip_multi_frames_wait (C++ host layer)",
                                 "workers": r["workers"], "batch": r["batch"], "frames": r["frames"], "ms_per_frame": r["ms_per_frame"],
                                 "Mvectors_per_s": r["Mvectors_per_s"]}
        except Exception as e:
            out["end_to_end"] = {"error": repr(e)[:300]}
    mismatch = False
    try:
        import oracle
        ok = True
        for k in range(N):                                           # one pair of every worker's range
            first, cnt = MultiDevice.pair_range(total, N, k)
            if not cnt:
                continue
            kp = first + (3 * k + 1) % cnt
            prev = host[0] if key_mode else host[kp]
            ent_o, _ = oracle.sad_flow(prev, host[kp + 1], B, R, threads=4)
            ok = ok and bool((got[kp].view(np.uint32) == ent_o.view(np.uint32)).all())
        out["parity_check"] = {"what": "one searched pair per worker vs the CPU oracle, bit for bit", "ranks": N, "ok": ok}
        mismatch = not ok
    except ImportError as e:
        out["parity_check"] = {"ok": None, "skipped": str(e)}
    print(json.dumps(out), flush=True)
    if mismatch:
        print("[bench] PARITY MISMATCH against the oracle -- the line above is not a valid measurement", file=sys.stderr)
        return 3
    return 0


def main(argv=None) -> int:
    args = parse(argv)
    if args.end_to_end_only:
        from ofps_amd import synth
        fr = synth.luma_sequence(5, args.width, args.height, max_step=args.search_range)
        print(json.dumps(end_to_end_leg(fr, args.width, args.height, args.block, args.search_range, 0)), flush=True)
        return 0
    from ofps_amd import distributed as D
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.launcher == "threads":
        return run_threads(args)
    if args.gpus > 1 and not D.launched_by_torchrun():
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU, RCCL)
        if not args.stub:
            import torch
            need = (max(args.device_map) + 1) if args.device_map else args.gpus
            if torch.cuda.device_count() < need:
                raise SystemExit(f"bench.py: --gpus {args.gpus} needs {need} GPU(s), this node has {torch.cuda.device_count()}")
        return D.launch_ranks(os.path.abspath(__file__), sys.argv[1:] if argv is None else list(argv), args.gpus)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
