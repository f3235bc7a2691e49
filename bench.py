#!/usr/bin/env python3
"""bench.py -- Mvectors/s of the flow hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the full-search SAD block matcher (N1, the dominant kernel of the hot path)
over one batch of P = 256 consecutive 1080p frame pairs (a 257-frame sequence) already resident in HBM: one launch of
sad_strip_kernel<16,16> through the C ABI (ofps_hip_sad_flow_dev).  Workload = BASELINE.json
configs[1] (1080p synthetic, 16x16 blocks, +-16 full search), one GPU's worth per rank (weak scaling:
independent frame pairs per GPU, no data-path collective -- SURVEY.md 8e).

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline     -- algorithmic bytes per launch / average launch duration (HIP events on the launch
                  stream) against the 8 TB/s HBM peak, plus the packed-SAD VALU view of the same
                  launch (the kernel is VALU-bound);
  cpu_baseline -- the oracle's scalar full search (OpenMP over block rows) on the host cores, timed
                  on a bounded sample of the same frames (rank 0, N=1 only).
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
# packed-SAD VALU peak: tools/ubench_sad (profiles/ubench_sad_r01.txt) measures v_qsad_pk_u16_u8 at ~16 and
# v_sad_u8 at ~4 cycles per wave64 instruction per SIMD, i.e. the SAD unit retires 64 |a-b| per clock per
# SIMD either way: 256 CU x 4 SIMD x 64 x 2.4 GHz = 157 T|a-b|/s.
SAD_ABSDIFF_PER_CLK_PER_SIMD = 64.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=256,
                    help="frame pairs per step = per launch: a 257-frame 1080p sequence resident in HBM (0.53 GB), made of "
                         "--gen-pairs generated pairs traversed forward and backward")
    ap.add_argument("--gen-pairs", type=int, default=64, help="distinct frame pairs generated on the host (65 frames)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--block", type=int, default=16)
    ap.add_argument("--range", dest="search_range", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--sad-mode", choices=["exhaustive", "pruned"], default="exhaustive",
                    help="search strategy of the SAD kernel (both return the same bits; pruned is content-dependent)")
    ap.add_argument("--content", choices=["regions", "camera"], default="regions",
                    help="regions (default, SURVEY.md 8d): an independent integer displacement per 64x64 region and frame.  "
                         "camera: one global translation per frame + sensor noise +-1 (smooth camera motion, the decoder's "
                         "real input) -- the content the opt-in pruned mode is for")
    ap.add_argument("--ref-mode", choices=["pairs", "key"], default="pairs",
                    help="pairs: frame k vs k+1 (default, no collective).  key: every frame vs one shared key frame that rank 0 "
                         "broadcasts to all ranks over RCCL inside the timed step (SURVEY.md 8e, north_star's shared-reference case)")
    ap.add_argument("--pipeline", action="store_true",
                    help="also time the fused tail (detect + Almeida LSQ) per step; reported under 'pipeline'")
    return ap.parse_args()


def cpu_baseline(frames: np.ndarray, block: int, rng: int, budget_s: float):
    """The oracle (kind 'port': C restatement, gcc -O3 -fopenmp) on all host cores, bounded sample."""
    import oracle
    nblk = (frames.shape[2] // block) * (frames.shape[1] // block)
    # pick the thread count that is actually fastest on this host (cgroup limits, SMT): one pair each
    cands = sorted({t for t in (oracle.num_threads(), 128, 64, 32, 16, 8) if 1 < t <= oracle.num_threads()} | {1})
    best_t, best_dt = 1, None
    single_dt = None
    for t in cands:
        t0 = time.perf_counter()
        oracle.sad_flow(frames[0], frames[1], block, rng, threads=t)
        dt = time.perf_counter() - t0
        if t == 1:
            single_dt = dt
        if best_dt is None or dt < best_dt:
            best_t, best_dt = t, dt
    threads = best_t
    done = 0
    t0 = time.perf_counter()
    k = 0
    while True:
        oracle.sad_flow(frames[k % (len(frames) - 1)], frames[k % (len(frames) - 1) + 1], block, rng, threads=threads)
        done += 1; k += 1
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
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
    return {"value": round(done * nblk / el / 1e6, 4), "unit": "Mvectors/s", "cores": threads, "kind": "port", "tail_single_thread": tail,
            "sample": f"{done} frame-pair searches cycling over the bench sequence, {frames.shape[2]}x{frames.shape[1]}, "
                      f"{block}x{block} blocks, +-{rng}, {el:.1f} s wall",
            "ms_per_pair": round(el / done * 1e3, 2),
            # how the reference runs an estimator / detector: one thread per plugin call (tracking/worker.rs:347-352)
            "single_thread": {"value": round(nblk / single_dt / 1e6, 4), "ms_per_pair": round(single_dt * 1e3, 1), "sample": "1 pair"}}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # under torchrun (RANK/MASTER_PORT set) the process group is always created, also at world size 1, so the
    # RCCL init / barrier / max-reduce path of an N-GPU run can be exercised on a 1-GPU box
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)

    from ofps_amd import synth
    from ofps_amd.runtime import HipContext

    W, H, B, R, P = args.width, args.height, args.block, args.search_range, args.pairs
    stride = (W + 63) // 64 * 64
    # independent sequence per rank (weak scaling): same generator, different seed
    # G generated pairs (G+1 frames); the resident sequence of P+1 frames walks them forward and backward (every
    # consecutive pair of it is a generated pair), so a step is long enough for the GPU to sit at its sustained clock
    # -- a 64-pair step (1 ms) left the first steps of a short run ~10 % slow -- without 150 s of host-side generation
    G = max(1, min(P, args.gen_pairs))
    gen = dict(max_step=R) if args.content == "regions" else dict(max_step=min(R, 12), region=1 << 14, noise=1)
    frames = synth.luma_sequence(G + 1, W, H, seed=synth.SEED0 + 1000 * rank, stride=stride, **gen)
    walk = np.abs(((np.arange(P + 1) + G) % (2 * G)) - G) if G > 1 else np.arange(P + 1) % 2
    d_frames = torch.from_numpy(frames).cuda(non_blocking=False)[torch.from_numpy(walk).cuda()].contiguous()
    nbx, nby = W // B, H // B
    nblk = nbx * nby
    d_out = torch.empty((P, nblk, 4), dtype=torch.float32, device="cuda")

    ctx = HipContext(local_rank)
    ctx.use_torch_stream()            # launches go to torch's current stream: torch events see them
    ctx.set_sad_mode(ctx.SAD_PRUNED if args.sad_mode == "pruned" else ctx.SAD_EXHAUSTIVE)

    key_mode = args.ref_mode == "key"

    def step():
        if key_mode and use_dist:
            # the shared reference frame travels rank 0 -> all ranks (RCCL broadcast, on torch's current stream like the
            # search that follows it); its slot is frame 0 of every rank's resident sequence
            dist.broadcast(d_frames[0], src=0)
        ctx.sad_flow_dev(d_frames.data_ptr(), P + 1, W, H, stride, stride * H, 1 if key_mode else 0, B, R, d_out.data_ptr(), None)

    token = torch.zeros(1, dtype=torch.int32, device="cuda") if use_dist else None

    def barrier():
        # an all-reduce of one int on the launch stream (dist.barrier() builds a fresh tensor and device-synchronises by
        # itself: measured ~1 ms per call under RCCL, which lands inside the timed region)
        if use_dist:
            dist.all_reduce(token)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # ---- per-launch duration with HIP events on the launch stream (roofline leg)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in evs:
        a.record(); step(); b.record()
    torch.cuda.synchronize()
    launch_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))

    out = None
    if rank == 0:
        vectors_per_step = world * P * nblk
        ms_per_step = el / args.steps * 1e3
        algo_bytes = P * (2 * W * H + 16 * nblk)                         # SURVEY.md 8d, per pair x pairs per launch
        achieved = algo_bytes / (launch_ms * 1e-3) / 1e9
        abs_diffs = P * nblk * B * B * (2 * R + 1) ** 2
        valu_peak = 256 * 4 * 2.4e9 * SAD_ABSDIFF_PER_CLK_PER_SIMD
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                per_pair = tj.get(f"sad_{W}x{H}_b{B}_r{R}", {}).get("hbm_bytes_per_pair")
                traffic = per_pair * P if per_pair is not None else None
            except Exception:
                traffic = None
        out = {
            "metric": ("Mvectors/s, 1080p 16x16 blocks +-16 full-search SAD" if (W, H, B, R) == (1920, 1080, 16, 16)
                       else f"Mvectors/s, {W}x{H} {B}x{B} blocks +-{R} full-search SAD"),
            "value": round(vectors_per_step * args.steps / el / 1e6, 3),
            "unit": "Mvectors/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "ms_per_frame_pair": round(ms_per_step / P, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": ({(1920, 1080, 16, 16): "cfg2 (BASELINE.json configs[1]): ",
                                     (3840, 2160, 8, 32): "cfg4 (BASELINE.json configs[3]): ",
                                     (640, 360, 16, 8): "cfg1 geometry (BASELINE.json configs[0]): "}.get((W, H, B, R), "")
                                    + f"{W}x{H} synthetic luma, {B}x{B} blocks, +-{R} full-search SAD"),
                       "pairs_per_step": P, "generated_pairs": G, "content": args.content, "vectors_per_pair": nblk, "parallelism": (f"frame-pair sharding x{world}" + (", key frame broadcast from rank 0 per step (RCCL)" if key_mode and use_dist else "")),
                       "ref_mode": args.ref_mode,
                       "kernel": (f"sad_strip_kernel<{B},{R}>" if args.sad_mode == "exhaustive" else "sad_pde_kernel + sad_strip_kernel<16,16> on overflow strips"),
                       "sad_mode": args.sad_mode},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes, "launch_ms": round(launch_ms, 5),
                         "note": "full search is VALU-bound (SURVEY.md 8d): see 'valu'",
                         "valu": {"abs_diffs_per_launch": abs_diffs,
                                  "achieved_Tops": round(abs_diffs / (launch_ms * 1e-3) / 1e12, 3),
                                  "peak_Tops": round(valu_peak / 1e12, 3),
                                  "frac": round(abs_diffs / (launch_ms * 1e-3) / valu_peak, 4),
                                  "peak_basis": "SAD unit: 64 |a-b| per clock per SIMD (v_qsad_pk_u16_u8 16 cyc, measured)"}},
        }
        if args.sad_mode == "pruned":
            out["roofline"]["valu"]["note"] = ("exhaustive-equivalent rate: the pruned search returns the same winners but "
                                               "evaluates fewer |a-b|, so this fraction can exceed 1")
            out["roofline"]["traffic"] = None             # the committed PMC traffic figure belongs to the exhaustive kernel

    if args.pipeline:
        d_res = torch.empty((P, 4), dtype=torch.int32, device="cuda")
        dim = ctx.block_dim(0.05, 3)
        d_field = torch.empty((P, dim * dim, 2), dtype=torch.float32, device="cuda")
        d_quat = torch.empty((P, 4), dtype=torch.float32, device="cuda")

        def full():
            step()
            ctx.detect_dev(d_out.data_ptr(), nblk, P, 0.05, 3, 0.003, d_res.data_ptr(), d_field.data_ptr())
            ctx.almeida_dev(d_out.data_ptr(), nblk, P, W / H, 39.6 * H / W, False, 0, 0.05, 0, 0, d_quat.data_ptr())
        for _ in range(2):
            full()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            full()
        torch.cuda.synchronize()
        pel = time.perf_counter() - t1
        if out is not None:
            # (a second, high-priority HIP stream for the tail was measured: no gain -- the SAD grid occupies every
            # CU and the tail's ~40 dependent small launches only trickle through; the serial chain is reported)
            out["pipeline"] = {"stages": "sad -> block-motion detect -> almeida LSQ (device resident)",
                               "ms_per_step": round(pel / args.steps * 1e3, 4),
                               "Mvectors_per_s_per_gpu": round(P * nblk * args.steps / pel / 1e6, 3)}

    # ---- outside the timed region: every rank checks one pair of the batch it just searched against the CPU oracle
    # (the oracle is the checker here, never the thing measured); rank 0 reports whether all ranks agreed
    ok = None
    if not key_mode:
        try:
            import oracle
            k = (3 * rank + 1) % P                               # a different pair on every rank
            ent_o, _ = oracle.sad_flow(np.ascontiguousarray(frames[walk[k]][:, :W]), np.ascontiguousarray(frames[walk[k + 1]][:, :W]),
                                       B, R, threads=4)
            ok = bool((d_out[k].cpu().numpy().view(np.uint32) == ent_o.view(np.uint32)).all())
        except Exception as e:                                   # no oracle on this machine: report "not checked"
            print(f"[bench] parity check skipped on rank {rank}: {e}", file=sys.stderr)
    if use_dist:
        t = torch.tensor([-1 if ok is None else int(ok)], dtype=torch.int32, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = None if int(t.item()) < 0 else bool(t.item())
    if out is not None:
        out["parity_check"] = {"what": "one searched pair per rank vs the CPU oracle, bit for bit", "ranks": world, "ok": ok}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(frames[:, :, :W].copy() if stride != W else frames, B, R, args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
