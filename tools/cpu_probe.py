#!/usr/bin/env python3
"""Sustained rate of the CPU baseline (oracle full search, AVX2 inner loop) per thread count on this host, 2 s each;
prints the cgroup CPU quota next to it (a throttled container makes short bursts look faster than it can sustain)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from ofps_amd import synth
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(p): print(p, open(p).read().strip())
print("sched_getaffinity", len(os.sched_getaffinity(0)), "omp max", oracle.num_threads())
fr = synth.luma_sequence(17, 1920, 1080, max_step=16)
for t in (8, 16, 32, 48, 64, 96, 128):
    if t > oracle.num_threads(): break
    oracle.sad_flow(fr[0], fr[1], 16, 16, threads=t)
    n = 0; t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        oracle.sad_flow(fr[n % 16], fr[n % 16 + 1], 16, 16, threads=t); n += 1
    el = time.perf_counter() - t0
    print(f"threads {t:4d}: {el / n * 1e3:8.3f} ms/pair sustained, {8040 * n / el / 1e6:7.2f} Mvectors/s")
