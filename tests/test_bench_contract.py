"""bench.py's JSON line must keep the driver's contract.  The line is built by bench.build_line() from measured times;
here it is fed synthetic timings (no GPU) and checked key by key, and the argument defaults are checked against
BASELINE.json's configs[1] by parsing, not by grepping the source."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _line(argv, world, el, launch_ms, counts):
    args = bench.parse(argv)
    nblk = (args.width // args.block) * (args.height // args.block)
    return args, json.loads(json.dumps(bench.build_line(args, world, el, launch_ms, counts[0], counts, nblk, world, traffic_per_pair=4_000_000)))


def test_line_from_synthetic_timings_follows_the_contract():
    # 20 steps of 4.1 ms, 256 pairs of 8,040 vectors on one rank
    args, d = _line([], 1, 20 * 4.1e-3, 4.08, [256])
    for k, t in {"metric": str, "value": (int, float), "unit": str, "n_gpus": int, "steps": int, "warmup": int,
                 "ms_per_step": (int, float), "higher_is_better": bool, "scaling": str, "dtype": str, "data": str,
                 "config": dict, "roofline": dict}.items():
        assert k in d and isinstance(d[k], t), k
    assert "vs_baseline" in d and d["vs_baseline"] is None          # BASELINE.md publishes no number for this metric
    assert d["unit"] == "Mvectors/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "u8"
    assert d["data"] == "synthetic" and d["n_gpus"] == 1 and d["ranks_seen"] == 1
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["workload"].startswith("cfg2")
    assert abs(d["value"] - 256 * 8040 / 4.1e-3 / 1e6) < 1e-2       # vectors / time
    assert abs(d["ms_per_step"] - 4.1) < 1e-9 and abs(d["ms_per_frame_pair"] - 4.1 / 256) < 1e-4
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    algo = 256 * (2 * 1920 * 1080 + 16 * 8040)                      # 4,275,840 B per pair
    assert r["algorithmic_bytes_per_launch"] == algo
    assert abs(r["achieved"] - algo / 4.08e-3 / 1e9) < 0.01 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["traffic"] == 4_000_000 * 256
    assert r["valu"]["abs_diffs_per_launch"] == 256 * 8040 * 256 * 33 * 33


def test_weak_and_strong_lines_count_pairs_the_right_way():
    # weak, 8 ranks: every rank its own 256 pairs -> value is the whole-job aggregate
    _, w = _line(["--gpus", "8"], 8, 20 * 4.1e-3, 4.08, [256] * 8)
    assert w["n_gpus"] == 8 and w["scaling"] == "weak" and w["config"]["pairs_per_step"] == 2048
    assert abs(w["value"] - 8 * 256 * 8040 / 4.1e-3 / 1e6) < 0.1
    # strong, cfg4 preset: ONE batch of 64 pairs split 8 ways; rank 0's launch covers 8 pairs
    args, s = _line(["--gpus", "8", "--config", "cfg4"], 8, 20 * 2.4e-3, 2.3, [8] * 8)
    assert (args.width, args.height, args.block, args.search_range, args.pairs, args.scaling) == (3840, 2160, 8, 32, 64, "strong")
    assert s["scaling"] == "strong" and s["config"]["pairs_per_step"] == 64 and s["config"]["pairs_per_rank"] == [8] * 8
    assert s["config"]["workload"].startswith("cfg4") and "gather_results" in s["config"]["step"]
    assert abs(s["value"] - 64 * 129600 / 2.4e-3 / 1e6) < 0.5
    assert s["roofline"]["pairs_per_launch"] == 8
    assert s["roofline"]["algorithmic_bytes_per_launch"] == 8 * (2 * 3840 * 2160 + 16 * 129600)


def test_defaults_are_baseline_configs_1():
    a = bench.parse([])
    assert (a.width, a.height, a.block, a.search_range, a.gpus, a.scaling, a.pairs, a.ref_mode) == \
        (1920, 1080, 16, 16, 1, "weak", 256, "pairs")
    cfg = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"][1]
    assert "1080p" in cfg and "16" in cfg                           # the config the defaults stand for
    # explicit flags win over a preset
    b = bench.parse(["--config", "cfg4", "--pairs", "16"])
    assert b.pairs == 16 and b.width == 3840


def test_device_map_rules():
    """gloo with real compute is the one-GPU rehearsal of an N-rank run: allowed, with one device per rank; a
    repeated device needs gloo (RCCL refuses two ranks on one GPU); the line then says it is a rehearsal."""
    import pytest
    a = bench.parse(["--gpus", "2", "--backend", "gloo", "--device-map", "0,0"])
    assert a.device_map == [0, 0] and a.backend == "gloo" and not a.stub
    assert bench.parse(["--gpus", "2", "--device-map", "1,0"]).device_map == [1, 0]            # distinct devices: RCCL is fine
    for bad in (["--gpus", "2", "--device-map", "0,0"],                                         # repeated device under nccl
                ["--gpus", "2", "--backend", "gloo", "--device-map", "0"],                      # one entry per rank
                ["--gpus", "1", "--backend", "gloo", "--device-map", "x"]):
        with pytest.raises(SystemExit):
            bench.parse(bad)
    args = bench.parse(["--gpus", "2", "--backend", "gloo", "--device-map", "0,0", "--config", "cfg4"])
    d = bench.build_line(args, 2, 1.0, 1.0, 32, [32, 32], 129600, 2)
    assert "rehearsal" in d and "NOT a scaling measurement" in d["rehearsal"]
    assert "rehearsal" not in bench.build_line(bench.parse([]), 1, 1.0, 1.0, 256, [256], 8040, 1)


def test_line_says_what_each_row_is_pinned_to():
    _, d = _line([], 1, 20 * 4.1e-3, 4.08, [256])
    pin = d["parity_pin"]
    assert any(k.startswith("N1") and "build-defined spec" in v for k, v in pin.items())
    assert any(k.startswith("N2") and "build-defined spec" in v for k, v in pin.items())
    assert any(k.startswith("A6-A12") and "reference-held" in v for k, v in pin.items())
    assert any(k.startswith("A1-A5") and "hand-derived" in v for k, v in pin.items())


def test_traffic_is_flagged_stale_when_the_kernel_source_changed(tmp_path, monkeypatch):
    import json as _json
    v, stale = bench.committed_traffic_per_pair(1920, 1080, 16, 16)
    d = _json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))["sad_1920x1080_b16_r16"]
    assert v == d["hbm_bytes_per_pair"]
    assert stale == (d.get("kernel_source_sha16") != bench.kernel_source_sha16())
    monkeypatch.setattr(bench, "kernel_source_sha16", lambda name="sad.hip": "0" * 16)
    assert bench.committed_traffic_per_pair(1920, 1080, 16, 16)[1] is True
    _, line = _line([], 1, 1.0, 1.0, [256])
    assert "traffic_stale" in line["roofline"]
