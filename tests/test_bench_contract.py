"""The bench line committed as evidence (profiles/r01/bench_n1.json = stdout of `python bench.py` on an MI355X) must keep
the driver's contract: one JSON object with the required keys, the roofline and cpu_baseline objects, sane values."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_follows_the_contract():
    line = open(os.path.join(ROOT, "profiles", "r01", "bench_n1.json")).read().strip().splitlines()[-1]
    d = json.loads(line)
    for k, t in {"metric": str, "value": (int, float), "unit": str, "n_gpus": int, "steps": int, "warmup": int,
                 "ms_per_step": (int, float), "higher_is_better": bool, "scaling": str, "dtype": str, "data": str,
                 "config": dict, "roofline": dict, "cpu_baseline": dict}.items():
        assert k in d and isinstance(d[k], t), k
    assert "vs_baseline" in d and d["vs_baseline"] is None          # BASELINE.md publishes no number for this metric
    assert d["unit"] == "Mvectors/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "u8"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["traffic"] is None or r["traffic"] > 0
    # value is consistent with ms_per_step and the workload
    vec = d["config"]["pairs_per_step"] * d["config"]["vectors_per_pair"] * d["n_gpus"]
    assert abs(d["value"] - vec / d["ms_per_step"] / 1e3) / d["value"] < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    assert d["value"] / c["value"] >= 30.0                          # north_star target: >= 30x the CPU path of the same box
    assert d["parity_check"]["ok"] is True                           # the searched batch was checked against the oracle


def test_bench_defaults_match_baseline_config():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for needle in ('"--width", type=int, default=1920', '"--height", type=int, default=1080', '"--block", type=int, default=16',
                   'dest="search_range", type=int, default=16', '"--gpus", type=int, default=1'):
        assert needle in src, needle
