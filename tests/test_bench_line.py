"""The one line bench.py prints is what the driver records: it keeps the contract keys, `config`, `roofline` and `cpu_baseline`
whole, only the NAMES of other keys, and cuts long lines (round 3 lost BASELINE's second metric that way).  compact_line()
is checked here on a full record of a real run (profiles/r04_bench_default.json, the detail file of the committed
collection) and on the shapes the other launch modes produce."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")


def test_compact_line_carries_both_baseline_metrics_and_the_targets():
    b = bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default.json")))
    line = b.compact_line(full)
    text = json.dumps(line)
    assert len(text) < 6000                                   # the driver keeps an 8 KB tail
    for k in CONTRACT:
        assert k in line, k
    assert line["metric"].startswith("hops/sec (1024-hop HPR") and line["unit"] == "hops/s" and line["dtype"] == "f32"
    assert line["vs_baseline"] is None and line["scaling"] == "weak" and line["higher_is_better"] is True
    assert "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] and 0.2 < r["valu_issue_frac"] < 1.0
    # BASELINE's second metric as flat scalars of `roofline` (the driver's record keeps the scalars of `roofline` and `config`,
    # cuts strings at ~160 characters and drops nested objects), the detail nested beside them
    assert 0.5 < r["median47_frac"] < 1.0 and r["median47_launches"] > 1000 and "8 B/element" in r["median47_is"]
    assert abs(r["median47_traffic"] / (25840 * 4096) - 8.0) < 0.1
    assert r["median47"]["sustained_seconds"] >= 1.0
    for k, v in r.items():
        if isinstance(v, str):
            assert len(v) <= 200, (k, len(v))
    cfg = line["config"]                                       # the north star's offline target, and the host-vector figure
    assert cfg["offline_batch_x_realtime"] > 10000 and cfg["offline_long_x_realtime"] > 10000
    assert cfg["offline_host_x_realtime"] > 10000 and cfg["per_hop_api_us"] > 0 and cfg["per_hop_api_resident_us"] > 0
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "hops/s" and "sample" in c
    assert 0.5 < line["legs"]["offline_host"]["frac_of_link_roof"] < 1.0


def test_compact_line_without_legs_and_at_several_ranks():
    b = bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default.json")))
    bare = {k: v for k, v in full.items() if k not in ("offline_batch", "offline_long", "offline_host", "all_outputs", "s_noise",
                                                        "sse_block", "realtime", "roofline_median", "cpu_baseline", "gpu_over_cpu")}
    line = b.compact_line(bare)                                # --no-legs --no-cpu-baseline --no-realtime
    assert "legs" not in line and "cpu_baseline" not in line and "median47" not in line["roofline"]
    assert line["value"] == full["value"]
    many = dict(bare, n_gpus=8, offline_batch_sharded={"x_realtime": 2.0e6, "ms_per_step": 7.5, "value": 4.0e8, "ranks_reported": 8,
                                                        "config": {"clips_total": 512}})
    line = b.compact_line(many)                                # --gpus 8: the sharded figure where the record keeps it
    assert line["config"]["offline_batch_sharded_x_realtime"] == 2.0e6
    assert line["config"]["offline_batch_sharded_ranks_reported"] == 8 and line["config"]["offline_batch_sharded_clips_total"] == 512


def test_compact_line_of_round_5_carries_the_per_hop_sweep():
    """Round 5: the per-hop call of the reference's API at the other hops of its sweep and on the SSE path (BASELINE configs[4],
    hop 512) inside `config`, per launch and with the resident kernel -- from the full record of the round's final collection."""
    b = bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))
    line = b.compact_line(full)
    assert len(json.dumps(line)) < 6000
    cfg = line["config"]
    for key in ("per_hop_api_us_by_hop", "per_hop_api_resident_us_by_hop"):
        assert set(cfg[key]) == {"256", "512", "2048", "4096", "sse_512"}, cfg[key]
        assert all(0 < v < 100 for v in cfg[key].values())
    assert cfg["per_hop_api_resident_us_by_hop"]["sse_512"] < 15.0          # VERDICT r4's target for the resident SSE hop
    assert cfg["per_hop_api_resident_us_by_hop"]["2048"] < 30.0 and cfg["per_hop_api_resident_us_by_hop"]["4096"] < 40.0
    recorded = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default_line.json")))
    assert recorded["config"]["per_hop_api_us_by_hop"] == cfg["per_hop_api_us_by_hop"]
