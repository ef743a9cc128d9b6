"""`python bench.py --gpus 2` with WORLD_SIZE unset must start its own rank processes (fresh children, before the
parent touches torch/HIP), relay rank 0's JSON line and propagate failures.  Runs here on CPU through the launcher's
self-test hook (gloo, stand-in step function): the spawn logic, rank plumbing and sharding.py exchange are real."""
import json
import os
import subprocess
import sys

from conftest import REPO

BENCH = os.path.join(REPO, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_launcher_spawns_ranks_and_reports_exchange():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "3",
                        "--frames", "5", "--selftest-launcher"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                 # exactly one JSON line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["samples_per_step"] == 2 * 3 * 5 * 256
    ex = line["exchange"]
    for k in ("broadcast_ms", "scatter_ms", "gather_ms"):
        assert ex[k] >= 0.0
    assert ex["gather_bytes"] == 2 * 3 * 5 * 256 * 4
    assert "SELF-TEST" in line["data"]                     # can never be mistaken for a measurement


def test_launcher_eight_ranks_reports_every_ranks_step_time():
    """The node's shape (8 ranks): one line, per-rank step times (min / max / all) so that stragglers are visible, and the
    scatter / gather legs of the exchange over 8 uneven-free shards."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--frames", "4", "--selftest-launcher"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["config"]["samples_per_step"] == 8 * 2 * 4 * 256
    ranks = line["ms_per_step_ranks"]
    assert len(ranks["all"]) == 8 and ranks["min"] <= ranks["max"] <= line["ms_per_step"] * 1.0001
    ex = line["exchange"]
    assert ex["gather_bytes"] == 8 * 2 * 4 * 256 * 4 and ex["scatter_dtype"] == "float32" and ex["gather_ms"] >= 0
    assert "rows" not in line                                      # secondary rows are an N = 1 matter


def test_launcher_propagates_rank_failure():
    # rank 1 dies before the first collective; rank 0 would wait for it forever: the launcher must end it and fail
    env = _env()
    env["CTTS_BENCH_SELFTEST_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2",
                        "--frames", "4", "--selftest-launcher"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 7
    assert "rank 1 exited with code 7" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_parent_does_not_import_torch_before_spawning():
    """The launcher role runs before any torch import (a GPU-initialised parent must never fork/exec)."""
    code = (
        "import sys; sys.argv = ['bench.py', '--gpus', '2']\n"
        "import bench\n"
        "seen = {}\n"
        "def fake(n, argv, **kw):\n"
        "    seen['n'] = n; seen['torch'] = 'torch' in sys.modules; return 0\n"
        "bench.launch_ranks = fake\n"
        "try:\n"
        "    bench.main()\n"
        "except SystemExit as e:\n"
        "    assert e.code == 0\n"
        "assert seen == {'n': 2, 'torch': False}, seen\n")
    r = subprocess.run([sys.executable, "-c", code], env=_env(), cwd=REPO, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]


def test_launcher_keeps_per_rank_stderr_and_echoes_the_failing_rank(tmp_path):
    env = _env()
    env["CTTS_BENCH_SELFTEST_FAIL_RANK"] = "1"
    env["CTTS_BENCH_LOG_DIR"] = str(tmp_path)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2",
                        "--frames", "4", "--selftest-launcher"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 7
    assert (tmp_path / "rank0.err").exists() and "fails on purpose" in (tmp_path / "rank1.err").read_text()
    assert "tail of" in r.stderr and "fails on purpose" in r.stderr


def test_launcher_wall_clock_limit_ends_its_children(tmp_path):
    # rank 1 never reaches the first collective and never exits: the launcher's own limit must end both ranks
    import time
    env = _env()
    env["CTTS_BENCH_SELFTEST_HANG_RANK"] = "1"
    env["CTTS_BENCH_LOG_DIR"] = str(tmp_path)
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2",
                        "--frames", "4", "--selftest-launcher", "--launch-timeout", "20"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 124 and time.time() - t0 < 120
    assert "wall-clock limit" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_defaults_and_profiler_guard(monkeypatch):
    """Defaults the driver relies on: N = 1, the secondary rows on, the CPU aggregate on 300-frame utterances; and the guard that
    keeps a profiled run (rocprofv3's library initialises the GPU before Python starts) from starting child processes."""
    sys.path.insert(0, REPO)
    import bench
    a = bench.parse_args([])
    assert a.gpus == 1 and a.rows == "config3,config3f16,f16server,bf16x6,config4,config5" and not a.no_rows
    assert a.cpu_frames == 900 and a.cpu_aggregate_frames == 300 and a.cpu_budget <= 30
    for k in list(os.environ):
        if k.startswith(("ROCPROF", "ROCP_")):
            monkeypatch.delenv(k)
    monkeypatch.delenv("LD_PRELOAD", raising=False)
    monkeypatch.delenv("HSA_TOOLS_LIB", raising=False)
    assert not bench.under_profiler()
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    assert bench.under_profiler()
    monkeypatch.delenv("LD_PRELOAD")
    monkeypatch.setenv("ROCPROFILER_SOMETHING", "1")
    assert bench.under_profiler()
