#!/usr/bin/env python3
"""Headline benchmark: WaveGlow inference throughput (audio samples/s at 22.05 kHz).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path - ``WaveGlow.infer`` (glow.py:314-350 semantics) of
BASELINE.json config 2: full WaveGlow (12 flows, 512 WN channels, 8 groups), fp32, on a batch
of 8 synthetic 80x900 mels already resident in HBM, random-init weights from the deterministic
recipe in cookietts_amd/synthetic.py.  ``--dtype bf16 --batch 32`` is config 3.

Multi-GPU (one process per GPU, RCCL): for N > 1 every rank runs its own batch (utterance-batch
sharding, weak scaling, no data-path collective); the timed region is bracketed by barrier +
synchronize and the MAX over ranks is used.  Rank 0 prints ONE JSON line.  Two ways in:

  * ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` (WORLD_SIZE set), or
  * plain ``python bench.py --gpus N``: this process then acts as the launcher - it starts N fresh
    rank processes BEFORE touching torch/HIP itself, relays rank 0's JSON line and exits non-zero if
    any rank fails.

Rank 0 synthesises the weights; for N > 1 the other ranks receive them through
``cookietts_amd.sharding.broadcast_state_dict`` (the once-per-model RCCL broadcast of SURVEY.md §8e).
Beside the timed weak-scaling steps the per-request exchange of §8e is timed once per run:
``scatter_mels`` (rank 0 -> ranks) and ``gather_waves`` (direct point-to-point into rank 0), reported as
``exchange.{broadcast_ms, scatter_ms, gather_ms}``.

Extra objects on the line:
  roofline     - the dominant kernel (WN in-layer conv-GEMM: dilated conv + cond + gate),
                 algorithmic FLOPs per launch / mean launch time measured with HIP events
                 inside the library on the launch stream, vs the MFMA peak of the dtype; beside it the
                 res GEMM, the deferred skip GEMM and the whole step against the same peak.
  rows         - N = 1 only, AFTER the timed region: short runs of config 3 (bf16, B=32, with its own
                 roofline), the bf16x6 GEMM mode, config 4 (WaveFlow, B=1 and B=8) and config 5 (Tacotron2
                 decoder, 900 forced steps, chained into the headline's vocoder), ~7 s; a failing row is
                 recorded in place and never takes the headline down (--no-rows / --rows '' to skip).
  ms_per_step_ranks - every rank's own time for the K steps (min / max / all): stragglers at N > 1.
  cpu_baseline - oracle/waveglow_torch_cpu.py (a reference-free torch-CPU restatement pinned to the
                 reference's goldens; test infrastructure) timed on this box's physical cores on ONE
                 full 80x900 utterance of the same model: 1 warm-up + best-of-N (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
import types

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

METRIC = "audio samples/sec (22.05kHz) WaveGlow infer, 80×900 mel, 1/2/4/8 GPU; real-time factor"
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA
# HBM bytes per in-layer launch from committed PMC passes (2 x FETCH_SIZE (gfx950 half-count correction,
# calibrated on flow_tail) + WRITE_SIZE) of the exact launch shapes named in the file; not re-measured inside a bench run
# (PMC collection needs its own rocprofv3 passes: scripts/pmc3.sh).
TRAFFIC_FILES = ["r6_pmc_traffic.json", "r5_pmc_traffic.json", "r4_pmc_traffic.json", "r3_pmc_traffic.json", "r1_17_pmc_traffic.json"]


def load_traffic():
    for name in TRAFFIC_FILES:
        try:
            with open(os.path.join(REPO, "profiles", name)) as f:
                return {k: v for k, v in json.load(f).items() if isinstance(v, dict)}, "profiles/" + name
        except Exception:
            continue
    return {}, None


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="utterances per GPU (config 2: 8; config 3: 32)")
    ap.add_argument("--frames", type=int, default=900)
    ap.add_argument("--config", default="full", help="key of cookietts_amd.synthetic.WAVEGLOW_CONFIGS")
    ap.add_argument("--cpu-frames", type=int, default=900,
                    help="mel frames of the CPU-baseline utterance (0 = skip; default = the metric's 900)")
    ap.add_argument("--cpu-budget", type=float, default=25.0, help="wall-time budget (s) of the CPU-baseline repeats")
    ap.add_argument("--cpu-aggregate-frames", type=int, default=300,
                    help="mel frames per utterance of the concurrent-instances leg of the CPU baseline (the single-instance leg "
                         "runs --cpu-frames; the shorter utterance keeps the whole CPU part near 90 s; samples/s is what is "
                         "compared, and the WN convolutions are linear in the utterance length)")
    ap.add_argument("--rows", default="config3,config3f16,f16server,bf16x6,config4,config5",
                    help="N = 1 only: short secondary rows run AFTER the headline's timed region and attached to the JSON "
                         "line as \"rows\" (config3 = bf16 B=32, bf16x6 = six-product loop at the headline batch, config4 = "
                         "WaveFlow B=8 and B=1, config5 = Tacotron2 900 forced steps B=4 + chained vocoder); empty = none")
    ap.add_argument("--no-rows", action="store_true", help="same as --rows ''")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="launcher role (--gpus N > 1): wall-clock seconds after which the rank processes are terminated (exit 124)")
    ap.add_argument("--no-exchange-dry-run", action="store_true",
                    help="N = 1: skip the world-size-1 RCCL pass over the broadcast / scatter / gather code (outside the timed steps)")
    ap.add_argument("--no-cpu-aggregate", action="store_true",
                    help="skip the concurrent-instances leg of the CPU baseline (single-instance figure only)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-exchange", action="store_true", help="skip the scatter/gather timing for N > 1")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f16", "bf16x3"],
                    help="f32 = BASELINE config 2 (default, the headline); bf16 = config 3 (use --batch 32); f16 = config 3's "
                         "kernels on IEEE-half storage / v_mfma_f32_32x32x16_f16 (the reference's own half mode; extra row); bf16x3 = "
                         "split-bf16 (hi + lo operands, three bf16 MFMA products per contraction, fp32 accumulate): an "
                         "extra row, never the headline")
    ap.add_argument("--gemm-mode", default="f32", choices=["f32", "bf16x3", "bf16x6"],
                    help="main loop of the fp32 conv-GEMM (--dtype f32 only): f32 = fp32 MFMA products (default, the "
                         "headline); bf16x3 = operands split in registers into hi + lo bf16, three bf16 MFMA products, "
                         "fp32 accumulate; bf16x6 = hi + mid + lo (24 mantissa bits), the six products >= 2^-16 - "
                         "extra rows, labelled in the line's dtype.  Travels in the model's config "
                         "struct; there is no process-wide default")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="test hook for tests/test_bench_launcher.py: gloo on CPU with a stand-in step function; "
                         "exercises the launcher, the rank plumbing and sharding.py only - measures nothing")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------- launcher ----
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, timeout_s=1500.0, log_dir=None):
    """Start ``n`` fresh rank processes of this script.  The parent has not imported torch and makes
    no HIP call, so nothing GPU-initialised is ever forked or exec'd.  Every rank's stderr goes to
    ``<log_dir>/rank<r>.err`` (the tail of a failing rank's file is echoed); a wall-clock ``timeout_s``
    ends exactly the children this function started and returns 124.  Returns the exit code."""
    port = _free_port()
    log_dir = log_dir or os.environ.get("CTTS_BENCH_LOG_DIR") or os.path.join(REPO, "gpurun_out", "bench_logs")
    os.makedirs(log_dir, exist_ok=True)
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        errs.append(open(os.path.join(log_dir, f"rank{r}.err"), "w"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=errs[-1], text=True))

    def end_children(which):
        for r in which:                       # exactly these PIDs
            if procs[r].poll() is None:
                procs[r].terminate()
        for r in which:
            try:
                procs[r].wait(20)
            except subprocess.TimeoutExpired:
                procs[r].kill()

    def echo_tail(r, lines=30):
        errs[r].flush()
        try:
            with open(os.path.join(log_dir, f"rank{r}.err")) as f:
                tail = f.read().splitlines()[-lines:]
        except OSError:
            tail = []
        sys.stderr.write(f"[bench launcher] ---- tail of {log_dir}/rank{r}.err ----\n" + "\n".join(tail) + "\n")

    failed = None
    out0 = None
    pending = set(range(n))
    t_start = time.monotonic()
    while pending and failed is None:
        for r in sorted(pending):
            if r == 0 and out0 is None and procs[0].poll() is not None:
                out0 = procs[0].stdout.read()
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0:
                failed = (r, rc)
                break
        if pending and failed is None:
            if time.monotonic() - t_start > timeout_s:
                sys.stderr.write(f"[bench launcher] wall-clock limit of {timeout_s:.0f} s reached with ranks {sorted(pending)} "
                                 f"still running: terminating them\n")
                end_children(sorted(pending))
                for r in sorted(pending):
                    echo_tail(r, 10)
                for f in errs:
                    f.close()
                return 124
            time.sleep(0.2)
            if out0 is None and procs[0].poll() is not None:
                out0 = procs[0].stdout.read()
    if failed is not None:
        end_children(sorted(pending))         # the others would hang in a collective
        sys.stderr.write(f"[bench launcher] rank {failed[0]} exited with code {failed[1]}\n")
        echo_tail(failed[0])
        if out0 is None and procs[0].stdout is not None:
            try:
                out0 = procs[0].stdout.read()
            except Exception:
                out0 = ""
        sys.stdout.write(out0 or "")
        for f in errs:
            f.close()
        return failed[1] if failed[1] > 0 else 1
    if out0 is None:
        out0 = procs[0].stdout.read()
    for f in errs:
        f.close()
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    for ln in out0.splitlines():
        if not ln.startswith("{"):
            sys.stderr.write(ln + "\n")
    if not lines:
        sys.stderr.write("[bench launcher] rank 0 printed no JSON line\n")
        echo_tail(0)
        return 1
    print(lines[-1])
    return 0


# ------------------------------------------------------------------------------- worker ----
def cpu_baseline(cfg, sd, frames, seed, budget_s, config_key=None, aggregate=True, aggregate_frames=None):
    """Single-instance figure (best thread count of a probe) and, beside it, the aggregate of as many concurrent
    single-utterance instances as fill the host (the metric's batch is 8 utterances): `value` is the better of the two
    - the best this box's CPU does on the workload - and `cores` the threads that figure used."""
    from oracle import waveglow_torch_cpu as wt
    r = wt.timed_baseline(sd, cfg, frames, seed, budget_s=budget_s)
    out = {"value": r["value"], "unit": "samples/s", "cores": r["cores"], "kind": "port",
           "sample": f"oracle/waveglow_torch_cpu.py (torch CPU conv ops, reference-free restatement pinned to the "
                     f"reference goldens), same 12x512 model, 1 utterance x {frames} mel frames ({r['samples']} samples), "
                     f"torch.set_num_threads({r['threads_set']}) (fastest of a probe over {r['probe']} s per "
                     f"thread count on a short utterance; {r['physical']} physical cores), warm-up + best of {r['runs']} "
                     f"= {r['best_s']:.2f} s -> {r['value']:.0f} samples/s single instance",
           "single_instance": {"value": r["value"], "threads": r["cores"], "best_s": r["best_s"], "runs": r["runs"]}}
    if aggregate and config_key is not None:
        inst = max(1, min(16, r["physical"] // max(r["cores"], 1)))
        if inst > 1:
            try:
                af = min(frames, aggregate_frames or frames)
                g = wt.timed_aggregate(config_key, af, seed, r["cores"], inst)
                g["frames"] = af
                out["aggregate"] = g
                out["sample"] += (f"; aggregate: {inst} concurrent fresh processes x {r['cores']} threads, one {af}-frame "
                                  f"utterance each, started together: {g['samples']} samples in {g['span_s']:.2f} s = "
                                  f"{g['value']:.0f} samples/s ({inst * r['cores']} threads)")
                if g["value"] > out["value"]:
                    out["value"], out["cores"] = g["value"], inst * r["cores"]
            except Exception as e:       # the single-instance figure stands; say why the aggregate is missing
                out["sample"] += f"; aggregate run failed: {e!r}"[:300]
    return out


def wn_roofline(prof, cfg, dtype, gemm_mode, B, F, steps, elapsed, config_key="full"):
    """Roofline object of the WN kernels from the library's own HIP-event timing of the steps just run (``prof``: the
    caller-owned ``_lib.Profile`` that was bound while they ran; its slots are collected and cleared here): the in-layer conv-GEMM against the MFMA peak of the arithmetic it executes; in bf16 also
    the res and deferred-skip GEMMs and the whole step against HBM."""
    import ctypes
    from cookietts_amd import _lib
    wn = cfg["WN_config"]
    C, n_layers = wn["n_channels"], wn["n_layers"]
    L = F * cfg["hop_length"] // cfg["n_group"]
    traffic, traffic_src = load_traffic()
    n_in, ms_in = prof.collect(_lib.PROF_WN_IN)
    n, ms = types.SimpleNamespace(value=n_in), types.SimpleNamespace(value=ms_in)
    # algorithmic MACs per time step of ONE in-layer launch: dilated conv
    # C*2C*3 plus this layer's slice of the conditioning projection 256*2C
    mac = 3 * C * 2 * C + 256 * 2 * C
    # bf16x3 executes three bf16 products per algorithmic MAC; the roofline counts the EXECUTED bf16 flops
    products = 3 if dtype == "bf16x3" else {"f32": 1, "bf16x3": 3, "bf16x6": 6}[gemm_mode] if dtype == "f32" else 1
    flop_per_launch = 2.0 * mac * B * L * products
    mean_s = ms.value / max(n.value, 1) * 1e-3
    achieved = flop_per_launch / mean_s / 1e12
    peak = FP32_MFMA_PEAK_TFLOPS if dtype == "f32" else BF16_MFMA_PEAK_TFLOPS
    half = dtype in ("bf16", "f16")
    kname = "conv_gemm_f32_kernel<GATE>" if dtype == "f32" else ("conv_gemm_bf16_pp_kernel<GATE, F16>" if dtype == "f16" else "conv_gemm_bf16_pp_kernel<GATE>")
    if dtype == "f32" and gemm_mode != "f32":
        kname, peak = f"conv_gemm_f32_kernel<GATE, X{products}>", BF16_MFMA_PEAK_TFLOPS
    # (the IEEE-half kernels have their own PMC entries: their traffic is not the bf16 instantiation's)
    key = ("f32_" + gemm_mode) if (dtype == "f32" and gemm_mode != "f32") else dtype
    e = traffic.get(key)
    tbytes = e.get("hbm_bytes_per_launch") if (e and config_key == "full" and F == 900 and B == e.get("batch", 8)) else None
    roofline = {"kernel": kname + " (WN in-layer: dilated conv + cond + tanh*sigmoid)",
                "bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "traffic": tbytes,
                "traffic_source": (f"{traffic_src}: committed rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes of this "
                                   f"launch shape, not re-measured in this run") if traffic_src else None,
                "launches": int(n.value), "mean_launch_ms": round(mean_s * 1e3, 4),
                "flop_per_launch": flop_per_launch}
    if dtype == "f32" and not _lib.tuning_active("CTTS_F32_NO_ROUND_SPLIT"):
        # round-aligned launches (gemm_f32.hip launch_gemm_f32): the timed span of a launch covers BOTH kernels it may consist of
        roofline["launch_parts"] = ("a launch = conv_gemm_f32_kernel on the column tiles of its whole rounds of 2 x CUs workgroups "
                                    "+, first, conv_gemm_f32_small_kernel on the few tiles beyond them (config 2: 1792 + 8 of 1800 "
                                    "tiles); mean_launch_ms spans both, flop_per_launch is the whole launch's")
    if products > 1:
        roofline["note"] = (f"flop_per_launch counts the {products} executed bf16 products per algorithmic MAC; "
                            f"algorithmic flops are 1/{products} of it")
    # the other WN launches of the step: the res GEMM per layer (K = C) and the deferred skip GEMM per four layers (K = 4 C);
    # MFMA-bound in fp32, memory-bound in bf16.  (CTTS_F32_NO_DEFER_SKIP: one res/skip GEMM per layer, no skip slot.)
    slots = {}
    for which in (_lib.PROF_WN_RS, _lib.PROF_WN_SKIP):
        n2, ms2 = prof.collect(which)
        if n2 > 0:
            slots[which] = (n2, ms2 / n2 * 1e-3)
    if half:
        # In bf16 the in-layer GEMM stays MFMA-bound (1430 FLOP/B vs a ridge of ~312); the memory-bound WN kernels are the
        # res GEMM (per layer: act read + x read-modify-write = 3*C*2 B per time step) and the deferred skip GEMM (4 act
        # reads + the skip sum written, and re-read by the second launch = 5.5*C*2 B per time step on average).
        kt = "<SPLIT, F16>" if dtype == "f16" else "<SPLIT>"
        if _lib.PROF_WN_RS in slots:
            n2, mean2 = slots[_lib.PROF_WN_RS]
            bytes2 = float(3.0 * C * 2) * B * L
            e2 = traffic.get(dtype + "_res")
            t2 = e2.get("hbm_bytes_per_launch") if (e2 and config_key == "full" and F == 900 and B == e2.get("batch")) else None
            roofline["res_hbm"] = {"kernel": f"conv_gemm_bf16_ps_kernel{kt} (WN res 1x1: x += W_res act; persistent form)", "bound": "hbm",
                                   "achieved": round(bytes2 / mean2 / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                   "frac": round(bytes2 / mean2 / 8e12, 4), "launches": n2, "mean_launch_ms": round(mean2 * 1e3, 4),
                                   "bytes_per_launch": bytes2, "traffic": t2}
        if _lib.PROF_WN_SKIP in slots:
            # the deferred skip GEMM (K = 4 C over four layers' gated activations): 2 * C * 4C flop against 5.5 * C * 2 B per time
            # step = 372 FLOP/B at C = 512, above the ~312 ridge: an MFMA-bound launch (it was priced against HBM until round 5)
            n2, mean2 = slots[_lib.PROF_WN_SKIP]
            flop2 = 2.0 * C * 4 * C * B * L
            bytes2 = float(5.5 * C * 2) * B * L
            e2 = traffic.get(dtype + "_skip")
            t2 = e2.get("hbm_bytes_per_launch") if (e2 and config_key == "full" and F == 900 and B == e2.get("batch")) else None
            roofline["skip_mfma"] = {"kernel": f"conv_gemm_bf16_pp_kernel{kt} (WN skip sum over 4 layers' act, K = 4 C)", "bound": "mfma",
                                     "achieved": round(flop2 / mean2 / 1e12, 1), "peak": peak, "unit": "TFLOP/s",
                                     "frac": round(flop2 / mean2 / 1e12 / peak, 4), "launches": n2, "mean_launch_ms": round(mean2 * 1e3, 4),
                                     "flop_per_launch_mean": flop2, "bytes_per_launch": bytes2,
                                     "hbm_GBps_algorithmic": round(bytes2 / mean2 / 1e9, 1), "traffic": t2}
    else:
        deferred = _lib.PROF_WN_SKIP in slots
        if _lib.PROF_WN_RS in slots:
            n2, mean2 = slots[_lib.PROF_WN_RS]
            flop2 = (2.0 * C * C if deferred else 2.0 * (2 * C * C * (n_layers - 1) + C * C) / n_layers) * B * L * products
            roofline["res_skip_mfma"] = {
                "kernel": "conv_gemm_f32_kernel<SPLIT> (" + ("WN res 1x1: x += W_res act" if deferred else "WN res/skip 1x1") +
                          ", read-modify-write epilogue)", "bound": "mfma", "achieved": round(flop2 / mean2 / 1e12, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(flop2 / mean2 / 1e12 / peak, 4), "launches": n2,
                "mean_launch_ms": round(mean2 * 1e3, 4), "flop_per_launch_mean": flop2}
        if deferred:
            n2, mean2 = slots[_lib.PROF_WN_SKIP]
            flop2 = 2.0 * C * 4 * C * B * L * products
            roofline["skip_mfma"] = {"kernel": "conv_gemm_f32_kernel<SPLIT> (deferred skip GEMM over four layers' act, K = 4 C)",
                                     "bound": "mfma", "achieved": round(flop2 / mean2 / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
                                     "frac": round(flop2 / mean2 / 1e12 / peak, 4), "launches": n2,
                                     "mean_launch_ms": round(mean2 * 1e3, 4), "flop_per_launch": flop2}
    if half:
        # whole step against HBM with SURVEY 8d's per-layer-kernel byte count (7*C*2 B per step per layer; the deferred-skip form moves ~5.4*C*2)
        step_bytes = 7.0 * C * 2 * n_layers * cfg["n_flows"] * B * L
        roofline["step_hbm_algorithmic"] = {"bytes_per_step": step_bytes, "achieved": round(step_bytes * steps / elapsed / 1e9, 1),
                                            "peak": 8000.0, "unit": "GB/s", "frac": round(step_bytes * steps / elapsed / 8e12, 4)}
    flop_step = 2.0 * 18.85e6 * B * L * cfg["n_flows"] if (C == 512 and n_layers == 8) else None     # 18.85 M MAC / step / flow
    if flop_step and dtype == "f32" and gemm_mode == "f32":
        roofline["step_mfma_algorithmic"] = {"flop_per_step": flop_step, "achieved": round(flop_step * steps / elapsed / 1e12, 2),
                                             "peak": peak, "unit": "TFLOP/s", "frac": round(flop_step * steps / elapsed / 1e12 / peak, 4)}
    return roofline


def under_profiler():
    """rocprofv3's preloaded library initialises the GPU before Python starts: nothing may be forked from such a process."""
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "HSA_TOOLS_LIB")):
        return True
    return any(k.startswith(("ROCPROF", "ROCPROFILER", "ROCP_")) for k in os.environ)


def run_rows(which, model, cfg, lib, args, device):
    """Short secondary rows after the headline's timed region (N = 1 only).  Each row is guarded: a failure is recorded
    in the row and never takes the headline line down.  ``model`` is the headline's WaveGlow (config 2 weights)."""
    import time as _t
    import types
    import torch
    sys.path.insert(0, os.path.join(REPO, "scripts"))
    import bench_rows
    from cookietts_amd import _lib, synthetic
    rows = []
    t_all = _t.perf_counter()
    prof = _lib.Profile()                                  # the rows' own timing slots (caller-owned handle, ABI 6)

    def guard(name, fn):
        t0 = _t.perf_counter()
        try:
            out = fn()
            for r in (out if isinstance(out, list) else [out]):
                r["row_wall_s"] = round(_t.perf_counter() - t0, 1)
                rows.append(r)
        except Exception as e:                                    # noqa: BLE001 - recorded, never fatal
            rows.append({"row": name, "error": repr(e)[:400]})

    def timed_infer(mel, steps, warmup):
        for _ in range(warmup):
            model.infer(mel, sigma=0.6)
        torch.cuda.synchronize(device)
        with prof:
            t0 = _t.perf_counter()
            for _ in range(steps):
                out = model.infer(mel, sigma=0.6)
            torch.cuda.synchronize(device)
            dt = _t.perf_counter() - t0
        assert bool(torch.isfinite(out).all())
        return dt, out

    F = 900
    T = F * cfg["hop_length"]

    def config3(fmt="bf16"):
        B3 = 32
        mel = torch.from_numpy(synthetic.synthetic_mel(B3, F, seed=4321)).to(device)
        model.set_compute_dtype(torch.bfloat16 if fmt == "bf16" else torch.float16)
        try:
            dt, out = timed_infer(mel, 3, 1)
            roof = wn_roofline(prof, cfg, fmt, "f32", B3, F, 3, dt, args.config)
        finally:
            model.set_compute_dtype(torch.float32)
        return {"row": "A/config3 (1-GPU shard)" + ("" if fmt == "bf16" else ", IEEE-half storage"), "metric": METRIC,
                "value": B3 * T * 3 / dt, "unit": "samples/s",
                "rtf": B3 * T * 3 / dt / 22050.0, "ms_per_step": dt / 3 * 1e3, "steps": 3, "warmup": 1, "dtype": fmt,
                "batch": B3, "frames": F, "roofline": roof,
                "note": ("BASELINE config 3's per-GPU shard (256 utterances / 8 GPUs = 32): bf16 MFMA WN stacks, fp32 tails" if fmt == "bf16" else
                         "config 3's shard on the same kernels with IEEE-half storage and v_mfma_f32_32x32x16_f16 (the reference's own "
                         "reduced-precision mode, glow.py:343): 8x closer to the fp32 reference than bf16, inside the 1e-3 bound")}

    def f16_server():
        """The shapes the _5_infer slot sees (text2speech.py:658-665: <= 16 mels per vocoder call, .half()): 80 x 900 at 1 / 4 / 16."""
        out_rows = []
        model.set_compute_dtype(torch.float16)
        try:
            for Bs in (1, 4, 16):
                mel = torch.from_numpy(synthetic.synthetic_mel(Bs, F, seed=4400 + Bs)).to(device)
                dt, out = timed_infer(mel, 5, 2)
                roof = wn_roofline(prof, cfg, "f16", "f32", Bs, F, 5, dt, args.config)
                out_rows.append({"batch": Bs, "ms_per_step": dt / 5 * 1e3, "samples_per_s": Bs * T * 5 / dt, "rtf": Bs * T * 5 / dt / 22050.0,
                                 "roofline": roof})
        finally:
            model.set_compute_dtype(torch.float32)
        return {"row": "A/f16 at the server's batch sizes", "metric": METRIC, "value": out_rows[0]["samples_per_s"], "unit": "samples/s (batch 1)",
                "dtype": "f16", "frames": F, "steps": 5, "warmup": 2, "batches": out_rows,
                "note": "WaveGlowVocoder.half() mode (IEEE-half storage + fp16 MFMA from fp32 masters, inside the 1e-3 waveform bound); one 80 x 900 "
                        "utterance = 452 tiles of the 256 x 256 in-layer block on 256 CUs = 1.77 rounds: the launch runs two"}

    def bf16x6():
        mel = torch.from_numpy(synthetic.synthetic_mel(args.batch, F, seed=4322)).to(device)
        model.set_f32_gemm_mode("bf16x6")
        try:
            dt, out = timed_infer(mel, 2, 1)
            roof = wn_roofline(prof, cfg, "f32", "bf16x6", args.batch, F, 2, dt, args.config)
        finally:
            model.set_f32_gemm_mode(args.gemm_mode)
        return {"row": "A/config2 under --gemm-mode bf16x6", "metric": METRIC, "value": args.batch * T * 2 / dt, "unit": "samples/s",
                "ms_per_step": dt / 2 * 1e3, "steps": 2, "warmup": 1, "batch": args.batch, "frames": F,
                "dtype": "f32 tensors, split-bf16 GEMM products (bf16x6)", "roofline": roof}

    ns = types.SimpleNamespace(steps=3, warmup=1, batches="")
    for name in which:
        if name == "config3" and args.dtype == "f32":
            guard("A/config3", config3)
        elif name == "config3f16" and args.dtype == "f32":
            guard("A/config3f16", lambda: config3("f16"))
        elif name == "f16server" and args.dtype == "f32":
            guard("A/f16server", f16_server)
        elif name == "bf16x6" and args.dtype == "f32" and args.gemm_mode == "f32":
            guard("A/bf16x6", bf16x6)
        elif name == "config4":
            guard("B/config4", lambda: bench_rows.row_waveflow(ns))
        elif name == "config5":
            guard("C/config5", lambda: bench_rows.row_tacotron(types.SimpleNamespace(steps=2, warmup=1, batches=""), vocoder=model))
    return rows, round(_t.perf_counter() - t_all, 1)


def worker(args, pre=None):
    """``pre`` = {"sd": ..., "cpu": ...} computed by main() BEFORE this process touched the GPU (N = 1 only)."""
    import ctypes
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    selftest = args.selftest_launcher
    from cookietts_amd import sharding, synthetic
    if selftest:
        device = torch.device("cpu")
    else:
        device = torch.device("cuda", local_rank)
        torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if selftest:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    def sync():
        if not selftest:
            torch.cuda.synchronize(device)

    def fence():
        sync()
        if world > 1:
            dist.barrier()
            sync()

    seed = 1234
    cfg = synthetic.WAVEGLOW_CONFIGS["toy" if selftest else args.config]
    B, F = args.batch, args.frames
    T = F * cfg["hop_length"]
    exchange = None
    sd = None
    if selftest:
        lib = None
        model = torch.nn.Linear(8, 8)                      # something to broadcast
        if os.environ.get("CTTS_BENCH_SELFTEST_FAIL_RANK") == str(rank):
            sys.stderr.write("selftest: this rank fails on purpose\n")
            raise SystemExit(7)                            # lets the test see a rank failure propagate
        if os.environ.get("CTTS_BENCH_SELFTEST_HANG_RANK") == str(rank):
            time.sleep(3600)                               # lets the test see the wall-clock limit end the ranks

        def step_on(m):                                   # stand-in with the vocoder's shape contract
            return m.mean(dim=1, keepdim=True).repeat_interleave(cfg["hop_length"], dim=2).reshape(m.shape[0], -1)
    else:
        from cookietts_amd import WaveGlow, _lib
        lib = _lib.lib()
        model = WaveGlow(**cfg)
        if rank == 0:                                      # rank 0 owns the checkpoint ...
            sd = pre["sd"] if pre else synthetic.waveglow_state_dict(cfg, seed=seed)
            model.load_state_dict(synthetic.to_torch(sd))
        model = model.to(device).eval()
        model.set_f32_gemm_mode(args.gemm_mode)                  # in the model's config struct (ABI 4): explicit, never
                                                                 # inherited from the environment or a process default
        if args.dtype == "bf16":
            model.set_compute_dtype(torch.bfloat16)
        elif args.dtype == "f16":
            model.set_compute_dtype(torch.float16)
        elif args.dtype == "bf16x3":
            model.set_compute_dtype("bf16x3")

        def step_on(m):
            return model.infer(m, sigma=0.6)

    if world > 1:                                          # ... every other rank gets it over xGMI
        fence()
        t0 = time.perf_counter()
        nbytes = sharding.broadcast_state_dict(model, src=0)
        fence()
        exchange = {"broadcast_ms": (time.perf_counter() - t0) * 1e3, "broadcast_bytes": int(nbytes)}

    mel = torch.from_numpy(synthetic.synthetic_mel(B, F, seed=seed + rank)).to(device)   # resident in HBM

    for _ in range(args.warmup):
        out = step_on(mel)
    fence()
    timing = not args.no_kernel_timing and lib is not None
    prof = _lib.Profile() if timing else None
    if timing:
        prof.bind()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step_on(mel)
    sync()
    elapsed_own = time.perf_counter() - t0             # this rank's K steps, before waiting for the others
    fence()
    elapsed = time.perf_counter() - t0
    if timing:
        prof.unbind()
    assert out.shape == (B, T) and bool(torch.isfinite(out).all())

    rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        # every rank's own time for the same K steps (the barrier-to-barrier span is the max): stragglers show up here
        mine = torch.tensor([elapsed_own], dtype=torch.float64, device=device)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(t.item()) / args.steps * 1e3 for t in every]
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # the per-request exchange of SURVEY 8e, outside the timed steps: rank 0 holds world*B mels, scatters the
    # slices, every rank vocodes its slice, waves are gathered point-to-point into rank 0
    if world > 1 and not args.no_exchange:
        n_mel = cfg["n_mel_channels"]
        # config 3 ships bf16 mels (32 x 80 x 900 bf16 = 4.6 MB per rank); the fp32 path ships fp32
        xdtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
        all_mels = None
        if rank == 0:
            all_mels = torch.cat([torch.from_numpy(synthetic.synthetic_mel(B, F, seed=seed + r)) for r in range(world)]).to(device)
        for it in range(2):                                # pass 0 opens the point-to-point connections
            fence()
            t0 = time.perf_counter()
            local, counts = sharding.scatter_mels(all_mels, n_mel, device, src=0, dtype=xdtype)
            fence()
            t1 = time.perf_counter()
            wave = step_on(local.float())
            fence()
            t2 = time.perf_counter()
            waves = sharding.gather_waves(wave, counts, dst=0)
            fence()
            t3 = time.perf_counter()
        if rank == 0:
            assert waves.shape == (world * B, T) and bool(torch.isfinite(waves).all())
            if selftest:
                assert torch.equal(waves, step_on(all_mels))
        exchange.update(scatter_ms=(t1 - t0) * 1e3, infer_ms=(t2 - t1) * 1e3, gather_ms=(t3 - t2) * 1e3,
                        scatter_bytes=int(world * B * n_mel * F * (2 if xdtype == torch.bfloat16 else 4)),
                        scatter_dtype=str(xdtype).replace("torch.", ""), gather_bytes=int(world * B * T * 4),
                        note="steady state (second pass; the first opens the RCCL point-to-point channels); "
                             "barrier + synchronize on both sides of each leg, so each figure includes one barrier")

    # N = 1: the same exchange code on a world-size-1 RCCL group, outside the timed steps, so that broadcast / scatter /
    # gather have run on RCCL on this box before the first multi-GPU lease.  A failure here is
    # recorded, it never takes the headline line down.
    if world == 1 and not selftest and not args.no_exchange and not args.no_exchange_dry_run:
        try:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
            sync()
            t0 = time.perf_counter()
            nbytes = sharding.broadcast_state_dict(model, src=0)
            sync()
            t1 = time.perf_counter()
            n_mel = cfg["n_mel_channels"]
            local, counts = sharding.scatter_mels(mel, n_mel, device, src=0)
            sync()
            t2 = time.perf_counter()
            wave = step_on(local)
            sync()
            t3 = time.perf_counter()
            waves = sharding.gather_waves(wave, counts, dst=0)
            sync()
            t4 = time.perf_counter()
            assert waves.shape == (B, T) and bool(torch.isfinite(waves).all())
            exchange = {"dry_run_world_1": True, "backend": "nccl (RCCL)", "broadcast_ms": (t1 - t0) * 1e3,
                        "broadcast_bytes": int(nbytes), "scatter_ms": (t2 - t1) * 1e3, "infer_ms": (t3 - t2) * 1e3,
                        "gather_ms": (t4 - t3) * 1e3, "scatter_bytes": int(B * n_mel * F * 4), "gather_bytes": int(B * T * 4),
                        "note": "world size 1: the code path of the multi-GPU exchange on RCCL, not an xGMI measurement"}
            dist.destroy_process_group()
        except Exception as e:
            exchange = {"dry_run_world_1": True, "error": repr(e)[:300]}

    if rank == 0:
        samples = world * B * T * args.steps
        value = samples / elapsed
        wn = cfg["WN_config"]
        C, n_layers = wn["n_channels"], wn["n_layers"]
        roofline = wn_roofline(prof, cfg, args.dtype, args.gemm_mode, B, F, args.steps, elapsed, args.config) if timing else None
        cpu = pre["cpu"] if pre else None      # timed by main() before the GPU was touched (its aggregate leg starts children)
        rows, rows_s = None, None
        which_rows = [] if (args.no_rows or world > 1 or selftest) else [r for r in args.rows.split(",") if r]
        if which_rows:
            rows, rows_s = run_rows(which_rows, model, cfg, lib, args, device)
        line = {
            "metric": METRIC, "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_ranks": {"min": min(rank_ms), "max": max(rank_ms), "all": [round(x, 3) for x in rank_ms]},
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype if (args.gemm_mode == "f32" or args.dtype != "f32") else f"f32 tensors, split-bf16 GEMM products ({args.gemm_mode})",
            "data": "synthetic" if not selftest else "LAUNCHER SELF-TEST (gloo/CPU stand-in step; not a measurement)",
            "rtf": value / 22050.0,
            "config": {"workload": f"WaveGlow {args.config} ({cfg['n_flows']} flows, {C} WN ch, "
                                   f"{cfg['n_group']} groups, {n_layers} layers) { {'f32': 'fp32', 'bf16': 'bf16-MFMA', 'f16': 'fp16-MFMA (IEEE-half storage)', 'bf16x3': 'split-bf16 (3 bf16 MFMA products, fp32 accumulate)'}[args.dtype] } infer, batch {B} x (80x{F}) "
                                   f"mel per GPU, sigma 0.6, random-init weights",
                       "batch_per_gpu": B, "frames": F, "samples_per_step": world * B * T,
                       "parallelism": f"utterance-batch shard x{world}"},
            "roofline": roofline, "cpu_baseline": cpu, "exchange": exchange,
        }
        if rows is not None:
            line["rows"] = rows
            line["rows_wall_s"] = rows_s
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launcher role: nothing below this line in THIS process imports torch or touches HIP
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], timeout_s=args.launch_timeout))
    pre = None
    if args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.selftest_launcher and args.dtype != "cpu":
        # The CPU baseline runs FIRST, while this process has made no HIP call: its aggregate leg starts child
        # processes, and nothing GPU-initialised may ever be forked / exec'd on this pool.  (Profiler runs pass
        # --cpu-frames 0: rocprofv3's preloaded library initialises the GPU before Python starts.)
        from cookietts_amd import synthetic
        cfg = synthetic.WAVEGLOW_CONFIGS[args.config]
        pre = {"sd": synthetic.waveglow_state_dict(cfg, seed=1234), "cpu": None}
        if args.cpu_frames > 0 and under_profiler():
            # rocprofv3's preloaded library has initialised the GPU already: no child processes, and the CPU legs would
            # only be profiled idle time
            pre["cpu"] = {"value": None, "unit": "samples/s", "cores": 0, "kind": "port",
                          "sample": "skipped: running under rocprofv3 (its library initialises the GPU before Python starts, "
                                    "so the aggregate leg's child processes may not be started)"}
        elif args.cpu_frames > 0:
            pre["cpu"] = cpu_baseline(cfg, pre["sd"], args.cpu_frames, 1234, args.cpu_budget, config_key=args.config,
                                      aggregate=not args.no_cpu_aggregate, aggregate_frames=args.cpu_aggregate_frames)
    worker(args, pre)


if __name__ == "__main__":
    main()
