"""The WaveFlow row queue (gemm_f32_small.hip ``wf_row_persistent_kernel``): the rows of a flow - each row its fused layers and a
tail stage (end conv, affine update, next row's start conv) - as ONE launch whose workgroups take (row, stage, tile) items in
order and wait for the neighbouring tiles of the previous stage.  It runs the same tile bodies and the same per-column tail
arithmetic as the per-layer launches, so every form must agree BIT FOR BIT; its bounded wait must fail loudly."""
import numpy as np
import pytest
import torch

from cookietts_amd import synthetic

pytestmark = pytest.mark.gpu


def _model():
    from cookietts_amd import WaveFlow
    cfg = synthetic.WAVEFLOW_CONFIGS["full"]
    m = WaveFlow(**cfg)
    m.load_state_dict(synthetic.to_torch(synthetic.waveflow_state_dict(cfg, seed=78)))
    return m.cuda().eval()


def _inputs(B, F, seed):
    mel = torch.from_numpy(synthetic.synthetic_mel(B, F + 1, seed=seed)).cuda()
    g = torch.Generator().manual_seed(seed)
    z = (torch.randn(B, F * 256, generator=g) * 0.6).cuda()
    return z, mel


@pytest.mark.parametrize("B,F,splitk", [(1, 120, False), (3, 333, False), (8, 240, False), (1, 120, True), (2, 333, True)])
def test_row_queue_equals_per_layer_launches(hip_lib_path, tuning, B, F, splitk):
    """Queue forced on at every size (ragged last tile at F = 333) vs one launch per layer: items of 128 x 128 tiles against the
    128 x 128 shape, items of the split-K body (what the queue takes below 600 items per layer) against the split-K shape."""
    from cookietts_amd import _lib
    m = _model()
    z, mel = _inputs(B, F, seed=11 + B)
    if not splitk:
        tuning.set("CTTS_F32_NO_SPLITK")                   # the split-K shape sums K in another order
    tuning.set("CTTS_WF_NO_ROW_QUEUE")
    tuning.set("CTTS_WF_NO_REGION_SPLIT")
    ref, _ = m.inverse(z, mel, return_CPU=False)
    code = _lib.lib().ctts_last_gemm_loop()
    assert not code & 64 and bool(code & 32) == splitk
    tuning.clear("CTTS_WF_NO_ROW_QUEUE")
    tuning.set("CTTS_WF_ROW_QUEUE_MIN", "1")
    for _ in range(3):                                     # epochs, counters and flags are re-armed per call
        got, _ = m.inverse(z, mel, return_CPU=False)
        code = _lib.lib().ctts_last_gemm_loop()
        assert (code & 192) == 192 and bool(code & 32) == splitk, "the whole-flow queue did not run (or not with the expected body)"
        assert torch.isfinite(got).all() and torch.equal(got, ref)
    tuning.set("CTTS_WF_QUEUE_DEBUG", "128")               # one launch per ROW (start / tail kernels between) instead of per flow
    got, _ = m.inverse(z, mel, return_CPU=False)
    assert (_lib.lib().ctts_last_gemm_loop() & 192) == 64 and torch.equal(got, ref)


def test_row_queue_full_size_default_and_forms(hip_lib_path, tuning):
    """Config 4 at B = 8 x 900 frames takes the queue by default; region split and single launches give the same bits."""
    from cookietts_amd import _lib
    m = _model()
    z, mel = _inputs(8, 900, seed=5)
    got, _ = m.inverse(z, mel, return_CPU=False)
    assert _lib.lib().ctts_last_gemm_loop() & 64
    tuning.set("CTTS_WF_NO_ROW_QUEUE")
    split, _ = m.inverse(z, mel, return_CPU=False)
    assert not _lib.lib().ctts_last_gemm_loop() & 64
    assert torch.equal(got, split)
    tuning.set("CTTS_WF_NO_REGION_SPLIT")
    single, _ = m.inverse(z, mel, return_CPU=False)
    assert torch.equal(got, single)


def test_row_queue_abort_is_loud(hip_lib_path, tuning):
    """With the abort word set (what an expired wait does) every workgroup leaves; the audio is NaN, not noise - and the
    abort is a STATUS (ABI 6): a call that returns to the host raises at once, a call that stays on the
    device is reported by the next call on that workspace (CTTS_E_ABORT, once), after which the workspace is clean again."""
    from cookietts_amd import _lib
    m = _model()
    z, mel = _inputs(2, 100, seed=3)
    tuning.set("CTTS_WF_ROW_QUEUE_MIN", "1")
    good, _ = m.inverse(z, mel, return_CPU=False)
    tuning.set("CTTS_WF_INJECT_ABORT")
    with pytest.raises(_lib.HipLibraryError, match="row queue .* aborted"):       # synchronising call: raises itself
        m.inverse(z, mel, return_CPU=True)
    bad, _ = m.inverse(z, mel, return_CPU=False)                                  # device-side call: NaN now ...
    assert torch.isnan(bad).all()
    tuning.clear("CTTS_WF_INJECT_ABORT")
    with pytest.raises(_lib.HipLibraryError, match="rc=-4"):                      # ... CTTS_E_ABORT from the next call
        m.inverse(z, mel, return_CPU=False)
    again, _ = m.inverse(z, mel, return_CPU=False)                                # reported once; clean afterwards
    assert torch.equal(again, good)
    cpu, _ = m.inverse(z, mel, return_CPU=True)
    assert torch.equal(cpu, good.cpu())


def test_two_threads_time_two_models_into_disjoint_profiles(hip_lib_path):
    """Kernel-timing slots are caller-owned handles bound per thread (ABI 6): two threads running two WaveGlow models
    concurrently each collect exactly their own launches."""
    import threading
    from cookietts_amd import WaveGlow, _lib
    cfg = synthetic.WAVEGLOW_CONFIGS["toy_early"]
    n_layers, n_flows = cfg["WN_config"]["n_layers"], cfg["n_flows"]
    results, errors = {}, []

    def run(tag, calls):
        try:
            torch.cuda.set_device(0)
            m = WaveGlow(**cfg)
            m.load_state_dict(synthetic.to_torch(synthetic.waveglow_state_dict(cfg, seed=7 + calls)))
            m = m.cuda().eval()
            mel = torch.from_numpy(synthetic.synthetic_mel(1, 12, seed=calls)).cuda()
            zz = torch.from_numpy(synthetic.synthetic_noise(1, cfg["n_group"], 12 * 32, seed=calls)).cuda()
            stream = torch.cuda.Stream()
            prof = _lib.Profile()
            with torch.cuda.stream(stream):
                m.infer_from_noise(mel, zz)                      # not bound yet: must not be recorded anywhere
                with prof:
                    for _ in range(calls):
                        m.infer_from_noise(mel, zz)
                m.infer_from_noise(mel, zz)                      # unbound again
            stream.synchronize()
            results[tag] = prof.collect(_lib.PROF_WN_IN)
            prof.close()
        except Exception as e:                                   # noqa: BLE001 - reported by the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=run, args=("a", 2)), threading.Thread(target=run, args=("b", 5))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert results["a"][0] == 2 * n_flows * n_layers and results["b"][0] == 5 * n_flows * n_layers
    assert results["a"][1] > 0.0 and results["b"][1] > 0.0


@pytest.mark.parametrize("key", ["toy", "toy_dilations", "toy_conv_early", "kh1", "merge"])
@pytest.mark.parametrize("precond", [False, True])
def test_row_queue_on_the_goldens_configs(hip_lib_path, tuning, key, precond):
    """Queue forced on small golden configurations, with the conditioning folded into the in-layer GEMM (a K segment) and handed
    over precomputed (shift_spect / scale_spect turn the fold off: the per-element addend, whose row stride and pad are defaults
    that launch_gemm_f32 fills in - the queue launches the tile body without it and once forgot them)."""
    from cookietts_amd import _lib
    from cookietts_amd.waveglow_ax import WaveGlow
    # "kh1": a 1 x 3 kernel = at most four K segments per layer: the kernels' four-segment instantiation
    # "merge": merge_res_skip with the tanh-sigmoid unit (every layer reads the start outputs, all rows are skip rows)
    if key == "kh1":
        cfg = synthetic.waveflow_config(n_flows=2, n_group=8, n_layers=3, kernel_size_h=1)
    elif key == "merge":
        cfg = synthetic.waveflow_config(n_flows=2, n_group=8, n_layers=3, WN=dict(merge_res_skip=True))
    else:
        cfg = dict(synthetic.WAVEFLOW_CONFIGS[key])
    if precond:
        cfg.update(shift_spect=2.0, scale_spect=0.5)
    m = WaveGlow(**cfg)
    m.load_state_dict(synthetic.to_torch(synthetic.waveflow_state_dict(cfg, seed=31)))
    m = m.cuda().eval()
    assert m._folded != precond
    F = 9
    mel = torch.from_numpy(np.pad(synthetic.synthetic_mel(2, F, seed=31)[:, :cfg["n_mel_channels"]], ((0, 0), (0, 0), (0, 1)))).cuda()
    z = (torch.randn(2, F * cfg["hop_length"], generator=torch.Generator().manual_seed(31)) * 0.7).cuda()
    tuning.set("CTTS_WF_NO_ROW_QUEUE")
    ref, _ = m.inverse(z, mel, return_CPU=False)
    tuning.clear("CTTS_WF_NO_ROW_QUEUE")
    tuning.set("CTTS_WF_ROW_QUEUE_MIN", "1")
    got, _ = m.inverse(z, mel, return_CPU=False)
    assert _lib.lib().ctts_last_gemm_loop() & 64, "the row queue did not run"
    assert torch.equal(got, ref)


def test_row_queue_two_streams_share_the_chip(hip_lib_path, tuning):
    """Two models' calls on two streams, queued back to back from one host thread, so that row launches of both are resident at
    the same time and neither gets the chip to itself: the queue needs no co-residency (items are claimed in order), each call
    has its own counters / flags in its own workspace, and the results equal the calls run alone."""
    ma, mb = _model(), _model()
    za, mela = _inputs(3, 150, seed=41)
    zb, melb = _inputs(2, 210, seed=42)
    tuning.set("CTTS_WF_ROW_QUEUE_MIN", "1")
    ref_a, _ = ma.inverse(za, mela, return_CPU=False)
    ref_b, _ = mb.inverse(zb, melb, return_CPU=False)
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(3):
        with torch.cuda.stream(sa):
            got_a, _ = ma.inverse(za, mela, return_CPU=False)
        with torch.cuda.stream(sb):
            got_b, _ = mb.inverse(zb, melb, return_CPU=False)
        outs.append((got_a, got_b))
    torch.cuda.synchronize()
    for got_a, got_b in outs:
        assert torch.equal(got_a, ref_a) and torch.equal(got_b, ref_b)


def test_split_bf16_modes_keep_the_per_layer_launches(hip_lib_path, tuning):
    """The queue's tile bodies are fp32-MFMA only: a model that asks for the split-bf16 loops at a queue-eligible size runs
    its layers one launch at a time (and still agrees with the fp32 result to the mode's accuracy) instead of failing."""
    from conftest import rms_rel_err
    from cookietts_amd import _lib
    m = _model()
    z, mel = _inputs(5, 600, seed=9)                       # 5 x 600 frames: 375 tiles of 128 columns per layer (> 1 per CU)
    ref, _ = m.inverse(z, mel, return_CPU=False)
    assert _lib.lib().ctts_last_gemm_loop() & 64
    for mode, tol in (("bf16x3", 1e-4), ("bf16x6", 5e-6)):
        m.set_f32_gemm_mode(mode)
        got, _ = m.inverse(z, mel, return_CPU=False)
        code = _lib.lib().ctts_last_gemm_loop()
        assert not code & 64 and (code & 15) == (3 if mode == "bf16x3" else 6)
        d = rms_rel_err(got.cpu().numpy(), ref.cpu().numpy())
        print(f"config 4, 5 x 600 frames under {mode}: rms rel diff vs fp32 MFMA {d:.2e}")
        assert d < tol
    m.set_f32_gemm_mode("f32")
