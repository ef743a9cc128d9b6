"""One architecture of the published WaveFlow table at batch 1 (scripts/bench_rows.py row_waveflow_table), for rocprofv3:
python scripts/debug/wf_table_one.py <n_group> <n_flows> <n_channels> <separable 0|1> [calls] [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cookietts_amd import synthetic  # noqa: E402
from cookietts_amd.waveglow_ax import WaveGlow  # noqa: E402


def main():
    G, n_flows, C, sep = (int(x) for x in sys.argv[1:5])
    calls = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    batch = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    hop = 256 if G == 8 else 300
    cfg = synthetic.waveflow_config(n_flows=n_flows, n_group=G, n_channels=C, hop_length=hop, win_length=4 * hop,
                                    WN=dict(seperable_conv=bool(sep)))
    m = WaveGlow(**cfg)
    m.load_state_dict(synthetic.to_torch(synthetic.waveflow_state_dict(cfg, seed=1234)))
    m = m.cuda().eval()
    F = 220500 // hop
    mel = torch.from_numpy(synthetic.synthetic_mel(batch, F)).cuda()
    m.infer(mel, sigma=0.6, return_CPU=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        m.infer(mel, sigma=0.6, return_CPU=False)
    torch.cuda.synchronize()
    print(f"G={G} flows={n_flows} C={C} sep={sep} B={batch}: {(time.perf_counter() - t0) / calls * 1e3:.1f} ms per call")


if __name__ == "__main__":
    main()
