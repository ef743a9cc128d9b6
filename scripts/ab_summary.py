"""One summary line of a bench.py JSON line (scripts/gpu.sh ab): step time and the WN launches' means, any dtype."""
import json
import sys


def main(path, label):
    for line in open(path):
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        r = d.get("roofline") or {}
        parts = [label, f"{d['ms_per_step']:.2f} ms/step", f"in {r.get('mean_launch_ms')} ms frac {r.get('frac')}"]
        for key in ("res_skip_mfma", "skip_mfma", "res_hbm", "skip_hbm", "step_mfma_algorithmic", "step_hbm_algorithmic"):
            e = r.get(key)
            if e:
                parts.append(f"{key} {e.get('mean_launch_ms', '')} {e.get('frac')}")
        print("  ".join(str(p) for p in parts))
        return
    print(label, "no JSON line in", path)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
