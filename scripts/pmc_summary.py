#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes: per kernel (template arguments kept, parameter list dropped) the number of dispatches,
the mean duration and the mean value per dispatch of every counter found in the given output directories.

    python scripts/pmc_summary.py --label "bench.py f32" --min-us 50 gpurun_out/r3_pmc_f32_* > profiles/r3_pmc_f32.json

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; they are converted to bytes here.  No correction factor is
applied: MI355X_MICROARCH.md's gfx950 note (FETCH_SIZE counts half of 16 B/lane streaming reads) is applied by the reader
(bench.py: traffic = 2 * fetch + write), after calibration on a streaming kernel of known bytes in the same pass."""
import argparse
import collections
import csv
import glob
import json
import os
import re
import sys

KIB = {"FETCH_SIZE", "WRITE_SIZE"}


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in name:                       # cut the parameter list: first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).replace("ctts::", "")


def derive(e):
    """GRBM_GUI_ACTIVE is summed over the 8 XCDs (18.9e9 / s on the headline kernel = 8 x 2.36 GHz); SQ_VALU_MFMA_BUSY_CYCLES
    is summed over the 1024 SIMDs and equals 64 x the number of v_mfma_f32_32x32x2_f32 issued (checked on the headline:
    845.57 GFLOP / 4096 flop x 64 = 13 212 057 600 exactly).  Busy fraction of a SIMD's matrix pipe =
    busy / (1024 x GUI_ACTIVE / 8); shader clock under the kernel = GUI_ACTIVE / 8 / duration."""
    if e.get("GRBM_GUI_ACTIVE"):
        e["shader_clock_ghz_under_pmc"] = e["GRBM_GUI_ACTIVE"] / 8.0 / (e["mean_us_under_pmc"] * 1e3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
            e["mfma_busy_frac_per_simd"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] * 128.0)
    if e.get("SQ_WAVE_CYCLES"):
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if n in e:
                e[n + "_frac_of_wave_cycles"] = e[n] / e["SQ_WAVE_CYCLES"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--label", default="")
    ap.add_argument("--min-us", type=float, default=20.0, help="drop kernels whose mean duration is below this")
    ap.add_argument("--match", default="", help="regex a kernel name must match")
    ap.add_argument("--rederive", action="store_true", help="the arguments are summary .json files: recompute the derived fields in place")
    a = ap.parse_args()
    if a.rederive:
        for path in a.dirs:
            d = json.load(open(path))
            for e in d["kernels"].values():
                derive(e)
            json.dump(d, open(path, "w"), indent=1)
        return
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    dur = collections.defaultdict(lambda: [0.0, 0])
    meta = {}
    for d in a.dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            with open(path) as f:
                for r in csv.DictReader(f):
                    k = short(r["Kernel_Name"])
                    v = float(r["Counter_Value"]) * (1024.0 if r["Counter_Name"] in KIB else 1.0)
                    c = acc[k][r["Counter_Name"]]
                    c[0] += v
                    c[1] += 1
                    key = (path, r["Dispatch_Id"])
                    if key not in seen:
                        seen.add(key)
                        dur[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                        dur[k][1] += 1
                    meta[k] = {"grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"]), "lds": int(r["LDS_Block_Size"]),
                               "vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                               "scratch": int(r["Scratch_Size"])}
    out = {}
    for k, counters in acc.items():
        mean_us = dur[k][0] / max(dur[k][1], 1)
        if mean_us < a.min_us or (a.match and not re.search(a.match, k)):
            continue
        e = {"dispatches_per_pass": max(c[1] for c in counters.values()), "mean_us_under_pmc": round(mean_us, 2), **meta[k]}
        for n, c in sorted(counters.items()):
            e[n] = c[0] / c[1]
        derive(e)
        out[k] = e
    json.dump({"label": a.label, "source_dirs": a.dirs, "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["mean_us_under_pmc"] * kv[1]["dispatches_per_pass"]))},
              sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
