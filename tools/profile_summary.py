#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of `tools/gpu_check.sh prof pmc` (under gpurun_out/) into the committed summaries under profiles/:

    python tools/profile_summary.py --tag r1_e

  * profiles/<tag>_kernel_stats.csv          copy of the --kernel-trace --stats summary
  * profiles/<tag>_hbm_traffic.json          per-family HBM bytes of the B=32 forward from the FETCH_SIZE / WRITE_SIZE passes
                                             (FETCH_SIZE doubled, MI355X_MICROARCH.md section HBM); also written to
                                             profiles/latest_hbm_traffic.json, which bench.py reads for roofline.traffic
  * prints the average duration of the HiFi-GAN conv_gemm launches per B=32 forward (must agree with bench.py's avg_launch_ms)
"""
import argparse
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
N_DEC = 17          # decoder launches of the same template in front of the vocoder's (4 layers x 4 GEMMs + to_mel)


def forwards(rows):
    """Split a kernel trace into forwards (ending at conv_post); return the launches of each B=32 forward (the long ones)."""
    rows = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
    out, cur = [], []
    for r in rows:
        cur.append(r)
        if "conv_post" in r["Kernel_Name"]:
            if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 100000:
                out.append(cur)
            cur = []
    return out


def voc_gemms(fwd):
    """HiFi-GAN launches of conv_gemm_kernel<f16>: conv_pre, 4 ups and the ResBlock convs that are not in a fused pair kernel."""
    g = [r for r in fwd if "conv_gemm_kernelIDF16" in r["Kernel_Name"]]
    return g[N_DEC:]


def pmc_family(path, counter):
    rows = list(csv.DictReader(open(path)))
    by_disp = {}
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        d = by_disp.setdefault(r["Dispatch_Id"], dict(r, value=0.0))
        d["value"] += float(r["Counter_Value"])
    fw = forwards(list(by_disp.values()))
    f = fw[-1]
    gem = voc_gemms(f)
    pair = [r for r in f if "resblock_pair_c" in r["Kernel_Name"]]
    return sum(r["value"] for r in gem), len(gem), sum(r["value"] for r in pair), len(pair)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    args = ap.parse_args()
    pdir = os.path.join(ROOT, "profiles")
    shutil.copy(os.path.join(OUT, "prof", "r1_kernel_stats.csv"), os.path.join(pdir, args.tag + "_kernel_stats.csv"))
    trace = list(csv.DictReader(open(os.path.join(OUT, "prof", "r1_kernel_trace.csv"))))
    avgs = []
    for f in forwards(trace):
        g = voc_gemms(f)
        avgs.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in g) / len(g) / 1e6)
    print("rocprof kernel trace: avg duration of the %d vocoder conv_gemm launches per B=32 forward (ms):" % len(g),
          ", ".join("%.4f" % a for a in avgs))
    fg, ng, fp, npair = pmc_family(os.path.join(OUT, "pmc_fetch", "f_counter_collection.csv"), "FETCH_SIZE")
    wg, _, wp, _ = pmc_family(os.path.join(OUT, "pmc_write", "w_counter_collection.csv"), "WRITE_SIZE")
    fam = {
        "conv_gemm_f16_vocoder": dict(launches=ng, fetch_size_kb_raw=fg, write_size_kb_raw=wg,
                                      hbm_bytes_per_forward=(2 * fg + wg) * 1024, hbm_bytes_per_launch=(2 * fg + wg) * 1024 / ng),
        "resblock_pair_c32_c64": dict(launches=npair, fetch_size_kb_raw=fp, write_size_kb_raw=wp,
                                  hbm_bytes_per_forward=(2 * fp + wp) * 1024, hbm_bytes_per_launch=(2 * fp + wp) * 1024 / max(npair, 1)),
    }
    frames = 32768
    doc = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --steps 1 --warmup 1 --cpu-utts 0`, the B=32 x 1024-frame forward",
        "correction": "FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md section HBM); WRITE_SIZE as reported",
        "families": fam,
        "frames": frames,
        "vocoder_hbm_bytes_per_frame": sum(v["hbm_bytes_per_forward"] for v in fam.values()) / frames,
        "algorithmic_contract_bytes_per_frame": 2026000.0,
        "hbm_bytes_per_launch": fam["conv_gemm_f16_vocoder"]["hbm_bytes_per_launch"],
        "rocprof_avg_launch_ms_per_forward": avgs,
    }
    for name in (args.tag + "_hbm_traffic.json", "latest_hbm_traffic.json"):
        json.dump(doc, open(os.path.join(pdir, name), "w"), indent=1)
    print(json.dumps({k: v for k, v in doc.items() if k != "families"}, indent=1))
    for k, v in fam.items():
        print(k, {a: (round(b / 1e9, 3) if a.startswith("hbm") else b) for a, b in v.items()})


if __name__ == "__main__":
    main()
