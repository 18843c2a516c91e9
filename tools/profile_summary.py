#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of `tools/gpu_check.sh prof pmc pmcsq pmccal bench` (under gpurun_out/) into the committed summaries
under profiles/:

    python tools/profile_summary.py --tag r2_d

  * profiles/<tag>_kernel_stats.csv     copy of the --kernel-trace --stats summary (prof stage)
  * profiles/<tag>_hbm_traffic.json     per-family HBM bytes of the B=32 forward from the FETCH_SIZE / WRITE_SIZE passes, corrected
                                        with the factors measured by the calibration stage (tools/pmc_calibrate.py: dispatches of
                                        exactly 1 GiB read + 1 GiB written); falls back to "FETCH_SIZE x 2" (MI355X_MICROARCH.md
                                        section HBM) when no calibration run is present.  Also written to
                                        profiles/latest_hbm_traffic.json, which bench.py reads for roofline.traffic
  * profiles/<tag>_sq_counters.json     per-family SQ counters of the same forward: MFMA-busy, wait and LDS bank-conflict ratios
  * profiles/<tag>_launches.json        per-launch table of bench.py's profiled step (kernel, shape, ms, TF/s, GB/s)
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
N_DEC_MX = 8        # "mx" precision: the decoder's conv-FFN launches (4 layers x 2) of conv_gemm_mx_kernel in front of the generator's
GIB = float(1 << 30)


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


def _is_f16_gemm(name):
    return "conv_gemm_kernelIDF16" in name or "conv_gemm_phased_kernel" in name


def _is_mx_gemm(name):
    return "conv_gemm_mx_kernel" in name or "conv_gemm_mx_group3_kernel" in name or "conv_gemm_mx_up_kernel" in name


FAMILIES = {
    # (the fp16 family = the 4-wave kernel + the phased 8-wave kernel that takes its MFMA-bound shapes)
    "conv_gemm_f16_vocoder": lambda f: [r for r in f if _is_f16_gemm(r["Kernel_Name"])][N_DEC:],
    "conv_gemm_f16_decoder": lambda f: [r for r in f if _is_f16_gemm(r["Kernel_Name"])][:N_DEC],
    "resblock_pair_c32_c64": lambda f: [r for r in f if "resblock_pair_c32_kernel" in r["Kernel_Name"] or "resblock_pair_c64_kernel" in r["Kernel_Name"]],
    "attention_mfma_f16": lambda f: [r for r in f if "attention_mfma_kernel" in r["Kernel_Name"]],
    "conv_gemm_split_token_rate": lambda f: [r for r in f if "conv_gemm_split_kernel" in r["Kernel_Name"] or "conv_gemm_x3_kernel" in r["Kernel_Name"]],
    "layernorm": lambda f: [r for r in f if "layernorm_kernel" in r["Kernel_Name"]],
    # round 3, "mx" precision (bench.py's default): the generator's three MX kernels, the decoder's conv-FFN on the same conv-GEMM kernel
    # (round 6: the same-level convs of a stage's three ResBlocks are ONE grouped launch of conv_gemm_mx_group3_kernel -- the family has 19 launches instead of 39)
    "conv_gemm_mx_vocoder": lambda f: [r for r in f if _is_mx_gemm(r["Kernel_Name"])][N_DEC_MX:],
    "conv_gemm_mx_decoder": lambda f: [r for r in f if _is_mx_gemm(r["Kernel_Name"])][:N_DEC_MX],
    # (round 4: the two-group variants conv_c64_mx2_kernel / resblock_pair_c32_mx2_kernel are the launchers' default; the fused C = 64 / k = 3 pair)
    # (round 4, second half: the k = 7 / 11 launches of stage 2 run on the streamed conv_gemm_mx64_kernel, the k = 3 up-conv on the persistent conv_c64_mx kernels)
    "conv_c64_mx": lambda f: [r for r in f if "conv_c64_mx" in r["Kernel_Name"] or "conv_gemm_mx64_kernel" in r["Kernel_Name"]],
    "resblock_pair_c64_mx": lambda f: [r for r in f if "resblock_pair_c64_mx" in r["Kernel_Name"]],
    # (round 6: the k = 3 pairs of stage 3 run on resblock_pair_c32_e5_kernel -- E5M2 activation operands -- and stay in this family)
    "resblock_pair_c32_mx": lambda f: [r for r in f if "resblock_pair_c32_mx" in r["Kernel_Name"] or "resblock_pair_c32_e5" in r["Kernel_Name"]],
    "attention_mfma_x3_lds": lambda f: [r for r in f if "attention_mfma_x3_lds_kernel" in r["Kernel_Name"]],
    "conv_post": lambda f: [r for r in f if "conv_post" in r["Kernel_Name"]],
    "mx_planes_kernel": lambda f: [r for r in f if "mx_planes_kernel" in r["Kernel_Name"]],
    "attention_mfma_f32": lambda f: [r for r in f if "attention_mfma_f32_kernel" in r["Kernel_Name"]],
}


def counter_rows(path):
    """{dispatch id: row with one float field per counter} of a rocprofv3 counter_collection.csv (values summed over instances)."""
    by = {}
    for r in csv.DictReader(open(path)):
        d = by.setdefault(r["Dispatch_Id"], dict(Kernel_Name=r["Kernel_Name"], Start_Timestamp=r.get("Start_Timestamp", "0"),
                                                 End_Timestamp=r.get("End_Timestamp", "0")))
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return list(by.values())


def find_csv(dirname, suffix):
    base = os.path.join(OUT, dirname)
    for root, _, files in os.walk(base):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(root, f)
    return None


def family_sums(path, counters):
    rows = counter_rows(path)
    fw = forwards(rows)
    if not fw:
        return {}
    f = fw[-1]
    out = {}
    for name, sel in FAMILIES.items():
        rs = sel(f)
        if rs:
            out[name] = dict(launches=len(rs), **{c: sum(r.get(c, 0.0) for r in rs) for c in counters})
    return out


def calibration():
    """bytes actually moved / counter value, from the dispatches of tools/pmc_calibrate.py (1 GiB read + 1 GiB written each)."""
    cal = {}
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        p = find_csv("pmc_cal_" + kind, "counter_collection.csv")
        if not p:
            continue
        per = {}
        for r in counter_rows(p):
            v = r.get(counter, 0.0)
            if v <= 0:
                continue
            nm = r["Kernel_Name"]
            key = "copyBuffer" if "copyBuffer" in nm else ("elementwise" if "elementwise" in nm else None)
            if key and v * 1024 > 0.2 * GIB:                    # the 1 GiB dispatches only
                per.setdefault(key, []).append(GIB / (v * 1024.0))
        cal[counter] = {k: sum(v) / len(v) for k, v in per.items()}
    return cal


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--mode", default="mx", choices=["mx", "f16"], help="precision the profiled bench.py ran in (its default is mx)")
    args = ap.parse_args()
    pdir = os.path.join(ROOT, "profiles")
    ks = find_csv("prof", "kernel_stats.csv")
    if ks:
        shutil.copy(ks, os.path.join(pdir, args.tag + "_kernel_stats.csv"))
    avgs = []
    kt = find_csv("prof", "kernel_trace.csv")
    if kt:
        for f in forwards(list(csv.DictReader(open(kt)))):
            g = FAMILIES["conv_gemm_mx_vocoder" if args.mode == "mx" else "conv_gemm_f16_vocoder"](f)
            if g:
                avgs.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in g) / len(g) / 1e6)
        print("rocprof kernel trace: avg duration of the vocoder conv_gemm launches per B=32 forward (ms):", ", ".join("%.4f" % a for a in avgs))
    cal = calibration()
    # engine kernels read / write with 16-byte accesses like the elementwise calibration kernel
    f_corr = cal.get("FETCH_SIZE", {}).get("elementwise", 2.0)
    w_corr = cal.get("WRITE_SIZE", {}).get("elementwise", 1.0)
    pf, pw = find_csv("pmc_fetch", "counter_collection.csv"), find_csv("pmc_write", "counter_collection.csv")
    if pf and pw:
        ff, fw_ = family_sums(pf, ["FETCH_SIZE"]), family_sums(pw, ["WRITE_SIZE"])
        fam = {}
        for name in ff:
            if name not in fw_:
                continue
            fb, wb = ff[name]["FETCH_SIZE"] * 1024 * f_corr, fw_[name]["WRITE_SIZE"] * 1024 * w_corr
            fam[name] = dict(launches=ff[name]["launches"], fetch_size_kb_raw=ff[name]["FETCH_SIZE"], write_size_kb_raw=fw_[name]["WRITE_SIZE"],
                             hbm_bytes_per_forward=fb + wb, hbm_bytes_per_launch=(fb + wb) / ff[name]["launches"])
        frames = 32768
        voc = sum(fam[k]["hbm_bytes_per_forward"] for k in ("conv_gemm_f16_vocoder", "resblock_pair_c32_c64", "conv_gemm_mx_vocoder", "conv_c64_mx",
                                                            "resblock_pair_c64_mx", "resblock_pair_c32_mx", "mx_planes_kernel") if k in fam)
        dom = "conv_gemm_mx_vocoder" if args.mode == "mx" else "conv_gemm_f16_vocoder"
        doc = {
            "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on `python bench.py --steps 1 --warmup 1 --cpu-utts 0 --no-other-precision`, the B=32 x 1024-frame forward",
            "correction": dict(fetch_factor=f_corr, write_factor=w_corr, calibration=cal,
                               note="bytes = counter x 1024 x factor; factors = 2^30 / (counter x 1024) measured on dispatches that read and write exactly 1 GiB "
                                    "(tools/pmc_calibrate.py); without a calibration run: FETCH_SIZE x 2 (MI355X_MICROARCH.md section HBM), WRITE_SIZE x 1"),
            "families": fam, "frames": frames, "vocoder_hbm_bytes_per_frame": voc / frames,
            "algorithmic_contract_bytes_per_frame": 2026000.0,
            "hbm_bytes_per_launch": fam.get(dom, {}).get("hbm_bytes_per_launch"),
            "rocprof_avg_launch_ms_per_forward": avgs, "mode": args.mode, "dominant_family": dom,
        }
        # latest_hbm_traffic.json: one entry per precision mode (bench.py reads roofline.traffic of its mode from it)
        lp = os.path.join(pdir, "latest_hbm_traffic.json")
        latest = json.load(open(lp)) if os.path.exists(lp) else {}
        if "families" in latest and "f16" not in latest:          # round-2 layout: the whole file was the fp16 measurement
            latest = {"f16": latest}
        latest[args.mode] = doc
        hp = os.path.join(OUT, "csrc_hash.txt")          # written on the GPU box by tools/gpu_check.sh: the kernel sources the passes ran on
        latest["csrc_hash"] = doc["csrc_hash"] = open(hp).read().strip() if os.path.exists(hp) else None
        json.dump(latest, open(lp, "w"), indent=1)
        json.dump(doc, open(os.path.join(pdir, args.tag + "_hbm_traffic.json"), "w"), indent=1)
        print(json.dumps({k: v for k, v in doc.items() if k not in ("families",)}, indent=1))
    sq = {}
    ps = find_csv("pmc_sq", "counter_collection.csv")
    if ps:
        cs = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"]
        for name, v in family_sums(ps, cs).items():
            d = dict(v)
            if v.get("GRBM_GUI_ACTIVE"):
                # SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over every SIMD of the chip (measured: exactly 16.0
                # per v_mfma_f32_16x16x32_f16 on a launch of known MFMA count); GRBM_GUI_ACTIVE is the sum of the 8 XCDs' active
                # cycles (measured 15.4 cycles / ns = 8 x 1.93 GHz).  Fraction of all matrix-pipe cycles that were busy =
                # busy / (per-XCD active cycles x 256 CUs x 4 SIMDs) -- rocprofv3's own MfmaUtil expression.
                d["mfma_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4)
            if v.get("SQ_BUSY_CU_CYCLES"):
                d["mfma_busy_over_busy_cu_cycles"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / v["SQ_BUSY_CU_CYCLES"]
            if v.get("SQ_WAVE_CYCLES"):
                d["wait_any_frac"] = v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"]
                d["wait_inst_any_frac"] = v.get("SQ_WAIT_INST_ANY", 0.0) / v["SQ_WAVE_CYCLES"]
                d["active_inst_any_frac"] = v.get("SQ_ACTIVE_INST_ANY", 0.0) / v["SQ_WAVE_CYCLES"]
            sq.setdefault(name, {}).update(d)
    pl = find_csv("pmc_lds", "counter_collection.csv")
    if pl:
        cs = ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_UNALIGNED_STALL", "GRBM_GUI_ACTIVE"]
        for name, v in family_sums(pl, cs).items():
            d = {k: v[k] for k in cs if k != "GRBM_GUI_ACTIVE"}
            if v.get("SQ_LDS_IDX_ACTIVE"):
                d["lds_bank_conflict_frac"] = v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]
            sq.setdefault(name, {}).update(d)
    if sq:
        json.dump(dict(source="rocprofv3 --pmc passes of tools/gpu_check.sh pmcsq (SQ counters, --kernel-trace only), last B=32 forward of the run; "
                              "sums over the launches of each kernel family", families=sq), open(os.path.join(pdir, args.tag + "_sq_counters.json"), "w"), indent=1)
        for k, v in sq.items():
            print(k, {a: (round(b, 4) if isinstance(b, float) and b < 10 else b) for a, b in v.items() if "frac" in a or "over" in a})
    for src, dst in (("launches.json", "_launches_f16.json"), ("launches.json.mx", "_launches_mx.json"), ("launches.json.x3", "_launches_strict.json")):
        p = os.path.join(OUT, src)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(pdir, args.tag + dst))
    bl = os.path.join(OUT, "bench.log")
    if os.path.exists(bl):
        for line in open(bl):
            if line.startswith("{"):
                open(os.path.join(pdir, args.tag + "_bench_line.json"), "w").write(line)


if __name__ == "__main__":
    main()
