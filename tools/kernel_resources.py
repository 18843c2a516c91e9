#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel in a built object (the spill check after touching a kernel: a scratch reload inside a counted-vmcnt
loop drains the DMA pipeline -- DESIGN.md section 4).

    python tools/kernel_resources.py [emotivoice_amd/csrc/build/ev_gemm.o] [--filter mx_kernel] [--spills]
"""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin"


def resources(obj):
    with tempfile.TemporaryDirectory() as td:
        # the host object embeds the gfx950 code object as an offload bundle
        subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + os.path.join(td, "fat.bin"), obj], check=True, capture_output=True)
        subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--unbundle", "--input=" + os.path.join(td, "fat.bin"),
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + os.path.join(td, "dev.co")], check=True, capture_output=True)
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", os.path.join(td, "dev.co")], capture_output=True, text=True).stdout
    out = []
    for blk in notes.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        out.append(dict(name=g("name"), vgpr=g("vgpr_count"), agpr=blk.split()[0], sgpr=g("sgpr_count"),
                        scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"), spill=g("vgpr_spill_count")))
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in out), capture_output=True, text=True).stdout.split("\n")
    for r, n in zip(out, names):
        r["name"] = n
    return out


if __name__ == "__main__":
    skip = {sys.argv.index("--filter") + 1} if "--filter" in sys.argv else set()
    args = [a for i, a in enumerate(sys.argv) if i >= 1 and i not in skip and not a.startswith("--")]
    obj = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "emotivoice_amd", "csrc", "build", "ev_gemm.o")
    flt = sys.argv[sys.argv.index("--filter") + 1] if "--filter" in sys.argv else ""
    for r in resources(obj):
        if flt and flt not in r["name"]:
            continue
        if "--spills" in sys.argv and r["scratch"] in ("0", "?") and r["spill"] in ("0", "?"):
            continue
        print("%-110s vgpr %3s agpr %3s sgpr %3s scratch %5s spill %3s lds %6s" % (r["name"][:110], r["vgpr"], r["agpr"], r["sgpr"], r["scratch"], r["spill"], r["lds"]))
