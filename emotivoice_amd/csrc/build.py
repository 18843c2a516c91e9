#!/usr/bin/env python3
"""Build libevhip.so for gfx950 with hipcc (in-tree: the .so travels to the GPU box with the snapshot).

    python emotivoice_amd/csrc/build.py [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["ev_gemm.hip", "ev_misc.hip", "ev_engine.cpp"]
HEADERS = sorted(f for f in os.listdir(HERE) if f.endswith(".h")) + ["../../include/evhip.h", "../../include/evhip_ops.h"]      # every header in csrc/: a stale .so cost round 3 an hour of wrong measurements
OUT = os.path.join(HERE, "libevhip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _compile(src, extra=(), tag=""):
    obj = os.path.join(HERE, "build", os.path.splitext(src)[0] + tag + ".o")
    deps = [os.path.join(HERE, src)] + [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    if _newer(obj, deps):
        return obj
    cmd = [HIPCC] + FLAGS + list(extra) + ["-x", "hip", "-c", os.path.join(HERE, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force=False):
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    if force:
        for f in os.listdir(os.path.join(HERE, "build")):
            os.remove(os.path.join(HERE, "build", f))
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if force or not _newer(OUT, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return OUT


def build_variant(tag, defines):
    """Tuning variants (e.g. --variant trace EV_TRACE): libevhip_<tag>.so next to the product library, loaded through the
    EVHIP_LIB environment variable by tools/; never used by the package itself."""
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    extra = ["-D" + d for d in defines]
    objs = [_compile(src, extra, "_" + tag) for src in SOURCES]
    out = os.path.join(HERE, "libevhip_%s.so" % tag)
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return out


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
    else:
        print(build(force="--force" in sys.argv))
