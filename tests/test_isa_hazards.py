"""ISA hazard lint as a CPU test: every v_mfma* of every kernel embedded in libevhip.so is checked against the gfx950 wait-state
table (tools/isa_hazard_lint.py) -- distance to the first VALU / DS / VMEM access of its destination registers, partial SrcC / vDst overlaps, and
accumulate chains ACROSS opcodes, which hipcc never pads and of which one pair measurably loses products on the MI355X (16x16x32 f16 <-> the legacy
16x16x16 form: profiles/r5_a_mfma_chain_check.txt).  Inline-asm MFMAs are opaque to hipcc's own hazard recogniser, so without this test the next
instantiation of an asm kernel can regress below every emulation tolerance (round 3 shipped one such kernel: mfma_asm_fence, ev_gemm.hip)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_hazard_lint as lint      # noqa: E402

LIB = os.path.join(ROOT, "emotivoice_amd", "csrc", "libevhip.so")


HIPCC = "/opt/rocm/bin/hipcc"


def _need_tools(*extra):
    """These are CPU tests, but they drive the ROCm binutils: skip (not error) on a host without them."""
    missing = [t for t in (lint.LLVM + "/llvm-objcopy", lint.LLVM + "/clang-offload-bundler", lint.LLVM + "/llvm-objdump") + extra if not os.path.exists(t)]
    if missing:
        pytest.skip("ROCm tools not installed: " + ", ".join(missing))


@pytest.fixture(scope="module")
def shipped():
    if not os.path.exists(LIB):
        pytest.skip("libevhip.so not built")
    _need_tools()
    return lint.run(LIB)


def test_every_shipped_mfma_respects_the_wait_state_table(shipped):
    viol, fns = shipped
    kernels = {k: v for k, v in fns.items() if v["mfma"]}
    assert len(kernels) >= 200 and sum(v["mfma"] for v in kernels.values()) >= 30000        # the whole library was seen, not one code object of it
    assert any("conv_gemm_mx_kernel" in k for k in kernels) and any("attention_mfma_x3_lds_kernel" in k for k in kernels)      # ev_gemm.hip AND ev_misc.hip
    assert not viol, "\n".join("%(kernel).100s: %(producer)s -> %(consumer)s (%(kind)s: %(wait_states)d of %(required)d states)" % v for v in viol[:10])


def test_no_unmeasured_cross_opcode_accumulate_chain(shipped):
    """The only chain across two MFMA opcodes in the library is f16 K = 32 <-> block-scaled fp4 (the MX kernels' pass hand-over), measured exact at 0
    states; in particular no kernel chains a 16x16x32 with a legacy 16x16x16 MFMA on one accumulator (the round-4 attention finding)."""
    _, fns = shipped
    pairs = set()
    for v in fns.values():
        for c in v["chains"]:
            a, b = c.split(" -> ")
            if a != b:
                pairs.add(frozenset((a, b)))
    assert pairs <= lint.CHAIN_MEASURED_OK, pairs - lint.CHAIN_MEASURED_OK
    assert not any("16x16x16" in op for p in pairs for op in p)


SELF_TEST = r'''
#include <hip/hip_runtime.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
// early_read: an inline-asm MFMA whose accumulator a VALU op reads two issue slots later (the bug class of round 3)
extern "C" __global__ void early_read(const h8* a, const h8* b, f4* o) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a[threadIdx.x]), "v"(b[threadIdx.x]));
    o[threadIdx.x] = acc * 2.0f;
}
// fenced: the same with the library's 20 wait states
extern "C" __global__ void fenced(const h8* a, const h8* b, f4* o) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a[threadIdx.x]), "v"(b[threadIdx.x]));
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc));
    o[threadIdx.x] = acc * 2.0f;
}
// mixed_chain: K = 32 then the legacy K = 16 form on one accumulator, as compiler builtins (hipcc pads nothing between them)
extern "C" __global__ void mixed_chain(const h8* a, const h8* b, const h4* a2, const h4* b2, f4* o) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a2[threadIdx.x], b2[threadIdx.x], acc, 0, 0, 0);
    o[threadIdx.x] = acc;
}
// partial_srcc: the second MFMA's SrcC (v[22:25]) overlaps the first one's destination (v[20:23]) only partially, one slot behind it
extern "C" __global__ void partial_srcc(const h8* a, const h8* b, float* o) {
    float r;
    asm volatile("v_mfma_f32_16x16x32_f16 v[20:23], %1, %2, 0\n\tv_mfma_f32_16x16x32_f16 v[24:27], %1, %2, v[22:25]\n\t"
                 "s_nop 7\n\ts_nop 7\n\ts_nop 3\n\tv_mov_b32 %0, v24"
                 : "=v"(r) : "v"(a[threadIdx.x]), "v"(b[threadIdx.x]) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
    o[threadIdx.x] = r;
}
// waw: the second MFMA overwrites the first one's destination (SrcC = 0: no chain) one slot behind it
extern "C" __global__ void waw(const h8* a, const h8* b, float* o) {
    float r;
    asm volatile("v_mfma_f32_16x16x32_f16 v[20:23], %1, %2, 0\n\tv_mfma_f32_16x16x32_f16 v[20:23], %2, %1, 0\n\t"
                 "s_nop 7\n\ts_nop 7\n\ts_nop 3\n\tv_mov_b32 %0, v20"
                 : "=v"(r) : "v"(a[threadIdx.x]), "v"(b[threadIdx.x]) : "v20", "v21", "v22", "v23");
    o[threadIdx.x] = r;
}
'''


def test_lint_flags_the_known_bad_patterns(tmp_path):
    """The lint must SEE what it is there for: a VALU read two slots behind an asm MFMA, and a K32 -> K16 builtin chain; and pass the fenced form."""
    _need_tools(HIPCC)
    src = tmp_path / "selftest.hip"
    src.write_text(SELF_TEST)
    obj = tmp_path / "selftest.o"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-c", str(src), "-o", str(obj)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    viol, fns = lint.run(str(obj))
    by = {}
    for v in viol:
        by.setdefault(v["kernel"].split("(")[0], []).append(v)
    assert "early_read" in by and by["early_read"][0]["kind"] == "non-MFMA access" and by["early_read"][0]["wait_states"] < 8
    assert "mixed_chain" in by and by["mixed_chain"][0]["kind"] == "accumulate chain across two MFMA opcodes"
    assert "fenced" not in by
    # (round 6) the partial-SrcC / write-after-write class the docstring promises: p + 3 = 7 states, both kernels sit 0 states behind the producer
    for k in ("partial_srcc", "waw"):
        assert k in by and by[k][0]["kind"] == "MFMA partial SrcC / vDst overlap" and by[k][0]["required"] == 7 and by[k][0]["wait_states"] == 0, (k, by.get(k))
    fenced = [v for k, v in fns.items() if k.startswith("fenced")]
    assert fenced and fenced[0]["mfma"] == 1 and (fenced[0]["min_slack"] is None or fenced[0]["min_slack"] >= 0)      # None: no access within the look-ahead
