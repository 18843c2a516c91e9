#!/usr/bin/env python3
"""ISA hazard lint for the shipped gfx950 code objects.

gfx9 has no hardware interlock between an MFMA's register write and a later VALU / LDS / VMEM access of the same registers: the software
must put wait states between them.  hipcc pads the MFMAs it schedules itself, but an inline-asm MFMA (mfma_inplace / mfma_mx_inplace: tied
accumulators, v_mfma_scale_f32_16x16x128_f8f6f4 has no builtin here) is opaque to it -- round 4 found a kernel that had shipped with a VALU read
two issue slots behind a block-scaled MFMA (mfma_asm_fence, ev_gemm.hip).  This tool disassembles every code object embedded in libevhip.so
(or in a .o) and checks, for EVERY v_mfma* instruction, the issue distance to the first instruction that touches its destination registers:

  consumer                                                                   required wait states (gfx950; p = passes of the producing MFMA)
  ------------------------------------------------------------------------   ------------------------------------------------------------
  next MFMA of the SAME opcode taking the destination WHOLE as SrcC          0      (the accumulate chain the hardware forwards)
  next MFMA of ANOTHER opcode taking it whole as SrcC                        0 for the pairs MEASURED exact at 0 states (CHAIN_MEASURED_OK: f16 K = 32 <-> fp4
                                                                                     block-scaled), p + 4 for every other pair -- tools/mfma_chain_check.hip: 16x16x32 f16
                                                                                     <-> legacy 16x16x16 loses products at 0 and 4 states, is exact from 5, and hipcc
                                                                                     (whose table says 0) pads nothing, builtins included
  MFMA whose SrcC overlaps it partially / whose vDst overlaps it (WAW)       p + 3  (LLVM GCNHazardRecognizer, GFX940 XDL -> XDL SrcC: p + 2, + 1 on gfx950)
  MFMA reading it as SrcA / SrcB / scale                                     p + 4  (XDL write -> SrcA/B: p + 3, + 1 on gfx950)
  anything else that reads or writes it (VALU, v_accvgpr_*, DS, VMEM, ...)   p + 4  (XDL write -> VALU / memory read, WAW: p + 3, + 1 on gfx950;
                                                                                     /opt/skills guide 5.7: "8-pass XDL: 12 states")
  the same for a non-XDL (SGEMM) producer, v_mfma_f32_16x16x4_f32            p + 2  (hipcc's own rule; these are only ever compiler-scheduled)

Passes (16 cycles = 4 passes per 16x16 tile at the double-K rate of gfx950): 16x16x32 f16 / bf16 and the legacy 16x16x16 forms 4, 32x32x16 8,
16x16x4 f32 8, v_mfma_scale_f32_16x16x128_f8f6f4 4 for fp4 / fp6 operands and 8 with an fp8 operand -- the lint does not decode cbsz / blgp and
prices every block-scaled MFMA (always inline asm in this library) at 8 passes = 12 states.

A wait state = one issued instruction; `s_nop N` = N + 1.  The scan follows fall-through, both sides of a conditional branch and the target of
s_branch, up to the required distance (<= 20 instructions), so loop back-edges are covered.  Output: one line per violation, and per kernel
the minimum slack (distance - requirement) over all its MFMAs and the mixed-opcode accumulate chains it contains.

    python tools/isa_hazard_lint.py [emotivoice_amd/csrc/libevhip.so] [--json out.json] [--verbose]
"""
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"

PASSES = [            # (opcode regex, passes, is_xdl)
    (re.compile(r"^v_mfma_scale_f32_16x16x128_f8f6f4"), 8, True),
    (re.compile(r"^v_mfma_scale_f32_32x32x64_f8f6f4"), 16, True),
    (re.compile(r"^v_mfma_f32_16x16x128_f8f6f4"), 8, True),
    (re.compile(r"^v_mfma_f32_32x32x64_f8f6f4"), 16, True),
    (re.compile(r"^v_mfma_f32_16x16x4_f32"), 8, False),
    (re.compile(r"^v_mfma_f32_32x32x2_f32"), 16, False),
    (re.compile(r"^v_mfma_f32_4x4x"), 2, True),
    (re.compile(r"^v_mfma_[a-z0-9]+_32x32x"), 8, True),
    (re.compile(r"^v_mfma_[a-z0-9]+_16x16x"), 4, True),
]


# Accumulate chains ACROSS opcodes (SrcC == vDst whole).  hipcc's hazard table has 0 wait states for them and pads nothing, the hardware is not that
# uniform: measured on the MI355X with exact integer operands (tools/mfma_chain_check.hip, profiles/r5_a_mfma_chain_check.txt, 4 waves / SIMD, 2 x 10^7 words):
#   v_mfma_f32_16x16x32_f16 <-> legacy v_mfma_f32_16x16x16_f16 : WRONG at 0 and 4 states (also as compiler builtins), exact at 5, 6, 7, 8, 16 -> not in this set
#   v_mfma_f32_16x16x32_f16 <-> v_mfma_scale_f32_16x16x128_f8f6f4 (fp4 operands): exact at 0 states in both orders -> the MX kernels' pass 0 -> pass 1 hand-over
#   the same with an E5M2 (bf8) B operand, `cbsz:4 blgp:1` (round 6, profiles/r6_a_mfma_chain_check.txt): exact at 0 states in both orders, and against the fp4 x fp4 form
# A block-scaled MFMA's chain key carries its operand formats (cbsz = A, blgp = B: 0 fp8, 1 bf8, 2 fp6, 3 bf6, 4 fp4): a pair is only as good as its measurement.
MX16 = "v_mfma_scale_f32_16x16x128_f8f6f4"
CHAIN_MEASURED_OK = {frozenset(("v_mfma_f32_16x16x32_f16", MX16 + " cbsz:4 blgp:4")),
                     frozenset(("v_mfma_f32_16x16x32_f16", MX16 + " cbsz:4 blgp:1")),
                     frozenset((MX16 + " cbsz:4 blgp:4", MX16 + " cbsz:4 blgp:1"))}


def chain_key(x):
    """Opcode as the accumulate-chain table sees it: block-scaled MFMAs with their operand formats."""
    if "f8f6f4" not in x.op:
        return x.op
    a, b = re.search(r"cbsz:(\d+)", x.args), re.search(r"blgp:(\d+)", x.args)
    return "%s cbsz:%s blgp:%s" % (x.op, a.group(1) if a else "0", b.group(1) if b else "0")


def mfma_info(op):
    for rx, p, xdl in PASSES:
        if rx.match(op):
            return p, xdl
    raise ValueError("no pass count for %s: extend PASSES" % op)


def device_objects(path, td):
    """Code objects (gfx950 ELF files) embedded in a host object / shared library."""
    fat = os.path.join(td, "fat.bin")
    subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, path], check=True, capture_output=True)
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
    out = []
    for i, s in enumerate(starts):
        part = os.path.join(td, "bundle%d.bin" % i)
        with open(part, "wb") as f:
            f.write(data[s:starts[i + 1] if i + 1 < len(starts) else len(data)])
        co = os.path.join(td, "dev%d.co" % i)
        r = subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--unbundle", "--input=" + part,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            out.append(co)
    return out


REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")
LINE = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
FUNC = re.compile(r"^[0-9a-f]+ <(.+)>:$")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        f = m.group(1)
        if m.group(2) is not None:
            out.add((f, int(m.group(2))))
        else:
            out.update((f, i) for i in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


class Ins:
    __slots__ = ("op", "args", "addr", "fn", "regs", "ops")

    def __init__(self, op, args, addr, fn):
        self.op, self.args, self.addr, self.fn = op, args, addr, fn
        self.regs = regs_of(args)
        self.ops = None


def split_operands(args):
    """Top-level comma split of an operand string (modifiers like op_sel_hi:[0,0,0] stay with the last operand)."""
    out, depth, cur = [], 0, ""
    for ch in args:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def parse(co):
    txt = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
    ins, fn = [], None
    for line in txt.splitlines():
        m = FUNC.match(line)
        if m:
            fn = m.group(1)
            continue
        m = LINE.match(line)
        if m and fn is not None:
            ins.append(Ins(m.group(1), m.group(2), int(m.group(3), 16), fn))
    return ins


def branch_target(i, ins, by_addr):
    """Index of the branch target (SOPP simm16 in dwords relative to the next instruction), or None."""
    m = re.match(r"^(-?\d+)", ins[i].args.strip())
    if not m:
        return None
    nxt = ins[i + 1].addr if i + 1 < len(ins) else ins[i].addr + 4
    simm = int(m.group(1))
    if simm >= 32768:
        simm -= 65536
    return by_addr.get(nxt + 4 * simm)


def lint(ins):
    by_addr = {x.addr: k for k, x in enumerate(ins)}
    violations, per_fn = [], {}
    for k, x in enumerate(ins):
        if not x.op.startswith("v_mfma"):
            continue
        passes, xdl = mfma_info(x.op)
        ops = split_operands(x.args)
        dst = regs_of(ops[0])
        st = per_fn.setdefault(x.fn, dict(mfma=0, min_slack=None, chains={}, worst=None))
        st["mfma"] += 1
        need_other = passes + (4 if xdl else 2)
        need_srcc = passes + 3
        horizon = need_other + 12          # look a little further than required, to report the slack of the tightest site
        # DFS over the instruction stream: (index, wait states issued so far)
        stack, seen = [(k + 1, 0)], set()
        while stack:
            j, ws = stack.pop()
            while j < len(ins) and ws < horizon:
                if (j, ws) in seen:
                    break
                seen.add((j, ws))
                y = ins[j]
                if y.fn != x.fn:
                    break
                touch = y.regs & dst
                if touch:
                    need, kind = need_other, "non-MFMA access"          # (per instruction: nothing is inherited from the previous one)
                    if y.op.startswith("v_mfma"):
                        yo = split_operands(y.args)
                        ydst, ya, yb, yc = regs_of(yo[0]), regs_of(yo[1]), regs_of(yo[2]), regs_of(yo[3])
                        yscale = set().union(*[regs_of(o) for o in yo[4:]]) if len(yo) > 4 else set()
                        if (ya | yb | yscale) & dst:
                            need, kind = need_other, "MFMA SrcA/B/scale"
                        elif yc == dst:
                            kx, ky = chain_key(x), chain_key(y)
                            key = "%s -> %s" % (kx, ky)
                            c = st["chains"].setdefault(key, [0, None])
                            c[0] += 1
                            c[1] = ws if c[1] is None else min(c[1], ws)
                            if ky == kx or frozenset((kx, ky)) in CHAIN_MEASURED_OK:
                                need, kind = 0, "chain"
                            else:
                                # a dependent chain across two opcodes that nobody has measured: priced like any other reader (p + 4), which is where the
                                # one pair measured BAD (16x16x32 f16 <-> legacy 16x16x16 f16: wrong at 0 and 4 states, exact from 5) is safe with margin
                                need, kind = need_other, "accumulate chain across two MFMA opcodes"
                        else:
                            # SrcC overlaps the producer's destination only partially, or only vDst does (write after write)
                            need, kind = need_srcc, "MFMA partial SrcC / vDst overlap"
                    slack = ws - need
                    if kind != "chain":
                        if st["min_slack"] is None or slack < st["min_slack"]:
                            st["min_slack"], st["worst"] = slack, "%x: %s -> %x: %s %s" % (x.addr, x.op, y.addr, y.op, y.args[:60])
                        if slack < 0:
                            violations.append(dict(kernel=x.fn, producer="%x: %s %s" % (x.addr, x.op, x.args[:70]), consumer="%x: %s %s" % (y.addr, y.op, y.args[:70]),
                                                   kind=kind, wait_states=ws, required=need))
                    break          # the first access decides (later ones are further away)
                if y.op == "s_endpgm" or y.op.startswith("s_setpc") or y.op.startswith("s_swappc"):
                    break
                step = 1
                if y.op == "s_nop":
                    step = int(y.args.strip() or 0) + 1
                if y.op == "s_branch":
                    t = branch_target(j, ins, by_addr)
                    if t is None:
                        break
                    j, ws = t, ws + step
                    continue
                if y.op.startswith("s_cbranch"):
                    t = branch_target(j, ins, by_addr)
                    if t is not None:
                        stack.append((t, ws + step))
                j, ws = j + 1, ws + step
    return violations, per_fn


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def run(path):
    with tempfile.TemporaryDirectory() as td:
        cos = device_objects(path, td)
        if not cos:
            raise RuntimeError("no gfx950 code object found in " + path)
        allv, allfn = [], {}
        for co in cos:
            v, f = lint(parse(co))
            allv += v
            allfn.update(f)
    names = demangle(list(allfn))
    for v in allv:
        v["kernel"] = names.get(v["kernel"], v["kernel"])
    return allv, {names.get(k, k): v for k, v in allfn.items()}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    skip = set()
    if "--json" in sys.argv:
        skip.add(sys.argv[sys.argv.index("--json") + 1])
    args = [a for a in args if a not in skip]
    path = args[0] if args else os.path.join(ROOT, "emotivoice_amd", "csrc", "libevhip.so")
    viol, fns = run(path)
    kernels = {k: v for k, v in fns.items() if v["mfma"]}
    print("%d kernels with MFMAs, %d MFMA instructions, %d violations" % (len(kernels), sum(v["mfma"] for v in kernels.values()), len(viol)))
    mixed = {}
    for k, v in kernels.items():
        for c, (n, dmin) in v["chains"].items():
            a, b = c.split(" -> ")
            if a != b:
                m = mixed.setdefault(c, [0, 0, dmin])
                m[0] += n; m[1] += 1; m[2] = min(m[2], dmin)
    for c, (n, nk, dmin) in sorted(mixed.items()):
        print("accumulate chain across two opcodes: %-80s %6d sites in %3d kernels, closest pair %d states apart" % (c, n, nk, dmin))
    tight = sorted(((v["min_slack"], k, v["worst"]) for k, v in kernels.items() if v["min_slack"] is not None), key=lambda t: t[0])
    for s, k, w in tight[: (len(tight) if "--verbose" in sys.argv else 12)]:
        print("min slack %3d  %-100s %s" % (s, k[:100], w))
    for v in viol[:50]:
        print("VIOLATION %(kernel).90s\n    %(producer)s\n    %(consumer)s\n    %(kind)s: %(wait_states)d wait states, %(required)d required" % v)
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump(dict(library=os.path.relpath(path, ROOT), violations=viol, mixed_chains={c: dict(sites=n, kernels=nk, min_states_apart=d) for c, (n, nk, d) in mixed.items()},
                           kernels={k: dict(mfma=v["mfma"], min_slack=v["min_slack"], tightest=v["worst"]) for k, v in kernels.items()}), f, indent=1, sort_keys=True)
    sys.exit(1 if viol else 0)
