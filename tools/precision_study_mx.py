#!/usr/bin/env python3
"""CPU emulation of "one fp16 MFMA + block-scaled (MX) cross terms" recipes for the HiFi-GAN generator (tuning tool).

A product x.w with x = xh + xl, w = wh + wl (xh = fp16(x), xl = x - xh) needs xh.wh + xh.wl + xl.wh for fp32-class
accuracy (the "strict" mode: three fp16 MFMAs).  The two cross terms are 2^-11 of the result, so a few significant bits are
enough for the 1e-3 contract: this tool evaluates them in the OCP MX formats of `v_mfma_scale_f32_16x16x128_f8f6f4`
(fp4 e2m1 / fp6 e2m3 at 4x the fp16 MFMA rate, fp8 e4m3 at 2x), with one E8M0 scale per 32 consecutive K elements
(= 32 input channels of one tap), and reports the waveform error on the zero-mean reference fixture recipe.

    python tools/precision_study_mx.py [--phonemes 48]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import EVShapes, am_forward, hifigan_forward, synth_inputs, synth_state_dict  # noqa: E402
from oracle.jets_oracle import fold_weight_norm, to_torch_sd  # noqa: E402

FORMATS = {            # ebits, mbits, emax (unbiased exponent of the largest binade), max value
    "fp4": (2, 1, 2, 6.0),
    "fp6": (2, 3, 2, 7.5),
    "fp8": (4, 3, 8, 448.0),
    "bf8": (5, 2, 15, 57344.0),      # OCP E5M2 = the top byte of an IEEE half
}


def e5m2_hi(xh, rne):
    """Round 6 study: E5M2 code of an fp16 value WITHOUT a block scale -- the top byte of the half (truncation: one v_perm_b32 per four values), or
    round-to-nearest-even on the dropped byte (one packed 16-bit add per two values in front of it)."""
    bits = xh.half().numpy().view(np.uint16).astype(np.uint32)
    if rne:
        bits = bits + 0x7F + ((bits >> 8) & 1)          # no carry out of the exponent field for the activations at hand (|x| << 57344)
    return torch.from_numpy((bits & 0xFF00).astype(np.uint16).view(np.float16).astype(np.float32)).reshape(xh.shape)


def e5m2_lo(xl):
    """the remainder x - fp16(x) in E5M2 with ONE constant E8M0 scale 2^-11 for every block (|xl| <= 2^-11 2^e: xl 2^11 is inside E5M2's normal range
    for every fp16-normal x): v_cvt_scalef32_pk_bf8_f32, round-to-nearest-even, saturating; no block maximum."""
    return quant_elem(xl * 2048.0, "bf8") / 2048.0


def r16(x):
    return x.half().float()


def quant_elem(v, fmt):
    """round-to-nearest-even of v into a small float format (no block scale), saturating."""
    ebits, mbits, emax, vmax = FORMATS[fmt]
    emin = 2 - (1 << (ebits - 1))          # smallest normal exponent (bias = 2^(ebits-1) - 1)
    a = v.abs().clamp_min(1e-38)
    e = torch.floor(torch.log2(a)).clamp(emin, emax)
    quantum = torch.exp2(e - mbits)
    out = torch.round(v / quantum) * quantum
    return out.clamp(-vmax, vmax)


RULES = {"act": "ocp", "w": "ocp"}          # block-scale rule of the activation / weight quantisers (--act-rule / --w-rule)


def mx_quant(v, fmt, dim, side="act"):
    """MX block quantisation along `dim` in blocks of 32 (zero padded): shared power-of-two scale from the block max.
    Rules: "ocp" = 2^(floor(log2(amax)) - emax_elem) (the OCP MX rule: a block maximum in [6, 8) x scale saturates to 6 in fp4);
    "nosat" = 2^(floor(log2(amax x 4/3)) - emax_elem): the exponent is bumped when the maximum's mantissa is >= 1.5, nothing saturates (fp4 / fp6 only:
    their largest value is 1.5 / 1.875 x 2^emax); "best" = whichever of the two gives the smaller squared error in the block (an offline quantiser: weights)."""
    if fmt is None:
        return v
    rule = RULES[side]
    if rule != "ocp" and fmt in ("fp4", "fp6"):
        v_ = v.movedim(dim, -1)
        n_ = v_.shape[-1]
        pad_ = (-n_) % 32
        vp_ = F.pad(v_, (0, pad_))
        blk_ = vp_.reshape(*vp_.shape[:-1], -1, 32)
        amax_ = blk_.abs().amax(-1, keepdim=True)
        emax_ = FORMATS[fmt][2]
        e0 = torch.floor(torch.log2(amax_.clamp_min(1e-38))) - emax_
        e1 = torch.floor(torch.log2((amax_ * (4.0 / 3.0)).clamp_min(1e-38))) - emax_
        outs = []
        for e_ in (e0, e1) + ((e0 - 1, e0 + 1) if rule == "best3" else ()):
            sc_ = torch.where(amax_ > 0, torch.exp2(e_), torch.ones_like(amax_))
            outs.append(quant_elem(blk_ / sc_, fmt) * sc_)
        if rule == "nosat":
            q_ = outs[1]
        else:
            q_, er_ = outs[0], ((outs[0] - blk_) ** 2).sum(-1, keepdim=True)
            for o_ in outs[1:]:
                e2_ = ((o_ - blk_) ** 2).sum(-1, keepdim=True)
                q_ = torch.where(e2_ < er_, o_, q_)
                er_ = torch.minimum(e2_, er_)
        return q_.reshape(vp_.shape)[..., :n_].movedim(-1, dim)
    v = v.movedim(dim, -1)
    n = v.shape[-1]
    pad = (-n) % 32
    vp = F.pad(v, (0, pad))
    blk = vp.reshape(*vp.shape[:-1], -1, 32)
    amax = blk.abs().amax(-1, keepdim=True)
    emax = FORMATS[fmt][2]
    scale = torch.exp2(torch.floor(torch.log2(amax.clamp_min(1e-38))) - emax)
    scale = torch.where(amax > 0, scale, torch.ones_like(scale))
    qv = quant_elem(blk / scale, fmt) * scale
    return qv.reshape(vp.shape)[..., :n].movedim(-1, dim)


def mx_quant_lo_fixed(lo, hi, fmt, dim):
    """Round 5 study: the remainder plane of an ACTIVATION quantised with a scale DERIVED from the hi plane's block scale instead of its own block maximum
    (|a - fp16(a)| <= 2^-11 2^(e + 1) with e the exponent of the block's largest hi part: scale byte bl = bh - 11) -- saves the producer one block maximum."""
    lo, hi = lo.movedim(dim, -1), hi.movedim(dim, -1)
    n = lo.shape[-1]
    pad = (-n) % 32
    lp, hp = F.pad(lo, (0, pad)), F.pad(hi, (0, pad))
    lb, hb = lp.reshape(*lp.shape[:-1], -1, 32), hp.reshape(*hp.shape[:-1], -1, 32)
    amax = hb.abs().amax(-1, keepdim=True)
    scale = torch.exp2(torch.floor(torch.log2(amax.clamp_min(1e-38))) - FORMATS[fmt][2] - 11)
    scale = torch.where(amax > 0, scale, torch.ones_like(scale))
    qv = quant_elem(lb / scale, fmt) * scale
    return qv.reshape(lp.shape)[..., :n].movedim(-1, dim)


LO_FIXED = [False]          # set per recipe by emulate()


class Recipe:
    def __init__(self, name, cross=None, res32=True, op32=True, hi_from="f16", lo_terms=("xh_wl", "xl_wh"), res_planes=None, mrf16=False,
                 engine_flow=False, mrf_planes=None, lo_fixed=False, e5m2_stages=(), e5m2_rne=False, exact_stages=(), cross_w=None, cross_x=None, res32_stages=(),
                 nocross_stages=(), only_term_stages=None):
        # nocross_stages: the convs of these stages drop BOTH cross terms (one fp16 MFMA per product, fp32 storage as the recipe says) -- what a stage costs in
        # accuracy when it runs the fast mode's arithmetic on the mx mode's data flow; only_term_stages = (stages, term): keep only that cross term there
        self.nocross_stages = tuple(nocross_stages)
        self.only_term_stages = only_term_stages
        # cross_w / cross_x: attribution -- override the format of the WEIGHT / ACTIVATION side of the cross terms ("f16" = exact lo parts)
        self.cross_w, self.cross_x = cross_w, cross_x
        # res32_stages: the residual stream of these stages stays an fp32 tensor (no plane round trip) -- what a per-stage ev_config.mx_residual would buy
        self.res32_stages = tuple(res32_stages)
        # exact_stages: attribution -- the convs of these generator stages evaluate their cross terms exactly (fp16 lo parts), everything else as the recipe says
        self.exact_stages = tuple(exact_stages)
        # e5m2_stages: generator stages (0-3; conv_pre / the up-conv INTO stage i count as stage i's input) whose ACTIVATION cross-term operands are E5M2
        # without block maxima (weights stay block-scaled fp4): Q(xh) = top byte of the fp16 hi plane, Q(xl) = E5M2 of xl 2^11 at the constant scale 2^-11;
        # a residual rebuilt from planes in such a stage uses hi + that E5M2 remainder
        self.e5m2_stages, self.e5m2_rne = tuple(e5m2_stages), e5m2_rne
        self.lo_fixed = lo_fixed          # activation remainders (operand planes and residual planes) with the derived scale
        self.name, self.cross, self.res32, self.op32, self.lo_terms = name, cross, res32, op32, lo_terms
        # mrf16: the first two scaled ResBlock outputs of a stage are stored in fp16 and added in the third one's fp32 epilogue (the fast mode's MRF sum:
        # 8 bytes less HBM traffic per stage-output element than an fp32 running sum)
        # engine_flow: the engine's actual data flow -- stage 3 (C = 32) keeps fp32 in / out (fused pairs), conv_pre's output is fp32 (planes are cut from it)
        self.mrf16, self.engine_flow = mrf16, engine_flow
        # mrf_planes: the first two scaled ResBlock outputs of a stage stored as fp16 hi + remainder codes in this MX format (2.53 bytes per element)
        self.mrf_planes = mrf_planes
        # res_planes: the residual stream exists ONLY as the plane set of leaky_relu(x, .1) its consumers read anyway: fp16 hi plane + the
        # remainder in this format ("f16" = a second fp16 plane, "fp4" / "fp6" = the MX code plane itself); None = a separate fp32 tensor
        self.res_planes = res_planes


def planes_roundtrip(x, fmt):
    """x -> a = leaky_relu(x, .1) -> fp16(a) + Q(a - fp16(a)) -> leaky_relu^-1: what a consumer reconstructs from the plane set."""
    a = F.leaky_relu(x, 0.1)
    h = r16(a)
    lo = a - h
    lo = e5m2_lo(lo) if fmt == "e5m2" else r16(lo) if fmt == "f16" else (mx_quant_lo_fixed(lo, h, fmt, 1) if LO_FIXED[0] else mx_quant(lo, fmt, 1))
    a2 = h + lo
    return torch.where(a2 >= 0, a2, a2 * 10.0)


def conv_mx(x, w, b, rc, transposed=False, e5m2=False, exact=False, kw_nocross=False, only_term=None, **kw):
    """x [1, C, T] fp32, w fp32.  hi.hi in fp16 operands + cross terms in rc.cross (None = omitted, 'f16' = fp16 exact)."""
    op = F.conv_transpose1d if transposed else F.conv1d
    xh, wh = r16(x), r16(w)
    y = op(xh, wh, b, **kw)
    if rc.cross is None or kw_nocross:
        return y
    xl, wl = x - xh, w - wh
    kdim_w = 0 if transposed else 1      # conv_transpose weight is [Cin, Cout, k]
    if RULES["w"] == "joint" and rc.cross == "fp4" and not exact and not getattr(rc, "cross_w", None):
        # offline joint choice (round 6 study): the hi part of a weight may be EITHER fp16 neighbour of w -- the one whose remainder lands closer to a point of the
        # block's fp4 grid (scale: the better of the two candidates of the 'best' rule, fixed from the round-to-nearest remainders).  The fp16 pass then multiplies
        # that hi part, the remainder pass its code: the pair (wh', Q(w - wh')) is what approximates w.
        RULES["w"] = "best"
        q0 = mx_quant(wl, "fp4", kdim_w, "w")
        # block scale actually used: recover from a re-quantisation of the block maxima is awkward; recompute per block
        def blocks(t):
            t_ = t.movedim(kdim_w, -1)
            n_ = t_.shape[-1]
            pad_ = (-n_) % 32
            return F.pad(t_, (0, pad_)).reshape(*t_.shape[:-1], -1, 32), n_, t_.shape
        wb, n_, shp = blocks(w)
        whb, _, _ = blocks(wh)
        wlb = wb - whb
        amax_ = wlb.abs().amax(-1, keepdim=True)
        best = None
        for bump in (0.0, 1.0):
            e_ = torch.floor(torch.log2((amax_ * (4.0 / 3.0 if bump else 1.0)).clamp_min(1e-38))) - 2
            sc_ = torch.where(amax_ > 0, torch.exp2(e_), torch.ones_like(amax_))
            # candidate hi parts: nearest, and the other fp16 neighbour on the remainder's side
            ulp = (torch.nextafter(whb.half(), torch.full_like(whb, float("inf")).half()).float() - whb).abs()
            ulp_dn = (whb - torch.nextafter(whb.half(), torch.full_like(whb, -float("inf")).half()).float()).abs()
            alt = torch.where(wlb >= 0, whb + ulp, whb - ulp_dn)
            res = []
            for cand in (whb, alt):
                r_ = wb - cand
                qr = quant_elem(r_ / sc_, "fp4") * sc_
                res.append((cand, qr, (r_ - qr).abs()))
            pick = res[1][2] < res[0][2]
            hi_ = torch.where(pick, res[1][0], res[0][0])
            ql_ = torch.where(pick, res[1][1], res[0][1])
            err_ = ((wb - hi_ - ql_) ** 2).sum(-1, keepdim=True)
            if best is None:
                best = [hi_, ql_, err_]
            else:
                use = err_ < best[2]
                best = [torch.where(use, hi_, best[0]), torch.where(use, ql_, best[1]), torch.minimum(err_, best[2])]
        def unblocks(tb):
            return tb.reshape(*shp[:-1], -1)[..., :n_].movedim(-1, kdim_w)
        wh2, wlq = unblocks(best[0]), unblocks(best[1])
        y = op(xh, wh2, b, **kw)
        y = y + op(mx_quant(xh, "fp4", 1), wlq, None, **kw)
        y = y + op(mx_quant(xl, "fp4", 1), mx_quant(wh2, "fp4", kdim_w, "w"), None, **kw)
        RULES["w"] = "joint"
        return y
    if rc.cross == "f16" or exact:
        qa = qb = lambda t, d: r16(t)
    else:
        qa = lambda t, d: mx_quant(t, rc.cross, d)
        qb = lambda t, d: mx_quant(t, rc.cross, d, "w")
        if getattr(rc, "cross_x", None):
            qa = (lambda t, d: r16(t)) if rc.cross_x == "f16" else (lambda t, d: mx_quant(t, rc.cross_x, d))
        if getattr(rc, "cross_w", None):
            qb = (lambda t, d: r16(t)) if rc.cross_w == "f16" else (lambda t, d: mx_quant(t, rc.cross_w, d, "w"))
    if e5m2:
        return y + op(e5m2_hi(xh, rc.e5m2_rne), qb(wl, kdim_w), None, **kw) + op(e5m2_lo(xl), qb(wh, kdim_w), None, **kw)
    lo_terms = rc.lo_terms if only_term is None else (only_term,)
    if "xh_wl" in lo_terms:
        if RULES.get("wl_diffuse") and rc.cross == "fp4" and not getattr(rc, "cross_w", None):
            # error diffusion along K for the codes of the weight REMAINDER (its partner xh = leaky-relu outputs has a non-zero mean over the channels, so the
            # coherent part of sum_k xh_k err_k is mean(xh) sum_k err_k: diffusion drives sum_k err_k of a row to ~0).  Scales: the 'best' rule's.
            wq0 = qb(wl, kdim_w)
            wlm = wl.movedim(kdim_w, -1)
            n_ = wlm.shape[-1]
            pad_ = (-n_) % 32
            wp_ = F.pad(wlm, (0, pad_))
            blk_ = wp_.reshape(*wp_.shape[:-1], -1, 32)
            q0_ = F.pad(wq0.movedim(kdim_w, -1), (0, pad_)).reshape(blk_.shape)
            # recover each block's scale from the quantised block: the largest |q| is <= 6 scale; use the OCP / bumped pair and pick the one consistent with q0
            amax_ = blk_.abs().amax(-1, keepdim=True)
            e0 = torch.floor(torch.log2(amax_.clamp_min(1e-38))) - 2
            sc0 = torch.where(amax_ > 0, torch.exp2(e0), torch.ones_like(amax_))
            r0 = quant_elem(blk_ / sc0, "fp4") * sc0
            same0 = ((r0 - q0_).abs().amax(-1, keepdim=True) == 0)
            sc_ = torch.where(same0, sc0, sc0 * 2)
            flat = wp_.reshape(*wp_.shape[:-1], -1)
            scf = sc_.expand(blk_.shape).reshape(flat.shape)
            out = torch.empty_like(flat)
            carry = torch.zeros_like(flat[..., 0])
            alpha = float(RULES["wl_diffuse"])
            for k_ in range(flat.shape[-1]):
                t_ = flat[..., k_] + alpha * carry
                qk = quant_elem(t_ / scf[..., k_], "fp4") * scf[..., k_]
                out[..., k_] = qk
                carry = t_ - qk
            wq1 = out[..., :n_].movedim(-1, kdim_w)
            y = y + op(qa(xh, 1), wq1, None, **kw)
        else:
            y = y + op(qa(xh, 1), qb(wl, kdim_w), None, **kw)
    if "xl_wh" in lo_terms:
        xlq = mx_quant_lo_fixed(xl, xh, rc.cross, 1) if (LO_FIXED[0] and rc.cross not in (None, "f16")) else qa(xl, 1)
        y = y + op(xlq, qb(wh, kdim_w), None, **kw)
    return y


def emulate(sd, mel_ct, shapes, rc, prefix="generator"):
    LO_FIXED[0] = bool(getattr(rc, "lo_fixed", False))
    st = (lambda t: t) if rc.res32 else r16              # residual-stream / stage tensors
    if rc.res_planes:
        st = lambda t: planes_roundtrip(t, rc.res_planes)
    so = (lambda t: t) if rc.op32 else r16               # conv1 -> conv2 intermediates (operand-only tensors)

    stage = [0]

    def conv(x, name, **kw):
        ot = rc.only_term_stages[1] if (rc.only_term_stages and stage[0] in rc.only_term_stages[0]) else None
        return conv_mx(x, fold_weight_norm(sd, name), sd[name + ".bias"], rc, e5m2=stage[0] in rc.e5m2_stages, exact=stage[0] in rc.exact_stages,
                       kw_nocross=stage[0] in rc.nocross_stages, only_term=ot, **kw)

    x = mel_ct.unsqueeze(0)
    x = st(conv(x, prefix + ".conv_pre", padding=3))
    nk = len(shapes.rb_kernels)
    for i, (u, k) in enumerate(zip(shapes.up_rates, shapes.up_kernels)):
        stage[0] = i
        if rc.res_planes:
            fmt_i = "e5m2" if i in rc.e5m2_stages else rc.res_planes
            st = (lambda t, f=fmt_i: planes_roundtrip(t, f))
            if (rc.engine_flow and i + 1 == len(shapes.up_rates)) or i in rc.res32_stages:
                st = (lambda t: t)
        x = F.leaky_relu(x, 0.1)
        # the up-conv INTO stage i reads stage i - 1's output planes
        x = conv_mx(x, fold_weight_norm(sd, f"{prefix}.ups.{i}"), sd[f"{prefix}.ups.{i}.bias"], rc, transposed=True, e5m2=max(i - 1, 0) in rc.e5m2_stages, exact=max(i - 1, 0) in rc.exact_stages,
                    stride=u, padding=(k - u) // 2)
        x = st(x)
        xs = None
        for j, (rk, dils) in enumerate(zip(shapes.rb_kernels, shapes.rb_dils)):
            r = f"{prefix}.resblocks.{i * nk + j}"
            y = x
            for d_i, d in enumerate(dils):
                xt = conv(F.leaky_relu(y, 0.1), f"{r}.convs1.{d_i}", dilation=d, padding=(rk * d - d) // 2)
                xt = so(F.leaky_relu(xt, 0.1))
                xt = conv(xt, f"{r}.convs2.{d_i}", dilation=1, padding=(rk - 1) // 2)
                y = st(xt + y)
            y = y / nk
            if rc.mrf16 and j + 1 < nk:
                y = r16(y)
            if rc.mrf_planes and j + 1 < nk:          # partial sums as fp16 hi + MX remainder codes (no leaky-relu: the raw scaled branch)
                hq = r16(y)
                y = hq + mx_quant(y - hq, rc.mrf_planes, 1)
            xs = y if xs is None else xs + y
        x = st(xs)
    x = F.leaky_relu(x)
    x = F.conv1d(x, fold_weight_norm(sd, prefix + ".conv_post"), sd[prefix + ".conv_post.bias"], padding=3)
    return torch.tanh(x).view(-1)


def errs(a, b):
    a, b = a.double().numpy(), b.double().numpy()
    d = np.linalg.norm(a - b)
    return d / np.linalg.norm(b), d / np.linalg.norm(b - b.mean())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--phonemes", type=int, default=48)
    ap.add_argument("--seed", type=int, default=21)
    ap.add_argument("--mel16", action="store_true", help="round the generator's mel input to fp16 first")
    ap.add_argument("--weights", default="parity", help="synthetic weight recipe: parity (conv_post bias re-centred here for a zero-mean waveform) or "
                                                        "parity_zdc_hot (trained-like gains, the worst reference fixture's recipe)")
    ap.add_argument("--only", default=None, help="substring filter on the recipe names")
    ap.add_argument("--act-rule", default="ocp", choices=("ocp", "nosat", "best"), help="block-scale rule of the activation quantisers (see mx_quant)")
    ap.add_argument("--w-rule", default="ocp", choices=("ocp", "nosat", "best", "best3", "joint"), help="block-scale rule of the weight quantiser (offline: 'best' / 'joint' are free)")
    ap.add_argument("--wl-diffuse", type=float, default=0.0, help="error-diffusion factor along K for the weight remainder's codes (0 = round to nearest)")
    ap.add_argument("--wseed", type=int, default=0, help="weight seed (round 6: the parity suite runs three draws of the weights)")
    args = ap.parse_args()
    torch.set_num_threads(8)
    RULES["act"], RULES["w"] = args.act_rule, args.w_rule
    RULES["wl_diffuse"] = args.wl_diffuse
    shapes = EVShapes()
    sd = to_torch_sd(synth_state_dict(args.wseed, args.weights))
    utt = synth_inputs(args.seed, [args.phonemes], [7])[0]
    with torch.no_grad():
        am = am_forward(sd, torch.from_numpy(utt["ling"]), 7, torch.from_numpy(utt["style"]), torch.from_numpy(utt["content"]), shapes)
        mel = am["dec_outputs"].t().contiguous()
        lo, hi = -20.0, 20.0
        for _ in range(0 if "zdc" in args.weights else 30):
            mid = 0.5 * (lo + hi)
            sd["generator.conv_post.bias"] = torch.tensor([mid])
            if hifigan_forward(sd, mel, shapes).mean().item() > 0:
                hi = mid
            else:
                lo = mid
        ref = hifigan_forward(sd, mel, shapes)
        print("frames %d  wav mean %.4f  std %.4f" % (mel.shape[1], ref.mean().item(), ref.std().item()))
        cases = [
            Recipe("hi.hi only, fp32 storage (1 MFMA)", None),
            Recipe("hi.hi only, fp16 storage (= fast)", None, res32=False, op32=False),
            Recipe("+ fp16 cross terms, fp32 storage (= strict)", "f16"),
            Recipe("+ fp16 cross terms, fp16 residual stream", "f16", res32=False),
            Recipe("+ MX-fp8 cross terms, fp32 storage", "fp8"),
            Recipe("+ MX-fp6 cross terms, fp32 storage", "fp6"),
            Recipe("+ MX-fp4 cross terms, fp32 storage", "fp4"),
            Recipe("+ MX-fp4 cross terms, fp16 residual stream", "fp4", res32=False),
            Recipe("+ MX-fp4 cross terms, residual = hi plane + fp16 remainder plane", "fp4", res_planes="f16"),
            Recipe("+ MX-fp4 cross terms, residual = hi plane + fp6 remainder codes", "fp4", res_planes="fp6"),
            Recipe("+ MX-fp4 cross terms, residual = hi plane + fp4 remainder codes", "fp4", res_planes="fp4"),
            Recipe("ENGINE r4: MX-fp4, residual from fp4 planes (stages 0-2), fp32 stage 3", "fp4", res_planes="fp4", engine_flow=True),
            Recipe("ENGINE r4 + fp16 MRF partial sums", "fp4", res_planes="fp4", engine_flow=True, mrf16=True),
            Recipe("MX-fp4, fp32 residual + fp16 MRF partial sums", "fp4", mrf16=True),
            Recipe("ENGINE r4 + MRF partials as hi + fp4 remainder", "fp4", res_planes="fp4", engine_flow=True, mrf_planes="fp4"),
            Recipe("ENGINE r4 + MRF partials as hi + fp4 remainder, stage 3 from planes too", "fp4", res_planes="fp4", mrf_planes="fp4"),
            Recipe("ENGINE r4 + MRF partials, remainder scale DERIVED from the hi scale (bl = bh - 11)", "fp4", res_planes="fp4", engine_flow=True, mrf_planes="fp4", lo_fixed=True),
            Recipe("E5M2 r6: ENGINE r4 + MRF partials, stage 3 activations E5M2 (truncated hi byte)", "fp4", res_planes="fp4", engine_flow=True, mrf_planes="fp4", e5m2_stages=(3,)),
            Recipe("E5M2 r6: same, hi byte rounded to nearest", "fp4", res_planes="fp4", engine_flow=True, mrf_planes="fp4", e5m2_stages=(3,), e5m2_rne=True),
            Recipe("E5M2 r6: stages 2 + 3 (truncated)", "fp4", res_planes="fp4", engine_flow=True, mrf_planes="fp4", e5m2_stages=(2, 3)),
            Recipe("E5M2 r6: stages 2 + 3 (nearest)", "fp4", res_planes="fp4", engine_flow=True, mrf_planes="fp4", e5m2_stages=(2, 3), e5m2_rne=True),
            Recipe("E5M2 r6: every stage (truncated)", "fp4", res_planes="fp4", engine_flow=True, mrf_planes="fp4", e5m2_stages=(0, 1, 2, 3)),
            Recipe("E5M2 r6: every stage (nearest)", "fp4", res_planes="fp4", engine_flow=True, mrf_planes="fp4", e5m2_stages=(0, 1, 2, 3), e5m2_rne=True),
            Recipe("ATTR r6: ENGINE + MRF partials, stage 0 cross terms exact", "fp4", res_planes="fp4", engine_flow=True, mrf_planes="fp4", exact_stages=(0,)),
            Recipe("ATTR r6: stage 1 exact", "fp4", res_planes="fp4", engine_flow=True, mrf_planes="fp4", exact_stages=(1,)),
            Recipe("ATTR r6: stage 2 exact", "fp4", res_planes="fp4", engine_flow=True, mrf_planes="fp4", exact_stages=(2,)),
            Recipe("ATTR r6: stage 3 exact", "fp4", res_planes="fp4", engine_flow=True, mrf_planes="fp4", exact_stages=(3,)),
            Recipe("ATTR r6: every stage's cross terms exact (what the plane-set residual / MRF partials alone cost)", "fp4", res_planes="fp4", engine_flow=True, mrf_planes="fp4", exact_stages=(0, 1, 2, 3)),
            Recipe("ATTR r6: fp4 cross terms, fp32 residual and MRF sums (what the cross terms alone cost)", "fp4"),
            Recipe("RES r6: ENGINE + MRF partials, fp32 residual in stage 0 only", "fp4", res_planes="fp4", engine_flow=True, mrf_planes="fp4", res32_stages=(0,)),
            Recipe("RES r6: fp32 residual in stage 1 only", "fp4", res_planes="fp4", engine_flow=True, mrf_planes="fp4", res32_stages=(1,)),
            Recipe("RES r6: fp32 residual in stage 2 only", "fp4", res_planes="fp4", engine_flow=True, mrf_planes="fp4", res32_stages=(2,)),
            Recipe("RES r6: fp32 residual in stages 0-2, MRF partials as planes", "fp4", res_planes="fp4", engine_flow=True, mrf_planes="fp4", res32_stages=(0, 1, 2)),
            Recipe("RES r6: residual from planes, fp32 MRF sums", "fp4", res_planes="fp4", engine_flow=True),
            Recipe("ATTR r6: fp32 residual / MRF, fp4 ACTIVATION codes, exact weight lo parts", "fp4", cross_w="f16"),
            Recipe("ATTR r6: fp32 residual / MRF, exact activation parts, fp4 WEIGHT codes", "fp4", cross_x="f16"),
            Recipe("ATTR r6: fp32 residual / MRF, fp4 activations, fp6 weights", "fp4", cross_w="fp6"),
            Recipe("FAST3 r6: ENGINE + MRF partials, stage 3 without cross terms (fp16 operands, fp32 rows)", "fp4", res_planes="fp4", engine_flow=True, mrf_planes="fp4", nocross_stages=(3,)),
            Recipe("FAST3 r6: stage 3 with the weight correction only (xh.wl)", "fp4", res_planes="fp4", engine_flow=True, mrf_planes="fp4", only_term_stages=((3,), "xh_wl")),
            Recipe("FAST3 r6: stage 3 with the operand correction only (xl.wh)", "fp4", res_planes="fp4", engine_flow=True, mrf_planes="fp4", only_term_stages=((3,), "xl_wh")),
            Recipe("FAST3 r6: stages 2 + 3 without cross terms", "fp4", res_planes="fp4", engine_flow=True, mrf_planes="fp4", nocross_stages=(2, 3)),
            Recipe("+ MX-fp4, only xh.wl (weight correction)", "fp4", lo_terms=("xh_wl",)),
            Recipe("+ MX-fp4, only xl.wh (operand correction)", "fp4", lo_terms=("xl_wh",)),
        ]
        m_in = r16(mel) if args.mel16 else mel
        for rc in cases:
            if args.only and args.only not in rc.name:
                continue
            out = emulate(sd, m_in, shapes, rc)
            e, eac = errs(out, ref)
            print("%-52s wav %.3e   wav_ac %.3e" % (rc.name, e, eac), flush=True)


if __name__ == "__main__":
    main()
