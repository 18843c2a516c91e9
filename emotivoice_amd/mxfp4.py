"""OCP MX-fp4 (e2m1 elements, E8M0 power-of-two block scales) quantiser on the host, numpy only.

Used by ``packer.py`` for the weight side of the "mx" precision mode and by the tests as the statement of what the device
planes contain.  In that mode a product x.w (x = xh + xl, w = wh + wl, the fp16 hi parts and their fp32 remainders) is
evaluated as

    xh.wh                      one fp16 MFMA (v_mfma_f32_16x16x32_f16)
  + Q(wl).Q(xh) + Q(wh).Q(xl)  two block-scaled fp4 MFMAs (v_mfma_scale_f32_16x16x128_f8f6f4, 4x the fp16 rate)

The cross terms are 2^-11 of the result, so the ~2 significant bits of fp4 leave a relative error of ~3e-5 per product instead
of fp16's 3e-4 (tools/precision_study_mx.py: waveform 3.5e-4 against 2.3e-3 for plain fp16 operands, bar 1e-3).

Element code (4 bits): s e e m, values +-{0, .5, 1, 1.5, 2, 3, 4, 6}; two codes per byte, element 2j in the low nibble of
byte j (checked against the instruction on the device: tools/mfma_ubench.hip).  Scale byte b means 2^(b - 127); the block scale
is 2^(floor(log2(amax)) - 2) (OCP MX v1.0: the largest element lands in the top binade, values above 6 saturate).
"""
from __future__ import annotations

import numpy as np

FP4_VALUES = np.array([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0], np.float32)


def scale_bytes(amax: np.ndarray) -> np.ndarray:
    """E8M0 byte of the block scale for a block whose largest magnitude is ``amax`` (fp32): biased exponent - 2, at least 1
    (0 would be 2^-127, a subnormal the conversion instruction cannot divide by; all-zero blocks get 1)."""
    bits = np.ascontiguousarray(amax, np.float32).view(np.uint32)
    e = ((bits >> 23) & 0xFF).astype(np.int32) - 2
    return np.clip(e, 1, 254).astype(np.uint8)


def encode_fp4(y: np.ndarray) -> np.ndarray:
    """fp32 (already divided by the block scale) -> 4-bit codes, round to nearest even, saturating at 6."""
    a = np.abs(y)
    c = ((a > 0.25).astype(np.uint8) + (a >= 0.75) + (a > 1.25) + (a >= 1.75) + (a > 2.5) + (a >= 3.5) + (a > 5.0)).astype(np.uint8)
    return (c | (np.signbit(y).astype(np.uint8) << 3)).astype(np.uint8)


def decode_fp4(codes: np.ndarray) -> np.ndarray:
    v = FP4_VALUES[codes & 7]
    return np.where(codes & 8, -v, v).astype(np.float32)


def quantize(v: np.ndarray, block: int = 32, rule: str = "ocp"):
    """v [..., K] fp32 (K % block == 0, block % 32 == 0) -> (codes uint8 [..., K/2], scales uint8 [..., K/block]).
    One scale per ``block`` consecutive elements (the instruction takes one per 32; a coarser block repeats it).

    rule "ocp": the scale of ``scale_bytes`` -- what the device quantisers of the ACTIVATIONS compute (ev_mxq.h), bit for bit.
    rule "best" (round 6, the WEIGHT planes, packed once on the host): a block whose maximum has a mantissa >= 1.5 saturates under the OCP rule (values in
    (6, 8) x scale become 6); the next scale up represents it exactly but doubles every other element's quantum.  Per block, whichever of the two leaves the
    smaller squared error is kept.  Any E8M0 byte is a valid operand of the block-scaled MFMA, so this costs nothing at run time; emulated on the
    trained-like weight draws (tools/precision_study_mx.py --w-rule best) it takes 7-17 % off the waveform error of the whole generator."""
    v = np.ascontiguousarray(v, np.float32)
    K = v.shape[-1]
    assert K % block == 0 and block % 32 == 0 and rule in ("ocp", "best")
    blk = v.reshape(v.shape[:-1] + (K // block, block))
    amax = np.abs(blk).max(-1)
    sb = scale_bytes(amax)

    def encode(sbytes):
        inv = np.ldexp(np.float32(1.0), 127 - sbytes.astype(np.int32)).astype(np.float32)
        return encode_fp4(blk * inv[..., None])

    codes = encode(sb)
    if rule == "best":
        sb1 = np.minimum(sb.astype(np.int32) + 1, 254).astype(np.uint8)
        codes1 = encode(sb1)

        def sqerr(c, sbytes):
            sc = np.ldexp(np.float32(1.0), sbytes.astype(np.int32) - 127).astype(np.float32)
            d = (decode_fp4(c) * sc[..., None]).astype(np.float64) - blk
            return (d * d).sum(-1)

        up = sqerr(codes1, sb1) < sqerr(codes, sb)
        codes = np.where(up[..., None], codes1, codes)
        sb = np.where(up, sb1, sb).astype(np.uint8)
    codes = codes.reshape(v.shape)
    packed = (codes[..., 0::2] | (codes[..., 1::2] << 4)).astype(np.uint8)
    return packed, sb


W_RULE = "best"         # block-scale rule of every weight plane (see quantize)


def dequantize(packed: np.ndarray, sb: np.ndarray, block: int = 32) -> np.ndarray:
    codes = np.empty(packed.shape[:-1] + (packed.shape[-1] * 2,), np.uint8)
    codes[..., 0::2] = packed & 15
    codes[..., 1::2] = packed >> 4
    v = decode_fp4(codes)
    K = v.shape[-1]
    sc = np.ldexp(np.float32(1.0), sb.astype(np.int32) - 127).astype(np.float32)
    return (v.reshape(v.shape[:-1] + (K // block, block)) * sc[..., None]).reshape(v.shape)


def split_hi_lo(x: np.ndarray):
    """fp32 -> (fp16 hi part as fp32, exact fp32 remainder)."""
    x = np.ascontiguousarray(x, np.float32)
    hi = x.astype(np.float16).astype(np.float32)
    return hi, x - hi


# ---- Round 6: maxima-free E5M2 operands for the ACTIVATION side of the cross terms (ev_pair_e5.h).  OCP E5M2 (bf8: s eeeee mm, bias 15) is the top byte of an
# IEEE half, so Q(xh) needs no conversion, no block maximum and no scale; the remainder x - fp16(x) is at most 2^-11 of x's binade, so xl 2^11 fits E5M2's range
# for every fp16-normal x and ONE constant E8M0 scale (2^-11) serves every block.
E5M2_LO_SHIFT = 11


def e5m2_decode(codes: np.ndarray) -> np.ndarray:
    """E5M2 bytes -> fp32 (exact: an E5M2 code is the top byte of the half with a zero low byte)."""
    return (np.ascontiguousarray(codes, np.uint8).astype(np.uint16) << 8).view(np.float16).astype(np.float32)


def e5m2_hi_codes(hi: np.ndarray) -> np.ndarray:
    """fp16 hi parts (any float dtype holding fp16-representable values) -> their E5M2 codes by TRUNCATION: the top byte of the half (v_perm_b32 on the device)."""
    return (np.ascontiguousarray(hi).astype(np.float16).view(np.uint16) >> 8).astype(np.uint8)


def e5m2_lo_codes(lo: np.ndarray) -> np.ndarray:
    """fp32 remainders x - fp16(x) -> E5M2 codes of lo 2^11, round to nearest even, saturating (v_cvt_scalef32_pk_bf8_f32 with scale 2^-11)."""
    y = np.ascontiguousarray(lo, np.float32).astype(np.float64) * float(1 << E5M2_LO_SHIFT)
    a = np.abs(y)
    e = np.floor(np.log2(np.maximum(a, 1e-300)))
    e = np.clip(e, -14, 15)                              # subnormals share the quantum of the smallest normal binade
    q = np.ldexp(1.0, (e - 2).astype(np.int64))
    r = np.minimum(np.rint(a / q) * q, 57344.0)          # np.rint: half to even
    h = np.where(np.signbit(y), -r, r).astype(np.float16)
    return (h.view(np.uint16) >> 8).astype(np.uint8)


def e5m2_lo_decode(codes: np.ndarray) -> np.ndarray:
    return e5m2_decode(codes) * np.float32(1.0 / (1 << E5M2_LO_SHIFT))


W_SCALE_BLOCK = 128     # weights: one scale per (output channel, tap, 128 input channels) = one per MFMA A-row


def pack_weight_planes(w: np.ndarray) -> np.ndarray:
    """GEMM-layout weight [N][taps][K] fp32 (N % 128 == 0, K % 128 == 0) -> one byte array
         [wl4: N x taps x K/2] [wh4: N x taps x K/2] [swl: N/128 x K/128 x taps x 128] [swh: same]
    wl4 / wh4: fp4 codes of the fp32 remainder w - fp16(w) and of fp16(w); scale planes are tile-major so that the scales of one
    (128-channel output tile, 128-channel K chunk) are ``taps * 128`` contiguous bytes (one or two LDS-DMA pieces)."""
    w = np.ascontiguousarray(w, np.float32)
    N, taps, K = w.shape
    assert N % 128 == 0 and K % 128 == 0, (N, K)
    hi, lo = split_hi_lo(w)
    out = []
    scales = []
    for part in (lo, hi):
        codes, sb = quantize(part, W_SCALE_BLOCK, W_RULE)              # sb [N][taps][K/128]
        out.append(codes.reshape(-1))
        scales.append(np.ascontiguousarray(sb.reshape(N // 128, 128, taps, K // 128).transpose(0, 3, 2, 1)).reshape(-1))
    return np.concatenate(out + scales)


def weight_planes_dequant(blob: np.ndarray, N: int, taps: int, K: int):
    """inverse of pack_weight_planes (tests): -> (Q(wl), Q(wh)) as fp32 [N][taps][K]."""
    nc = N * taps * K // 2
    ns = N * taps * (K // 128)
    res = []
    for i in range(2):
        codes = blob[i * nc:(i + 1) * nc].reshape(N, taps, K // 2)
        sb = blob[2 * nc + i * ns: 2 * nc + (i + 1) * ns].reshape(N // 128, K // 128, taps, 128).transpose(0, 3, 2, 1).reshape(N, taps, K // 128)
        res.append(dequantize(codes, sb, W_SCALE_BLOCK))
    return res[0], res[1]


def pack_pair_weight_planes(w: np.ndarray) -> np.ndarray:
    """GEMM-layout weight [32][taps][32] fp32 of a C = 32 ResBlock conv -> the fp4 planes of the fused MX pair kernel (ev_pair_mx.h):
         [Q(w - fp16(w)): KP x 32 x 16 B] [Q(fp16(w)): KP x 32 x 16 B] [scales of the first: KP x 32] [scales of the second: KP x 32]
    tap-major ([tap][output channel]), one scale per (output channel, tap) = per 32 input channels = per MFMA k-block; the taps are
    padded to KP = a multiple of four with zero codes (one v_mfma_scale_f32_16x16x128 covers four taps x 32 channels)."""
    w = np.ascontiguousarray(w, np.float32)
    N, taps, K = w.shape
    assert N == 32 and K == 32, (N, K)
    kp = (taps + 3) // 4 * 4
    hi, lo = split_hi_lo(w)
    codes, scales = [], []
    for part in (lo, hi):
        c, sb = quantize(part, 32, W_RULE)                           # [N][taps][16], [N][taps][1]
        cp = np.zeros((kp, N, 16), np.uint8)
        sp = np.ones((kp, N), np.uint8)
        cp[:taps] = c.transpose(1, 0, 2)
        sp[:taps] = sb[..., 0].T
        codes.append(cp.reshape(-1))
        scales.append(sp.reshape(-1))
    return np.concatenate(codes + scales)


def pair_weight_planes_dequant(blob: np.ndarray, taps: int):
    """inverse of pack_pair_weight_planes (tests): -> (Q(wl), Q(wh)) as fp32 [32][taps][32]."""
    kp = (taps + 3) // 4 * 4
    nc, ns = kp * 32 * 16, kp * 32
    res = []
    for i in range(2):
        c = blob[i * nc:(i + 1) * nc].reshape(kp, 32, 16)[:taps].transpose(1, 0, 2)
        sb = blob[2 * nc + i * ns:2 * nc + (i + 1) * ns].reshape(kp, 32)[:taps].T[..., None]
        res.append(dequantize(np.ascontiguousarray(c), np.ascontiguousarray(sb), 32))
    return res[0], res[1]


def pack_c64_weight_planes(w: np.ndarray) -> np.ndarray:
    """GEMM-layout weight [64][taps][64] fp32 of a C = 64 conv -> the fp4 planes of conv_c64_mx_kernel (ev_conv64_mx.h):
         codes  [plane: Q(w - fp16(w)), Q(fp16(w))][output-channel half][KP taps][32 co][32 B]
         scales [plane][half][KP][32 co][2]          (one per (output channel, tap, 32 input channels) = per MFMA k-block)
    taps padded to KP = a multiple of two with zero codes (one v_mfma_scale_f32_16x16x128 covers two taps x 64 channels)."""
    w = np.ascontiguousarray(w, np.float32)
    N, taps, K = w.shape
    assert N == 64 and K == 64, (N, K)
    kp = (taps + 1) // 2 * 2
    hi, lo = split_hi_lo(w)
    codes, scales = [], []
    for part in (lo, hi):
        c, sb = quantize(part, 32, W_RULE)                           # [64][taps][32 B], [64][taps][2]
        cp = np.zeros((2, kp, 32, 32), np.uint8)
        sp = np.ones((2, kp, 32, 2), np.uint8)
        cp[:, :taps] = c.reshape(2, 32, taps, 32).transpose(0, 2, 1, 3)
        sp[:, :taps] = sb.reshape(2, 32, taps, 2).transpose(0, 2, 1, 3)
        codes.append(cp.reshape(-1))
        scales.append(sp.reshape(-1))
    return np.concatenate(codes + scales)


def c64_weight_planes_dequant(blob: np.ndarray, taps: int):
    """inverse of pack_c64_weight_planes (tests): -> (Q(wl), Q(wh)) as fp32 [64][taps][64]."""
    kp = (taps + 1) // 2 * 2
    nc, ns = 2 * kp * 32 * 32, 2 * kp * 32 * 2
    res = []
    for i in range(2):
        c = blob[i * nc:(i + 1) * nc].reshape(2, kp, 32, 32)[:, :taps].transpose(0, 2, 1, 3).reshape(64, taps, 32)
        sb = blob[2 * nc + i * ns:2 * nc + (i + 1) * ns].reshape(2, kp, 32, 2)[:, :taps].transpose(0, 2, 1, 3).reshape(64, taps, 2)
        res.append(dequantize(np.ascontiguousarray(c), np.ascontiguousarray(sb), 32))
    return res[0], res[1]
