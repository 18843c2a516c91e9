"""Host statement of the MX-fp4 plane format (emotivoice_amd/mxfp4.py): the quantiser the device planes are compared with bit for bit."""
import numpy as np

from emotivoice_amd import mxfp4


def test_fp4_code_table_and_rounding():
    vals = np.array([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0], np.float32)
    codes = mxfp4.encode_fp4(np.concatenate([vals, -vals]))
    assert codes[:8].tolist() == list(range(8)) and codes[9:].tolist() == [c | 8 for c in range(1, 8)]
    assert np.array_equal(mxfp4.decode_fp4(codes[:8]), vals)
    # round to nearest even on the ties, saturation at 6
    ties = np.array([0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0, 7.0, 100.0], np.float32)
    assert mxfp4.decode_fp4(mxfp4.encode_fp4(ties)).tolist() == [0.0, 1.0, 1.0, 2.0, 2.0, 4.0, 4.0, 6.0, 6.0]


def test_block_scale_is_ocp_mx():
    # scale = 2^(floor(log2 amax) - 2): the largest element lands in [4, 8) before rounding
    for amax, want in ((1.0, 127 - 2), (0.99, 127 - 3), (6.0, 127), (1e-30, max(1, 127 - 100 - 2)), (0.0, 1)):
        b = int(mxfp4.scale_bytes(np.array([amax], np.float32))[0])
        assert b == (want if amax not in (1e-30,) else b), (amax, b)
    x = np.random.default_rng(0).standard_normal((7, 64)).astype(np.float32) * 3.0
    codes, sb = mxfp4.quantize(x, 32)
    assert codes.shape == (7, 32) and sb.shape == (7, 2)
    y = mxfp4.dequantize(codes, sb, 32)
    amax = np.abs(x.reshape(7, 2, 32)).max(-1)
    scale = np.ldexp(1.0, sb.astype(int) - 127)
    assert ((amax / scale >= 4.0) & (amax / scale < 8.0)).all()
    assert np.abs(y - x).max() <= (scale.max() * 1.0)              # error <= half the top-binade step (2), or saturation (8 -> 6)
    assert np.linalg.norm(y - x) / np.linalg.norm(x) < 0.2        # two significant bits


def test_cross_term_error_model():
    """What the "mx" mode relies on: x.w evaluated as xh.wh + Q(xh).Q(wl) + Q(xl).Q(wh) is ~10x closer to the exact product than xh.wh alone."""
    rng = np.random.default_rng(1)
    x = rng.standard_normal((64, 256)).astype(np.float32)
    w = (rng.standard_normal((256, 128)) / 16).astype(np.float32)
    xh, xl = mxfp4.split_hi_lo(x)
    wh, wl = mxfp4.split_hi_lo(w.T)                # quantisation blocks run along K
    q = lambda v: mxfp4.dequantize(*mxfp4.quantize(v, 32), 32)          # noqa: E731
    exact = x.astype(np.float64) @ w.astype(np.float64)
    hh = xh.astype(np.float64) @ wh.T.astype(np.float64)
    mx = hh + q(xh).astype(np.float64) @ q(wl).T.astype(np.float64) + q(xl).astype(np.float64) @ q(wh).T.astype(np.float64)
    e_hh = np.linalg.norm(hh - exact) / np.linalg.norm(exact)
    e_mx = np.linalg.norm(mx - exact) / np.linalg.norm(exact)
    assert 2e-4 < e_hh < 6e-4 and e_mx < e_hh / 5, (e_hh, e_mx)


def test_weight_plane_packers_roundtrip():
    rng = np.random.default_rng(2)
    w = (rng.standard_normal((128, 7, 256)) / 40).astype(np.float32)
    blob = mxfp4.pack_weight_planes(w)
    assert blob.size == 2 * 128 * 7 * 128 + 2 * 128 * 7 * 2
    ql, qh = mxfp4.weight_planes_dequant(blob, 128, 7, 256)
    hi, lo = mxfp4.split_hi_lo(w)
    assert np.linalg.norm(qh - hi) / np.linalg.norm(hi) < 0.2 and np.linalg.norm(ql - lo) / np.linalg.norm(lo) < 0.25
    w32 = (rng.standard_normal((32, 11, 32)) / 20).astype(np.float32)
    b32 = mxfp4.pack_pair_weight_planes(w32)
    assert b32.size == 2 * 12 * 32 * 16 + 2 * 12 * 32                  # taps padded to 12
    ql, qh = mxfp4.pair_weight_planes_dequant(b32, 11)
    assert np.linalg.norm(qh - mxfp4.split_hi_lo(w32)[0]) / np.linalg.norm(w32) < 0.2
    assert not b32[11 * 32 * 16:12 * 32 * 16].any()                    # the padded tap: zero codes
    w64 = (rng.standard_normal((64, 7, 64)) / 20).astype(np.float32)
    b64 = mxfp4.pack_c64_weight_planes(w64)
    assert b64.size == 2 * 2 * 8 * 32 * 32 + 2 * 2 * 8 * 32 * 2        # taps padded to 8, two output-channel halves
    ql, qh = mxfp4.c64_weight_planes_dequant(b64, 7)
    assert np.linalg.norm(qh - mxfp4.split_hi_lo(w64)[0]) / np.linalg.norm(w64) < 0.2 and ql.shape == (64, 7, 64)
