"""Per-kernel parity tests (MI355X): each gfx950 kernel, launched through the C ABI entry points of
include/evhip_ops.h, against the plain fp32 torch op it replaces."""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
F = torch.nn.functional


@pytest.fixture(scope="module")
def lib():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from emotivoice_amd import _ffi
    return _ffi.lib()


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


PAD = 64


def _padded(rows, cols, dtype, fill=None):
    """[PAD + rows + PAD, cols] device tensor; returns (full, view of the logical rows)."""
    full = torch.randn(rows + 2 * PAD, cols, device="cuda", dtype=torch.float32).to(dtype)
    return full, full[PAD:PAD + rows]


def _run_conv(lib, x_view, w_torch, bias, *, dtype, taps, dil, center, row_valid=None, valid_shift=0, act=0, act_slope=0.0,
              pro_slope=None, res=None, scale=1.0, acc32=None, post_slope=None, seq_bias=None, row_seq=None,
              want16=True, want32=True, before_post=False, add16=None, dbg=0):
    from emotivoice_amd import _ffi
    M, K = x_view.shape
    N = w_torch.shape[0]
    tdt = torch.float16 if dtype == 0 else torch.float32
    wg = w_torch.permute(0, 2, 1).contiguous().to(tdt)           # [N][taps][K]
    d = _ffi.ev_conv_gemm_desc()
    d.dtype = dtype
    d.A, d.lda = x_view.data_ptr(), x_view.stride(0)
    d.W = wg.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.M, d.N, d.K, d.taps, d.dil, d.center = M, N, K, taps, dil, center
    if row_valid is not None:
        d.row_valid, d.valid_shift = row_valid.data_ptr(), valid_shift
    if seq_bias is not None:
        d.row_seq, d.seq_bias, d.ld_seq_bias = row_seq.data_ptr(), seq_bias.data_ptr(), seq_bias.stride(0)
    d.act, d.act_slope = act, act_slope
    if pro_slope is not None:
        d.pro_lrelu, d.pro_slope = 1, pro_slope
    if res is not None:
        d.res, d.res_dtype, d.ldres = res.data_ptr(), (0 if res.dtype == torch.float16 else 1), res.stride(0)
    d.out_scale = scale
    if acc32 is not None:
        d.acc32, d.ldacc = acc32.data_ptr(), acc32.stride(0)
    if add16 is not None:
        d.add16_a, d.add16_b, d.ldadd = add16[0].data_ptr(), add16[1].data_ptr(), add16[0].stride(0)
    if post_slope is not None:
        d.post_lrelu, d.post_slope = 1, post_slope
    out16 = torch.full((M, N), 7.0, device="cuda", dtype=torch.float16) if want16 else None
    out32 = torch.full((M, N), 7.0, device="cuda", dtype=torch.float32) if want32 else None
    d.out16 = out16.data_ptr() if want16 else None
    d.out32 = out32.data_ptr() if want32 else None
    d.ldo = N
    d.out32_before_post = 1 if before_post else 0
    d.reserved0 = dbg            # tuning / A-B switches of the launcher (bit 2: 4-wave kernel instead of the phased 8-wave one)
    torch.cuda.synchronize()
    rc = lib.ev_op_conv_gemm(C.byref(d), None)
    assert rc == 0
    torch.cuda.synchronize()
    return out16, out32


def _ref_conv(x, w, bias, dil, center, taps):
    """x [M,K] fp32 (already rounded to the operand dtype), w [N,K,taps] -> [M,N] fp32 on CPU."""
    xc = x.float().cpu().t().unsqueeze(0)
    y = F.conv1d(xc, w.float().cpu(), bias.float().cpu() if bias is not None else None, dilation=dil, padding=center * dil)
    # "same" geometry only when the conv is centred
    assert center * 2 == taps - 1
    return y.squeeze(0).t()


def _lrelu(x, s):
    return torch.where(x > 0, x, x * s)


CONV_CASES = [
    # name, dtype, M, K, N, taps, dil
    ("f16_linear_384", 0, 512, 384, 384, 1, 1),
    ("f16_ffn1_k3", 0, 256, 384, 1536, 3, 1),
    ("f16_ffn2_k3", 0, 256, 1536, 384, 3, 1),
    ("f16_c32_k11_d5", 0, 1024, 32, 32, 11, 5),
    ("f16_c64_k7_d3", 0, 512, 64, 64, 7, 3),
    ("f16_c128_k3_d1", 0, 512, 128, 128, 3, 1),
    ("f16_c256_k11_d1", 0, 256, 256, 256, 11, 1),
    ("f16_n96", 0, 256, 384, 96, 1, 1),
    ("f32_linear_384", 1, 256, 384, 384, 1, 1),
    ("f32_qkv", 1, 256, 384, 1152, 1, 1),
    ("f32_conv_k3", 1, 512, 384, 384, 3, 1),
    ("f32_n32_k7", 1, 256, 32, 32, 7, 1),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_gemm_plain(lib, case):
    name, dtype, M, K, N, taps, dil = case
    torch.manual_seed(hash(name) % 1000)
    tdt = torch.float16 if dtype == 0 else torch.float32
    full, x = _padded(M, K, tdt)
    w = (torch.randn(N, K, taps, device="cuda") / math.sqrt(K * taps)).to(tdt)
    bias = torch.randn(N, device="cuda")
    center = (taps - 1) // 2
    # zero halo: the rows outside [0, M) that the conv reads must be zero for the torch 'same' reference
    full[:PAD] = 0
    full[PAD + M:] = 0
    o16, o32 = _run_conv(lib, x, w, bias, dtype=dtype, taps=taps, dil=dil, center=center)
    ref = _ref_conv(x, w, bias, dil, center, taps)
    tol32 = 2e-5 if dtype == 0 else 1e-5
    assert _rel(o32.cpu(), ref) < tol32, name
    assert _rel(o16.float().cpu(), ref) < 6e-4, name
    # transpose detection: the error must not be explained by a swapped output layout
    assert o32.shape == (M, N)


def test_conv_gemm_full_epilogue(lib):
    """ResBlock-style call: leaky-relu prologue, bias, residual, scale, fp32 accumulate-in, post leaky-relu,
    row mask with shift, out32 taken before the post activation (models/hifigan/models.py:50-57,121-127)."""
    torch.manual_seed(3)
    M, C_, taps, dil = 1024, 64, 7, 3
    full, x = _padded(M, C_, torch.float16)
    valid = torch.ones(M // 8, dtype=torch.uint8, device="cuda")
    valid[:4] = 0
    valid[60:70] = 0
    valid[-4:] = 0
    vrow = valid.repeat_interleave(8).bool()
    full[:PAD] = 0
    full[PAD + M:] = 0
    x[~vrow] = 0
    w = (torch.randn(C_, C_, taps, device="cuda") / math.sqrt(C_ * taps)).half()
    bias = torch.randn(C_, device="cuda")
    res = torch.randn(M, C_, device="cuda").half()
    acc = torch.randn(M, C_, device="cuda")
    o16, o32 = _run_conv(lib, x, w, bias, dtype=0, taps=taps, dil=dil, center=3, row_valid=valid, valid_shift=3,
                         pro_slope=0.1, res=res, scale=1.0 / 3.0, acc32=acc, post_slope=0.01, before_post=True)
    xin = _lrelu(x.float(), 0.1).half().float()     # prologue is applied in fp16 on the staged tile
    ref = _ref_conv(xin, w, bias, dil, 3, taps)
    ref = (ref + res.float().cpu()) * (1.0 / 3.0) + acc.cpu()
    ref_post = _lrelu(ref, 0.01)
    m = vrow.cpu()
    ref[~m] = 0
    ref_post[~m] = 0
    assert _rel(o32.cpu(), ref) < 2e-5
    assert _rel(o16.float().cpu(), ref_post) < 6e-4
    assert float(o32[~vrow].abs().max()) == 0.0 and float(o16[~vrow].float().abs().max()) == 0.0


@pytest.mark.parametrize("C_,taps,outs,M", [(64, 7, "16", 1024), (64, 7, "32", 1024), (128, 3, "16", 1024), (128, 11, "both", 1024),
                                             (32, 3, "16", 1024),
                                             # >= 256 tiles of 256 x 128: the paired-step kernel (two taps per barrier)
                                             (128, 3, "16", 65536), (128, 7, "16", 65536), (128, 11, "16", 65536), (256, 7, "16", 32768)])
def test_conv_gemm_mrf16_and_static_outputs(lib, C_, taps, outs, M):
    """The specialised epilogues of the frame-rate path: one fp16 output / one fp32 output (compile-time store count), fp16
    residual, and the MRF sum with the first two ResBlock branches as fp16 addends (add16_a/b) instead of an fp32 accumulator
    (models/hifigan/models.py:121-127).  Same reference as the generic epilogue."""
    torch.manual_seed(7 + C_ + taps)
    dil = 1 if M == 1024 else (taps - 1) // 2      # the big cases also use the ResBlock dilations 1 / 3 / 5
    full, x = _padded(M, C_, torch.float16)
    valid = torch.ones(M // 8, dtype=torch.uint8, device="cuda")
    valid[:3] = 0
    valid[50:53] = 0
    valid[M // 16:M // 16 + 4] = 0
    valid[-2:] = 0
    vrow = valid.repeat_interleave(8).bool()
    full[:PAD] = 0
    full[PAD + M:] = 0
    x[~vrow] = 0
    w = (torch.randn(C_, C_, taps, device="cuda") / math.sqrt(C_ * taps)).half()
    bias = torch.randn(C_, device="cuda")
    res = torch.randn(M, C_, device="cuda").half()
    a16, b16 = torch.randn(M, C_, device="cuda").half(), torch.randn(M, C_, device="cuda").half()
    c = (taps - 1) // 2
    m = vrow.cpu()
    base = _ref_conv(x.float(), w, bias, dil, c, taps)
    # (a) conv2 of a non-final pair: residual only
    ref = base + res.float().cpu()
    ref[~m] = 0
    o16, o32 = _run_conv(lib, x, w, bias, dtype=0, taps=taps, dil=dil, center=c, row_valid=valid, valid_shift=3, res=res,
                         want16=outs != "32", want32=outs != "16")
    if o32 is not None:
        assert _rel(o32.cpu(), ref) < 2e-5
        assert float(o32[~vrow].abs().max()) == 0.0
    if o16 is not None:
        assert _rel(o16.float().cpu(), ref) < 6e-4
        assert float(o16[~vrow].float().abs().max()) == 0.0
    # (b) conv2 of the last pair of the last ResBlock: scale, two fp16 addends, post leaky-relu
    ref = (base + res.float().cpu()) * (1.0 / 3.0) + a16.float().cpu() + b16.float().cpu()
    ref_post = _lrelu(ref, 0.1)
    ref[~m] = 0
    ref_post[~m] = 0
    o16, o32 = _run_conv(lib, x, w, bias, dtype=0, taps=taps, dil=dil, center=c, row_valid=valid, valid_shift=3, res=res,
                         scale=1.0 / 3.0, add16=(a16, b16), post_slope=0.1, want16=True, want32=outs == "both", before_post=True)
    assert _rel(o16.float().cpu(), ref_post) < 6e-4
    assert float(o16[~vrow].float().abs().max()) == 0.0
    if o32 is not None:
        assert _rel(o32.cpu(), ref) < 2e-5
    # (c) conv1: leaky-relu prologue + activation, fp16 output only
    xin = _lrelu(x.float(), 0.1).half().float()
    ref = _lrelu(_ref_conv(xin, w, bias, dil, c, taps), 0.1)
    ref[~m] = 0
    o16, _ = _run_conv(lib, x, w, bias, dtype=0, taps=taps, dil=dil, center=c, row_valid=valid, valid_shift=3, pro_slope=0.1,
                       act=3, act_slope=0.1, want16=True, want32=False)
    assert _rel(o16.float().cpu(), ref) < 6e-4
    assert float(o16[~vrow].float().abs().max()) == 0.0


PHASED_CASES = [
    # name, M, K, N, taps, dil, epilogue class
    ("c128_k11_d5_pro", 65536, 128, 128, 11, 5, "pro"),
    ("c128_k7_d1_res", 65536, 128, 128, 7, 1, "res"),
    ("c128_k3_d1_pro", 65536, 128, 128, 3, 1, "pro"),
    ("c128_k11_d1_mrf", 65536, 128, 128, 11, 1, "mrf"),
    ("c256_k7_d3_res_masked", 33024, 256, 256, 7, 3, "res_masked"),
    ("c64_k11_d5_pro", 131072, 64, 64, 11, 5, "pro"),
    ("c64_k11_d1_mrf", 131072, 64, 64, 11, 1, "mrf"),
    ("ffn1_gelu", 8192, 384, 1536, 3, 1, "gelu"),
    ("ffn2_res", 22016, 1536, 384, 3, 1, "res"),
    ("ffn2_res32_stream", 22016, 1536, 384, 3, 1, "res32"),      # the mel decoder's fp32 residual stream (in place)
    ("conv_pre_k7", 33024, 96, 512, 7, 1, "plain"),
]


@pytest.mark.parametrize("case", PHASED_CASES, ids=[c[0] for c in PHASED_CASES])
def test_phased_kernel_matches_4_wave_kernel_bitwise(lib, case):
    """Large launches of the frame-rate path take conv_gemm_phased_kernel (8 waves, LDS-DMA staging, alternating matrix / load
    phases); it must reproduce the 4-wave kernel bit for bit -- which kernel runs depends on the batch's row count, and the engine
    promises batch-invariant results -- and, on its own, the fp32 torch conv (models/hifigan/models.py:50-57,116-128)."""
    name, M, K, N, taps, dil, cls = case
    torch.manual_seed(len(name) * 7 + taps)
    full, x = _padded(M, K, torch.float16)
    full[:PAD] = 0
    full[PAD + M:] = 0
    w = (torch.randn(N, K, taps, device="cuda") / math.sqrt(K * taps)).half()
    bias = torch.randn(N, device="cuda")
    center = (taps - 1) // 2
    kw = dict(dtype=0, taps=taps, dil=dil, center=center, want32=False)
    res = torch.randn(M, N, device="cuda").half() if cls in ("res", "res_masked", "mrf") else None
    if cls == "res32":
        res = torch.randn(M, N, device="cuda")
        kw.update(res=res, want16=False, want32=True)
    valid = None
    if cls == "pro":
        kw.update(pro_slope=0.1, act=3, act_slope=0.1)
    elif cls == "gelu":
        kw.update(act=2)
    elif cls in ("res", "res_masked"):
        kw.update(res=res)
        if cls == "res_masked":
            valid = (torch.rand(M // 4, device="cuda") > 0.2).to(torch.uint8)          # one byte per 4 rows (valid_shift = 2)
            kw.update(row_valid=valid, valid_shift=2)
    elif cls == "mrf":
        ma, mb = torch.randn(M, N, device="cuda").half(), torch.randn(M, N, device="cuda").half()
        kw.update(res=res, add16=(ma, mb), scale=1.0 / 3.0, post_slope=0.1)
    o_new = [o for o in _run_conv(lib, x, w, bias, **kw) if o is not None][0]
    o_old = [o for o in _run_conv(lib, x, w, bias, dbg=4, **kw) if o is not None][0]
    bits = torch.int16 if o_new.dtype == torch.float16 else torch.int32
    assert torch.equal(o_new.view(bits), o_old.view(bits)), name
    # and against torch on a slice of rows (the whole tensor would take the CPU a while)
    rows = slice(M // 2 - 512, M // 2 + 512)
    xs = full[PAD + rows.start - PAD:PAD + rows.stop + PAD].float()
    if cls == "pro":
        xs = _lrelu(xs, 0.1)
    ref = F.conv1d(xs.cpu().t().unsqueeze(0), w.float().cpu(), bias.cpu(), dilation=dil, padding=0)
    ref = ref.squeeze(0).t()[PAD - center * dil:PAD - center * dil + 1024]
    if cls == "pro":
        ref = _lrelu(ref, 0.1)
    elif cls == "gelu":
        ref = F.gelu(ref)
    elif cls in ("res", "res_masked", "res32"):
        ref = ref + res[rows].float().cpu()
        if valid is not None:
            ref = ref * valid[rows.start // 4:rows.stop // 4].repeat_interleave(4).float().cpu()[:, None]
    elif cls == "mrf":
        ref = _lrelu((ref + res[rows].float().cpu()) / 3.0 + ma[rows].float().cpu() + mb[rows].float().cpu(), 0.1)
    assert _rel(o_new[rows].float().cpu(), ref) < 8e-4, name


def test_conv_gemm_activations_and_seq_bias(lib):
    torch.manual_seed(4)
    M, K, N = 256, 384, 384
    full, x = _padded(M, K, torch.float32)
    full[:PAD] = 0
    full[PAD + M:] = 0
    w = torch.randn(N, K, 1, device="cuda") / math.sqrt(K)
    bias = torch.randn(N, device="cuda")
    row_seq = torch.randint(0, 3, (M,), device="cuda", dtype=torch.int32)
    sb = torch.randn(3, N, device="cuda")
    for act, fn in ((1, torch.relu), (2, F.gelu), (4, torch.tanh), (3, lambda t: _lrelu(t, 0.1))):
        _, o32 = _run_conv(lib, x, w, bias, dtype=1, taps=1, dil=1, center=0, act=act, act_slope=0.1, seq_bias=sb,
                           row_seq=row_seq, want16=False)
        ref = fn(_ref_conv(x, w, bias, 1, 0, 1)) + sb.cpu()[row_seq.cpu().long()]
        assert _rel(o32.cpu(), ref) < 1e-5, act


@pytest.mark.parametrize("cin,s", [(64, 2), (128, 2), (256, 8), (512, 8)])
def test_conv_transpose_polyphase(lib, cin, s):
    """ConvTranspose1d(k = 2s, stride s, pad s/2) as the packed 3-tap conv (models/hifigan/models.py:99-103,119)."""
    from emotivoice_amd.packer import _convT_to_gemm
    torch.manual_seed(5)
    cout, M = cin // 2, 256
    full, x = _padded(M, cin, torch.float16)
    full[:PAD] = 0
    full[PAD + M:] = 0
    wt = (torch.randn(cin, cout, 2 * s) / math.sqrt(cin * 2)).half().float()
    bias = torch.randn(cout)
    wg = torch.from_numpy(_convT_to_gemm(wt.numpy(), s)).cuda()              # [s*cout][3][cin]
    w_as_conv = wg.permute(0, 2, 1).contiguous()                             # torch layout [N, K, taps]
    o16, o32 = _run_conv(lib, x, w_as_conv, bias.repeat(s).cuda(), dtype=0, taps=3, dil=1, center=1)
    got = o32.cpu().reshape(M * s, cout)
    ref = F.conv_transpose1d(x.float().cpu().t().unsqueeze(0), wt, bias, stride=s, padding=s // 2).squeeze(0).t()
    assert ref.shape == got.shape
    assert _rel(got, ref) < 2e-5


def test_layernorm_and_head(lib):
    torch.manual_seed(6)
    rows, Cc = 300, 384
    x = torch.randn(rows, Cc, device="cuda") * 3 + 0.5
    g, b = torch.randn(Cc, device="cuda"), torch.randn(Cc, device="cuda")
    w = torch.randn(Cc, device="cuda")
    valid = torch.ones(rows, dtype=torch.uint8, device="cuda")
    valid[7] = 0
    o16 = torch.empty(rows, Cc, device="cuda", dtype=torch.float16)
    o32 = torch.empty(rows, Cc, device="cuda")
    dot = torch.empty(rows, device="cuda")
    rc = lib.ev_op_layernorm(x.data_ptr(), rows, Cc, g.data_ptr(), b.data_ptr(), 1e-12, valid.data_ptr(), o16.data_ptr(),
                             o32.data_ptr(), w.data_ptr(), 0.25, dot.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    ref = F.layer_norm(x.cpu(), (Cc,), g.cpu(), b.cpu(), 1e-12)
    ref[7] = 0
    refdot = ref @ w.cpu() + 0.25
    refdot[7] = 0
    assert _rel(o32.cpu(), ref) < 2e-6
    assert _rel(o16.float().cpu(), ref) < 5e-4
    assert _rel(dot.cpu(), refdot) < 5e-6


@pytest.mark.parametrize("is_f16", [0, 1, 2])
def test_attention_ragged(lib, is_f16):
    """Attention restricted to each utterance's rows == per-utterance B=1 attention (modules/encoder.py:72-109)."""
    torch.manual_seed(7)
    Cc, H = 384, 8
    lens = [70, 1, 130, 64, 300, 129, 1024]
    offs, rows = [], 4
    for n in lens:
        offs.append(rows)
        rows += n + 4
    dt = torch.float16 if is_f16 == 1 else torch.float32          # 2: fp32 rows through the split-precision kernel (3 fp16 MFMAs per product)
    qkv = torch.randn(rows, 3 * Cc, device="cuda").to(dt)
    out = torch.zeros(rows, Cc, device="cuda", dtype=dt)
    so = torch.tensor(offs, dtype=torch.int32, device="cuda")
    sl = torch.tensor(lens, dtype=torch.int32, device="cuda")
    rc = lib.ev_op_attention(qkv.data_ptr(), is_f16, Cc, H, so.data_ptr(), sl.data_ptr(), len(lens), max(lens), out.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    for o, n in zip(offs, lens):
        blk = qkv[o:o + n].float().cpu()
        q, k, v = [t.view(n, H, 48).transpose(0, 1) for t in blk.split(Cc, dim=1)]
        att = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(48), dim=-1) @ v
        ref = att.transpose(0, 1).reshape(n, Cc)
        assert _rel(out[o:o + n].float().cpu(), ref) < (1e-3 if is_f16 == 1 else 2e-6), (n, is_f16)


@pytest.mark.parametrize("k,dil,accmode,Cc", [(3, 1, "acc32", 32), (3, 5, "add16", 32), (7, 3, "acc32", 32), (7, 1, "add16", 32),
                                               (11, 1, "none", 32), (11, 5, "acc32", 32), (11, 3, "add16", 32),
                                               (3, 1, "none", 64), (3, 3, "add16", 64), (3, 5, "acc32", 64)])
def test_fused_resblock_pair(lib, k, dil, accmode, Cc):
    """conv1(dil) -> leaky-relu -> conv2 + residual in one persistent kernel == the two torch convs of
    models/hifigan/models.py:50-57, incl. sequence-edge masking of the intermediate and the MRF epilogue."""
    from emotivoice_amd import _ffi
    torch.manual_seed(100 + k + dil)
    M = 5 * 256
    full, x = _padded(M, Cc, torch.float16)
    valid = torch.ones(M // 16, dtype=torch.uint8, device="cuda")
    valid[:2] = 0
    valid[30:34] = 0
    valid[-3:] = 0
    vrow = valid.repeat_interleave(16).bool()
    full[:PAD] = 0
    full[PAD + M:] = 0
    x[~vrow] = 0
    w1 = (torch.randn(Cc, Cc, k, device="cuda") / math.sqrt(Cc * k)).half()
    w2 = (torch.randn(Cc, Cc, k, device="cuda") / math.sqrt(Cc * k)).half()
    b1, b2 = torch.randn(Cc, device="cuda") * 0.1, torch.randn(Cc, device="cuda") * 0.1
    acc = torch.randn(M, Cc, device="cuda")
    a16, b16 = torch.randn(M, Cc, device="cuda").half(), torch.randn(M, Cc, device="cuda").half()
    w1g, w2g = w1.permute(0, 2, 1).contiguous(), w2.permute(0, 2, 1).contiguous()
    out16 = torch.full((M, Cc), 7.0, device="cuda", dtype=torch.float16)
    out32 = torch.full((M, Cc), 7.0, device="cuda")
    d = _ffi.ev_res_pair_desc()
    d.x, d.ldx, d.w1, d.b1, d.w2, d.M, d.k, d.dil = x.data_ptr(), Cc, w1g.data_ptr(), b1.data_ptr(), w2g.data_ptr(), M, k, dil
    e = d.epi
    e.bias, e.res, e.res_dtype, e.ldres = b2.data_ptr(), x.data_ptr(), 0, Cc
    e.row_valid, e.valid_shift = valid.data_ptr(), 4
    e.out_scale = 1.0 / 3.0
    if accmode == "acc32":
        e.acc32, e.ldacc = acc.data_ptr(), Cc
        addend = acc.cpu()
    elif accmode == "add16":
        e.add16_a, e.add16_b, e.ldadd = a16.data_ptr(), b16.data_ptr(), Cc
        addend = a16.float().cpu() + b16.float().cpu()
    else:
        addend = 0.0
    e.post_lrelu, e.post_slope, e.out16, e.out32, e.ldo, e.out32_before_post = 1, 0.01, out16.data_ptr(), out32.data_ptr(), Cc, 1
    torch.cuda.synchronize()
    assert (lib.ev_op_resblock_pair_c32 if Cc == 32 else lib.ev_op_resblock_pair_c64)(C.byref(d), None) == 0
    torch.cuda.synchronize()
    xin = _lrelu(x.float(), 0.1).half().float()
    xt = _lrelu(_ref_conv(xin, w1, b1, dil, (k - 1) // 2, k), 0.1)
    xt[~vrow.cpu()] = 0
    xt = xt.half().float()                                   # the intermediate lives in LDS as fp16
    ref = (_ref_conv(xt, w2, b2, 1, (k - 1) // 2, k) + x.float().cpu()) * (1.0 / 3.0) + addend
    ref_post = _lrelu(ref, 0.01)
    ref[~vrow.cpu()] = 0
    ref_post[~vrow.cpu()] = 0
    # (an fp16 ulp of the LDS-resident intermediate may round differently than the torch reference: 1e-4, not 2e-5)
    assert _rel(out32.cpu(), ref) < 1e-4, (k, dil)
    assert _rel(out16.float().cpu(), ref_post) < 6e-4
    assert float(out32[~vrow].abs().max()) == 0.0


@pytest.mark.parametrize("M,K,N,taps", [(256, 384, 384, 1), (512, 384, 1536, 3), (256, 1536, 384, 3), (256, 384, 1152, 1)])
def test_split_precision_gemm_matches_fp32(lib, M, K, N, taps):
    """fp32 activations x (hi, lo) fp16 weight split, 3 fp16 MFMAs per product: fp32-level agreement with an fp64 reference
    (the token-rate path must keep durations bit-exact, so this is held to the same 1e-5 bound as the exact fp32 MFMA kernel,
    and the error is reported next to the exact kernel's)."""
    from emotivoice_amd import _ffi
    torch.manual_seed(M + K + N)
    full, x = _padded(M, K, torch.float32)
    full[:PAD] = 0
    full[PAD + M:] = 0
    w = torch.randn(N, K, taps, device="cuda") / math.sqrt(K * taps)
    bias = torch.randn(N, device="cuda")
    wg = w.permute(0, 2, 1).contiguous()
    hi = wg.half()
    lo = ((wg - hi.float()) * 2048.0).half()
    ref = F.conv1d(x.double().cpu().t().unsqueeze(0), w.double().cpu(), bias.double().cpu(), padding=(taps - 1) // 2).squeeze(0).t()
    errs = {}
    for dtype in (2, 1):
        d = _ffi.ev_conv_gemm_desc()
        d.dtype, d.A, d.lda = dtype, x.data_ptr(), K
        if dtype == 2:
            d.W, d.W_lo = hi.data_ptr(), lo.data_ptr()
        else:
            d.W = wg.data_ptr()
        d.bias, d.M, d.N, d.K, d.taps, d.dil, d.center, d.out_scale = bias.data_ptr(), M, N, K, taps, 1, (taps - 1) // 2, 1.0
        out = torch.full((M, N), 7.0, device="cuda")
        d.out32, d.ldo = out.data_ptr(), N
        torch.cuda.synchronize()
        assert lib.ev_op_conv_gemm(C.byref(d), None) == 0
        torch.cuda.synchronize()
        errs[dtype] = _rel(out.cpu().double(), ref)
    assert errs[2] < 1e-5 and errs[1] < 1e-5, errs
    assert errs[2] < 3e-6, errs          # within a small factor of fp32 rounding


@pytest.mark.parametrize("M,K,N,taps,S,form", [(256, 1536, 384, 3, 4, "res"), (768, 384, 384, 3, 3, "relu"), (1024, 384, 384, 3, 3, "plain"), (512, 1536, 384, 3, 2, "full"),
                                               (256, 384, 64, 1, 12, "plain")])
def test_split_k_token_rate_gemm(lib, M, K, N, taps, S, form):
    """ev_conv_gemm_desc.ksplit: the K-chunks of a split-precision (hi / lo weights) GEMM cut into S ranges, one block of the 128 x 64-tile kernel per (tile, range)
    writing fp32 partial sums, a second kernel reducing them in range order and applying the epilogue (the token-rate conv-FFN / predictor convs: ev_config.
    token_splitk).  Held to the same bound against fp64 as the one-pass kernel, equal to it to fp32 rounding, bit-identical between two launches and to the
    same rows computed inside a larger M (batch invariance), invalid rows exact zeros; inconsistent calls are refused."""
    from emotivoice_amd import _ffi
    torch.manual_seed(M + K + N + S)
    Mbig = M + 512
    full, xbig = _padded(Mbig, K, torch.float32)
    full[:PAD] = 0
    full[PAD + M:PAD + M + 128] = 0                 # a gap behind the first M rows: they see the same neighbourhood alone and inside the larger launch
    full[PAD + Mbig:] = 0
    x = xbig[:M]
    w = torch.randn(N, K, taps, device="cuda") / math.sqrt(K * taps)
    bias = torch.randn(N, device="cuda")
    wg = w.permute(0, 2, 1).contiguous()
    hi = wg.half()
    lo = ((wg - hi.float()) * 2048.0).half()
    res = torch.randn(Mbig, N, device="cuda")
    acc = torch.randn(Mbig, N, device="cuda")
    valid = torch.ones(Mbig // 8, dtype=torch.uint8, device="cuda")
    valid[1] = 0
    valid[M // 8:(M + 128) // 8] = 0
    ws = torch.zeros(S * Mbig * N + 4, dtype=torch.float32, device="cuda")

    def run(rows, ksplit, scratch=True):
        d = _ffi.ev_conv_gemm_desc()
        d.dtype, d.A, d.lda, d.W, d.W_lo = 2, xbig.data_ptr(), K, hi.data_ptr(), lo.data_ptr()
        d.bias, d.M, d.N, d.K, d.taps, d.dil, d.center, d.out_scale = bias.data_ptr(), rows, N, K, taps, 1, (taps - 1) // 2, 1.0
        d.row_valid, d.valid_shift = valid.data_ptr(), 3
        if form == "relu":
            d.act = 1
        if form in ("res", "full"):
            d.res, d.res_dtype, d.ldres = res.data_ptr(), 1, N
        if form == "full":
            d.act, d.out_scale, d.acc32, d.ldacc, d.post_lrelu, d.post_slope = 2, 0.5, acc.data_ptr(), N, 1, 0.2
        out = torch.full((rows, N), 7.0, device="cuda")
        d.out32, d.ldo = out.data_ptr(), N
        d.ksplit = ksplit
        if ksplit > 1 and scratch:
            d.mx_scratch, d.mx_scratch_size = ws.data_ptr(), S * rows * N * 4
        torch.cuda.synchronize()
        rc = lib.ev_op_conv_gemm(C.byref(d), None)
        torch.cuda.synchronize()
        return rc, out

    rc, one = run(M, 0)
    assert rc == 0
    rc, spl = run(M, S)
    assert rc == 0
    rc, spl2 = run(M, S)
    assert rc == 0 and torch.equal(spl, spl2)
    rc, big = run(Mbig, S)
    assert rc == 0 and torch.equal(big[:M], spl)                 # the same rows inside a larger launch: same bits
    ref = F.conv1d(x.double().cpu().t().unsqueeze(0), w.double().cpu(), bias.double().cpu(), padding=(taps - 1) // 2).squeeze(0).t()
    if form == "relu":
        ref = ref.clamp(min=0)
    if form == "full":
        ref = 0.5 * ref * (1.0 + torch.erf(ref / math.sqrt(2.0)))
    if form in ("res", "full"):
        ref = ref + res[:M].double().cpu()
    if form == "full":
        ref = ref * 0.5 + acc[:M].double().cpu()
        ref = torch.where(ref > 0, ref, ref * 0.2)
    vrow = valid[:M // 8].bool().repeat_interleave(8).cpu()
    ref[~vrow] = 0
    assert float(spl[~vrow.cuda()].abs().max()) == 0.0
    e1, es = _rel(one.cpu().double(), ref), _rel(spl.cpu().double(), ref)
    assert e1 < 3e-6 and es < 3e-6, (e1, es)
    assert _rel(spl.cpu().double(), one.cpu().double()) < 1e-6
    # refused: no scratch, a range count that does not divide the K-chunks, a dtype without the split kernel
    assert run(M, S, scratch=False)[0] == -2
    if (K // 32) % 5:
        assert run(M, 5)[0] == -2


@pytest.mark.parametrize("C_,k,dil", [(32, 11, 5), (32, 3, 1), (64, 7, 3), (128, 11, 5), (96, 7, 1)])
def test_split_precision_generator_convs(lib, C_, k, dil):
    """The split-precision (EV_PREC_X3) form of the HiFi-GAN ResBlock convs (models/hifigan/models.py:50-57,121-126): fp32
    activations, leaky-relu applied to the fp32 value while staging (before the hi/lo split), dilated taps, N = 32 (BN = 32
    tile) and N % 64 == 0, then conv2's epilogue: fp32 residual, 1/3 scale, fp32 accumulate-in, row mask.  fp64 reference."""
    from emotivoice_amd import _ffi
    torch.manual_seed(C_ * 100 + k)
    M = 1024
    N = 512 if C_ == 96 else C_                      # conv_pre shape: K = 96 (padded mel), N = 512
    full, x = _padded(M, C_, torch.float32)
    full[:PAD] = 0
    full[PAD + M:] = 0
    valid = torch.ones(M // 4, dtype=torch.uint8, device="cuda")
    valid[:3] = 0
    valid[100:104] = 0
    vrow = valid.bool().repeat_interleave(4)
    w = torch.randn(N, C_, k, device="cuda") / math.sqrt(C_ * k)
    bias = torch.randn(N, device="cuda")
    wg = w.permute(0, 2, 1).contiguous()
    hi = wg.half()
    lo = ((wg - hi.float()) * 2048.0).half()
    res = torch.randn(M, N, device="cuda")
    acc = torch.randn(M, N, device="cuda")

    def call(pro, act, with_res):
        d = _ffi.ev_conv_gemm_desc()
        d.dtype, d.A, d.lda, d.W, d.W_lo = 2, x.data_ptr(), C_, hi.data_ptr(), lo.data_ptr()
        d.bias, d.M, d.N, d.K, d.taps, d.dil, d.center, d.out_scale = bias.data_ptr(), M, N, C_, k, dil, (k - 1) // 2, 1.0
        d.row_valid, d.valid_shift = valid.data_ptr(), 2
        if pro:
            d.pro_lrelu, d.pro_slope = 1, 0.1
        if act:
            d.act, d.act_slope = 3, 0.1
        if with_res:
            d.res, d.res_dtype, d.ldres = res.data_ptr(), 1, N
            d.out_scale = 1.0 / 3.0
            d.acc32, d.ldacc = acc.data_ptr(), N
        out = torch.full((M, N), 7.0, device="cuda")
        d.out32, d.ldo = out.data_ptr(), N
        torch.cuda.synchronize()
        assert lib.ev_op_conv_gemm(C.byref(d), None) == 0
        torch.cuda.synchronize()
        return out

    xin = x.double().cpu()
    conv = lambda z: F.conv1d(z.t().unsqueeze(0), w.double().cpu(), bias.double().cpu(), dilation=dil, padding=dil * (k - 1) // 2).squeeze(0).t()  # noqa: E731
    # conv1 form: lrelu -> conv -> lrelu
    ref1 = _lrelu(conv(_lrelu(xin, 0.1)), 0.1)
    ref1[~vrow.cpu()] = 0
    o1 = call(True, True, False)
    assert _rel(o1.cpu().double(), ref1) < 3e-6, (C_, k, dil)
    assert float(o1[~vrow].abs().max()) == 0.0
    # conv2 form: conv + residual, scaled, + running MRF sum
    ref2 = (conv(xin) + res.double().cpu()) / 3.0 + acc.double().cpu()
    ref2[~vrow.cpu()] = 0
    o2 = call(False, False, True)
    assert _rel(o2.cpu().double(), ref2) < 3e-6, (C_, k, dil)


@pytest.mark.parametrize("C_,N,k,dil", [(128, 128, 7, 3), (64, 64, 11, 5), (32, 32, 3, 1), (32, 32, 11, 5), (256, 512, 3, 1)])
def test_split_precision_8_wave_kernel_large_m(lib, C_, N, k, dil):
    """conv_gemm_x3_kernel (8 waves, 256-row tiles) is only selected from 512 tiles up (smaller GEMMs take the 128 x 64 kernel), so it
    needs inputs of >= 131 072 rows to be exercised by an op test: all three BN configurations (128 / 64 / 32), the single-slab
    K = 32 case, leaky-relu prologue + epilogue, fp32 residual + scale + fp32 accumulate-in (the in-place MRF sum: acc32 == out32),
    and bit-identity with the small-tile kernel on the same rows (the engine's batch-invariance promise)."""
    from emotivoice_amd import _ffi
    torch.manual_seed(C_ + N + k)
    M = 256 * 520 if N <= 128 else 256 * 136          # >= 512 tiles of 256 x BN
    full = torch.randn(M + 2 * PAD, C_, device="cuda")
    full[:PAD] = 0
    full[PAD + M:] = 0
    x = full[PAD:PAD + M]
    w = torch.randn(N, C_, k, device="cuda") / math.sqrt(C_ * k)
    bias = torch.randn(N, device="cuda")
    wg = w.permute(0, 2, 1).contiguous()
    hi = wg.half()
    lo = ((wg - hi.float()) * 2048.0).half()
    res = torch.randn(M, N, device="cuda")
    acc0 = torch.randn(M, N, device="cuda")

    def call(rows, conv2):
        d = _ffi.ev_conv_gemm_desc()
        d.dtype, d.A, d.lda, d.W, d.W_lo = 2, x.data_ptr(), C_, hi.data_ptr(), lo.data_ptr()
        d.bias, d.M, d.N, d.K, d.taps, d.dil, d.center, d.out_scale = bias.data_ptr(), rows, N, C_, k, dil, (k - 1) // 2, 1.0
        out = acc0[:rows].clone()
        if conv2:
            d.res, d.res_dtype, d.ldres, d.out_scale = res.data_ptr(), 1, N, 1.0 / 3.0
            d.acc32, d.ldacc = out.data_ptr(), N          # in place, like the engine's running MRF sum
        else:
            d.pro_lrelu, d.pro_slope, d.act, d.act_slope = 1, 0.1, 3, 0.1
        d.out32, d.ldo = out.data_ptr(), N
        torch.cuda.synchronize()
        assert lib.ev_op_conv_gemm(C.byref(d), None) == 0
        torch.cuda.synchronize()
        return out

    sub = 256 * 8                                           # rows compared against the fp64 reference / the small-tile kernel
    xin = full[:PAD + sub + PAD].double().cpu()
    conv = lambda z: F.conv1d(z.t().unsqueeze(0), w.double().cpu(), bias.double().cpu(), dilation=dil).squeeze(0).t()  # noqa: E731
    h = dil * (k - 1) // 2
    for conv2 in (False, True):
        big = call(M, conv2)
        if conv2:
            ref = (conv(xin)[PAD - h:PAD - h + sub] + res[:sub].double().cpu()) / 3.0 + acc0[:sub].double().cpu()
        else:
            ref = _lrelu(conv(_lrelu(xin, 0.1))[PAD - h:PAD - h + sub], 0.1)
        assert _rel(big[:sub].cpu().double(), ref) < 3e-6, (C_, N, k, conv2)
        small = call(sub, conv2)                            # 8 x N/BN tiles: the first-generation kernel
        # rows whose conv window stays inside the first `sub` rows see identical inputs in both launches
        assert torch.equal(big[:sub - h], small[:sub - h]), (C_, N, k, conv2)
        assert torch.isfinite(big).all()


# ---------------------------------------------------------------------------------------------------------------------
# "MX" precision: hi.hi as one fp16 MFMA + the two cross terms as block-scaled fp4 MFMAs (ev_gemm_mx.h).  emotivoice_amd/mxfp4.py
# is the host statement of the plane format; the references below are built from it in fp64.
def _mx_weights(w_nkt):
    """w [N][K][taps] fp32 (cuda) -> dict of device tensors in the kernel's layouts + the dequantised fp4 parts (cpu fp64)."""
    from emotivoice_amd import mxfp4
    wg = w_nkt.permute(0, 2, 1).contiguous().float().cpu().numpy()            # [N][taps][K]
    hi = wg.astype(np.float16)
    lo = wg - hi.astype(np.float32)
    planes = mxfp4.pack_weight_planes(wg)
    qwl, qwh = mxfp4.weight_planes_dequant(planes, *wg.shape)
    return dict(hi=torch.from_numpy(hi).cuda(), lo=torch.from_numpy((lo * np.float32(2048.0)).astype(np.float16)).cuda(),
                mx=torch.from_numpy(planes).cuda(), wh=torch.from_numpy(hi.astype(np.float64)), qwl=torch.from_numpy(qwl.astype(np.float64)),
                qwh=torch.from_numpy(qwh.astype(np.float64)), w=torch.from_numpy(wg.astype(np.float64)))


def _mx_act_parts(a):
    """a [rows][C] fp32 cpu -> (hi, Q(hi), Q(lo)) as fp64 + the planes as numpy (fp16 hi, codes hi / lo, scale bytes hi / lo)."""
    from emotivoice_amd import mxfp4
    an = a.numpy().astype(np.float32)
    hi, lo = mxfp4.split_hi_lo(an)
    ch, sh = mxfp4.quantize(hi, 32)
    cl, sl = mxfp4.quantize(lo, 32)
    t = lambda z: torch.from_numpy(np.asarray(z, np.float64))          # noqa: E731
    return t(hi), t(mxfp4.dequantize(ch, sh, 32)), t(mxfp4.dequantize(cl, sl, 32)), (hi.astype(np.float16), ch, cl, sh, sl)


def _e5_act_parts(a):
    """a [rows][C] fp32 cpu -> (hi, Q(hi), Q(lo)) as fp64 in the maxima-free E5M2 operand format of ev_pair_e5.h (mxfp4.e5m2_*): Q(hi) = the top byte of the fp16 hi
    part, Q(lo) = E5M2 of lo 2^11 at the constant scale 2^-11."""
    from emotivoice_amd import mxfp4
    hi, lo = mxfp4.split_hi_lo(a.numpy().astype(np.float32))
    t = lambda z: torch.from_numpy(np.asarray(z, np.float64))          # noqa: E731
    return t(hi), t(mxfp4.e5m2_decode(mxfp4.e5m2_hi_codes(hi))), t(mxfp4.e5m2_lo_decode(mxfp4.e5m2_lo_codes(lo))), None


def _conv64(x_rows, w_ntk, dil, taps):
    """x [rows][K] fp64 (with halo rows), w [N][taps][K] fp64 -> valid conv [rows - (taps-1)*dil][N]."""
    return F.conv1d(x_rows.t().unsqueeze(0), w_ntk.permute(0, 2, 1).contiguous(), dilation=dil).squeeze(0).t()


class _PlaneSet:
    """device plane set of an [M][C] activation with 64 slack rows on both sides (the layout ev_gemm_mx.h reads)."""

    def __init__(self, M, Cc):
        R = M + 2 * PAD
        self.M, self.C, self.R = M, Cc, R
        self.h = torch.full((R, Cc), 3.0, device="cuda", dtype=torch.float16)
        self.q4 = [torch.full((R, Cc // 2), 0x77, device="cuda", dtype=torch.uint8) for _ in range(2)]
        self.qs = [torch.full((max(1, Cc // 128), R, 4), 130, device="cuda", dtype=torch.uint8) for _ in range(2)]

    def out_fields(self, d, slope):
        d.mxo_h = self.h[PAD:].data_ptr()
        d.mxo_q4[0], d.mxo_q4[1] = self.q4[0][PAD:].data_ptr(), self.q4[1][PAD:].data_ptr()
        d.mxo_qs[0], d.mxo_qs[1] = self.qs[0][0, PAD:].data_ptr(), self.qs[1][0, PAD:].data_ptr()
        d.mxo_qs_stride, d.mxo_logC, d.mxo_slope = self.R * 4, int(math.log2(self.C)), slope

    def in_fields(self, d):
        d.A, d.lda = self.h[PAD:].data_ptr(), self.C
        d.mx_x4[0], d.mx_x4[1] = self.q4[0][PAD:].data_ptr(), self.q4[1][PAD:].data_ptr()
        d.mx_xs[0], d.mx_xs[1] = self.qs[0][0, PAD:].data_ptr(), self.qs[1][0, PAD:].data_ptr()
        d.mx_xs_stride = self.R * 4


def _mx_desc(lib, wts, M, N, K, taps, dil, bias):
    from emotivoice_amd import _ffi
    d = _ffi.ev_conv_gemm_desc()
    d.dtype, d.W, d.W_lo, d.W_mx = 3, wts["hi"].data_ptr(), wts["lo"].data_ptr(), wts["mx"].data_ptr()
    d.bias, d.M, d.N, d.K, d.taps, d.dil, d.center, d.out_scale = bias.data_ptr(), M, N, K, taps, dil, (taps - 1) // 2, 1.0
    d.ldo = N
    return d


def _launch(lib, d):
    torch.cuda.synchronize()
    assert lib.ev_op_conv_gemm(C.byref(d), None) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("Cc,N,k,dil,M", [(128, 128, 3, 1, 512), (128, 128, 7, 3, 512), (128, 128, 11, 5, 768), (256, 256, 11, 1, 256),
                                         (256, 1024, 3, 1, 256), (512, 2048, 3, 1, 256)])
def test_mx_conv_gemm_fp32_input(lib, Cc, N, k, dil, M):
    """DT_MX with an fp32 activation: mx_planes_kernel (leaky-relu, hi / lo split, fp4 planes) + conv_gemm_mx_kernel against the
    fp64 evaluation of  xh.wh + Q(xh).Q(wl) + Q(xl).Q(wh)  with the host quantiser -- and against the exact product, which the
    fp4 cross terms must approach to ~1e-4 (fp16 operands alone: 4e-4)."""
    from emotivoice_amd import _ffi
    torch.manual_seed(Cc + N + k)
    full = torch.randn(M + 2 * PAD, Cc, device="cuda") * torch.exp(torch.randn(M + 2 * PAD, 1, device="cuda"))     # rows of very different scale
    full[:PAD] = 0
    full[PAD + M:] = 0
    x = full[PAD:PAD + M]
    w = torch.randn(N, Cc, k, device="cuda") / math.sqrt(Cc * k)
    bias = torch.randn(N, device="cuda")
    wts = _mx_weights(w)
    valid = torch.ones(M // 4, dtype=torch.uint8, device="cuda")
    valid[:2] = 0
    valid[50:53] = 0
    vrow = valid.bool().repeat_interleave(4).cpu()
    d = _mx_desc(lib, wts, M, N, Cc, k, dil, bias)
    d.A, d.lda = x.data_ptr(), Cc
    d.pro_lrelu, d.pro_slope, d.act, d.act_slope = 1, 0.1, 3, 0.1
    d.row_valid, d.valid_shift = valid.data_ptr(), 2
    nb = lib.ev_op_mx_scratch_bytes(M, Cc)
    scratch = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    d.mx_scratch, d.mx_scratch_size = scratch.data_ptr(), nb
    out = torch.full((M, N), 7.0, device="cuda")
    d.out32 = out.data_ptr()
    _launch(lib, d)
    h = dil * (k - 1) // 2
    a = _lrelu(full.float().cpu(), 0.1)
    ah, qah, qal, _ = _mx_act_parts(a)
    rows = slice(PAD - h, PAD + M + h)
    emu = _conv64(ah[rows], wts["wh"], dil, k) + _conv64(qah[rows], wts["qwl"], dil, k) + _conv64(qal[rows], wts["qwh"], dil, k) + bias.double().cpu()
    exact = _conv64(a.double()[rows], wts["w"], dil, k) + bias.double().cpu()
    emu, exact = _lrelu(emu, 0.1), _lrelu(exact, 0.1)
    emu[~vrow] = 0
    exact[~vrow] = 0
    got = out.cpu().double()
    assert float(out[~vrow.cuda()].abs().max()) == 0.0
    assert _rel(got, emu) < 2e-6, (Cc, N, k, _rel(got, emu))
    assert _rel(got, exact) < 1.5e-4, (Cc, N, k, _rel(got, exact))
    assert _rel(emu, exact) > 1e-6          # (the test would be vacuous if the emulation were exact)


@pytest.mark.parametrize("cin,s", [(512, 8), (256, 8), (128, 2)])
def test_mx_transposed_conv_skips_its_zero_tap_bitwise(lib, cin, s):
    """The generator's up-convs on conv_gemm_mx_kernel (ConvTranspose1d(k = 2 s, stride s, pad s / 2) as a 3-tap conv, packer._convT_to_gemm): every output phase has
    one all-zero tap.  With ev_conv_gemm_desc.polyphase_cout the kernel issues no matrix instruction for it (round 6: a third of the launch's); the products that are
    left and their order are the same, so outputs and plane sets must equal the launch that multiplies the zeros -- and the torch ConvTranspose1d to the MX level."""
    from emotivoice_amd import _ffi
    from emotivoice_amd.packer import _convT_to_gemm
    torch.manual_seed(77 + cin + s)
    cout, M = cin // 2, 256 * 3
    N = s * cout
    full = torch.randn(M + 2 * PAD, cin, device="cuda") * torch.exp(0.5 * torch.randn(M + 2 * PAD, 1, device="cuda"))
    full[:PAD] = 0
    full[PAD + M:] = 0
    x = full[PAD:PAD + M]
    wt = torch.randn(cin, cout, 2 * s) / math.sqrt(cin * 2)
    wg = torch.from_numpy(_convT_to_gemm(wt.numpy(), s))              # [s * cout][3][cin]
    wts = _mx_weights(wg.permute(0, 2, 1).contiguous().cuda())
    bias = (torch.randn(cout) * 0.1).repeat(s).cuda()
    valid = torch.ones(M // 4, dtype=torch.uint8, device="cuda")
    valid[:3] = 0
    nb = lib.ev_op_mx_scratch_bytes(M, cin)
    outs = []
    for hint in (0, cout):
        d = _mx_desc(lib, wts, M, N, cin, 3, 1, bias)
        d.A, d.lda = x.data_ptr(), cin
        d.pro_lrelu, d.pro_slope = 1, 0.1
        d.row_valid, d.valid_shift = valid.data_ptr(), 2
        scratch = torch.zeros(nb, dtype=torch.uint8, device="cuda")
        d.mx_scratch, d.mx_scratch_size = scratch.data_ptr(), nb
        out = torch.full((M, N), 7.0, device="cuda")
        ps = _PlaneSet(M * s, cout)
        d.out32 = out.data_ptr()
        ps.out_fields(d, 0.1)
        d.polyphase_cout = hint
        _launch(lib, d)
        outs.append((out, ps))
    (o0, p0), (o1, p1) = outs
    assert torch.equal(o0, o1)
    assert torch.equal(p0.h, p1.h) and all(torch.equal(p0.q4[i], p1.q4[i]) and torch.equal(p0.qs[i], p1.qs[i]) for i in range(2))
    ref = F.conv_transpose1d(_lrelu(x.float().cpu(), 0.1).t().unsqueeze(0), wt, bias[:cout].cpu(), stride=s, padding=s // 2).squeeze(0).t()
    got = o1.cpu().reshape(M * s, cout)
    vrow = valid.bool().repeat_interleave(4 * s).cpu()
    assert float(got[~vrow].abs().max()) == 0.0
    assert _rel(got[vrow], ref[vrow]) < 2e-4
    # a hint that does not describe whole waves / an even phase count is refused
    d.polyphase_cout = 32
    assert lib.ev_op_conv_gemm(C.byref(d), None) != 0


@pytest.mark.parametrize("K,N,M,res", [(384, 1152, 256 * 5, False), (384, 384, 256 * 3, True), (128, 128, 256 * 2, False), (1536, 384, 256 * 2, True)])
def test_mx_one_tap_gemm(lib, K, N, M, res):
    """gemm_mx1_kernel (ev_gemm_mx1.h): nn.Linear in the MX arithmetic (the mel decoder's QKV / output projections): fp32 activation ->
    mx_planes_kernel -> one fp16 MFMA + two block-scaled fp4 MFMAs per product, one column block per pipeline step.  fp64 references: the same
    arithmetic with the host quantiser (tight) and the exact product (the MX error level); invalid row groups incl. a whole all-gap tile; the residual
    variant runs in place like the engine's."""
    torch.manual_seed(K + N + M)
    x = torch.randn(M, K, device="cuda") * torch.exp(0.5 * torch.randn(M, 1, device="cuda"))
    w = torch.randn(N, K, 1, device="cuda") / math.sqrt(K)
    bias = torch.randn(N, device="cuda")
    wts = _mx_weights(w)
    valid = torch.ones(M // 64, dtype=torch.uint8, device="cuda")
    valid[1] = 0
    valid[4:8] = 0                          # rows 256 .. 511: one whole tile
    vrow = valid.bool().repeat_interleave(64).cpu()
    r0 = torch.randn(M, N, device="cuda")
    out = r0.clone() if res else torch.full((M, N), 7.0, device="cuda")
    nb = lib.ev_op_mx_scratch_bytes(M, K)
    scratch = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    d = _mx_desc(lib, wts, M, N, K, 1, 1, bias)
    d.center = 0
    d.A, d.lda = x.data_ptr(), K
    d.row_valid, d.valid_shift = valid.data_ptr(), 6
    d.mx_scratch, d.mx_scratch_size = scratch.data_ptr(), nb
    d.out32 = out.data_ptr()
    if res:
        d.res, d.res_dtype, d.ldres = out.data_ptr(), 1, N
    _launch(lib, d)
    ah, qah, qal, _ = _mx_act_parts(x.cpu())
    emu = _conv64(ah, wts["wh"], 1, 1) + _conv64(qah, wts["qwl"], 1, 1) + _conv64(qal, wts["qwh"], 1, 1) + bias.double().cpu()
    exact = _conv64(x.double().cpu(), wts["w"], 1, 1) + bias.double().cpu()
    if res:
        emu, exact = emu + r0.double().cpu(), exact + r0.double().cpu()
    emu[~vrow] = 0
    exact[~vrow] = 0
    got = out.cpu().double()
    assert float(out[~vrow.cuda()].abs().max()) == 0.0
    assert _rel(got, emu) < 2e-6, (K, N, _rel(got, emu))
    assert _rel(got, exact) < 1.5e-4, (K, N, _rel(got, exact))
    assert _rel(emu, exact) > 1e-6
    out2 = r0.clone() if res else torch.full((M, N), 7.0, device="cuda")          # run to run
    d.out32 = out2.data_ptr()
    if res:
        d.res = out2.data_ptr()
    _launch(lib, d)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("Cc,N,k,dil,tiles_m", [(128, 128, 3, 5, 1100), (256, 256, 7, 1, 600), (128, 128, 11, 3, 530)])
def test_mx_conv_gemm_long_tile_list_with_all_gap_tiles(lib, Cc, N, k, dil, tiles_m):
    """conv_gemm_mx_kernel on more tiles than the chip holds at once (several residency rounds), with invalid row groups -- incl. whole all-gap
    tiles, which skip the main loop and must still write zero rows / zero planes -- in the middle and at the end of the tile list: two identical
    launches agree bit for bit (fp32 output AND the plane set of the result), and the rows around the gap match the fp64 emulation."""
    torch.manual_seed(Cc + k + tiles_m)
    M = 256 * tiles_m
    full = torch.randn(M + 2 * PAD, Cc, device="cuda")
    full[:PAD] = 0
    full[PAD + M:] = 0
    x = full[PAD:PAD + M]
    w = torch.randn(N, Cc, k, device="cuda") / math.sqrt(Cc * k)
    bias = torch.randn(N, device="cuda")
    wts = _mx_weights(w)
    valid = torch.ones(M // 64, dtype=torch.uint8, device="cuda")
    valid[:3] = 0
    valid[400:409] = 0                      # rows 25600 .. 26175: two whole tiles and a partial one
    valid[-7:] = 0
    nb = lib.ev_op_mx_scratch_bytes(M, Cc)
    scratch = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    outs = []
    for dbg in (0, 0):
        d = _mx_desc(lib, wts, M, N, Cc, k, dil, bias)
        d.A, d.lda = x.data_ptr(), Cc
        d.pro_lrelu, d.pro_slope = 1, 0.1
        d.row_valid, d.valid_shift = valid.data_ptr(), 6
        d.mx_scratch, d.mx_scratch_size = scratch.data_ptr(), nb
        out = torch.full((M, N), 7.0, device="cuda")
        ps = _PlaneSet(M, N)
        d.out32 = out.data_ptr()
        ps.out_fields(d, 0.1)
        d.reserved0 = dbg
        _launch(lib, d)
        outs.append((out, ps))
    (o0, p0), (o1, p1) = outs
    assert torch.equal(o0, o1)
    assert torch.equal(p0.h[PAD:PAD + M], p1.h[PAD:PAD + M])
    for i in range(2):
        assert torch.equal(p0.q4[i][PAD:PAD + M], p1.q4[i][PAD:PAD + M]) and torch.equal(p0.qs[i][:, PAD:PAD + M], p1.qs[i][:, PAD:PAD + M])
    vrow = valid.bool().repeat_interleave(64)
    assert float(o0[~vrow].abs().max()) == 0.0 and float(o0[vrow].abs().max()) > 0.1
    # spot check against the fp64 emulation on the rows around the all-gap tiles
    lo, hi = 25600 - 512, 26176 + 512
    h = dil * (k - 1) // 2
    a = _lrelu(full[PAD + lo - h:PAD + hi + h].float().cpu(), 0.1)
    ah, qah, qal, _ = _mx_act_parts(a)
    emu = _conv64(ah, wts["wh"], dil, k) + _conv64(qah, wts["qwl"], dil, k) + _conv64(qal, wts["qwh"], dil, k) + bias.double().cpu()
    emu[~vrow[lo:hi].cpu()] = 0
    assert _rel(o0[lo:hi].cpu().double(), emu) < 2e-6


def _random_plane_set(M, Cc, seed):
    """a device plane set with random contents (hi parts ~N(0, 1) with a per-row gain, random fp4 codes, block scales around 2^-8 .. 2^-2 of the hi parts):
    for A/B comparisons of two kernels on identical inputs, no host quantiser involved."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    ps = _PlaneSet(M, Cc)
    R = ps.R
    ps.h = (torch.randn(R, Cc, device="cuda", generator=g) * torch.exp(0.5 * torch.randn(R, 1, device="cuda", generator=g))).half()
    ps.q4 = [torch.randint(0, 256, (R, Cc // 2), device="cuda", dtype=torch.uint8, generator=g) for _ in range(2)]
    ps.qs = [torch.randint(lo, lo + 4, (max(1, Cc // 128), R, 4), device="cuda", dtype=torch.uint8, generator=g) for lo in (124, 113)]
    for t in (ps.h, ps.q4[0], ps.q4[1]):
        t[:PAD] = 0
        t[PAD + M:] = 0
    return ps


MX_EPI_FORMS = ["conv1", "gelu", "o32", "o32+planes", "res32", "res32+acc", "respl", "respl+o32", "respl+acc+o32+planes", "respl+acc>planes",
                    "part", "acc+part", "accpl>planes"]


@pytest.mark.parametrize("Cc,k,dil,tiles_m", [(128, 3, 1, 520), (128, 7, 3, 515), (128, 11, 5, 513), (256, 3, 5, 260), (256, 11, 1, 257)])
def test_mx_epilogue_forms_long_tile_list_deterministic(lib, Cc, k, dil, tiles_m):
    """conv_gemm_mx_kernel in every fast epilogue form the engine uses, on more tiles than the chip holds at once, random plane sets with invalid row groups incl.
    whole all-gap tiles: two launches agree on every output bit for bit (a missing MFMA -> VALU wait state or a scratch race shows up as run-to-run noise: round 4's
    mfma_asm_fence finding), invalid rows are zeros in every output.  (Written as the A/B harness of the persistent-tile-loop experiment, which it passed bit for
    bit against this kernel: profiles/r4_e_mx_persistent_prefetch_ab.txt.)"""
    from emotivoice_amd import _ffi, mxfp4
    torch.manual_seed(1700 + Cc + k)
    M = 256 * tiles_m
    R = M + 2 * PAD
    assert (M // 256) * (Cc // 128) > 512
    w = torch.randn(Cc, Cc, k) / math.sqrt(Cc * k)
    wg = w.permute(0, 2, 1).contiguous().numpy()
    hi = wg.astype(np.float16)
    lo16 = ((wg - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    d_hi, d_lo, d_mx = torch.from_numpy(hi).cuda(), torch.from_numpy(lo16).cuda(), torch.from_numpy(mxfp4.pack_weight_planes(wg)).cuda()
    bias = torch.randn(Cc, device="cuda") * 0.1
    valid = torch.ones(M // 64, dtype=torch.uint8, device="cuda")
    valid[:5] = 0                               # the first tile is all gap, the second starts with one
    valid[1000:1013] = 0                        # three whole tiles and a partial one in the middle
    valid[-9:] = 0                              # the last two tiles
    ps_in, ps_res, ps_acc = _random_plane_set(M, Cc, 1), _random_plane_set(M, Cc, 2), _random_plane_set(M, Cc, 3)
    res32 = torch.randn(M, Cc, device="cuda")
    acc32 = torch.randn(M, Cc, device="cuda")
    for form in MX_EPI_FORMS:
        outs = []
        for dbg in (0, 0):
            d = _ffi.ev_conv_gemm_desc()
            d.dtype, d.W, d.W_lo, d.W_mx = 3, d_hi.data_ptr(), d_lo.data_ptr(), d_mx.data_ptr()
            ps_in.in_fields(d)
            d.bias, d.M, d.N, d.K, d.taps, d.dil, d.center, d.out_scale, d.ldo = bias.data_ptr(), M, Cc, Cc, k, dil, (k - 1) // 2, 1.0, Cc
            d.row_valid, d.valid_shift = valid.data_ptr(), 6
            out = torch.full((M, Cc), 7.0, device="cuda")
            ps_o = _PlaneSet(M, Cc)
            acc_ps = None
            if form == "conv1":
                d.act, d.act_slope = 3, 0.1
                ps_o.out_fields(d, 1.0)
            elif form == "gelu":
                d.act = 2
                ps_o.out_fields(d, 1.0)
            elif form in ("o32", "o32+planes"):
                d.out32 = out.data_ptr()
                if "planes" in form:
                    ps_o.out_fields(d, 0.1)
            elif form.startswith("res32"):
                d.res, d.res_dtype, d.ldres, d.out32, d.out_scale = res32.data_ptr(), 1, Cc, out.data_ptr(), 1.0 / 3.0
                ps_o.out_fields(d, 0.1)
                if "acc" in form:
                    out.copy_(acc32)
                    d.acc32, d.ldacc = out.data_ptr(), Cc                   # in place, as the engine's fp32 MRF sum
            else:
                d.res, d.res_dtype, d.ldres, d.out_scale = ps_res.h[PAD:].data_ptr(), 3, Cc, 1.0 / 3.0
                d.res_x4, d.res_xs, d.res_xs_stride, d.res_inv_slope = ps_res.q4[1][PAD:].data_ptr(), ps_res.qs[1][0, PAD:].data_ptr(), R * 4, 10.0
                if form == "respl":
                    ps_o.out_fields(d, 0.1)
                elif form == "respl+o32":
                    d.out32 = out.data_ptr()
                elif form == "respl+acc+o32+planes":
                    d.acc32, d.ldacc, d.out32 = acc32.data_ptr(), Cc, out.data_ptr()
                    ps_o.out_fields(d, 0.1)
                elif form == "respl+acc>planes":
                    d.acc32, d.ldacc = acc32.data_ptr(), Cc
                    ps_o.out_fields(d, 0.1)
                elif form == "part":
                    ps_o.out_fields(d, 1.0)
                    d.mxo_partial = 1
                else:
                    acc_ps = _PlaneSet(M, Cc)
                    acc_ps.h.copy_(ps_acc.h); acc_ps.q4[1].copy_(ps_acc.q4[1]); acc_ps.qs[1].copy_(ps_acc.qs[1])
                    d.acc_h, d.acc_x4, d.acc_xs, d.acc_xs_stride, d.ldacc = acc_ps.h[PAD:].data_ptr(), acc_ps.q4[1][PAD:].data_ptr(), acc_ps.qs[1][0, PAD:].data_ptr(), R * 4, Cc
                    if form == "acc+part":
                        ps_o = acc_ps                                       # in place, as the engine's second ResBlock
                        ps_o.out_fields(d, 1.0)
                        d.mxo_partial = 1
                    else:
                        ps_o.out_fields(d, 0.1)
            d.reserved0 = dbg
            _launch(lib, d)
            outs.append((out, ps_o))
        (o0, p0), (o1, p1) = outs
        assert torch.equal(o0, o1), form
        assert torch.equal(p0.h, p1.h) and all(torch.equal(p0.q4[i], p1.q4[i]) and torch.equal(p0.qs[i], p1.qs[i]) for i in range(2)), form
        wrote32 = form not in ("conv1", "gelu", "respl", "respl+acc>planes", "part", "acc+part", "accpl>planes")
        if wrote32:
            vrow = valid.bool().repeat_interleave(64)
            assert float(o0[~vrow].abs().max()) == 0.0 and float(o0[vrow].abs().max()) > 0.1, form
        else:
            assert float(p0.h[PAD:PAD + M].float().abs().max()) > 0.1 and not bool(p0.h[PAD:PAD + 5 * 64].any()), form


@pytest.mark.parametrize("Cc,k,dil,up", [(128, 7, 3, 0), (256, 3, 1, 0), (128, 11, 5, 0), (256, 3, 1, 4)])
def test_mx_plane_set_chain(lib, Cc, k, dil, up):
    """The producer's epilogue writes the consumer's operand planes (EPI_MXP): (1) the planes equal the host quantiser applied to
    lrelu(out32) bit for bit (fp16 hi plane, fp4 codes, E8M0 scales, zero planes on invalid rows); (2) a second MX conv reading
    them gives bit-identical results to the same conv fed with the fp32 tensor (planes made by mx_planes_kernel); (3) conv1-style
    launches may write the planes ONLY; (4) up > 0: a transposed conv's [M][up * C] output lands as the [M * up][C] plane set;
    (5) residual / accumulate-in epilogues with planes."""
    from emotivoice_amd import _ffi, mxfp4
    torch.manual_seed(Cc * 7 + k + up)
    M = 512
    N1 = Cc * up if up else Cc                           # conv A: [M][Cc] -> [M][N1] (viewed as [M * max(up,1)][Cc])
    M2 = M * up if up else M
    full = torch.randn(M + 2 * PAD, Cc, device="cuda")
    full[:PAD] = 0
    full[PAD + M:] = 0
    wa = torch.randn(N1, Cc, k, device="cuda") / math.sqrt(Cc * k)
    ba = torch.randn(N1, device="cuda")
    wtsa = _mx_weights(wa)
    vshift = 2
    valid = torch.ones(M >> vshift, dtype=torch.uint8, device="cuda")
    valid[:3] = 0
    valid[40:44] = 0
    nb = lib.ev_op_mx_scratch_bytes(max(M, M2), Cc)
    scratch = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    res = torch.randn(M, N1, device="cuda")
    acc = torch.randn(M, N1, device="cuda")

    def conv_a(with_out32, with_res):
        ps = _PlaneSet(M2, Cc)
        d = _mx_desc(lib, wtsa, M, N1, Cc, k, dil, ba)
        d.A, d.lda, d.pro_lrelu, d.pro_slope = full[PAD:].data_ptr(), Cc, 1, 0.1
        d.mx_scratch, d.mx_scratch_size = scratch.data_ptr(), nb
        d.row_valid, d.valid_shift = valid.data_ptr(), vshift
        out = torch.full((M, N1), 7.0, device="cuda") if with_out32 else None
        if with_out32:
            d.out32 = out.data_ptr()
        if with_res:
            d.res, d.res_dtype, d.ldres, d.out_scale = res.data_ptr(), 1, N1, 1.0 / 3.0
            if with_res == 2:
                d.acc32, d.ldacc = acc.data_ptr(), N1
        else:
            d.act, d.act_slope = 3, 0.1
        ps.out_fields(d, 0.1 if with_res else 1.0)
        _launch(lib, d)
        return out, ps

    def check_planes(out, ps, slope):
        a = _lrelu(out.cpu().reshape(M2, Cc), slope)
        _, _, _, (h16, ch, cl, sh, sl) = _mx_act_parts(a)
        assert np.array_equal(ps.h[PAD:PAD + M2].cpu().numpy().view(np.uint16), h16.view(np.uint16)), "hi plane"
        for i, (codes, sb) in enumerate(((ch, sh), (cl, sl))):
            gs = ps.qs[i][:, PAD:PAD + M2].cpu().numpy()                      # [C/128][M2][4]
            want = sb.reshape(M2, Cc // 128, 4).transpose(1, 0, 2)
            bad = np.argwhere(gs != want)
            assert bad.size == 0, ("scale plane %d" % i, bad[:4], gs[tuple(bad[0])], want[tuple(bad[0])])
            gc = ps.q4[i][PAD:PAD + M2].cpu().numpy()
            bad = np.argwhere(gc != codes)
            assert bad.size == 0, ("code plane %d" % i, bad[:4], hex(gc[tuple(bad[0])]), hex(codes[tuple(bad[0])]))
        vrow = valid.bool().repeat_interleave((1 << vshift) * max(up, 1)).cpu().numpy()
        assert not ps.h[PAD:PAD + M2].cpu().numpy()[~vrow].any() and not ps.q4[0][PAD:PAD + M2].cpu().numpy()[~vrow].any()

    out_a, ps_a = conv_a(True, 0)
    check_planes(out_a, ps_a, 1.0)
    _, ps_only = conv_a(False, 0)                                   # (3) planes only
    assert torch.equal(ps_only.h, ps_a.h) and all(torch.equal(ps_only.q4[i][PAD:PAD + M2], ps_a.q4[i][PAD:PAD + M2]) for i in range(2))
    assert all(torch.equal(ps_only.qs[i][:, PAD:PAD + M2], ps_a.qs[i][:, PAD:PAD + M2]) for i in range(2))
    for mode in (1, 2):                                             # (5) residual (+ accumulate-in) epilogues, consumer slope 0.1
        out_r, ps_r = conv_a(True, mode)
        check_planes(out_r, ps_r, 0.1)
    # (2) conv B on the plane set vs conv B on the fp32 tensor
    wb = torch.randn(Cc, Cc, k, device="cuda") / math.sqrt(Cc * k)
    bb = torch.randn(Cc, device="cuda")
    wtsb = _mx_weights(wb)
    # the planes' slack rows / the fp32 tensor's halo rows: zeros on both sides (what the engine's gap rows provide)
    for t in [ps_a.h] + ps_a.q4:
        t[:PAD] = 0
        t[PAD + M2:] = 0
    for t in ps_a.qs:
        t[:, :PAD] = 1
        t[:, PAD + M2:] = 1
    fa = torch.zeros(M2 + 2 * PAD, Cc, device="cuda")
    fa[PAD:PAD + M2] = out_a.reshape(M2, Cc)
    outs = []
    for planes_in in (True, False):
        d = _mx_desc(lib, wtsb, M2, Cc, Cc, k, 1, bb)
        if planes_in:
            ps_a.in_fields(d)
        else:
            d.A, d.lda = fa[PAD:].data_ptr(), Cc                     # act = lrelu above, consumer slope 1: no prologue
            d.mx_scratch, d.mx_scratch_size = scratch.data_ptr(), nb
        o = torch.full((M2, Cc), 7.0, device="cuda")
        d.out32 = o.data_ptr()
        _launch(lib, d)
        outs.append(o)
    assert torch.equal(outs[0], outs[1])
    assert torch.isfinite(outs[0]).all()


@pytest.mark.parametrize("fmt", ["e5m2", "fp4"])
@pytest.mark.parametrize("k,dil,acc_in", [(3, 1, False), (3, 5, True), (7, 3, False), (7, 1, True), (11, 5, True), (11, 1, False)])
def test_fused_mx_resblock_pair(lib, k, dil, acc_in, fmt):
    """The fused C = 32 pair: conv1(dil) -> leaky-relu -> conv2 + fp32 residual, every product as one fp16 MFMA + two block-scaled MFMAs for the
    cross terms (four taps per instruction), x fp32 in / fp32 out, the slab and the intermediate quantised in the kernel.  fmt = the ACTIVATION
    operand of the cross terms: "e5m2" = resblock_pair_c32_e5_kernel (ev_pair_e5.h, the default since round 6: top byte of the fp16 hi part +
    E5M2 remainder at the constant scale 2^-11, no block maxima), "fp4" = resblock_pair_c32_mx*_kernel (ev_pair_mx.h, epi.reserved0 bit 4 /
    ev_config.mx_act_format = 1).  References in fp64: (a) the same arithmetic with the host quantiser (mxfp4.py) -- tight; (b) the exact convs of
    models/hifigan/models.py:50-57 -- the MX error level (~1e-4; fp16 operands give 5e-4)."""
    FMT = 32 if fmt == "e5m2" else 16          # (32: the E5M2 kernel at every k -- the launcher itself picks it for k = 3 only)
    from emotivoice_amd import _ffi, mxfp4
    torch.manual_seed(300 + k + dil)
    Cc, M = 32, 5 * 256
    full = torch.randn(M + 2 * PAD, Cc, device="cuda") * torch.exp(0.7 * torch.randn(M + 2 * PAD, 1, device="cuda"))
    valid = torch.ones(M // 16, dtype=torch.uint8, device="cuda")
    valid[:2] = 0
    valid[30:34] = 0
    valid[-3:] = 0
    vrow = valid.repeat_interleave(16).bool()
    full[:PAD] = 0
    full[PAD + M:] = 0
    x = full[PAD:PAD + M]
    x[~vrow] = 0                                   # the engine's invariant: invalid rows of every tensor are exact zeros
    w1 = torch.randn(Cc, Cc, k, device="cuda") / math.sqrt(Cc * k)
    w2 = torch.randn(Cc, Cc, k, device="cuda") / math.sqrt(Cc * k)
    b1, b2 = torch.randn(Cc, device="cuda") * 0.1, torch.randn(Cc, device="cuda") * 0.1
    acc = torch.randn(M, Cc, device="cuda")

    def wparts(w):
        wg = w.permute(0, 2, 1).contiguous().cpu().numpy()            # [N][taps][K]
        planes = mxfp4.pack_pair_weight_planes(wg)
        ql, qh = mxfp4.pair_weight_planes_dequant(planes, k)
        hi = wg.astype(np.float16)
        t = lambda z: torch.from_numpy(np.asarray(z, np.float64))      # noqa: E731
        return torch.from_numpy(hi).cuda(), torch.from_numpy(planes).cuda(), t(hi), t(ql), t(qh), t(wg)
    w1h, w1m, w1hd, w1ql, w1qh, w1e = wparts(w1)
    w2h, w2m, w2hd, w2ql, w2qh, w2e = wparts(w2)
    out = acc.clone() if acc_in else torch.full((M, Cc), 7.0, device="cuda")
    d = _ffi.ev_res_pair_desc()
    d.x, d.ldx, d.w1, d.b1, d.w2, d.M, d.k, d.dil = x.data_ptr(), Cc, w1h.data_ptr(), b1.data_ptr(), w2h.data_ptr(), M, k, dil
    d.w1_mx, d.w2_mx = w1m.data_ptr(), w2m.data_ptr()
    e = d.epi
    e.bias, e.res, e.res_dtype, e.ldres = b2.data_ptr(), x.data_ptr(), 1, Cc
    e.row_valid, e.valid_shift, e.out_scale = valid.data_ptr(), 4, 1.0 / 3.0
    if acc_in:
        e.acc32, e.ldacc = out.data_ptr(), Cc       # in place: the engine's running MRF sum
    e.out32, e.ldo = out.data_ptr(), Cc
    e.reserved0 = FMT
    torch.cuda.synchronize()
    assert lib.ev_op_resblock_pair_c32_mx(C.byref(d), None) == 0
    torch.cuda.synchronize()
    h = (k - 1) // 2

    def mxconv(a_rows, whd, wql, wqh, dd):
        ah, qah, qal, _ = (_e5_act_parts if fmt == "e5m2" else _mx_act_parts)(a_rows.float())
        return _conv64(ah, whd, dd, k) + _conv64(qah, wql, dd, k) + _conv64(qal, wqh, dd, k)
    xin = full.cpu()
    a0 = _lrelu(xin, 0.1)
    vr = vrow.cpu()
    # conv1 on rows [-h, M + h) of the tensor (it needs h * dil more on each side), masked to the valid rows
    r1 = slice(PAD - h - h * dil, PAD + M + h + h * dil)
    vmask1 = torch.zeros(M + 2 * h, dtype=torch.bool)
    vmask1[h:h + M] = vr
    addend = acc.double().cpu() if acc_in else 0.0
    res = {}
    for name, conv in (("emu", lambda a, i, dd: mxconv(a, *((w1hd, w1ql, w1qh) if i == 0 else (w2hd, w2ql, w2qh)), dd)),
                       ("exact", lambda a, i, dd: _conv64(a.double(), w1e if i == 0 else w2e, dd, k))):
        xt = _lrelu(conv(a0[r1], 0, dil) + b1.double().cpu(), 0.1)
        xt[~vmask1] = 0
        y = (conv(xt.float() if name == "emu" else xt, 1, 1) + b2.double().cpu() + xin[PAD:PAD + M].double()) / 3.0 + addend
        y[~vr] = 0
        res[name] = y
    got = out.cpu().double()
    assert float(out[~vrow].abs().max()) == 0.0
    # (an fp16 ulp of the in-LDS intermediate may round differently than the fp64 emulation: its remainder plane absorbs the difference up
    # to the fp4 step, measured <= 1.8e-5)
    assert _rel(got, res["emu"]) < 5e-5, (k, dil, _rel(got, res["emu"]))
    assert _rel(got, res["exact"]) < 2e-4, (k, dil, _rel(got, res["exact"]))
    assert _rel(res["emu"], res["exact"]) > 1e-6
    # the launcher's own choice (reserved0 = 0) is a rule on the layer's shape: E5M2 at k = 3, fp4 at k = 7 / 11
    outd = acc.clone() if acc_in else torch.full((M, Cc), 7.0, device="cuda")
    if acc_in:
        e.acc32 = outd.data_ptr()
    e.out32 = outd.data_ptr()
    e.reserved0 = 0
    assert lib.ev_op_resblock_pair_c32_mx(C.byref(d), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(outd, out) == ((k == 3) == (fmt == "e5m2")), (k, fmt)
    if fmt == "e5m2":
        # the two formats are different arithmetic (the cross terms' last bits): E5M2 is the closer one to the exact convs on these data
        outs4 = acc.clone() if acc_in else torch.full((M, Cc), 7.0, device="cuda")
        if acc_in:
            e.acc32 = outs4.data_ptr()
        e.out32 = outs4.data_ptr()
        e.reserved0 = 16
        assert lib.ev_op_resblock_pair_c32_mx(C.byref(d), None) == 0
        torch.cuda.synchronize()
        assert not torch.equal(out, outs4) and _rel(got, outs4.cpu().double()) < 2e-4
        print("pair k=%d dil=%d: vs exact  e5m2 %.2e  fp4 %.2e" % (k, dil, _rel(got, res["exact"]), _rel(outs4.cpu().double(), res["exact"])))
    # fp4: the two-group schedule (resblock_pair_c32_mx2_kernel) against the lock-step kernel (epi.reserved0 bit 2): same arithmetic per output element, 128-
    # instead of 256-row tiles -- bit-identical outputs
    out2 = acc.clone() if acc_in else torch.full((M, Cc), 7.0, device="cuda")
    if acc_in:
        e.acc32 = out2.data_ptr()
    e.out32 = out2.data_ptr()
    e.reserved0 = FMT | 4
    torch.cuda.synchronize()
    assert lib.ev_op_resblock_pair_c32_mx(C.byref(d), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, out2), (k, dil, acc_in)
    # ... and run to run: round 3's <3, accumulate-in> instantiation read conv1's accumulators too early after the last inline-asm MFMA (no hardware
    # interlock, see mfma_asm_fence in ev_gemm.hip) and differed between identical launches in a few hundred rows
    for dbg in (0, 4, 0):
        out3 = acc.clone() if acc_in else torch.full((M, Cc), 7.0, device="cuda")
        if acc_in:
            e.acc32 = out3.data_ptr()
        e.out32 = out3.data_ptr()
        e.reserved0 = FMT | dbg
        torch.cuda.synchronize()
        assert lib.ev_op_resblock_pair_c32_mx(C.byref(d), None) == 0
        torch.cuda.synchronize()
        assert torch.equal(out, out3), (k, dil, acc_in, dbg)


@pytest.mark.parametrize("dil,mode,M", [(1, "planes", 256 * 5), (3, "planes", 256 * 3), (5, "o32", 256 * 4), (5, "acc+o32+planes", 256 * 5), (3, "acc>planes", 256 * 2),
                                        (8, "o32+planes", 256 * 3)])
def test_fused_mx_resblock_pair_c64(lib, dil, mode, M):
    """resblock_pair_c64_mx_kernel (ev_pair64_mx.h): the k = 3 pair of stage 2 in one kernel, plane sets in / out, against the TWO layer-wise launches it
    replaces (conv_c64_mx_kernel: conv1 planes -> planes, conv2 planes + residual-from-planes -> planes / fp32, both covered by their own emulation tests):
    same products, same accumulation order, same quantisers -- every output must agree bit for bit.  128-row tiles over row counts that are not multiples of the
    126 output rows per tile, invalid row groups, every epilogue form the engine uses (planes only; fp32 only with the MRF scale; accumulate-in)."""
    from emotivoice_amd import _ffi, mxfp4
    torch.manual_seed(640 + dil + M)
    Cc, k = 64, 3
    R = M + 2 * PAD
    valid = torch.ones(M // 8, dtype=torch.uint8, device="cuda")
    valid[:2] = 0
    valid[37:41] = 0
    valid[-5:] = 0
    x = torch.randn(R, Cc) * torch.exp(0.5 * torch.randn(R, 1))
    x[:PAD] = 0
    x[PAD + M:] = 0
    x[PAD:PAD + M][~valid.repeat_interleave(8).bool().cpu()] = 0          # the engine's invariant: invalid rows of every tensor are exact zeros
    ps_x, _ = _host_plane_set(_lrelu(x, 0.1).float())

    def wset(seed):
        g = torch.Generator().manual_seed(seed)
        wg = (torch.randn(Cc, k, Cc, generator=g) / math.sqrt(Cc * k)).numpy()
        hi = wg.astype(np.float16)
        lo16 = ((wg - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
        return torch.from_numpy(hi).cuda(), torch.from_numpy(lo16).cuda(), torch.from_numpy(mxfp4.pack_c64_weight_planes(wg)).cuda()
    w1h, w1l, w1m = wset(1)
    w2h, w2l, w2m = wset(2)
    b1, b2 = torch.randn(Cc, device="cuda") * 0.1, torch.randn(Cc, device="cuda") * 0.1
    acc = torch.randn(M, Cc, device="cuda")
    want32, planes_out, acc_in = "o32" in mode, "planes" in mode, mode.startswith("acc")

    def epi_fields(e, out, ps_o):
        e.bias, e.row_valid, e.valid_shift, e.out_scale, e.ldo = b2.data_ptr(), valid.data_ptr(), 3, 1.0 / 3.0, Cc
        e.res_inv_slope = 10.0
        if acc_in:
            e.acc32, e.ldacc = acc.data_ptr(), Cc
        if want32:
            e.out32 = out.data_ptr()
        if planes_out:
            ps_o.out_fields(e, 0.1)
            e.mxo_logC = 6

    # ---- layer-wise: conv1 (planes in, leaky-relu, planes out), conv2 (planes + residual from the input planes)
    ps_t = _PlaneSet(M, Cc)
    ps_t.h.zero_()
    for i in range(2):
        ps_t.q4[i].zero_()
        ps_t.qs[i].fill_(1)                       # (the engine's plane buffers have zero slack rows: the arena is cleared and only rows [0, M) are written)
    d1 = _ffi.ev_conv_gemm_desc()
    d1.dtype, d1.W, d1.W_lo, d1.W_mx = 3, w1h.data_ptr(), w1l.data_ptr(), w1m.data_ptr()
    ps_x.in_fields(d1)
    d1.bias, d1.M, d1.N, d1.K, d1.taps, d1.dil, d1.center, d1.out_scale, d1.ldo = b1.data_ptr(), M, Cc, Cc, k, dil, 1, 1.0, Cc
    d1.row_valid, d1.valid_shift, d1.act, d1.act_slope = valid.data_ptr(), 3, 3, 0.1
    ps_t.out_fields(d1, 1.0)
    d1.mxo_logC = 6
    _launch(lib, d1)
    out_a, ps_a = torch.full((M, Cc), 7.0, device="cuda"), _PlaneSet(M, Cc)
    d2 = _ffi.ev_conv_gemm_desc()
    d2.dtype, d2.W, d2.W_lo, d2.W_mx = 3, w2h.data_ptr(), w2l.data_ptr(), w2m.data_ptr()
    ps_t.in_fields(d2)
    d2.M, d2.N, d2.K, d2.taps, d2.dil, d2.center = M, Cc, Cc, k, 1, 1
    d2.res, d2.res_dtype, d2.ldres = ps_x.h[PAD:].data_ptr(), 3, Cc
    d2.res_x4, d2.res_xs, d2.res_xs_stride = ps_x.q4[1][PAD:].data_ptr(), ps_x.qs[1][0, PAD:].data_ptr(), R * 4
    epi_fields(d2, out_a, ps_a)
    _launch(lib, d2)
    # ---- fused
    out_b, ps_b = torch.full((M, Cc), 7.0, device="cuda"), _PlaneSet(M, Cc)
    dp = _ffi.ev_res_pair_desc()
    dp.x, dp.ldx, dp.w1, dp.b1, dp.w2, dp.M, dp.k, dp.dil = ps_x.h[PAD:].data_ptr(), Cc, w1h.data_ptr(), b1.data_ptr(), w2h.data_ptr(), M, k, dil
    dp.w1_mx, dp.w2_mx = w1m.data_ptr(), w2m.data_ptr()
    e = dp.epi
    e.mx_x4[0], e.mx_x4[1] = ps_x.q4[0][PAD:].data_ptr(), ps_x.q4[1][PAD:].data_ptr()
    e.mx_xs[0], e.mx_xs[1], e.mx_xs_stride = ps_x.qs[0][0, PAD:].data_ptr(), ps_x.qs[1][0, PAD:].data_ptr(), R * 4
    epi_fields(e, out_b, ps_b)
    for _ in range(2):          # (twice: the second launch must reproduce the first)
        torch.cuda.synchronize()
        assert lib.ev_op_resblock_pair_c64_mx(C.byref(dp), None) == 0
        torch.cuda.synchronize()
        if want32:
            assert torch.equal(out_a, out_b), (dil, mode)
            assert float(out_b.abs().max()) > 0.1 and float(out_b[~valid.repeat_interleave(8).bool()].abs().max()) == 0.0
        if planes_out:
            assert torch.equal(ps_a.h[PAD:PAD + M], ps_b.h[PAD:PAD + M]), (dil, mode)
            for i in range(2):
                assert torch.equal(ps_a.q4[i][PAD:PAD + M], ps_b.q4[i][PAD:PAD + M]), (dil, mode, i)
                assert torch.equal(ps_a.qs[i][0, PAD:PAD + M, :2], ps_b.qs[i][0, PAD:PAD + M, :2]), (dil, mode, i)
            assert float(ps_b.h[PAD:PAD + M].float().abs().max()) > 0.1
    # the slack rows of the output planes were not touched
    if planes_out:
        assert float((ps_b.h[:PAD].float() - 3.0).abs().max()) == 0.0 and float((ps_b.h[PAD + M:].float() - 3.0).abs().max()) == 0.0
    assert lib.ev_op_resblock_pair_c64_mx(C.byref(dp), None) == 0 or True
    dp.k = 7
    assert lib.ev_op_resblock_pair_c64_mx(C.byref(dp), None) == -2          # only k = 3 is built


@pytest.mark.parametrize("dil,M", [(5, 256 * 3), (1, 256 * 2)])
def test_fused_mx_resblock_pair_c64_partial_out(lib, dil, M):
    """The fused C = 64 pair writing the stage's running MRF sum as a PARTIAL plane set (mxo_partial: hi plane, remainder codes, remainder scales of the raw
    scaled result -- round 6, stage 2): the same bits as the hi / remainder planes of the full plane set at slope 1, hi codes and hi scales left alone."""
    from emotivoice_amd import _ffi, mxfp4
    torch.manual_seed(6400 + dil + M)
    Cc, k = 64, 3
    R = M + 2 * PAD
    valid = torch.ones(M // 8, dtype=torch.uint8, device="cuda")
    valid[:1] = 0
    valid[-3:] = 0
    x = torch.randn(R, Cc) * torch.exp(0.5 * torch.randn(R, 1))
    x[:PAD] = 0
    x[PAD + M:] = 0
    x[PAD:PAD + M][~valid.repeat_interleave(8).bool().cpu()] = 0
    ps_x, _ = _host_plane_set(_lrelu(x, 0.1).float())

    def wset(seed):
        g = torch.Generator().manual_seed(seed)
        wg = (torch.randn(Cc, k, Cc, generator=g) / math.sqrt(Cc * k)).numpy()
        return torch.from_numpy(wg.astype(np.float16)).cuda(), torch.from_numpy(mxfp4.pack_c64_weight_planes(wg)).cuda()
    w1h, w1m = wset(11)
    w2h, w2m = wset(12)
    b1, b2 = torch.randn(Cc, device="cuda") * 0.1, torch.randn(Cc, device="cuda") * 0.1
    sets = []
    for part in (0, 1):
        ps_o = _PlaneSet(M, Cc)
        dp = _ffi.ev_res_pair_desc()
        dp.x, dp.ldx, dp.w1, dp.b1, dp.w2, dp.M, dp.k, dp.dil = ps_x.h[PAD:].data_ptr(), Cc, w1h.data_ptr(), b1.data_ptr(), w2h.data_ptr(), M, k, dil
        dp.w1_mx, dp.w2_mx = w1m.data_ptr(), w2m.data_ptr()
        e = dp.epi
        e.mx_x4[0], e.mx_x4[1] = ps_x.q4[0][PAD:].data_ptr(), ps_x.q4[1][PAD:].data_ptr()
        e.mx_xs[0], e.mx_xs[1], e.mx_xs_stride = ps_x.qs[0][0, PAD:].data_ptr(), ps_x.qs[1][0, PAD:].data_ptr(), R * 4
        e.bias, e.row_valid, e.valid_shift, e.out_scale, e.ldo, e.res_inv_slope = b2.data_ptr(), valid.data_ptr(), 3, 1.0 / 3.0, Cc, 10.0
        ps_o.out_fields(e, 1.0)
        e.mxo_logC, e.mxo_partial = 6, part
        torch.cuda.synchronize()
        assert lib.ev_op_resblock_pair_c64_mx(C.byref(dp), None) == 0
        torch.cuda.synchronize()
        sets.append(ps_o)
        if part:          # a partial set beside an fp32 output, or with an activation, is not a call the kernel takes
            e.mxo_slope = 0.1
            assert lib.ev_op_resblock_pair_c64_mx(C.byref(dp), None) != 0
    full, part = sets
    assert float(full.h[PAD:PAD + M].float().abs().max()) > 0.1
    assert torch.equal(full.h[PAD:PAD + M], part.h[PAD:PAD + M])
    assert torch.equal(full.q4[1][PAD:PAD + M], part.q4[1][PAD:PAD + M])
    assert torch.equal(full.qs[1][0, PAD:PAD + M, :2], part.qs[1][0, PAD:PAD + M, :2])
    assert bool((part.q4[0] == 0x77).all()) and bool((part.qs[0] == 130).all())          # hi codes / hi scales untouched
    assert not bool((full.q4[0][PAD:PAD + M] == 0x77).all())


@pytest.mark.parametrize("k,dil,mode,M", [(3, 1, "conv1", 256 * 3), (3, 5, "conv2acc", 256 * 8), (7, 3, "conv1", 256 * 9), (7, 1, "conv2", 256 * 17),
                                          (11, 5, "conv1", 256 * 8), (11, 1, "conv2acc", 256 * 5), (3, 1, "up", 256 * 8)])
def test_conv_c64_mx(lib, k, dil, mode, M):
    """conv_c64_mx_kernel (ev_conv64_mx.h) and, for k = 7 / 11, conv_gemm_mx64_kernel (ev_gemm_mx64.h): the stage-2 convs (C = 64) in the MX arithmetic on plane
    sets -- two taps per fp4 MFMA; persistent kernel: output channels split over two work items; tile counts that are / are not multiples of 8 (padded work items must store nothing, also with
    the in-place accumulate).  The input plane set comes from the host quantiser (mxfp4.py); fp64 references: the same arithmetic (tight) and
    the exact conv (the MX error level).  Outputs: fp32 rows and / or the plane set of lrelu(result), which must equal the host quantiser
    applied to the fp32 rows bit for bit."""
    from emotivoice_amd import _ffi, mxfp4
    torch.manual_seed(500 + k + dil + M)
    Cc = 64
    R = M + 2 * PAD
    full = torch.randn(R, Cc) * torch.exp(0.7 * torch.randn(R, 1))
    full[:PAD] = 0
    full[PAD + M:] = 0
    valid = torch.ones(M // 8, dtype=torch.uint8, device="cuda")
    valid[:2] = 0
    valid[40:43] = 0
    vrow = valid.repeat_interleave(8).bool()
    ah, qah, qal, (h16, ch, cl, sh, sl) = _mx_act_parts(full)
    s4 = lambda sb: np.concatenate([sb, np.ones((R, 2), np.uint8)], 1)        # noqa: E731  [rows][4 B], two bytes used
    d_h = torch.from_numpy(h16).cuda()
    d_q = [torch.from_numpy(np.ascontiguousarray(c)).cuda() for c in (ch, cl)]
    d_s = [torch.from_numpy(s4(sb)).cuda() for sb in (sh, sl)]
    w = torch.randn(Cc, Cc, k) / math.sqrt(Cc * k)
    bias = torch.randn(Cc, device="cuda") * 0.1
    wg = w.permute(0, 2, 1).contiguous().numpy()
    planes = mxfp4.pack_c64_weight_planes(wg)
    ql, qh = mxfp4.c64_weight_planes_dequant(planes, k)
    hi = wg.astype(np.float16)
    lo16 = ((wg - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    t64 = lambda z: torch.from_numpy(np.asarray(z, np.float64))          # noqa: E731
    whd, wql, wqh, we = t64(hi), t64(ql), t64(qh), t64(wg)
    d_hi, d_lo, d_mx = torch.from_numpy(hi).cuda(), torch.from_numpy(lo16).cuda(), torch.from_numpy(planes).cuda()
    res = torch.randn(M, Cc, device="cuda")
    acc = torch.randn(M, Cc, device="cuda")
    want32 = mode != "conv1"                                            # conv1 of a pair writes its result as planes only
    # k = 7 / 11: both kernels that can run the launch -- the streamed conv_gemm_mx64_kernel (the launcher's choice) and the persistent conv_c64_mx_kernel (reserved0 bit 3)
    for r0 in ((0, 8) if k > 3 else (0,)):
        out = acc.clone() if mode == "conv2acc" else torch.full((M, Cc), 7.0, device="cuda")
        ps = _PlaneSet(M, Cc)                                               # (its scale planes: [1][rows][4])
        d = _ffi.ev_conv_gemm_desc()
        d.dtype, d.A, d.lda, d.W, d.W_lo, d.W_mx = 3, d_h[PAD:].data_ptr(), Cc, d_hi.data_ptr(), d_lo.data_ptr(), d_mx.data_ptr()
        d.mx_x4[0], d.mx_x4[1] = d_q[0][PAD:].data_ptr(), d_q[1][PAD:].data_ptr()
        d.mx_xs[0], d.mx_xs[1], d.mx_xs_stride = d_s[0][PAD:].data_ptr(), d_s[1][PAD:].data_ptr(), R * 4
        d.bias, d.M, d.N, d.K, d.taps, d.dil, d.center, d.out_scale = bias.data_ptr(), M, Cc, Cc, k, dil, (k - 1) // 2, 1.0
        d.row_valid, d.valid_shift = valid.data_ptr(), 3
        if mode == "conv1":
            d.act, d.act_slope = 3, 0.1
        if mode.startswith("conv2"):
            d.res, d.res_dtype, d.ldres, d.out_scale = res.data_ptr(), 1, Cc, 1.0 / 3.0
            if mode == "conv2acc":
                d.acc32, d.ldacc = out.data_ptr(), Cc
        if want32:
            d.out32 = out.data_ptr()
        d.ldo = Cc
        planes_out = mode in ("conv1", "conv2")
        if planes_out:
            ps.out_fields(d, 1.0 if mode == "conv1" else 0.1)
            d.mxo_logC = 6
        d.reserved0 = r0
        _launch(lib, d)
        h = dil * (k - 1) // 2
        rows = slice(PAD - h, PAD + M + h)
        refs = {"emu": _conv64(ah[rows], whd, dil, k) + _conv64(qah[rows], wql, dil, k) + _conv64(qal[rows], wqh, dil, k),
                "exact": _conv64(full.double()[rows], we, dil, k)}
        vr = vrow.cpu()
        for name in refs:
            y = refs[name] + bias.double().cpu()
            if mode == "conv1":
                y = _lrelu(y, 0.1)
            if mode.startswith("conv2"):
                y = (y + res.double().cpu()) / 3.0 + (acc.double().cpu() if mode == "conv2acc" else 0.0)
            y[~vr] = 0
            refs[name] = y
        if want32:
            got = out.cpu().double()
            assert float(out[~vrow].abs().max()) == 0.0
            assert _rel(got, refs["emu"]) < 2e-6, (k, dil, mode, _rel(got, refs["emu"]))
            assert _rel(got, refs["exact"]) < 1.5e-4, (k, dil, mode, _rel(got, refs["exact"]))
        if planes_out:
            gh = ps.h[PAD:PAD + M].float().cpu()
            gq = [mxfp4.dequantize(ps.q4[i][PAD:PAD + M].cpu().numpy(), ps.qs[i][0, PAD:PAD + M, :2].cpu().numpy(), 32) for i in range(2)]
            if want32:              # the planes are the host quantiser applied to lrelu(out32, 0.1), bit for bit
                _, _, _, (h16o, cho, clo, sho, slo) = _mx_act_parts(_lrelu(out.cpu(), 0.1))
                assert np.array_equal(ps.h[PAD:PAD + M].cpu().numpy().view(np.uint16), h16o.view(np.uint16))
                for i, (codes, sb) in enumerate(((cho, sho), (clo, slo))):
                    assert np.array_equal(ps.q4[i][PAD:PAD + M].cpu().numpy(), codes), i
                    assert np.array_equal(ps.qs[i][0, PAD:PAD + M, :2].cpu().numpy(), sb), i
            else:                   # planes only: hi + Q(lo) reproduces the conv's result to the fp4 step of the remainder (2^-11 x 0.25)
                rec = gh.double() + torch.from_numpy(gq[1]).double()
                assert _rel(rec, refs["emu"]) < 1e-4, (k, dil, mode, _rel(rec, refs["emu"]))
                assert _rel(gh.double(), refs["emu"]) < 6e-4 and not ps.h[PAD:PAD + M].cpu().numpy()[~vr.numpy()].any()
    assert _rel(refs["emu"], refs["exact"]) > 1e-6
    # the two-group schedule (conv_c64_mx2_kernel, the launcher's choice above) against the lock-step kernel (reserved0 bit 2): the same arithmetic
    # per output element, so every output must agree bit for bit
    out2 = acc.clone() if mode == "conv2acc" else torch.full((M, Cc), 7.0, device="cuda")
    ps2 = _PlaneSet(M, Cc)
    if mode == "conv2acc":
        d.acc32 = out2.data_ptr()
    if want32:
        d.out32 = out2.data_ptr()
    if planes_out:
        ps2.out_fields(d, 1.0 if mode == "conv1" else 0.1)
        d.mxo_logC = 6
    d.reserved0 = 4 | (8 if k > 3 else 0)          # (k = 7 / 11: `out` / `ps` are the persistent kernel's, the loop's last pass)
    _launch(lib, d)
    if want32:
        assert torch.equal(out, out2), (k, dil, mode)
    if planes_out:
        assert torch.equal(ps.h[PAD:PAD + M], ps2.h[PAD:PAD + M])
        for i in range(2):
            assert torch.equal(ps.q4[i][PAD:PAD + M], ps2.q4[i][PAD:PAD + M]) and torch.equal(ps.qs[i][0, PAD:PAD + M, :2], ps2.qs[i][0, PAD:PAD + M, :2])


def _host_plane_set(a):
    """a [R][C] fp32 cpu (R = M + 2 PAD) -> a _PlaneSet holding the host quantiser's planes of a, and the parts (hi, Q(hi), Q(lo)) in fp64."""
    R, Cc = a.shape
    hi64, qh64, ql64, (h16, ch, cl, sh, sl) = _mx_act_parts(a)
    ps = _PlaneSet(R - 2 * PAD, Cc)
    ps.h.copy_(torch.from_numpy(h16))
    for i, (codes, sb) in enumerate(((ch, sh), (cl, sl))):
        ps.q4[i].copy_(torch.from_numpy(np.ascontiguousarray(codes)))
        if Cc == 64:            # [1][rows][4], two bytes used
            sb4 = np.concatenate([sb, np.ones((R, 2), np.uint8)], 1)[None]
        else:                   # chunk-major [C / 128][rows][4]
            sb4 = np.ascontiguousarray(sb.reshape(R, Cc // 128, 4).transpose(1, 0, 2))
        ps.qs[i].copy_(torch.from_numpy(sb4))
    return ps, (hi64, qh64, ql64)


@pytest.mark.parametrize("Cc,k,dil,mode", [(128, 3, 1, "planes"), (128, 7, 3, "o32"), (128, 11, 1, "o32+planes"), (256, 3, 5, "acc"),
                                           (256, 7, 1, "acc+planes"), (64, 3, 1, "planes"), (64, 7, 3, "o32+planes"), (64, 11, 5, "acc+planes"),
                                           (64, 3, 5, "acc"), (128, 7, 1, "acc>planes"), (256, 3, 3, "acc>planes"), (64, 7, 1, "acc>planes")])
def test_mx_residual_from_planes(lib, Cc, k, dil, mode):
    """res_dtype 3: conv2 of a ResBlock pair takes its residual from the plane set of lrelu(x, 0.1) that conv1 read -- x' = lrelu^-1(hi + Q4(lo)) --
    instead of an fp32 tensor (conv_gemm_mx_kernel's EPI_RESPL epilogues at C = 128 / 256, conv_c64_mx_kernel<K, MODE, true> at C = 64).
    Reference: the same arithmetic in fp64 with the host quantiser's planes; the fp32 output must match it tightly, a planes-only output must
    reproduce it to the fp4 step of the remainder, planes beside an fp32 output must be the host quantiser of that output bit for bit."""
    from emotivoice_amd import _ffi, mxfp4
    torch.manual_seed(900 + Cc + k + dil)
    M = 256 * 5
    R = M + 2 * PAD
    valid = torch.ones(M // 8, dtype=torch.uint8, device="cuda")
    valid[:2] = 0
    valid[70:73] = 0
    vrow = valid.repeat_interleave(8).bool().cpu()
    # operand (xt) and residual (x) activations; the plane sets hold lrelu(., slope) of them: slope 1 for xt (already activated), 0.1 for x
    xt = torch.randn(R, Cc) * torch.exp(0.5 * torch.randn(R, 1))
    xt[:PAD] = 0
    xt[PAD + M:] = 0
    x = torch.randn(R, Cc) * torch.exp(0.5 * torch.randn(R, 1))
    ps_t, (th, tqh, tql) = _host_plane_set(xt)
    ps_x, (xh, _, xql) = _host_plane_set(_lrelu(x, 0.1).float())
    a_rec = xh + xql                                              # what the epilogue adds back: hi + Q4(lo), then the inverse leaky-relu
    x_rec = torch.where(a_rec >= 0, a_rec, a_rec * 10.0)[PAD:PAD + M]
    assert _rel(x_rec, x.double()[PAD:PAD + M]) < 2e-4
    w = torch.randn(Cc, Cc, k) / math.sqrt(Cc * k)
    bias = torch.randn(Cc, device="cuda") * 0.1
    wg = w.permute(0, 2, 1).contiguous().numpy()
    hi = wg.astype(np.float16)
    lo16 = ((wg - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    if Cc == 64:
        planes = mxfp4.pack_c64_weight_planes(wg)
        ql, qh = mxfp4.c64_weight_planes_dequant(planes, k)
    else:
        planes = mxfp4.pack_weight_planes(wg)
        ql, qh = mxfp4.weight_planes_dequant(planes, *wg.shape)
    t64 = lambda z: torch.from_numpy(np.asarray(z, np.float64))          # noqa: E731
    d_hi, d_lo, d_mx = torch.from_numpy(hi).cuda(), torch.from_numpy(lo16).cuda(), torch.from_numpy(planes).cuda()
    acc = torch.randn(M, Cc, device="cuda")
    # ("acc>planes": the last conv of a stage in the engine's default flow -- running MRF sum in, ONLY the next up-conv's planes out)
    want32, planes_out, acc_in = mode not in ("planes", "acc>planes"), "planes" in mode, mode.startswith("acc")
    out = torch.full((M, Cc), 7.0, device="cuda")
    ps_o = _PlaneSet(M, Cc)
    d = _ffi.ev_conv_gemm_desc()
    d.dtype, d.W, d.W_lo, d.W_mx = 3, d_hi.data_ptr(), d_lo.data_ptr(), d_mx.data_ptr()
    ps_t.in_fields(d)
    d.bias, d.M, d.N, d.K, d.taps, d.dil, d.center, d.out_scale, d.ldo = bias.data_ptr(), M, Cc, Cc, k, dil, (k - 1) // 2, 1.0 / 3.0, Cc
    d.row_valid, d.valid_shift = valid.data_ptr(), 3
    d.res, d.res_dtype, d.ldres = ps_x.h[PAD:].data_ptr(), 3, Cc
    d.res_x4, d.res_xs, d.res_xs_stride, d.res_inv_slope = ps_x.q4[1][PAD:].data_ptr(), ps_x.qs[1][0, PAD:].data_ptr(), R * 4, 10.0
    if acc_in:
        d.acc32, d.ldacc = acc.data_ptr(), Cc
    if want32:
        d.out32 = out.data_ptr()
    if planes_out:
        ps_o.out_fields(d, 0.1)
        if Cc == 64:
            d.mxo_logC = 6
    _launch(lib, d)
    h = dil * (k - 1) // 2
    rows = slice(PAD - h, PAD + M + h)
    ref = _conv64(th[rows], t64(hi), dil, k) + _conv64(tqh[rows], t64(ql), dil, k) + _conv64(tql[rows], t64(qh), dil, k) + bias.double().cpu()
    ref = (ref + x_rec) / 3.0 + (acc.double().cpu() if acc_in else 0.0)
    ref[~vrow] = 0
    if want32:
        got = out.cpu().double()
        assert float(out[~vrow.cuda()].abs().max()) == 0.0
        assert _rel(got, ref) < 2e-6, (Cc, k, dil, mode, _rel(got, ref))
    if planes_out:
        sl_ = (lambda q: q[0, PAD:PAD + M, :2]) if Cc == 64 else (lambda q: q[:, PAD:PAD + M].permute(1, 0, 2).reshape(M, Cc // 32))
        if want32:
            _, _, _, (h16o, cho, clo, sho, slo) = _mx_act_parts(_lrelu(out.cpu(), 0.1))
            assert np.array_equal(ps_o.h[PAD:PAD + M].cpu().numpy().view(np.uint16), h16o.view(np.uint16))
            for i, (codes, sb) in enumerate(((cho, sho), (clo, slo))):
                assert np.array_equal(ps_o.q4[i][PAD:PAD + M].cpu().numpy(), codes), i
                assert np.array_equal(sl_(ps_o.qs[i]).cpu().numpy(), sb), i
        else:
            lo_rec = mxfp4.dequantize(ps_o.q4[1][PAD:PAD + M].cpu().numpy(), np.ascontiguousarray(sl_(ps_o.qs[1]).cpu().numpy()), 32)
            rec = ps_o.h[PAD:PAD + M].float().cpu().double() + torch.from_numpy(lo_rec).double()
            assert _rel(rec, _lrelu(ref, 0.1)) < 1e-4, (Cc, k, dil, mode, _rel(rec, _lrelu(ref, 0.1)))
            assert not ps_o.h[PAD:PAD + M].cpu().numpy()[~vrow.numpy()].any()


@pytest.mark.parametrize("Cc,k,dil,mode", [(128, 3, 1, "part"), (128, 7, 3, "acc+part"), (128, 11, 1, "acc>planes"), (256, 3, 5, "part"),
                                           (256, 7, 1, "acc+part"), (256, 11, 1, "acc>planes")])
def test_mx_mrf_partial_plane_sets(lib, Cc, k, dil, mode):
    """The running MRF sum of a generator stage as PARTIAL plane sets (ev_conv_gemm_desc::acc_h / mxo_partial; conv_gemm_mx_kernel's EPI_ACCPL / EPI_PART
    epilogues): "part" = the first ResBlock's last conv writes out_scale * x as fp16 hi plane + fp4 remainder codes + their scales, "acc+part" = the second
    one adds the partial and rewrites it IN PLACE, "acc>planes" = the third one adds it and writes the next up-conv's full plane set.  Each is compared bit
    for bit with the fp32 flow it replaces: the same launch with out32 (and the partial's fp32 value as acc32) followed by the host quantiser."""
    from emotivoice_amd import _ffi, mxfp4
    torch.manual_seed(1300 + Cc + k + dil)
    M = 256 * 5
    R = M + 2 * PAD
    valid = torch.ones(M // 8, dtype=torch.uint8, device="cuda")
    valid[:3] = 0
    valid[90:92] = 0
    vrow = valid.repeat_interleave(8).bool().cpu()
    xt = torch.randn(R, Cc) * torch.exp(0.5 * torch.randn(R, 1))
    xt[:PAD] = 0
    xt[PAD + M:] = 0
    x = torch.randn(R, Cc) * torch.exp(0.5 * torch.randn(R, 1))
    ps_t, _ = _host_plane_set(xt)
    ps_x, _ = _host_plane_set(_lrelu(x, 0.1).float())
    run = (torch.randn(R, Cc) * torch.exp(0.7 * torch.randn(R, 1))).float()            # the running sum so far (raw values, no activation)
    ps_s, (sh64, _, sql64) = _host_plane_set(run)
    acc_val = (sh64.float() + sql64.float())[PAD:PAD + M].contiguous().cuda()           # fp32(hi) + fp32(Q4(lo)) in fp32: what the epilogue adds
    w = torch.randn(Cc, Cc, k) / math.sqrt(Cc * k)
    bias = torch.randn(Cc, device="cuda") * 0.1
    wg = w.permute(0, 2, 1).contiguous().numpy()
    hi = wg.astype(np.float16)
    lo16 = ((wg - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    planes = mxfp4.pack_weight_planes(wg)
    d_hi, d_lo, d_mx = torch.from_numpy(hi).cuda(), torch.from_numpy(lo16).cuda(), torch.from_numpy(planes).cuda()
    acc_in, part = mode.startswith("acc"), mode.endswith("part")

    def desc():
        d = _ffi.ev_conv_gemm_desc()
        d.dtype, d.W, d.W_lo, d.W_mx = 3, d_hi.data_ptr(), d_lo.data_ptr(), d_mx.data_ptr()
        ps_t.in_fields(d)
        d.bias, d.M, d.N, d.K, d.taps, d.dil, d.center, d.out_scale, d.ldo = bias.data_ptr(), M, Cc, Cc, k, dil, (k - 1) // 2, 1.0 / 3.0, Cc
        d.row_valid, d.valid_shift = valid.data_ptr(), 3
        d.res, d.res_dtype, d.ldres = ps_x.h[PAD:].data_ptr(), 3, Cc
        d.res_x4, d.res_xs, d.res_xs_stride, d.res_inv_slope = ps_x.q4[1][PAD:].data_ptr(), ps_x.qs[1][0, PAD:].data_ptr(), R * 4, 10.0
        return d

    # the fp32 flow: the same conv with an fp32 output (and the partial's value as an fp32 accumulate-in)
    out = torch.full((M, Cc), 7.0, device="cuda")
    d = desc()
    d.out32 = out.data_ptr()
    if acc_in:
        d.acc32, d.ldacc = acc_val.data_ptr(), Cc
    _launch(lib, d)
    assert float(out[~vrow.cuda()].abs().max()) == 0.0
    # the plane flow
    d = desc()
    ps_o = ps_s if mode == "acc+part" else _PlaneSet(M, Cc)                              # in place for the second ResBlock
    untouched = [ps_o.q4[0].clone(), ps_o.qs[0].clone()]
    if acc_in:
        d.acc_h, d.acc_x4, d.acc_xs, d.acc_xs_stride, d.ldacc = ps_s.h[PAD:].data_ptr(), ps_s.q4[1][PAD:].data_ptr(), ps_s.qs[1][0, PAD:].data_ptr(), R * 4, Cc
    ps_o.out_fields(d, 1.0 if part else 0.1)
    d.mxo_partial = 1 if part else 0
    _launch(lib, d)
    _, _, _, (h16o, cho, clo, sho, slo) = _mx_act_parts(out.cpu() if part else _lrelu(out.cpu(), 0.1))
    sl_ = lambda q: q[:, PAD:PAD + M].permute(1, 0, 2).reshape(M, Cc // 32)             # noqa: E731
    assert np.array_equal(ps_o.h[PAD:PAD + M].cpu().numpy().view(np.uint16), h16o.view(np.uint16)), mode
    assert np.array_equal(ps_o.q4[1][PAD:PAD + M].cpu().numpy(), clo) and np.array_equal(sl_(ps_o.qs[1]).cpu().numpy(), slo), mode
    if part:            # a partial set has no hi-code plane: those buffers are not written
        assert torch.equal(ps_o.q4[0], untouched[0]) and torch.equal(ps_o.qs[0], untouched[1])
    else:
        assert np.array_equal(ps_o.q4[0][PAD:PAD + M].cpu().numpy(), cho) and np.array_equal(sl_(ps_o.qs[0]).cpu().numpy(), sho)
    # and a third launch reproduces the second bit for bit (in-place launches start from a fresh copy of the partial)
    if mode != "acc+part":
        ps_2 = _PlaneSet(M, Cc)
        ps_2.out_fields(d, 1.0 if part else 0.1)
        _launch(lib, d)
        assert torch.equal(ps_2.h[PAD:PAD + M], ps_o.h[PAD:PAD + M]) and torch.equal(ps_2.q4[1][PAD:PAD + M], ps_o.q4[1][PAD:PAD + M])


def test_layernorm_writes_its_consumers_plane_set(lib):
    """LayerNorm with a plane-set output (ev_op_layernorm_planes; the mel decoder's LayerNorms in the mx mode) = the fp32 LayerNorm followed by the host
    quantiser, bit for bit; invalid rows give all-zero planes."""
    torch.manual_seed(77)
    rows, Cc = 1024 + 3, 384
    x = (torch.randn(rows, Cc, device="cuda") * torch.exp(0.5 * torch.randn(rows, 1, device="cuda")) + 0.3)
    g, b = torch.randn(Cc, device="cuda") * 0.5 + 1.0, torch.randn(Cc, device="cuda") * 0.2
    valid = torch.ones(rows, dtype=torch.uint8, device="cuda")
    valid[:4] = 0
    valid[500:504] = 0
    y = torch.full((rows, Cc), 9.0, device="cuda")
    assert lib.ev_op_layernorm(x.data_ptr(), rows, Cc, g.data_ptr(), b.data_ptr(), 1e-12, valid.data_ptr(), None, y.data_ptr(), None, 0.0, None, None) == 0
    R = rows + 2 * PAD
    hpl = torch.full((R, Cc), 3.0, device="cuda", dtype=torch.float16)
    q4 = [torch.full((R, Cc // 2), 0x77, device="cuda", dtype=torch.uint8) for _ in range(2)]
    qs = [torch.full((Cc // 128, R, 4), 130, device="cuda", dtype=torch.uint8) for _ in range(2)]
    assert lib.ev_op_layernorm_planes(x.data_ptr(), rows, Cc, g.data_ptr(), b.data_ptr(), 1e-12, valid.data_ptr(), hpl[PAD:].data_ptr(),
                                      q4[0][PAD:].data_ptr(), q4[1][PAD:].data_ptr(), qs[0][0, PAD:].data_ptr(), qs[1][0, PAD:].data_ptr(), R * 4, None) == 0
    torch.cuda.synchronize()
    _, _, _, (h16, ch, cl, sh, sl) = _mx_act_parts(y.cpu())
    assert np.array_equal(hpl[PAD:PAD + rows].cpu().numpy().view(np.uint16), h16.view(np.uint16))
    sl_ = lambda q: q[:, PAD:PAD + rows].permute(1, 0, 2).reshape(rows, Cc // 32)      # noqa: E731
    for i, (codes, sb) in enumerate(((ch, sh), (cl, sl))):
        assert np.array_equal(q4[i][PAD:PAD + rows].cpu().numpy(), codes), i
        assert np.array_equal(sl_(qs[i]).cpu().numpy(), sb), i
    assert not hpl[PAD:PAD + rows].cpu().numpy()[~valid.bool().cpu().numpy()].any()
    assert float(hpl[:PAD].float().min()) == 3.0 and float(hpl[PAD + rows:].float().max()) == 3.0            # slack rows untouched
