"""CPU tests of the weight packer (emotivoice_amd/packer.py): weight-norm folding (both key styles),
[N][taps][K] conv layout, the 3-tap polyphase form of ConvTranspose1d, q/k/v fusion, embed_projection1
split, blob table round trip.  The GEMM kernel's arithmetic contract is emulated in numpy here
(out[m,n] = sum_{tap,k} A[m + (tap-center)*dil, k] * W[n][tap][k])."""
import json
import struct

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from emotivoice_amd import packer
from emotivoice_amd.config import EVShapes
from oracle import synth_state_dict
from oracle.jets_oracle import fold_weight_norm, to_torch_sd


def emu_conv_gemm(a, w, center, dil):
    """a [M,K] (rows outside read as zero), w [N][taps][K] -> [M,N]"""
    M, K = a.shape
    N, taps, _ = w.shape
    out = np.zeros((M, N), np.float64)
    for t in range(taps):
        sh = (t - center) * dil
        src = np.zeros((M, K))
        lo, hi = max(0, -sh), min(M, M - sh)
        src[lo:hi] = a[lo + sh:hi + sh]
        out += src @ w[:, t, :].T.astype(np.float64)
    return out


def parse_blob(blob):
    assert blob[:4] == b"EVW1"
    n = struct.unpack_from("<I", blob, 8)[0]
    out = {}
    for i in range(n):
        name, dtype, ndim, d0, d1, d2, d3, off, nbytes = packer._ENTRY.unpack_from(blob, 16 + i * packer._ENTRY.size)
        name = name.rstrip(b"\0").decode()
        dims = [d0, d1, d2, d3][:ndim]
        arr = np.frombuffer(blob, np.float16 if dtype == 0 else np.float32, int(np.prod(dims)), off).reshape(dims)
        assert off % 256 == 0 and nbytes == arr.nbytes
        out[name] = arr
    return out


@pytest.fixture(scope="module")
def packed():
    sd = synth_state_dict(0, "parity")
    blob, man = packer.pack_state_dict(sd, pe_len=512)
    return sd, to_torch_sd(sd), parse_blob(blob), json.loads(man)


def test_blob_table_and_manifest(packed):
    sd, tsd, t, man = packed
    assert set(t) == set(man)
    assert t["pe"].shape == (512, 384) and t["tok_emb"].shape == (502, 384) and t["spk_emb"].shape == (2014, 384)
    assert t["enc.0.qkv.w32"].shape == (1152, 1, 384) and t["dec.3.ffn2.w16"].shape == (384, 3, 1536)
    assert t["to_mel.w16"].shape == (96, 1, 384) and np.all(t["to_mel.w32"][80:] == 0)
    assert t["voc.pre.w16"].shape == (512, 7, 96) and np.all(t["voc.pre.w16"][:, :, 80:] == 0)
    assert t["voc.up0.w16"].shape == (8 * 256, 3, 512) and t["voc.up3.w16"].shape == (2 * 32, 3, 64)
    assert t["voc.rb11.c2.2.w16"].shape == (32, 11, 32) and t["voc.post.w"].shape == (7, 32)
    for k, v in t.items():
        if k.endswith(("mx",)):            # .wmx / .wpmx / .wcmx
            continue                                   # opaque fp4 planes stored under an fp16-typed entry (test_mx_weight_planes)
        assert np.isfinite(v.astype(np.float32)).all(), k


def test_pe_table_matches_reference_formula(packed):
    from oracle.jets_oracle import sinusoid_table
    assert np.array_equal(packed[2]["pe"], sinusoid_table(512, 384).numpy())


def test_qkv_fusion_and_projection_split(packed):
    sd, tsd, t, _ = packed
    x = np.random.default_rng(0).standard_normal((5, 384)).astype(np.float32)
    got = emu_conv_gemm(x, t["dec.1.qkv.w32"], 0, 1) + t["dec.1.qkv.b"]
    for i, nm in enumerate("qkv"):
        ref = F.linear(torch.from_numpy(x), tsd[f"am.decoder.encoders.1.self_attn.linear_{nm}.weight"],
                       tsd[f"am.decoder.encoders.1.self_attn.linear_{nm}.bias"]).numpy()
        assert np.allclose(got[:, i * 384:(i + 1) * 384], ref, atol=1e-5)
    cond = np.random.default_rng(1).standard_normal(1920).astype(np.float32)
    full = np.concatenate([np.tile(x, 1), np.tile(cond, (5, 1))], 1)
    ref = F.linear(torch.from_numpy(full), tsd["am.embed_projection1.weight"], tsd["am.embed_projection1.bias"]).numpy()
    got = emu_conv_gemm(x, t["proj.w32"], 0, 1) + (t["proj.wcond"].astype(np.float64) @ cond + t["proj.b"])
    assert np.allclose(got, ref, atol=2e-5)


@pytest.mark.parametrize("rb,grp,d", [(0, "c1", 2), (4, "c2", 1), (11, "c1", 1)])
def test_resblock_conv_layout_and_weight_norm(packed, rb, grp, d):
    sd, tsd, t, _ = packed
    shapes = EVShapes()
    k = shapes.rb_kernels[rb % 3]
    dil = shapes.rb_dils[rb % 3][d] if grp == "c1" else 1
    w = t[f"voc.rb{rb}.{grp}.{d}.w16"].astype(np.float32)
    C = w.shape[0]
    x = np.random.default_rng(2).standard_normal((70, C)).astype(np.float32)
    pre = f"generator.resblocks.{rb}.{'convs1' if grp == 'c1' else 'convs2'}.{d}"
    wref = fold_weight_norm(tsd, pre).half().float()
    ref = F.conv1d(torch.from_numpy(x).t().unsqueeze(0), wref, None, dilation=dil, padding=(k - 1) // 2 * dil).squeeze(0).t().numpy()
    got = emu_conv_gemm(x, w, (k - 1) // 2, dil)
    assert np.allclose(got, ref, atol=3e-3)


@pytest.mark.parametrize("i", [0, 1, 2, 3])
def test_conv_transpose_polyphase_equals_torch(packed, i):
    sd, tsd, t, _ = packed
    shapes = EVShapes()
    s = shapes.up_rates[i]
    w = t[f"voc.up{i}.w16"].astype(np.float32)       # [s*cout][3][cin]
    cin, cout = w.shape[2], w.shape[0] // s
    x = np.random.default_rng(3).standard_normal((19, cin)).astype(np.float32)
    wt = fold_weight_norm(tsd, f"generator.ups.{i}").half().float()
    ref = F.conv_transpose1d(torch.from_numpy(x).t().unsqueeze(0), wt, tsd[f"generator.ups.{i}.bias"], stride=s,
                             padding=(shapes.up_kernels[i] - s) // 2).squeeze(0).t().numpy()
    got = (emu_conv_gemm(x, w, 1, 1) + t[f"voc.up{i}.b"]).reshape(19 * s, cout)
    assert got.shape == ref.shape
    assert np.allclose(got, ref, atol=3e-3)


def test_legacy_weight_g_v_keys_and_module_prefix(packed):
    sd, _, t, _ = packed
    legacy = {}
    for k, v in sd.items():
        k2 = k.replace(".parametrizations.weight.original0", ".weight_g").replace(".parametrizations.weight.original1", ".weight_v")
        legacy["module." + k2] = v
    blob2, _ = packer.pack_state_dict(legacy, pe_len=512)
    t2 = parse_blob(blob2)
    for k in t:
        assert t[k].tobytes() == t2[k].tobytes(), k          # (bytes: the fp4 planes are not numbers)


def test_missing_key_raises(packed):
    sd = dict(packed[0])
    del sd["am.to_mel.bias"]
    with pytest.raises(KeyError):
        packer.pack_state_dict(sd, pe_len=64)


def test_mx_weight_planes(packed):
    """The "mx" precision's fp4 planes: present exactly for the generator convs the MX kernels take (N, K % 128 == 0: stages 0-1 and their
    up-convs; C = 32: the fused pair kernel), bytes = mxfp4's packers applied to the folded fp32 weight, and the dequantised planes
    reproduce lo = w - fp16(w) and hi = fp16(w) to fp4 accuracy (relative L2 ~0.1: two significant bits)."""
    from emotivoice_amd import mxfp4
    sd, tsd, t, man = packed
    mx = sorted(k for k in t if k.endswith(".wmx"))
    pmx = sorted(k for k in t if k.endswith(".wpmx"))
    assert [k for k in mx if k.startswith("voc.up")] == ["voc.up0.wmx", "voc.up1.wmx", "voc.up2.wmx"]
    # decoder: the conv-FFN (3 taps: conv_gemm_mx_kernel) and, since round 4, the QKV / output projections (1 tap: gemm_mx1_kernel)
    assert sorted(k for k in mx if k.startswith("dec.")) == sorted(["dec.%d.ffn%d.wmx" % (i, j) for i in range(4) for j in (1, 2)] +
                                                                   ["dec.%d.%s.wmx" % (i, n) for i in range(4) for n in ("qkv", "out")])
    assert not [k for k in mx if k.startswith("enc.")]              # the token-rate encoder stays fp32-class (bit-exact durations)
    wq = t["dec.0.qkv.w16"].astype(np.float32) + t["dec.0.qkv.w32l"].astype(np.float32) / 2048.0
    assert wq.shape == (1152, 1, 384)
    blob = t["dec.0.qkv.wmx"].view(np.uint8)
    want = mxfp4.pack_weight_planes(wq)
    assert blob.size in (want.size, want.size + 1)
    ql, qh = mxfp4.weight_planes_dequant(blob[:want.size], 1152, 1, 384)
    w16q = t["dec.0.qkv.w16"].astype(np.float32)
    assert np.linalg.norm(qh - w16q) / np.linalg.norm(w16q) < 0.2 and np.linalg.norm(ql - (wq - w16q)) / np.linalg.norm(wq - w16q) < 0.25
    assert len([k for k in mx if ".rb" in k]) == 2 * 3 * 3 * 2 and all(int(k.split(".")[1][2:]) < 6 for k in mx if ".rb" in k)
    cmx = sorted(k for k in t if k.endswith(".wcmx"))            # C = 64: stage 2 + the last up-conv
    assert len(cmx) == 3 * 3 * 2 + 1 and "voc.up3.wcmx" in cmx and all(6 <= int(k.split(".")[1][2:]) < 9 for k in cmx if ".rb" in k)
    assert len(pmx) == 3 * 3 * 2 and all(int(k.split(".")[1][2:]) >= 9 for k in pmx)
    for name, shape in (("voc.rb0.c1.1", (256, 3, 256)), ("voc.rb5.c2.0", (128, 11, 128)), ("voc.up1", (1024, 3, 256))):
        w16 = t[name + ".w16"].astype(np.float32)
        lo = t[name + ".w16l"].astype(np.float32) / 2048.0
        w = w16 + lo                                                        # the folded fp32 weight up to 2^-22
        assert w16.shape == shape
        blob = t[name + ".wmx"].view(np.uint8)
        want = mxfp4.pack_weight_planes(w)
        assert blob.size in (want.size, want.size + 1)
        ql, qh = mxfp4.weight_planes_dequant(blob[:want.size], *shape)
        assert np.linalg.norm(qh - w16) / np.linalg.norm(w16) < 0.2
        assert np.linalg.norm(ql - lo) / np.linalg.norm(lo) < 0.25
    w16 = t["voc.rb11.c1.2.w16"].astype(np.float32)
    ql, qh = mxfp4.pair_weight_planes_dequant(t["voc.rb11.c1.2.wpmx"].view(np.uint8), 11)
    assert np.linalg.norm(qh - w16) / np.linalg.norm(w16) < 0.2 and ql.shape == (32, 11, 32)
