"""End-to-end parity on MI355X: libevhip.so (through the C ABI) vs the CPU oracle and the committed
golden fixtures (outputs of the reference itself).

Tolerances (BASELINE.json north_star): mel / waveform <= 1e-3 relative L2 against the fp32 reference;
durations / mel lengths bit-exact.  Stage taps of the fp32 token-rate path are held to 1e-4.

Two precisions of the frame-rate path are tested (include/evhip.h, EV_PREC_*):
  * "strict" (decoder + generator in split precision, fp32 activations): every output and every Appendix-C tap is held to
    TOL_STRICT = 2e-5 (measured <= 3.1e-6) -- fifty times inside the contract, also on the DC-free fixture;
  * "mx" (the contract mode: strict's data flow, cross terms of the >= 128-channel generator layers as block-scaled fp4 MFMAs):
    mel as strict; waveform AND its DC-free measure <= FAST_HOT = 6e-3         # fp16 operands on the trained-like ("_hot") zero-mean fixtures: same mechanism, larger activations (weight seed 0: 2.6e-3; the second draw of
                        # the weights, round 6, is 1.7x harder in EVERY mode: fast 4.3e-3, mx 8.6e-4, strict 5e-6)
TOL_MX = 1e-3 on every fixture, the zero-mean ones included (no exemption);
  * "fast" (fp16 MFMA operands, the precision BASELINE.json's bf16 / fp16 configs name): mel and waveform <= 1e-3 on the
    synthetic-weight fixtures, whose waveform carries a DC offset ~3x its AC amplitude.  On a zero-mean waveform
    (tests/golden/n28_zero_dc.npz) fp16 operands measure ~2.2e-3: tools/precision_study.py attributes that evenly to the ~150
    fp16 roundings of weights and stored activations (no stage dominates), so it is asserted at its measured level, FAST_ZDC.
No test skips an utterance: where a predicted duration sits within NEAR_EPS of a rounding boundary and flips, the utterance is
re-synthesised with the oracle's durations (EV_FLAG_FORCED_DURATIONS) and compared all the same; the flipped tokens are counted.
"""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT, rel_l2

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

TOL_OUT = 1e-3          # north_star tolerance for mel and waveform
TOL_F32_TAP = 1e-4      # fp32 token-rate taps
TOL_STRICT = 2e-5       # every frame-rate quantity in the split-precision mode (measured <= 3.1e-6 over all tests)
FAST_ZDC = 3e-3         # fp16 operands on a zero-mean waveform (measured 2.2e-3; see module docstring)
FAST_HOT = 6e-3         # fp16 operands on the trained-like ("_hot") zero-mean fixtures: same mechanism, larger activations (weight seed 0: 2.6e-3; the second draw of
                        # the weights, round 6, is 1.7x harder in EVERY mode: fast 4.3e-3, mx 8.6e-4, strict 5e-6)
TOL_MX = 1e-3           # "mx" mode: the north_star bound on EVERY fixture, zero-mean ones included, on the DC-free measure too
NEAR_EPS = 2e-5         # |frac(exp(log_d) - 1) - 0.5| below which a duration may legitimately flip (log_d agrees to ~1e-6)
REPORT = {}


def _report(key, val):
    REPORT[key] = val
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_report.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


def rel_l2_ac(a, b):
    """relative L2 after removing the mean of the reference (DC-insensitive, stricter for wav)."""
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    mu = b.mean()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b - mu), 1e-30))


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.set_num_threads(min(16, os.cpu_count() or 1))      # the CPU oracle: more threads are slower on the 2 x 64-core box
    return True


_CACHE = {}
MODES = {          # name -> (decoder_precision, vocoder_precision)
    "fast": ("f16", "f16"),
    "f32dec": ("f32", "f16"),      # exact-fp32 decoder (debug mode for indexing) + fp16 generator
    "strict": ("x3", "x3"),
    "mx": ("mx", "mx"),            # the contract mode: fp4 cross terms in the decoder's conv-FFN and the generator; the rest split precision
    "mx32": ("mx", "mx"),          # the same with ev_config.mx_residual = 1: a separate fp32 residual tensor beside the planes (round 3's flow)
}


def _weights(mode):
    """mode = a synthetic.py weight recipe, optionally "<recipe>@<weight seed>" (round 6: the suite runs three draws of the weights; no suffix = seed 0)."""
    from oracle import synth_state_dict
    from oracle.jets_oracle import to_torch_sd
    from emotivoice_amd.packer import pack_state_dict
    if mode not in _CACHE:
        recipe, _, seed = mode.partition("@")
        sd = synth_state_dict(int(seed or 0), recipe)
        blob, man = pack_state_dict(sd)
        _CACHE[mode] = (to_torch_sd(sd), blob, man)
    return _CACHE[mode]


def _engine(wmode, prec="fast", keep=True):
    from emotivoice_amd.engine import EVEngine
    key = ("eng", wmode, prec, keep)
    if key not in _CACHE:
        dp, vp = MODES[prec]
        # "mx32": round 3's flow -- fp32 residual stream, and (round 4) the token-rate conv-FFN in one pass instead of split-K: the opt-outs stay under test
        # ... and (round 6) block-scaled fp4 activation operands in the fused 32-channel k = 3 pairs instead of E5M2 (ev_config.mx_act_format = 1), one launch per conv
        # instead of the grouped levels (ev_config.mx_group = 1)
        eng = EVEngine(decoder_precision=dp, vocoder_precision=vp, keep_stages=keep, mx_residual="fp32" if prec == "mx32" else "planes",
                       token_splitk=prec != "mx32", mx_act_format="fp4" if prec == "mx32" else "e5m2", mx_group=prec != "mx32")
        _, blob, man = _weights(wmode)
        eng.load_blob(blob, man)
        _CACHE[key] = eng
    return _CACHE[key]


def _drop_engines(only_big=True):
    """Close cached engines (their workspaces stay at the largest batch they have seen): tests that are about to run a BASELINE config at full
    size in several precisions call this first -- three 128-mel workspaces on top of the cached ones exceeded the 288 GB in round 3."""
    for key in [k for k in _CACHE if isinstance(k, tuple) and k[0] == "eng" and (not only_big or not k[3])]:
        _CACHE.pop(key).close()


def _oracle(wmode, utt, taps=None, durations=None, vocoder=True):
    from oracle import EVShapes, am_forward, jets_forward
    sd, _, _ = _weights(wmode)
    if taps is None and durations is None:          # the plain per-utterance reference is shared by the precision modes of a test (CPU seconds each)
        key = ("ref", wmode, np.asarray(utt["ling"]).tobytes(), int(utt["speaker"]), np.asarray(utt["style"]).tobytes()[:64], vocoder)
        full = key[:-1] + (True,)                   # (a reference with the waveform serves a mel-only request too)
        if key not in _CACHE and full in _CACHE:
            return _CACHE[full]
        if key not in _CACHE:
            _CACHE[key] = _oracle_uncached(sd, utt, vocoder)
        return _CACHE[key]
    return _oracle_uncached(sd, utt, vocoder, taps, durations)


def _drop_refs():
    for key in [k for k in _CACHE if isinstance(k, tuple) and k[0] == "ref"]:
        _CACHE.pop(key)


def _oracle_uncached(sd, utt, vocoder, taps=None, durations=None):
    from oracle import EVShapes, am_forward, jets_forward
    if not vocoder:
        with torch.no_grad():
            return am_forward(sd, torch.from_numpy(np.asarray(utt["ling"])).long(), int(utt["speaker"]), torch.from_numpy(utt["style"]).float(),
                              torch.from_numpy(utt["content"]).float(), EVShapes(), taps=taps, durations=durations)
    return jets_forward(sd, utt["ling"], utt["speaker"], utt["style"], utt["content"], EVShapes(), taps=taps, durations=durations)


def _near_boundary(log_d, eps=NEAR_EPS):
    v = np.exp(np.asarray(log_d, np.float64)) - 1.0
    frac = v - np.floor(v)
    return np.abs(frac - 0.5) < eps


def _check_durations(got, ref_dur, ref_logd):
    """Durations must be bit-exact except on tokens whose pre-round value is within NEAR_EPS of a rounding boundary.
    Returns the number of flipped (forgiven) tokens."""
    got, ref_dur = np.asarray(got), np.asarray(ref_dur)
    diff = got != ref_dur
    near = _near_boundary(ref_logd)
    assert not (diff & ~near).any(), "duration differs away from a rounding boundary: %s" % np.nonzero(diff & ~near)[0][:8]
    return int(diff.sum())


def _tap_stride(g, prefix):
    for k in g.files:
        if k.startswith(prefix + "__ax1_s"):
            return int(k.rsplit("_s", 1)[1])
    return 0


GOLDEN = sorted(p for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")) if not os.path.basename(p).startswith("simbert_"))


@pytest.mark.parametrize("prec", ["fast", "f32dec", "strict", "mx"])
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_golden_fixture(gpu, path, prec):
    """HIP path vs outputs of the reference itself (tests/golden/make_golden.py)."""
    g = np.load(path)
    wseed = int(g["weight_seed"])
    eng = _engine(str(g["dur_mode"]) + ("@%d" % wseed if wseed else ""), prec)
    utt = dict(ling=g["in_ling"], speaker=int(g["in_speaker"]), style=g["in_style"], content=g["in_content"])
    # (a fixture generated with alpha != 1 pins that the reference's inference branch ignores alpha: the engine runs at 1.0)
    out = eng.synthesize([utt])
    name = os.path.basename(path)[:-4] + "/" + prec
    assert not _near_boundary(g["log_dur"], 2e-4).any(), "fixture has a duration on a rounding boundary"
    assert np.array_equal(out["durations"], g["dur"]), name
    assert int(out["mel_lens"][0]) == int(g["mel_len"])
    e = dict(log_dur=rel_l2(out["log_durations"], g["log_dur"]), pitch=rel_l2(out["pitch"], g["pitch"]),
             energy=rel_l2(out["energy"], g["energy"]), mel=rel_l2(out["mel"], g["mel"]), wav=rel_l2(out["wav"], g["wav"]),
             wav_ac=rel_l2_ac(out["wav"], g["wav"]), wav_mean_over_std=float(abs(g["wav"].mean()) / g["wav"].std()))
    _report("golden/" + name, e)
    assert e["log_dur"] < TOL_F32_TAP and e["pitch"] < TOL_F32_TAP and e["energy"] < TOL_F32_TAP, e
    zero_dc = "_zdc" in str(g["dur_mode"])
    hot = str(g["dur_mode"]).endswith("_hot")       # trained-like generator gains: stage activations up to 2.5e3 (synthetic.HOT_*)
    if hot:
        assert np.abs(g["tap_voc_up3__ax1_s%d" % _tap_stride(g, "tap_voc_up3")]).max() > 300.0 if _tap_stride(g, "tap_voc_up3") else True
    if prec == "strict":
        assert e["mel"] < TOL_STRICT and e["wav"] < TOL_STRICT and e["wav_ac"] < TOL_STRICT, e
    elif prec == "mx":
        assert e["mel"] < TOL_MX and e["wav"] < TOL_MX and e["wav_ac"] < TOL_MX, e
    else:
        assert e["mel"] < (TOL_OUT if prec == "fast" else TOL_STRICT), e
        # fp16 storage neither saturates nor flushes at trained-like magnitudes: the error stays at the fp16-rounding level
        assert e["wav"] < (FAST_HOT if hot else FAST_ZDC if zero_dc else TOL_OUT), e
    assert out["wav"].shape[0] == 256 * int(g["mel_len"])
    assert np.isfinite(out["wav"]).all() and np.abs(out["wav"]).max() <= 1.0


F32_TAPS = ["tok_emb", "enc_l0", "enc_l1", "enc_l2", "enc_l3", "enc_out", "x_proj", "x_var", "upsampled"]
DEC_TAPS = ["dec_l0", "dec_l1", "dec_l2", "dec_l3", "dec_out", "mel"]
VOC_TAPS = ["voc_pre", "voc_up0", "voc_mrf0", "voc_up1", "voc_mrf1", "voc_up2", "voc_mrf2", "voc_up3", "voc_mrf3"]


@pytest.mark.parametrize("wmode", ["parity", "parity_zdc"])
@pytest.mark.parametrize("prec", ["fast", "f32dec", "strict", "mx", "mx32"])
def test_stage_taps_vs_oracle(gpu, prec, wmode):
    """Every Appendix-C stage tap of one 48-phoneme utterance against the oracle (plain and zero-DC weights)."""
    from oracle import synth_inputs
    eng = _engine(wmode, prec)
    utt = synth_inputs(21, [48], [7])[0]
    taps = {}
    ref = _oracle(wmode, utt, taps)
    out = eng.synthesize([utt])
    assert np.array_equal(out["durations"], ref["log_duration_predictions"].numpy())
    errs = {}
    for name in F32_TAPS + DEC_TAPS:
        got = eng.get_stage(name).reshape(taps[name].shape)
        errs[name] = rel_l2(got, taps[name].numpy())
    for name in VOC_TAPS:
        r = taps[name].numpy().T     # oracle keeps (C, L); engine is channels-last (L, C)
        got = eng.get_stage(name).reshape(r.shape)
        errs[name] = rel_l2(got, r)
    errs["wav"] = rel_l2(out["wav"], taps["wav"].numpy())
    errs["wav_ac"] = rel_l2_ac(out["wav"], taps["wav"].numpy())
    _report("taps/%s/%s" % (wmode, prec), errs)
    for name in F32_TAPS:
        assert errs[name] < TOL_F32_TAP, (name, errs)
    dec_tol = TOL_OUT if prec == "fast" else TOL_MX if prec in ("mx", "mx32") else TOL_STRICT
    for name in DEC_TAPS:
        assert errs[name] < dec_tol, (name, errs)
    if prec == "strict":
        for name in VOC_TAPS + ["wav", "wav_ac"]:
            assert errs[name] < TOL_STRICT, (name, errs)
    elif prec in ("mx", "mx32"):
        for name in VOC_TAPS + ["wav", "wav_ac"]:
            assert errs[name] < TOL_MX, (name, errs)
    else:
        # fp16 generator: its intermediate taps are fp16 tensors in HBM and sit at 1-2e-3 (they carry no DC to hide behind);
        # the contract figure is the waveform
        for name in VOC_TAPS:
            assert errs[name] < 2e-3, (name, errs)
        assert errs["wav"] < (FAST_ZDC if wmode.endswith("_zdc") else TOL_OUT), errs


def _tols(prec):
    """(mel tolerance, waveform tolerance) of a precision mode."""
    return {"strict": (TOL_STRICT, TOL_STRICT), "mx": (TOL_MX, TOL_MX), "mx32": (TOL_MX, TOL_MX)}.get(prec, (TOL_OUT, TOL_OUT))


def _tol_ac(prec):
    """bound on the DC-free waveform measure (the parity weights carry a DC offset of ~3x the AC amplitude, which flatters the plain one):
    the contract modes must meet it, the fp16 mode is asserted on the plain measure only (2.4e-3 DC-free, module docstring)."""
    return {"strict": TOL_STRICT, "mx": TOL_MX, "mx32": TOL_MX}.get(prec)


def _compare_utterances(eng, wmode, utts, out, wav_idx, tol, tol_wav, tag, tol_ac=None):
    """Per-utterance check of a batch result against the oracle: durations (bit-exact up to boundary flips, which are
    counted and re-run with forced durations), mel of every utterance, waveform of the utterances in wav_idx."""
    cu = out["cu_seqlens"]
    rep = dict(n=len(utts), flipped_tokens=0, forced_reruns=0, mel_max=0.0, wav_max=0.0, wav_ac_max=0.0, near_tokens=0)
    for b, u in enumerate(utts):
        need_wav = b in wav_idx
        ref = _oracle(wmode, u, vocoder=need_wav)
        logd = ref["log_dur_raw"].numpy()
        rdur = ref["log_duration_predictions"].numpy()
        d = out["durations"][cu[b]:cu[b + 1]]
        rep["near_tokens"] += int(_near_boundary(logd).sum())
        flips = _check_durations(d, rdur, logd)
        if flips:
            rep["flipped_tokens"] += flips
            rep["forced_reruns"] += 1
            solo = eng.synthesize([u], forced_durations=rdur, vocoder=need_wav)
            mel, wav = solo["mel"], solo.get("wav")
        else:
            assert int(out["mel_lens"][b]) == int(ref["mel_len"]), b
            mel, wav = out["mel_list"][b], (out["wav_list"][b] if need_wav else None)
        em = rel_l2(mel, ref["dec_outputs"].numpy())
        rep["mel_max"] = max(rep["mel_max"], em)
        assert em < tol, (tag, b, em)
        if need_wav:
            ew, ea = rel_l2(wav, ref["wav_predictions"].numpy()), rel_l2_ac(wav, ref["wav_predictions"].numpy())
            rep["wav_max"], rep["wav_ac_max"] = max(rep["wav_max"], ew), max(rep["wav_ac_max"], ea)
            assert ew < tol_wav, (tag, b, ew)
            assert tol_ac is None or ea < tol_ac, (tag, b, "wav_ac", ea)
    # the forgiveness window is ~20x the measured log-duration error: over a whole BASELINE config it may catch a handful of
    # tokens, and only a fraction of those flip
    assert rep["flipped_tokens"] <= max(2, rep["near_tokens"]), rep
    _report(tag, rep)
    return rep


@pytest.mark.parametrize("prec", ["fast", "strict", "mx"])
def test_ragged_batch_equals_per_utterance_reference(gpu, prec):
    """Reference semantics are B = 1 per utterance: a ragged batch must reproduce the
    per-utterance oracle, including utterances that straddle GEMM tile boundaries and a 1-phoneme utterance."""
    from oracle import synth_inputs
    eng = _engine("parity", prec)
    lens = [64, 9, 130, 1, 257, 40]
    utts = synth_inputs(31, lens, [0, 3, 2013, 77, 5, 1000])
    out = eng.synthesize(utts)
    tol, tol_wav = _tols(prec)
    _compare_utterances(eng, "parity", utts, out, set(range(len(utts))), tol, tol_wav, "ragged/" + prec, _tol_ac(prec))


@pytest.mark.parametrize("prec", ["fast", "strict", "mx"])
def test_batch_invariance_bit_exact(gpu, prec):
    """An utterance synthesised alone and inside a batch gives bit-identical outputs (per-utterance B=1
    semantics; no cross-utterance leakage through conv halos, attention or the length regulator)."""
    from oracle import synth_inputs
    eng = _engine("parity", prec)
    tol = TOL_STRICT if prec == "strict" else TOL_OUT
    utts = synth_inputs(41, [50, 120, 33], [1, 2, 3])
    batch = eng.synthesize(utts)
    wavs = [w.copy() for w in batch["wav_list"]]
    mels = [m.copy() for m in batch["mel_list"]]
    cu = batch["cu_seqlens"]
    durs = [batch["durations"][cu[b]:cu[b + 1]].copy() for b in range(3)]
    for b, u in enumerate(utts):
        solo = eng.synthesize([u])
        assert np.array_equal(solo["durations"], durs[b])
        assert np.array_equal(solo["mel"], mels[b]), b
        assert np.array_equal(solo["wav"], wavs[b]), b


@pytest.mark.parametrize("prec", ["fast", "strict", "mx"])
def test_shortest_utterances(gpu, prec):
    """1-, 2- and 3-phoneme utterances (fewer rows than any conv's taps, one attention key) inside a batch, against the
    oracle (which matches the reference on exactly these inputs: 0 mel difference on CPU)."""
    from oracle import synth_inputs
    eng = _engine("parity", prec)
    tol, tol_wav = _tols(prec)
    utts = [synth_inputs(30 + n, [n], [n])[0] for n in (1, 2, 3)] + synth_inputs(34, [17], [9])
    out = eng.synthesize(utts)
    cu = out["cu_seqlens"]
    for b, u in enumerate(utts):
        ref = _oracle("parity", u)
        assert np.array_equal(out["durations"][cu[b]:cu[b + 1]], ref["log_duration_predictions"].numpy()), b
        assert int(out["mel_lens"][b]) == int(ref["mel_len"])
        assert rel_l2(out["mel_list"][b], ref["dec_outputs"].numpy()) < tol, b
        assert rel_l2(out["wav_list"][b], ref["wav_predictions"].numpy()) < tol_wav, b


@pytest.mark.parametrize("prec", ["mx", "fast", "strict"])
def test_forced_durations_and_zero_duration_guard(gpu, prec):
    """Teacher-forced durations incl. zeros, and the all-zero guard of alignment.py:187-191."""
    from oracle import synth_inputs
    eng = _engine("parity", prec)
    tol = _tols(prec)[0]
    utt = synth_inputs(51, [20], [9])[0]
    dur = np.array([0, 3, 0, 0, 7, 1, 2, 0, 5, 4, 0, 0, 0, 6, 2, 2, 1, 0, 9, 3], np.int64)
    ref = _oracle("parity", utt, durations=torch.from_numpy(dur))
    out = eng.synthesize([utt], forced_durations=dur)
    assert int(out["mel_lens"][0]) == int(dur.sum()) == int(ref["mel_len"])
    assert rel_l2(out["mel"], ref["dec_outputs"].numpy()) < tol
    assert rel_l2(out["wav"], ref["wav_predictions"].numpy()) < tol
    zero = np.zeros(20, np.int64)
    ref0 = _oracle("parity", utt, durations=torch.from_numpy(zero))
    out0 = eng.synthesize([utt], forced_durations=zero)
    assert int(out0["mel_lens"][0]) == 20 == int(ref0["mel_len"])     # every duration becomes 1
    assert rel_l2(out0["mel"], ref0["dec_outputs"].numpy()) < tol


@pytest.mark.parametrize("prec", ["fast", "strict", "mx"])
def test_vocoder_only_and_int16(gpu, prec):
    """ev_vocoder on oracle mels (fp32 and fp16 inputs, ragged) + the caller's int16 epilogue."""
    from oracle import EVShapes, hifigan_forward, synth_inputs
    from oracle.jets_oracle import wav_to_int16
    eng = _engine("parity", prec)
    tol = _tols(prec)[1]
    sd, _, _ = _weights("parity")
    rng = np.random.default_rng(5)
    mels = [(1.25 * rng.standard_normal((80, T)) + 0.08).astype(np.float32) for T in (37, 5, 150)]
    refs = [hifigan_forward(sd, torch.from_numpy(m), EVShapes()).numpy() for m in mels]
    out = eng.vocoder(mels, want_int16=True)
    errs = [rel_l2(w, r) for w, r in zip(out["wav_list"], refs)]
    _report("vocoder_only_f32in/" + prec, errs)
    assert max(errs) < tol, errs
    assert np.array_equal(out["wav_i16"], wav_to_int16(out["wav"]))
    out16 = eng.vocoder([m.astype(np.float16) for m in mels])
    refs16 = [hifigan_forward(sd, torch.from_numpy(m.astype(np.float16).astype(np.float32)), EVShapes()).numpy() for m in mels]
    assert max(rel_l2(w, r) for w, r in zip(out16["wav_list"], refs16)) < tol


def test_config2_shape_properties(gpu):
    """BASELINE configs[1] at full size with the BENCH weights (B=32 x 256 phonemes, exactly 4 frames / phoneme): exact
    durations and lengths, bounded finite audio, first / last utterance bit-identical to their stand-alone synthesis."""
    from oracle import synth_inputs
    eng = _engine("bench", "fast", keep=False)
    utts = synth_inputs(1, [256] * 32, [0] * 32)
    out = eng.synthesize(utts)
    assert (out["durations"] == 4).all()
    assert (out["mel_lens"] == 1024).all()
    assert out["wav"].shape[0] == 32 * 1024 * 256
    assert np.isfinite(out["wav"]).all() and np.abs(out["wav"]).max() <= 1.0
    first, last = out["wav_list"][0].copy(), out["wav_list"][31].copy()
    assert np.array_equal(eng.synthesize([utts[0]])["wav"], first)
    assert np.array_equal(eng.synthesize([utts[31]])["wav"], last)


@pytest.mark.parametrize("prec", ["fast", "strict", "mx"])
def test_config2_every_utterance_vs_oracle(gpu, prec):
    """BASELINE configs[1] (B = 32 x 256 phonemes, one speaker) with the PARITY weights (predicted durations vary, the
    round / prefix-sum path is live): durations and mel of ALL 32 utterances and the waveform of 8 of them against the CPU
    oracle.  No utterance is skipped (boundary flips are re-run with forced durations and counted)."""
    from oracle import synth_inputs
    eng = _engine("parity", prec, keep=False)
    utts = synth_inputs(1, [256] * 32, [0] * 32)
    out = eng.synthesize(utts)
    tol, tol_wav = _tols(prec)
    rep = _compare_utterances(eng, "parity", utts, out, set(range(0, 32, 4)), tol, tol_wav, "config2_all/" + prec, _tol_ac(prec))
    assert rep["n"] == 32


@pytest.mark.parametrize("wseed", [1, 2])
@pytest.mark.parametrize("recipe", ["parity_zdc", "parity_zdc_hot"])
def test_config2_weight_draws(gpu, recipe, wseed):
    """BASELINE configs[1]'s shape on a SECOND and a THIRD draw of the weights (every other number of this suite is weight seed 0), contract mode,
    zero-mean waveforms, plain and trained-like generator gains: durations + mel of all 32 utterances, the waveform of 4 of them, against the CPU oracle.
    The worst DC-free waveform error over the draws is what README / DESIGN quote as the margin (report keys config2_draw/...)."""
    from oracle import synth_inputs
    wmode = "%s@%d" % (recipe, wseed)
    eng = _engine(wmode, "mx", keep=False)
    utts = synth_inputs(40 + wseed, [256] * 32, [0] * 32)
    out = eng.synthesize(utts)
    tol, tol_wav = _tols("mx")
    rep = _compare_utterances(eng, wmode, utts, out, {0, 9, 18, 27}, tol, tol_wav, "config2_draw/%s/w%d/mx" % (recipe, wseed), _tol_ac("mx"))
    assert rep["n"] == 32
    _drop_engines()
    _drop_refs()


@pytest.mark.parametrize("wseed", [3, 4, 5])
def test_further_trained_like_weight_draws(gpu, wseed):
    """Three more draws of the trained-like generator gains (round 6: draws 0-2 put the mx mode between 3.9e-4 and 8.3e-4, i.e. one draw decides the margin, so the
    suite keeps sampling): four 256-phoneme utterances per draw in the contract mode -- the large-batch path -- durations, mel and waveform of each against the CPU
    oracle.  Report keys hot_draw/w<seed>/mx; README / DESIGN quote the worst over all six draws."""
    from oracle import synth_inputs
    wmode = "parity_zdc_hot@%d" % wseed
    eng = _engine(wmode, "mx", keep=False)
    utts = synth_inputs(60 + wseed, [256] * 4, [0] * 4)
    out = eng.synthesize(utts)
    tol, tol_wav = _tols("mx")
    rep = _compare_utterances(eng, wmode, utts, out, {0, 1, 2, 3}, tol, tol_wav, "hot_draw/w%d/mx" % wseed, _tol_ac("mx"))
    assert rep["n"] == 4
    _drop_engines()
    _drop_refs()


@pytest.mark.parametrize("prec", ["mx", "fast", "strict"])
def test_config3_ragged_256_every_utterance(gpu, prec):
    """BASELINE configs[2] at full size: batch 256, lengths 64 + (i*7919 mod 449), speakers i mod 2000 (length-regulator
    ragged stress).  Size-independent properties (lengths consistent, bounded finite audio, shortest / longest / last utterance
    bit-identical to stand-alone synthesis) AND, against the CPU oracle: durations + mel of ALL 256 utterances, the waveform
    of 16 of them; nothing is skipped."""
    from oracle import synth_inputs
    eng = _engine("parity", prec, keep=False)
    lens = [64 + (i * 7919) % 449 for i in range(256)]
    utts = synth_inputs(3, lens, [i % 2000 for i in range(256)])
    out = eng.synthesize(utts)
    cu = out["cu_seqlens"]
    dsum = np.array([out["durations"][cu[b]:cu[b + 1]].sum() for b in range(256)])
    assert np.array_equal(dsum, out["mel_lens"]) and (out["durations"] >= 0).all()
    assert out["wav"].shape[0] == 256 * int(out["mel_lens"].sum()) and out["mel"].shape == (int(out["mel_lens"].sum()), 80)
    assert np.isfinite(out["wav"]).all() and np.abs(out["wav"]).max() <= 1.0 and np.isfinite(out["mel"]).all()
    picks = [int(np.argmin(lens)), int(np.argmax(lens)), 255]
    keep = {b: (out["wav_list"][b].copy(), out["mel_list"][b].copy()) for b in picks}
    out = dict(out, wav_list=[w.copy() for w in out["wav_list"]], mel_list=[m.copy() for m in out["mel_list"]],
               durations=out["durations"].copy(), mel_lens=out["mel_lens"].copy())       # the solo calls below re-use the handle
    for b in picks:
        solo = eng.synthesize([utts[b]])
        assert np.array_equal(solo["wav"], keep[b][0]) and np.array_equal(solo["mel"], keep[b][1]), b
    tol, tol_wav = _tols(prec)
    wav_idx = set([picks[0]] + list(range(1, 256, 17)))          # 16 utterances incl. the shortest
    rep = _compare_utterances(eng, "parity", utts, out, wav_idx, tol, tol_wav, "config3_all/" + prec, _tol_ac(prec))
    assert rep["n"] == 256
    _drop_engines()
    if prec == "strict":
        _drop_refs()            # (the last mode of this test: the 256 cached references are not needed again)


@pytest.mark.parametrize("prec", ["mx", "fast", "strict"])
def test_config5_vocoder_only_full_size(gpu, prec):
    """BASELINE configs[4] per-GPU share: 128 pre-computed 80 x 1024 fp16 mels through ev_vocoder.  (a) AGAINST THE ORACLE at full size:
    two of the eight distinct mels (262 144 samples each) vs hifigan_forward on the same fp16-rounded mel, in every precision mode
    (zero-mean weights: the DC-free measure is the contract figure); (b) the size-independent properties -- linearity does not hold
    for a GAN vocoder --: exact lengths, bounded finite audio, every mel's waveform bit-identical to the same mel anywhere else in the
    batch and vocoded alone (no cross-utterance leakage at any of the 4 upsampling stages)."""
    from oracle import EVShapes, hifigan_forward
    _drop_engines()
    eng = _engine("parity_zdc", prec, keep=False)
    sd, _, _ = _weights("parity_zdc")
    rng = np.random.default_rng(9)
    base = (1.25 * rng.standard_normal((8, 80, 1024)) + 0.08).astype(np.float16)
    mels = [base[i % 8] for i in range(128)]
    out = eng.vocoder(mels)
    assert out["wav"].shape[0] == 128 * 1024 * 256 and np.isfinite(out["wav"]).all() and np.abs(out["wav"]).max() <= 1.0
    first = [out["wav_list"][i].copy() for i in range(8)]
    for i in range(8, 128):
        assert np.array_equal(out["wav_list"][i], first[i % 8]), i          # identical mels -> identical audio anywhere in the batch
    errs = {}
    for i in (0, 5):
        ref = hifigan_forward(sd, torch.from_numpy(base[i].astype(np.float32)), EVShapes()).numpy()
        errs[i] = (rel_l2(first[i], ref), rel_l2_ac(first[i], ref))
    _report("config5_full_size/" + prec, {str(k): v for k, v in errs.items()})
    bound = {"strict": TOL_STRICT, "mx": TOL_MX, "fast": FAST_ZDC}[prec]
    assert max(max(v) for v in errs.values()) < bound, errs
    solo = eng.vocoder([mels[3]])
    assert np.array_equal(solo["wav"], first[3])
    _drop_engines()


@pytest.mark.parametrize("prec", ["mx", "fast", "strict"])
def test_long_utterance_extends_positional_table(gpu, prec):
    """Utterances longer than the packed sinusoid table (the reference auto-extends its table, encoder.py:216-237):
    forced durations of 12 frames x 400 phonemes = 4800 frames > 4096."""
    from oracle import synth_inputs
    eng = _engine("parity", prec, keep=False)
    tol = _tols(prec)[0]
    utt = synth_inputs(71, [400], [11])[0]
    dur = np.full(400, 12, np.int64)
    ref = _oracle("parity", utt, durations=torch.from_numpy(dur))
    out = eng.synthesize([utt], forced_durations=dur, vocoder=False)
    assert int(out["mel_lens"][0]) == 4800
    e = rel_l2(out["mel"], ref["dec_outputs"].numpy())
    _report("long_utt_mel/" + prec, e)
    assert e < tol


@pytest.mark.parametrize("prec", ["mx", "fast", "strict"])
def test_random_ragged_batches_are_batch_invariant(gpu, prec):
    """Randomised batches (lengths 1-300, random speakers, shuffled order): every utterance's mel / waveform / durations inside any
    batch are bit-identical to its stand-alone synthesis -- whatever tile, kernel generation or epilogue variant the batch's row
    counts select (the property that caught the fma-contraction difference between epilogue variants in round 2)."""
    from oracle import synth_inputs
    eng = _engine("parity", prec, keep=False)
    rng = np.random.default_rng(2024)
    pool_lens = [1, 2, 7, 31, 64, 65, 127, 128, 129, 200, 255, 256, 300]
    pool = synth_inputs(77, pool_lens, [int(s) for s in rng.integers(0, 2014, len(pool_lens))])
    solo = []
    for u in pool:
        r = eng.synthesize([u])
        solo.append((r["wav"].copy(), r["mel"].copy(), r["durations"].copy()))
    for trial in range(4):
        idx = rng.permutation(len(pool))[: int(rng.integers(2, len(pool) + 1))]
        out = eng.synthesize([pool[i] for i in idx])
        cu = out["cu_seqlens"]
        for b, i in enumerate(idx):
            assert np.array_equal(out["durations"][cu[b]:cu[b + 1]], solo[i][2]), (trial, int(i))
            assert np.array_equal(out["mel_list"][b], solo[i][1]), (trial, int(i))
            assert np.array_equal(out["wav_list"][b], solo[i][0]), (trial, int(i))


@pytest.mark.parametrize("prec", ["mx", "fast", "strict"])
def test_chunked_vocoding_is_bit_identical(gpu, prec):
    """Streaming vocoder (EVEngine.vocoder_chunked): chunks with 16 frames of context reproduce whole-utterance vocoding
    bit for bit; with too little context they do not (the receptive field is 14 frames per side)."""
    eng = _engine("parity", prec, keep=False)
    rng = np.random.default_rng(17)
    mel = (1.25 * rng.standard_normal((80, 700)) + 0.08).astype(np.float32)
    full = eng.vocoder([mel])["wav"]
    for chunk in (256, 100):
        parts = list(eng.vocoder_chunked(mel, chunk_frames=chunk))
        assert np.array_equal(np.concatenate(parts), full), chunk
    short = np.concatenate(list(eng.vocoder_chunked(mel, chunk_frames=256, context=8)))
    assert short.shape == full.shape and not np.array_equal(short, full)
    exact14 = np.concatenate(list(eng.vocoder_chunked(mel, chunk_frames=256, context=14)))
    assert np.array_equal(exact14, full)


@pytest.mark.parametrize("prec", [None, "fast"], ids=["default", "fast"])
def test_chunked_stage_execution_is_bit_identical(gpu, prec):
    """The generator's ResBlocks run on row chunks sized for the Infinity Cache (ev_config.vocoder_chunk_mb, overlapped
    tiling over the 6-conv chain of a ResBlock): any chunk size must give the same bits as whole-tensor execution, with chunk
    borders falling inside utterances and next to the zero gaps between them.  The chunked schedule exists for the fp16 flow ("fast");
    in the ABI's default precision (the plane-set flow) the field is accepted and must not change a bit either."""
    from emotivoice_amd.engine import EVEngine
    from oracle import synth_inputs
    _, blob, man = _weights("parity")
    utts = synth_inputs(21, [96, 33, 120, 64, 77, 128, 50, 101], [1, 2, 3, 4, 5, 6, 7, 8])
    outs = []
    for mb in (0, 1, 3, 7):       # whole tensors (default) / ~1 MB chunks (dozens of chunks per stage) / ~3 MB / ~7 MB
        eng = EVEngine(vocoder_chunk_mb=mb, precision=prec)
        eng.load_blob(blob, man)
        r = eng.synthesize(utts)
        outs.append(r["wav"].copy())
        eng.close()
    assert float(np.abs(outs[0]).max()) > 0
    for other in outs[1:]:
        assert other.shape == outs[0].shape and np.array_equal(outs[0], other)

