"""MI355X: remaining C-ABI surface in the ABI's DEFAULT precision (EVEngine() = ev_default_config() = the contract mode) -- duration scale alpha, device-pointer inputs on an external stream, borrowed device
weight blob (the post-broadcast path), exact-fp32 token-rate option, profiling counters, error reporting."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ctx():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from emotivoice_amd.engine import EVEngine
    from emotivoice_amd.packer import pack_state_dict
    from emotivoice_amd.synthetic import synth_inputs, synth_state_dict
    from oracle.jets_oracle import to_torch_sd
    sd = synth_state_dict(0, "parity")
    blob, man = pack_state_dict(sd)
    eng = EVEngine()
    eng.load_blob(blob, man)
    return dict(eng=eng, blob=blob, man=man, sd=to_torch_sd(sd), utts=synth_inputs(81, [37, 64], [3, 4]))


def test_alpha_scales_durations_like_the_reference(ctx):
    """ev_synthesize's alpha is GaussianUpsampling.forward's own alpha (alignment.py:183: ds = ds * alpha (float), T = int(sum),
    centres from the scaled durations).  The reference's inference branch never forwards JETSGenerator.forward's alpha to it
    (model_open_source.py:142), so this is an extension of the C ABI; the Python mirror passes 1.0 (test_gpu_generator.py)."""
    from oracle import EVShapes, jets_forward
    u = ctx["utts"][0]
    for alpha in (1.3, 0.6):
        ref = jets_forward(ctx["sd"], u["ling"], u["speaker"], u["style"], u["content"], EVShapes(), duration_scale=alpha)
        out = ctx["eng"].synthesize([u], alpha=alpha, vocoder=False)
        assert np.array_equal(out["durations"], ref["log_duration_predictions"].numpy())     # unscaled integer durations
        assert int(out["mel_lens"][0]) == int(ref["mel_len"])
        assert rel_l2(out["mel"], ref["dec_outputs"].numpy()) < 1e-3, alpha


def test_device_inputs_on_external_stream_match_host_inputs(ctx):
    from emotivoice_amd import _ffi
    eng, utts = ctx["eng"], ctx["utts"]
    host = eng.synthesize(utts)
    ling = torch.from_numpy(np.concatenate([u["ling"] for u in utts])).cuda()
    cu = np.array([0, 37, 101], np.int32)
    spk = torch.tensor([u["speaker"] for u in utts], dtype=torch.int64).cuda()
    style = torch.from_numpy(np.stack([u["style"] for u in utts])).cuda()
    content = torch.from_numpy(np.stack([u["content"] for u in utts])).cuda()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        eng.set_stream(s.cuda_stream)
        res = eng.synthesize_raw(2, ling.data_ptr(), cu, spk.data_ptr(), style.data_ptr(), content.data_ptr(), 1.0,
                                 _ffi.EV_FLAG_DEVICE_INPUTS | _ffi.EV_FLAG_WANT_INT16)
        dev = eng.result_to_numpy(res, want_int16=True)
    eng.set_stream(0)
    assert np.array_equal(dev["wav"], host["wav"]) and np.array_equal(dev["durations"], host["durations"])
    assert np.array_equal(dev["wav_i16"], (dev["wav"] * np.float32(32768.0)).astype(np.int64).astype(np.int16))


def test_borrowed_device_blob_equals_host_blob(ctx):
    """ev_load_weights_device: the path every rank takes after the RCCL broadcast."""
    from emotivoice_amd.engine import EVEngine
    t = torch.from_numpy(np.frombuffer(ctx["blob"], np.uint8).copy()).cuda()
    eng2 = EVEngine()
    eng2.load_blob_device(t.data_ptr(), t.numel(), keepalive=t)
    a = ctx["eng"].synthesize(ctx["utts"][:1])
    b = eng2.synthesize(ctx["utts"][:1])
    assert np.array_equal(a["wav"], b["wav"]) and np.array_equal(a["mel"], b["mel"])
    eng2.close()


def test_exact_fp32_token_rate_option(ctx):
    """token_rate='f32' (v_mfma_f32_16x16x4_f32) and the default split-precision path agree to fp32 rounding and give the
    same integer durations."""
    from emotivoice_amd.engine import EVEngine
    eng32 = EVEngine(token_rate="f32")
    eng32.load_blob(ctx["blob"], ctx["man"])
    a = ctx["eng"].synthesize(ctx["utts"], vocoder=False)
    b = eng32.synthesize(ctx["utts"], vocoder=False)
    assert np.array_equal(a["durations"], b["durations"])
    assert rel_l2(a["log_durations"], b["log_durations"]) < 5e-6 and rel_l2(a["pitch"], b["pitch"]) < 5e-6
    eng32.close()


def test_profiling_counters_and_errors(ctx):
    from emotivoice_amd.engine import EVEngine, EVError
    eng = ctx["eng"]
    eng.set_profiling(True)
    eng.synthesize(ctx["utts"])
    t = eng.timings()
    ks = {k["name"]: k for k in eng.kernel_stats()}
    eng.set_profiling(False)
    assert t["total"] > 0 and t["vocoder"] > 0 and t["am"] > 0 and t["total"] >= t["vocoder"]
    # the handle is the ABI's own default = the contract mode (mx): conv_pre on the split-precision kernel; 3 up-convs + the 36 ResBlock convs of
    # stages 0-1 on the MX conv-GEMM -- per stage five grouped launches of three convs each + the ResBlocks' three last convs (round 6: ev_config.mx_group) --;
    # stage 2: the last up-conv + 12 layer-wise k = 7 / 11 convs + 3 fused k = 3 pairs; stage 3: 9 fused pairs
    assert ks["voc_conv_gemm_x3"]["launches"] == 1 and ks["voc_conv_gemm_mx"]["launches"] == 3 + 2 * (5 + 3)
    assert ks["voc_conv_c64_mx"]["launches"] == 13 and ks["voc_resblock_pair_c64_mx"]["launches"] == 3
    assert ks["voc_resblock_pair_c32_mx"]["launches"] == 9
    assert ks["voc_conv_gemm_mx"]["flops"] > 0 and ks["dec_f32_attention"]["launches"] == 4 and ks["dec_mx_gemm"]["launches"] == 16
    assert "voc_conv_gemm_f16" not in ks and "dec_f16_gemm" not in ks      # nothing of the fp16 mode runs unless it is asked for
    with pytest.raises(EVError, match="unknown stage"):
        eng.get_stage("no_such_tap")
    fresh = EVEngine()
    with pytest.raises(EVError, match="weights not loaded"):
        fresh.synthesize(ctx["utts"][:1])
    with pytest.raises(EVError, match="bad magic"):
        fresh.load_blob(b"garbage-garbage-garbage-garbage!")
    fresh.close()


def test_integration_md_ctypes_recipe_meets_the_contract():
    """INTEGRATION.md section 2, literally: raw ctypes, ev_default_config with NO field overridden, ev_create, ev_load_weights,
    ev_synthesize, ev_memcpy_d2h -- on the zero-mean fixture the reference itself produced (tests/golden/n28_zero_dc.npz), where fp16
    operands measure 2.4e-3.  The ABI's own default must be inside north_star's 1e-3 on the plain AND the DC-free waveform measure
    (reference call site inference_am_vocoder_joint.py:70-74,122-131)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import os
    from conftest import GOLDEN_DIR
    from emotivoice_amd import _ffi
    from emotivoice_amd.packer import pack_state_dict
    from emotivoice_amd.synthetic import synth_state_dict
    g = np.load(os.path.join(GOLDEN_DIR, "n28_zero_dc.npz"))
    lib = _ffi.lib()
    cfg = _ffi.ev_config()
    lib.ev_default_config(C.byref(cfg))
    assert (cfg.decoder_precision, cfg.vocoder_precision) == (_ffi.EV_PREC_MX, _ffi.EV_PREC_MX)
    h = C.c_void_p()
    assert lib.ev_create(0, C.byref(cfg), C.byref(h)) == 0, lib.ev_last_error(None)
    try:
        blob, manifest = pack_state_dict(synth_state_dict(int(g["weight_seed"]), str(g["dur_mode"])))
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        assert lib.ev_load_weights(h, C.cast(buf, C.c_void_p), len(blob), manifest.encode() if isinstance(manifest, str) else None) == 0, lib.ev_last_error(h)
        ling = np.ascontiguousarray(g["in_ling"], np.int64)
        cu = np.array([0, ling.size], np.int32)
        spk = np.array([int(g["in_speaker"])], np.int64)
        style = np.ascontiguousarray(g["in_style"], np.float32).reshape(1, -1)
        content = np.ascontiguousarray(g["in_content"], np.float32).reshape(1, -1)
        res = _ffi.ev_result()
        rc = lib.ev_synthesize(h, 1, ling.ctypes.data, cu.ctypes.data, spk.ctypes.data, style.ctypes.data, content.ctypes.data,
                               C.c_float(1.0), _ffi.EV_FLAG_WANT_INT16, C.byref(res))
        assert rc == 0, lib.ev_last_error(h)
        assert res.batch == 1 and res.total_frames == int(g["mel_len"]) and res.total_samples == g["wav"].size
        wav = np.empty(res.total_samples, np.float32)
        dur = np.empty(res.total_tokens, np.int64)
        i16 = np.empty(res.total_samples, np.int16)
        assert lib.ev_memcpy_d2h(h, wav.ctypes.data, res.wav, wav.nbytes) == 0
        assert lib.ev_memcpy_d2h(h, dur.ctypes.data, res.durations, dur.nbytes) == 0
        assert lib.ev_memcpy_d2h(h, i16.ctypes.data, res.wav_i16, i16.nbytes) == 0
    finally:
        lib.ev_destroy(h)
    assert np.array_equal(dur, g["dur"])
    d = wav.astype(np.float64) - g["wav"].astype(np.float64)
    ref = g["wav"].astype(np.float64)
    e_wav = float(np.linalg.norm(d) / np.linalg.norm(ref))
    e_ac = float(np.linalg.norm(d) / np.linalg.norm(ref - ref.mean()))
    assert abs(ref.mean()) / ref.std() < 0.2            # the fixture IS zero-mean audio
    assert e_wav < 1e-3 and e_ac < 1e-3, (e_wav, e_ac)
    assert np.array_equal(i16, (wav * np.float32(32768.0)).astype(np.int64).astype(np.int16))


def test_small_batch_three_stream_generator_equals_single_stream_bitwise(ctx):
    """in the default precision a small batch (<= 2048 frame rows) runs the three ResBlocks of a generator stage on three streams with per-ResBlock
    plane sets, the running MRF sum ordered by events (stages 0-1: partial plane sets, `mrf_pl`; stages 2-3: the fp32 sum); ev_config.vocoder_streams = 1 keeps
    everything on the handle's stream.  Same arithmetic, same order of the MRF additions: the waveforms must be bit-identical -- for a single utterance (the
    reference's own call pattern) and for a batch that crosses into the serial large-batch path."""
    from emotivoice_amd.engine import EVEngine
    from emotivoice_amd.synthetic import synth_inputs
    one = EVEngine(vocoder_streams=1)
    one.load_blob(ctx["blob"], ctx["man"])
    assert one.decoder_precision == "mx" and one.vocoder_precision == "mx"
    for utts in (ctx["utts"][:1], ctx["utts"], synth_inputs(83, [200, 180, 190, 170], [5, 6, 7, 8])):
        a = ctx["eng"].synthesize(utts)
        b = one.synthesize(utts)
        assert np.array_equal(a["durations"], b["durations"]) and np.array_equal(a["mel"], b["mel"])
        assert np.array_equal(a["wav"], b["wav"]), len(utts)
    # ... and repeated single-utterance calls on the three-stream path are stable run to run (the event ordering is not a race)
    first = ctx["eng"].synthesize(ctx["utts"][:1])["wav"].copy()
    for _ in range(5):
        assert np.array_equal(ctx["eng"].synthesize(ctx["utts"][:1])["wav"], first)
    one.close()


def test_grouped_level_launches_equal_one_launch_per_conv_bitwise(ctx):
    """Round 6 (ev_config.mx_group): above the small-batch threshold the conv_gemm_mx_kernel stages of the default generator issue the three ResBlocks' same-level
    convs as ONE grouped grid (conv_gemm_mx_group3_kernel: the k = 11 tiles, then the k = 7 tiles, then the k = 3 tiles, each running its own instantiation's code
    on its own problem, one set of intermediates per ResBlock); mx_group = False issues one launch per conv, ResBlock after ResBlock.  Same products in the same order:
    bit-identical waveforms -- and the grouped path must really be taken: its launches appear in the profile as voc_conv_gemm_mx records whose taps are 3 + 7 + 11."""
    from emotivoice_amd.engine import EVEngine
    from emotivoice_amd.synthetic import synth_inputs
    one = EVEngine(mx_group=False)
    one.load_blob(ctx["blob"], ctx["man"])
    utts = synth_inputs(84, [200, 180, 190, 170, 60], [5, 6, 7, 8, 9])          # > 2048 frame rows: the large-batch path
    a = ctx["eng"].synthesize(utts)
    b = one.synthesize(utts)
    assert int(sum(a["mel_lens"])) > 2048
    assert np.array_equal(a["durations"], b["durations"]) and np.array_equal(a["mel"], b["mel"])
    assert np.array_equal(a["wav"], b["wav"])
    for _ in range(3):          # run to run
        assert np.array_equal(ctx["eng"].synthesize(utts)["wav"], a["wav"])
    ctx["eng"].set_profiling(True)
    try:
        c = ctx["eng"].synthesize(utts)
        recs = ctx["eng"].launch_records()
    finally:
        ctx["eng"].set_profiling(False)
    assert np.array_equal(c["wav"], a["wav"])
    grouped = [r for r in recs if r["name"] == "voc_conv_gemm_mx" and r["taps"] == 21]
    assert len(grouped) == 10, len(grouped)          # stages 0 and 1: five grouped levels each
    one.close()
