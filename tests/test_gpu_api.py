"""MI355X: remaining C-ABI surface -- duration scale alpha, device-pointer inputs on an external stream, borrowed device
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
    # conv_pre + 4 ups + the 48 ResBlock convs of stages 0-2 that are not in a fused pair kernel; 3 fused pairs at C = 64, 9 at C = 32
    assert ks["voc_conv_gemm_f16"]["launches"] == 53 and ks["voc_resblock_pair_c64"]["launches"] == 3
    assert ks["voc_resblock_pair_c32"]["launches"] == 9
    assert ks["voc_conv_gemm_f16"]["flops"] > 0 and ks["dec_f16_attention"]["launches"] == 4
    with pytest.raises(EVError, match="unknown stage"):
        eng.get_stage("no_such_tap")
    fresh = EVEngine()
    with pytest.raises(EVError, match="weights not loaded"):
        fresh.synthesize(ctx["utts"][:1])
    with pytest.raises(EVError, match="bad magic"):
        fresh.load_blob(b"garbage-garbage-garbage-garbage!")
    fresh.close()
