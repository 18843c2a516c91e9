"""Pin the CPU oracle (oracle/jets_oracle.py) against outputs of the REFERENCE ITSELF
(tests/golden/*.npz, produced by tests/golden/make_golden.py with the reference imported
from /root/reference).  Integer outputs must match exactly; fp32 outputs to 2e-5 rel-L2
(same arithmetic, different op order only)."""
import glob
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN_DIR, rel_l2
from oracle import EVShapes, jets_forward, synth_state_dict
from oracle.jets_oracle import to_torch_sd, wav_to_int16

CASES = sorted(p for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")) if not os.path.basename(p).startswith("simbert_"))
_SD = {}


def _sd(seed, mode):
    if (seed, mode) not in _SD:
        _SD[(seed, mode)] = to_torch_sd(synth_state_dict(seed, mode))
    return _SD[(seed, mode)]


def test_fixtures_present():
    assert len(CASES) >= 5


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_oracle_matches_reference(path):
    g = np.load(path)
    sd = _sd(int(g["weight_seed"]), str(g["dur_mode"]))
    taps = {}
    alpha = float(g["alpha"]) if "alpha" in g.files else 1.0
    out = jets_forward(sd, g["in_ling"], int(g["in_speaker"]), g["in_style"], g["in_content"], EVShapes(), alpha=alpha, taps=taps)
    # integer path: bit exact
    assert np.array_equal(out["log_duration_predictions"].numpy(), g["dur"])
    assert int(out["mel_len"]) == int(g["mel_len"])
    tol = 2e-5
    assert rel_l2(out["log_dur_raw"].numpy(), g["log_dur"]) < tol
    assert rel_l2(out["pitch_predictions"].numpy(), g["pitch"]) < tol
    assert rel_l2(out["energy_predictions"].numpy(), g["energy"]) < tol
    assert rel_l2(out["dec_outputs"].numpy(), g["mel"]) < tol
    assert rel_l2(out["wav_predictions"].numpy(), g["wav"]) < tol
    assert out["wav_predictions"].numel() == 256 * int(g["mel_len"])
    # stage taps (possibly strided along time in the fixture)
    n_taps = 0
    for key in g.files:
        if not key.startswith("tap_"):
            continue
        m = re.match(r"tap_(\w+?)(?:__ax(\d)_s(\d+))?$", key)
        name, ax, stride = m.group(1), m.group(2), m.group(3)
        mine = taps[name].numpy()
        if name.startswith("voc_") and mine.ndim == 2 and mine.shape[0] != g[key].shape[0]:
            mine = mine.T
        if stride is not None:
            sl = [slice(None)] * mine.ndim
            sl[int(ax)] = slice(None, None, int(stride))
            mine = mine[tuple(sl)]
        assert mine.shape == g[key].shape, (key, mine.shape, g[key].shape)
        assert rel_l2(mine, g[key]) < tol, key
        n_taps += 1
    assert n_taps >= 15


def test_alpha_is_ignored_but_duration_scale_is_not():
    """model_open_source.py:142 drops alpha in the inference branch; the oracle's duration_scale is GaussianUpsampling's own
    alpha (alignment.py:183-195: float scaling, mel_len = int(sum))."""
    g = np.load(os.path.join(GOLDEN_DIR, "tiny_parity.npz"))
    sd = _sd(int(g["weight_seed"]), str(g["dur_mode"]))
    args = (sd, g["in_ling"], int(g["in_speaker"]), g["in_style"], g["in_content"], EVShapes())
    a = jets_forward(*args, alpha=0.5)
    assert int(a["mel_len"]) == int(g["mel_len"]) and rel_l2(a["dec_outputs"].numpy(), g["mel"]) < 2e-5
    b = jets_forward(*args, duration_scale=1.3)
    assert int(b["mel_len"]) == int(np.float32(g["dur"].astype(np.float32) * np.float32(1.3)).sum().astype(np.int32))
    assert np.array_equal(b["log_duration_predictions"].numpy(), g["dur"])


def test_int16_epilogue_truncates_toward_zero():
    # inference_am_vocoder_joint.py:130-131: numpy astype('int16') == C cast
    x = np.array([0.0, 0.99997, -0.99997, 1.5 / 32768, -1.5 / 32768, 0.5], np.float32)
    assert wav_to_int16(x).tolist() == [0, 32767, -32767, 1, -1, 16384]
