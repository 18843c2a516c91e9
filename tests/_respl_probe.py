"""Subprocess body of tests/test_gpu_parity.py::test_residual_from_planes_flow_probe: the mx engine with EV_MX_RESPL=1 (ResBlock residuals rebuilt
from the plane sets, DESIGN.md "Next" item 0c) against the reference's own fixtures; prints one JSON line of errors.  keep_stages off = the
production flow (no fp32 copies at all), then on (stage taps still written)."""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    assert os.environ.get("EV_MX_RESPL") == "1"
    from conftest import GOLDEN_DIR, rel_l2
    from emotivoice_amd.engine import EVEngine
    from emotivoice_amd.packer import pack_state_dict
    from oracle import synth_state_dict
    res = {}
    engines = {}
    for path in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        name = os.path.basename(path)[:-4]
        if name.startswith("simbert_"):
            continue
        g = np.load(path)
        for keep in (False, True):
            key = (str(g["dur_mode"]), keep)
            if key not in engines:
                eng = EVEngine(precision="mx", keep_stages=keep)
                eng.load_blob(*pack_state_dict(synth_state_dict(0, key[0])))
                engines[key] = eng
            out = engines[key].synthesize([dict(ling=g["in_ling"], speaker=int(g["in_speaker"]), style=g["in_style"], content=g["in_content"])])
            ref = np.asarray(g["wav"], np.float64)
            d = np.linalg.norm(np.asarray(out["wav"], np.float64) - ref)
            res["%s/keep%d" % (name, keep)] = dict(wav=float(rel_l2(out["wav"], g["wav"])), wav_ac=float(d / max(np.linalg.norm(ref - ref.mean()), 1e-30)),
                                                    mel=float(rel_l2(out["mel"], g["mel"])), dur_ok=bool(np.array_equal(out["durations"], g["dur"])))
    print("RESPL_PROBE " + json.dumps(res, sort_keys=True))


if __name__ == "__main__":
    main()
